#!/usr/bin/env python3
"""Command line of lamp_amd/isa_guard.py (the ISA check of the hand-scheduled kernels that lamp_amd.build runs on every build):

    python tools/check_untracked_loads.py [file.s | file.hip] [kernel-name substring ...]      exit 1 on a finding
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lamp_amd.isa_guard import *  # noqa: F401,F403,E402
from lamp_amd.isa_guard import main  # noqa: E402

if __name__ == '__main__':
    main()
