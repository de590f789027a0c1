#!/usr/bin/env python3
"""The control plane of bench.py and run_eval (lamp_amd.sharding.ControlPlane: gloo rendezvous, then an "nccl" = RCCL group for the barrier and the
gathers) as a ONE-rank group, so that the exact calls run on a one-GPU box (RCCL refuses two ranks on one device; the
2-rank tests therefore run on gloo).

    python tools/check_rccl_control_plane.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29544')
os.environ['LAMP_FORCE_DIST'] = '1'
import bench  # noqa: E402
from lamp_amd import sharding  # noqa: E402

torch.cuda.set_device(0)
cp = sharding.ControlPlane(0, 1, torch.device('cuda', 0), 'nccl')
assert cp.backend == 'nccl', cp.note
cp.barrier()
rows = cp.gather(torch.tensor([0., 0., 1.5, 640.], dtype=torch.float64))
assert len(rows) == 1 and rows[0].tolist() == [0., 0., 1.5, 640.]
logits = cp.gather(torch.arange(12, dtype=torch.float32, device='cuda').view(3, 4))
assert logits[0].tolist() == torch.arange(12.).view(3, 4).tolist()
assert cp.gather_objects(bench.device_identity(0)) == [bench.device_identity(0)] and cp.ranks_in_group() == 1
cp.close()
print('rccl control plane ok (%s, device %s)' % (cp.backend, bench.device_identity(0)))
