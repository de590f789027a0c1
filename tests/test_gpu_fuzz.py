"""-m gpu: a seeded slice of the randomised differential campaign (tests/fuzz_parity.py) inside the suite, so that the
driver's GPU run exercises it too: 60 random eval shapes (1-8 heads, head widths 4..256, every mask kind, ragged lengths,
1-3 + 1-3 layers, with / without label self-attention) against the fp64 oracle incl. every attention map, padding
invariance bit for bit; 6 training shapes, every parameter's gradient against torch.autograd on the fp64 oracle."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_fuzz_slice_has_no_mismatch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'fuzz_parity.py'), '60', '6'], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', '')))
    assert r.returncode == 0, r.stderr[-3000:]
    ev = re.search(r'eval cases (\d+) bad (\d+)', r.stdout)
    tr = re.search(r'train cases (\d+) bad (\d+)', r.stdout)
    assert ev and tr, r.stdout[-2000:]
    assert (int(ev.group(1)), int(ev.group(2))) == (60, 0), r.stdout[-3000:]
    # one flagged training case in the campaign's history was a ReLU pre-activation of ~1e-7 flipping sign between fp32 and
    # fp64 (tests/fuzz_parity.py header); the six seeds run here have none
    assert (int(tr.group(1)), int(tr.group(2))) == (6, 0), r.stdout[-3000:]
