"""In-container only (needs /root/reference): the reference's UNMODIFIED main.py -test_only, run against the
dropin/lamp shim on a synthetic dataset in the reference's on-disk format.  It must get through the imports
(main.py:6-8), process_data, LAMP(**kwargs) with every keyword main.py:57-88 passes, get_trainable_parameters
/ Adam, runner.run_model and into test_epoch's `model(src, adj, None, None, ...)` (test.py:41) -- where, on this
GPU-less box, our forward must stop with its "HIP device only" error instead of silently computing on the CPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

MAKE_DATA = r'''
import argparse, random, torch, os
rng = random.Random(3)
n_words, n_labels, max_len = 50, 9, 12
src = {'<blank>': 0, '<unk>': 1, '<s>': 2, '</s>': 3}; src.update({'w%d' % i: 4 + i for i in range(n_words)})
tgt = {'<blank>': 0, '<unk>': 1, '<s>': 2, '</s>': 3}; tgt.update({'l%d' % i: 4 + i for i in range(n_labels)})
def sample(force=None):
    n = rng.randint(1, max_len)
    s = [2] + [rng.randint(4, 4 + n_words - 1) for _ in range(n)] + [3]
    labels = sorted(rng.sample(range(4, 4 + n_labels), rng.randint(1, 3)))
    if force is not None and force not in labels: labels = sorted(labels + [force])
    return s, [2] + labels + [3]
splits = {}
for name, n in (('train', 40), ('valid', 40), ('test', 40)):
    items = [sample(4 + (i % n_labels) if name == 'train' else None) for i in range(n)]
    splits[name] = {'src': [a for a, _ in items], 'tgt': [b for _, b in items]}
data = {'settings': argparse.Namespace(max_seq_len=max_len + 2), 'dict': {'src': src, 'tgt': tgt}, **splits}
os.makedirs('data/synth', exist_ok=True)
torch.save(data, 'data/synth/train_valid_test.pt')
'''

RUN_MAIN = r'''
import sys, runpy, functools, torch
sys.dont_write_bytecode = True
torch.load = functools.partial(torch.load, weights_only=False)  # main.py:23 unpickles a Namespace
sys.argv = ['main.py', '-dataroot', 'data/', '-dataset', 'synth', '-batch_size', '8', '-d_model', '32',
            '-d_inner_hid', '64', '-n_layers_enc', '2', '-n_layers_dec', '2', '-n_head', '2', '-encoder', 'graph',
            '-decoder', 'graph', '-label_mask', 'prior', '-test_only', '-overwrite', '-no_cuda']
runpy.run_path('%s/main.py', run_name='__main__')
''' % REF


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, 'main.py')), reason='reference not present')
def test_reference_main_reaches_our_forward_through_the_shim(tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1',
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, 'dropin'), ROOT, REF]))
    r = subprocess.run([sys.executable, '-c', MAKE_DATA], cwd=tmp_path, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, '-c', RUN_MAIN], cwd=tmp_path, env=env, capture_output=True, text=True,
                       timeout=600)
    err = r.stderr
    assert r.returncode != 0
    assert 'lamp_amd runs on an MI355X HIP device only' in err, err[-3000:]
    assert 'reference/test.py", line 41' in err          # reached the hot-path call site of the eval loop
    assert os.path.join('lamp_amd', 'Models.py') in err   # ... inside OUR LAMP.forward, not the reference's
    assert 'reference/lamp/' not in err                   # the reference's own lamp package was never imported


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, 'main.py')), reason='reference not present')
def test_reference_training_loop_reaches_our_training_forward(tmp_path):
    """Same, without -test_only: main.py -> runner.run_model -> train_epoch's `model(src, adj, None, gold_binary, ...)`
    (train.py:36) in train() mode, i.e. the autograd-recording path of lamp_amd/training.py."""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1',
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, 'dropin'), ROOT, REF]))
    r = subprocess.run([sys.executable, '-c', MAKE_DATA], cwd=tmp_path, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    # train.py:34 hard-codes `.cuda()` on the targets; on this GPU-less box make it a no-op so the loop gets to line 36
    script = RUN_MAIN.replace("'-test_only', ", "'-epoch', '1', ").replace(
        "sys.argv = [", "torch.Tensor.cuda = lambda self, *a, **k: self\nsys.argv = [")
    r = subprocess.run([sys.executable, '-c', script], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    err = r.stderr
    assert r.returncode != 0
    assert 'lamp_amd runs on an MI355X HIP device only' in err, err[-3000:]
    assert 'reference/train.py", line 36' in err
    assert os.path.join('lamp_amd', 'Models.py') in err
    assert 'reference/lamp/' not in err
