import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..', 'tests'))
import test_gpu_training as T
from lamp_amd import training
dev = torch.device('cuda:0')
for dropout in (0.0, 0.1):
    m, sd, blocked, seq, spos, h, tgt = T.build(T.CASES['reuters_like'], dev, dropout=dropout)
    m.train()
    res = {}
    for comp in (False, True):
        training.COMPOSITE_CALLS = comp
        training.DEFER_WEIGHT_GRADS = False
        m.zero_grad(set_to_none=True)
        torch.manual_seed(3)
        logits, enc, _ = m((seq.to(dev), spos.to(dev)), None, None, tgt.to(dev))
        F.binary_cross_entropy_with_logits(logits, tgt.to(dev)).backward()
        res[comp] = (logits.detach().clone(), enc.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    print('dropout', dropout, 'logits equal', torch.equal(res[0][0], res[1][0]), 'enc equal', torch.equal(res[0][1], res[1][1]))
    for n in res[0][2]:
        a, b = res[0][2][n], res[1][2][n]
        if not torch.equal(a, b):
            print('   differs', n, (a - b).abs().max().item(), a.abs().max().item())
