#!/usr/bin/env python3
"""The exact torch.distributed calls bench.py makes on its "nccl" (= RCCL) control plane -- init with device_id, barrier
with device_ids, all_gather of a float64 device tensor -- as a one-rank group, so the code path can be exercised on a
one-GPU box (RCCL refuses two ranks on one device; the 2-rank tests therefore run on gloo).

    python tools/check_rccl_control_plane.py
"""
import datetime
import os

import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29544')
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=600))
dist.barrier(device_ids=[0])
mine = torch.tensor([0., 0., 1.5, 640.], dtype=torch.float64, device=dev)
rows = [torch.empty_like(mine)]
dist.all_gather(rows, mine)
assert rows[0].cpu().tolist() == [0., 0., 1.5, 640.]
dist.barrier(device_ids=[0])
dist.destroy_process_group()
print('rccl control plane ok')
