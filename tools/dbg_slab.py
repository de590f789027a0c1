import ctypes, sys, os, torch
sys.path.insert(0, '/root/repo')
os.environ.setdefault('LAMP_HIP_LIBRARY', '/root/repo/lamp_amd/liblamp_hip_tuning.so')
from lamp_amd import _native as N
dev = torch.device('cuda:0')
lib = N.lib()
fn = lib.lamp_debug_slab_gemm
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong,
               ctypes.c_void_p, ctypes.c_void_p]
K = Nn = 512
for G in [int(a) for a in os.environ.get("GS", "1,2,3,4,5,8,9,10").split(",")]:
    M = 256 * 4 * G - 5
    g = torch.Generator().manual_seed(G)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(Nn, K, generator=g) / K ** 0.5).to(dev)
    wq = N.weight_pack(w, 1)
    want = N.linear(x, w)
    out = torch.zeros(M, Nn, device=dev)
    N.check(fn(x.data_ptr(), M, K, K, wq.data_ptr(), None, Nn, None, None, Nn, 0, out.data_ptr(), None, Nn, None, N.stream()), 'slab')
    torch.cuda.synchronize()
    bad = (out != want)
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    rin = sorted(set((rows % (4 * G)).tolist()))
    print('G=%2d M=%5d equal %s bad rows %d cols %d; bad row-in-slab %s; maxdiff %.3g' % (G, M, torch.equal(out, want), rows.numel(), cols.numel(), rin[:40],
          (out - want).abs().max().item()))
print('---- pattern at G=3')
G = 3
M = 256 * 4 * G
g = torch.Generator().manual_seed(7)
x = torch.randn(M, K, generator=g).to(dev)
w = (torch.randn(Nn, K, generator=g) / K ** 0.5).to(dev)
wq = N.weight_pack(w, 1)
want = N.linear(x, w)
for rep in range(3):
    out = torch.zeros(M, Nn, device=dev)
    N.check(fn(x.data_ptr(), M, K, K, wq.data_ptr(), None, Nn, None, None, Nn, 0, out.data_ptr(), None, Nn, None, N.stream()), 'slab')
    torch.cuda.synchronize()
    bad = (out != want).view(256, 12, 8, 64)
    wg_wave = bad.any(3).any(1)            # [wg, wave]
    print('rep', rep, 'bad workgroups', int(wg_wave.any(1).sum()), 'bad (wg, wave) pairs', int(wg_wave.sum()), 'first', wg_wave.nonzero()[:12].tolist())
    b0 = wg_wave.nonzero()[0].tolist()
    blk = bad[b0[0], :, b0[1], :]
    print('   in that block: bad rows', blk.any(1).nonzero().flatten().tolist(), 'bad lanes', blk.any(0).nonzero().flatten().tolist()[:70])
    d = (out - want).view(256, 12, 8, 64)[b0[0], :, b0[1], :]
    print('   diffs row0', d[0, :8].tolist())
print('---- which k quad is missing')
out = torch.zeros(M, Nn, device=dev)
N.check(fn(x.data_ptr(), M, K, K, wq.data_ptr(), None, Nn, None, None, Nn, 0, out.data_ptr(), None, Nn, None, N.stream()), 'slab')
torch.cuda.synchronize()
bad = (out != want).view(256, 12, 8, 64).any(3).any(1).nonzero()
xd, wd = x.double(), w.double()
import collections
hist = collections.Counter()
for wg, wv in bad.tolist():
    rows = slice(wg * 12, wg * 12 + 12); cols = slice(wv * 64, wv * 64 + 64)
    diff = (out[rows, cols] - want[rows, cols]).double()
    res = []
    for q in range(128):
        contrib = xd[rows, 4 * q:4 * q + 4] @ wd[cols, 4 * q:4 * q + 4].t()
        res.append(((diff + contrib).norm().item(), q))
    res.sort()
    hist[(res[0][1] // 4, res[0][1] % 4, res[0][0] < 1e-3)] += 1
print('(chunk, quad, exact match) -> count:', sorted(hist.items()))
