import ctypes, sys, os, torch
sys.path.insert(0, '/root/repo')
os.environ['LAMP_HIP_LIBRARY'] = '/root/repo/lamp_amd/liblamp_hip_tuning.so'
from lamp_amd import _native as N
dev = torch.device('cuda:0')
lib = N.lib()
M, d, dff, L = 2880, 512, 512, 90
g = torch.Generator().manual_seed(0)
rnd = lambda *sh: torch.randn(*sh, generator=g).to(dev)
A, Y, T = rnd(M, d), rnd(M, d), rnd(L, d)
wfc, w1, w2 = rnd(d, d) / d ** 0.5, rnd(dff, d) / d ** 0.5, rnd(d, dff) / dff ** 0.5
b1, b2, g1, be1, g2, be2 = rnd(dff), rnd(d), rnd(d), rnd(d), rnd(d), rnd(d)
fnc = lib.lamp_debug_launch_chain
fnc.restype = ctypes.c_int
fnc.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong,
                ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_void_p] * 6
pk = [N.weight_pack(w, f) for f in (0, 1) for w in (wfc, w1, w2)]
geom = lib.lamp_debug_chain_geometry
geom.argtypes = [ctypes.c_int]; geom.restype = None
def run(gi, res, r_mod, ffn=True):
    out = torch.zeros(M, d, device=dev)
    geom(gi)
    N.check(fnc(A.data_ptr(), d, d, res.data_ptr() if res is not None else None, r_mod, M, d, wfc.data_ptr(), g1.data_ptr(), be1.data_ptr(),
                w1.data_ptr() if ffn else None, b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), g2.data_ptr(), be2.data_ptr(), dff, out.data_ptr(), N.stream(),
                *[t.data_ptr() for t in pk]), 'chain')
    torch.cuda.synchronize()
    return out
for label, res, r_mod, ffn in (('row residual', Y, 0, True), ('modulo residual', T, L, True), ('no residual', None, 0, True),
                               ('modulo residual, no ffn', T, L, False), ('row residual, no ffn', Y, 0, False)):
    ref = run(8, res, r_mod, ffn)
    for gi in (0, 12, 13, 11, 14, 15, 16, 17, 18, 19, 20):
        try:
            got = run(gi, res, r_mod, ffn)
        except Exception as e:
            print(label, gi, 'error', e); continue
        got2 = run(gi, res, r_mod, ffn)
        bad = (got != ref)
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print('%-24s geom %2d equal %s repeat %s bad rows %d (first %s) bad cols %d (first %s)' % (label, gi, torch.equal(got, ref), torch.equal(got, got2),
              rows.numel(), rows[:6].tolist(), cols.numel(), cols[:6].tolist()))
geom(-1)
