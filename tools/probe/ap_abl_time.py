"""Debug: launch time of lbm_affine_post_f32 (30 x 1500, 16 384 epochs, standardized output only) for the tree's library and every
tools/probe/liblbm_ap_*.so (experiment builds of tools/probe/build_flag_variant.sh; ablations give garbage by design)."""
import ctypes as C, glob, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
E, Ch, T = 16384, 30, 1500
x = torch.randn(E, Ch, T, device=dev, generator=g) * 1e-5
A = torch.randn(Ch, Ch, device=dev, generator=g) / 5
b = torch.randn(Ch, device=dev, generator=g) * 1e-6
y = torch.empty(E + 1, Ch, T, device=dev)[:E];    # (one epoch of slack: the coalesced-address ablation writes past the last row)
rs = torch.empty(E, Ch, device=dev); keep = torch.empty(E, dtype=torch.uint8, device=dev)
paths = [os.path.join(ROOT, "lbm_amd", "liblbm_hip.so")] + sorted(glob.glob(os.path.join(ROOT, "tools", "probe", "liblbm_ap_*.so")))
libs = []
for path in paths:
    l = C.CDLL(path)
    l.lbm_affine_post_f32.argtypes = [C.c_void_p] * 7 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_void_p]
    libs.append((os.path.basename(path)[7:-3], l))
res = {n: [] for n, _ in libs}
ref = None
for rnd in range(3):          # round 0 is dropped; libraries interleaved
    for n, l in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for r in range(5):
            e0.record()
            assert l.lbm_affine_post_f32(x.data_ptr(), A.data_ptr(), b.data_ptr(), None, y.data_ptr(), rs.data_ptr(), keep.data_ptr(), E, Ch, Ch, T, 251,
                                         150e-6, 1, 1e-8, None) == 0
            e1.record(); torch.cuda.synchronize()
            if rnd and r: res[n].append(e0.elapsed_time(e1))
        if rnd == 1:
            if ref is None: ref = y.clone()
            print(f"{n:14s} equal to the tree's output: {bool(torch.equal(y, ref))}", flush=True)
for n, _ in libs:
    print(f"{n:14s} {np.mean(res[n]):.3f} (min {min(res[n]):.3f})", flush=True)
