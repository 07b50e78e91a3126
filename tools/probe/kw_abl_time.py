"""Debug: launch time of lbm_conv1d_k4s2_wino_f32 on the encoder's layers 1 and 2 for the tree's library and every tools/probe/liblbm_kw_*.so
(ablation builds of build_kw_abl.sh: results are garbage by design)."""
import ctypes as C, glob, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
E = 16384
with np.load(os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")) as f:
    W = {k: torch.from_numpy(f[k]).to(dev) for k in f.files}
paths = [os.path.join(ROOT, "lbm_amd", "liblbm_hip.so")] + sorted(glob.glob(os.path.join(ROOT, "tools", "probe", "liblbm_kw_*.so")))
data = {}
for name, Cin, Lin, Cout, wk in (("conv1", 30, 1500, 64, "encoder.0"), ("conv2", 64, 750, 128, "encoder.2")):
    PX, Lout = (Lin + 3) & ~3, Lin // 2
    P = (Lout + 3) & ~3
    data[name] = (torch.randn(E, Cin, PX, device=dev, generator=g).clamp_min(0), W[wk + ".weight"], W[wk + ".bias"], torch.empty(E, Cout, P, device=dev), Cin, Lin, Cout, PX, P)
libs = []
for path in paths:
    l = C.CDLL(path)
    l.lbm_conv1d_k4s2_wino_f32.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    libs.append((os.path.basename(path)[7:-3], l))
res = {n: {k: [] for k in data} for n, _ in libs}
for rnd in range(3):          # round 0 warms the chip up and is dropped; libraries interleaved so that drift hits all of them alike
    for n, l in libs:
        for name, (x, w, b, y, Cin, Lin, Cout, PX, P) in data.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for r in range(5):
                e0.record()
                assert l.lbm_conv1d_k4s2_wino_f32(x.data_ptr(), PX, w.data_ptr(), b.data_ptr(), y.data_ptr(), P, E, Cin, Lin, Cout, 1, None) == 0
                e1.record(); torch.cuda.synchronize()
                if rnd and r: res[n][name].append(e0.elapsed_time(e1))
for n, _ in libs:
    print(f"{n:14s} " + "   ".join(f"{k} {np.mean(v):.3f} (min {min(v):.3f})" for k, v in res[n].items()), flush=True)
