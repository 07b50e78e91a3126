"""Debug: launch time of lbm_conv1d_wino3_f32 on the encoder's third layer (128 -> 256, L = 375, 16 384 epochs, post-ReLU-like inputs) for the
tree's library and every tools/probe/liblbm_w3_*.so (tools/probe/build_flag_variant.sh conv_wino_ws.hip w3_<name>), interleaved over three
rounds, the first dropped; also says whether each build's output equals the tree's bit for bit."""
import ctypes as C, glob, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
E, Cin, Cout, L, P = 16384, 128, 256, 375, 376
with np.load(os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")) as f:
    w = torch.from_numpy(f["encoder.4.weight"]).to(dev); b = torch.from_numpy(f["encoder.4.bias"]).to(dev)
x = torch.randn(E, Cin, P, device=dev, generator=g).clamp_min(0)
y = torch.empty(E, Cout, P, device=dev)
paths = [os.path.join(ROOT, "lbm_amd", "liblbm_hip.so")] + sorted(glob.glob(os.path.join(ROOT, "tools", "probe", "liblbm_w3_*.so")))
libs = []
for path in paths:
    l = C.CDLL(path)
    l.lbm_conv1d_wino3_tables_bytes.restype = C.c_size_t
    l.lbm_conv1d_wino3_tables_bytes.argtypes = [C.c_int32, C.c_int32]
    l.lbm_conv1d_wino3_tables_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    l.lbm_conv1d_wino3_f32.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    tab = torch.empty(l.lbm_conv1d_wino3_tables_bytes(Cin, Cout), dtype=torch.uint8, device=dev)
    assert l.lbm_conv1d_wino3_tables_init(tab.data_ptr(), w.data_ptr(), Cin, Cout, None) == 0
    libs.append((os.path.basename(path)[7:-3], l, tab))
res = {n: [] for n, _, _ in libs}
ref = None
for rnd in range(3):
    for n, l, tab in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for r in range(5):
            e0.record()
            assert l.lbm_conv1d_wino3_f32(x.data_ptr(), P, tab.data_ptr(), b.data_ptr(), y.data_ptr(), P, E, Cin, L, Cout, 1, None) == 0
            e1.record(); torch.cuda.synchronize()
            if rnd and r: res[n].append(e0.elapsed_time(e1))
        if rnd == 1:
            if ref is None: ref = y.clone()
            print(f"{n:14s} equal to the tree's output: {bool(torch.equal(y[..., :L], ref[..., :L]))}", flush=True)
for n, _, _ in libs:
    print(f"{n:14s} {np.mean(res[n]):.3f} ms (min {min(res[n]):.3f})", flush=True)
