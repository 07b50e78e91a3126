"""Debug: launch time of lbm_fir_fft_f32 (825 taps, 30 x 1500 x 16 384) for the tree's library and every tools/probe/liblbm_firx_*.so
(experiment builds of tools/probe/build_flag_variant.sh), libraries interleaved over three rounds, the first dropped; also says whether
each build's output equals the tree's bit for bit."""
import ctypes as C, glob, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lbm_amd
dev = torch.device("cuda", 0)
taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
x = torch.randn(16384 * 30, 1500, device=dev) * 1e-5
y = torch.empty_like(x)
tp = torch.as_tensor(np.asarray(taps, dtype=np.float32), device=dev)
paths = [os.path.join(ROOT, "lbm_amd", "liblbm_hip.so")] + sorted(glob.glob(os.path.join(ROOT, "tools", "probe", "liblbm_firx_*.so")))
libs = []
for path in paths:
    l = C.CDLL(path)
    l.lbm_fir_fft_tables_bytes.restype = C.c_size_t
    l.lbm_fir_fft_tables_bytes.argtypes = [C.c_int32, C.c_int32]
    l.lbm_fir_fft_tables_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    l.lbm_fir_fft_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    nb = l.lbm_fir_fft_tables_bytes(1500, len(taps))
    tab = torch.empty((nb + 3) // 4, dtype=torch.float32, device=dev)
    assert l.lbm_fir_fft_tables_init(tab.data_ptr(), tp.data_ptr(), 1500, len(taps), None) == 0
    libs.append((os.path.basename(path)[7:-3], l, tab))
res = {n: [] for n, _, _ in libs}
ref = None
for rnd in range(3):
    for n, l, tab in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for r in range(5):
            e0.record()
            assert l.lbm_fir_fft_f32(x.data_ptr(), y.data_ptr(), x.shape[0], 1500, tab.data_ptr(), len(taps), 0, None) == 0
            e1.record(); torch.cuda.synchronize()
            if rnd and r: res[n].append(e0.elapsed_time(e1))
        if rnd == 1:
            if ref is None: ref = y.clone()
            print(f"{n:20s} equal to the tree's output: {bool(torch.equal(y, ref))}  max |diff| {float((y - ref).abs().max()):.3e}", flush=True)
for n, _, _ in libs:
    print(f"{n:20s} {np.mean(res[n]):.3f} ms (min {min(res[n]):.3f})", flush=True)
