"""Debug: lbm_fir_fft_f32 of the tree's library against a saved copy (tools/probe/liblbm_base.so), bit for bit, + timing."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lbm_amd
libs = {"base": C.CDLL(os.path.join(ROOT, "tools", "probe", "liblbm_base.so")), "new": C.CDLL(os.path.join(ROOT, "lbm_amd", "liblbm_hip.so"))}
for l in libs.values():
    l.lbm_fir_fft_tables_bytes.restype = C.c_size_t
    l.lbm_fir_fft_tables_bytes.argtypes = [C.c_int32, C.c_int32]
    l.lbm_fir_fft_tables_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    l.lbm_fir_fft_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(7)
def run(lib, x, taps, pad):
    rows, T = x.shape
    tp = torch.as_tensor(np.asarray(taps, dtype=np.float32), device=dev)
    nb = lib.lbm_fir_fft_tables_bytes(T, len(taps))
    assert nb
    tab = torch.empty((nb + 3) // 4, dtype=torch.float32, device=dev)
    assert lib.lbm_fir_fft_tables_init(tab.data_ptr(), tp.data_ptr(), T, len(taps), None) == 0
    y = torch.full_like(x, float("nan"))
    assert lib.lbm_fir_fft_f32(x.data_ptr(), y.data_ptr(), rows, T, tab.data_ptr(), len(taps), pad, None) == 0
    torch.cuda.synchronize()
    return y, tab
cases = [(500.0, 4.0, 30.0, 4096 * 30, 1500), (500.0, 4.0, 30.0, 7, 1500), (250.0, 1.0, 40.0, 33, 257), (500.0, 4.0, 30.0, 5, 60001), (500.0, 4.0, 30.0, 3, 60000),
         (1000.0, 4.0, 30.0, 64, 10000), (500.0, 8.0, 30.0, 31, 1499), (500.0, 4.0, 30.0, 9, 300), (128.0, 1.0, 30.0, 11, 513), (500.0, 0.5, 45.0, 4, 4000)]
bad = 0
for fs, lo, hi, rows, T in cases:
    taps = lbm_amd.create_filter(fs, lo, hi)
    x = torch.randn(rows, T, device=dev, generator=g) * 1e-5 + 3e-5
    for pad in (0, 1):
        if libs["base"].lbm_fir_fft_tables_bytes(T, len(taps)) == 0:
            print(f"fs {fs} T {T} ntaps {len(taps)}: no plan"); break
        ya, _ = run(libs["base"], x, taps, pad)
        yb, _ = run(libs["new"], x, taps, pad)
        same = torch.equal(ya.view(torch.int32), yb.view(torch.int32))
        bad += not same
        print(f"fs {fs} T {T} ntaps {len(taps)} rows {rows} pad {pad}: {'identical' if same else 'DIFFERENT max %.3g' % (ya - yb).abs().max().item()}  nan {int(torch.isnan(yb).sum())}")
# timing on the bench shape
taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
x = torch.randn(16384 * 30, 1500, device=dev, generator=g) * 1e-5
for name, lib in libs.items():
    y, tab = run(lib, x, taps, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for r in range(6):
        e0.record()
        lib.lbm_fir_fft_f32(x.data_ptr(), y.data_ptr(), x.shape[0], 1500, tab.data_ptr(), len(taps), 0, None)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{name}: {np.mean(ts[1:]):.3f} ms per 16384 epochs (min {min(ts):.3f})")
sys.exit(1 if bad else 0)
