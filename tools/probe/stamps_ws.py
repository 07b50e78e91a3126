"""Debug: phase times of conv_wino3_ws_kernel's tile walk (needs tools/probe/liblbm_stamps_ws.so from build_stamps_ws.sh)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LBM_AMD_LIB"] = os.path.join(ROOT, "tools", "probe", "liblbm_stamps_ws%s.so" % os.environ.get("WS_ABLATE", ""))
from lbm_amd import _lib
lib, p = _lib.load(), _lib.ptr
lib.lbm_debug_ws_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
E, L, ld, Cout = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 375, 376, 256
x = torch.randn(E, 128, ld, device=dev).clamp_min(0); w = torch.randn(Cout, 128, 3, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
tab = torch.empty(lib.lbm_conv1d_wino3_tables_bytes(128, Cout), dtype=torch.uint8, device=dev)
lib.lbm_conv1d_wino3_tables_init(p(tab), p(w), 128, Cout, None)
y = torch.empty(E, Cout, ld, device=dev)
nblk = 256
st = torch.zeros(nblk * 8 * 12, dtype=torch.int64, device=dev)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
for rep in range(3):
    st.zero_(); lib.lbm_debug_ws_stamps(st.data_ptr())
    t0.record()
    assert lib.lbm_conv1d_wino3_f32(p(x), ld, p(tab), p(b), p(y), ld, E, 128, L, Cout, 1, None) == 0
    t1.record(); torch.cuda.synchronize()
print(f"launch: {t0.elapsed_time(t1):.3f} ms")
s = st.cpu().numpy().reshape(nblk, 8, 12).astype(np.float64)
ntiles = E * 3 / 64
names = ["wait for half 0 + edges", "barrier A", "request half 1", "output of the previous tile", "MFMA half 0", "wait for half 1 + edges", "barrier B",
         "request next tile's half 0", "MFMA half 1", "exchange write"]
tot = s[:, :, :10].sum(2).mean() / ntiles
print(f"{E} epochs: {ntiles:.0f} tiles per workgroup, {tot:.0f} cycles per tile (128 MFMAs per wave = 8192 cycles of matrix time; two waves per SIMD: 16384)")
for w_ in (0, 3, 7):
    print(f" wave {w_}:")
    for i, n in enumerate(names):
        v = s[:, w_, i].mean() / ntiles
        print(f"   {n:32s} {v:8.0f}  {100 * v / tot:5.1f} %")
