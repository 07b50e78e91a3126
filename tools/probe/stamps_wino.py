"""Debug: phase times of conv_wino3_kernel's chunk loop (needs tools/probe/liblbm_stamps_wino.so)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LBM_AMD_LIB"] = os.path.join(ROOT, "tools", "probe", "liblbm_stamps_wino.so")
from lbm_amd import _lib
lib = _lib.load()
lib.lbm_debug_wino.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
E, cin, lin, cout = 4096, 128, 375, 256
ldx = ldy = 376
x = torch.randn(E, cin, ldx, device=dev); w = torch.randn(cout, cin, 3, device=dev) * 0.05; b = torch.randn(cout, device=dev)
y = torch.empty(E, cout, ldy, device=dev)
nblk = E * 3 * 2
st = torch.zeros(nblk * 10, dtype=torch.int64, device=dev)
for rep in range(2):
    lib.lbm_debug_wino(st.data_ptr())
    rc = lib.lbm_conv1d_pitched_f32(x.data_ptr(), ldx, w.data_ptr(), b.data_ptr(), y.data_ptr(), ldy, E, cin, lin, cout, 3, 1, 1, 1, None)
    assert rc == 0
    torch.cuda.synchronize()
s = st.cpu().numpy().reshape(-1, 10).astype(np.float64)
d = s[:, :8]
ok = (d >= 0).all(axis=1) & (d < 1e7).all(axis=1) & (s[:, 8] > 0)
d = d[ok]
names = ["prologue", "mfma half 1 (x15)", "store_chunk (x15)", "load_chunk (x15)", "mfma half 2 (x15)", "barrier (x15)", "last chunk", "epilogue"]
tot = d.sum(1).mean()
print(f"blocks {len(d)}, ticks per tile {tot:.0f}; kernel span {s[ok, 9].max() - s[ok, 8].min():.0f}")
for n, v in zip(names, d.mean(0)):
    print(f"  {n:22s} {v:9.0f}  {100 * v / tot:5.1f} %")
