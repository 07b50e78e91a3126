"""Debug: per-block phase stamps of the conv kernel (needs tools/probe/liblbm_stamps.so, a debug build)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LBM_AMD_LIB"] = os.path.join(ROOT, "tools", "probe", "liblbm_stamps.so")
from lbm_amd import _lib
lib = _lib.load()
lib.lbm_debug_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
E = 4096
cfgs = {"conv4": (256, 375, 256, 1, 1, 0), "conv3": (128, 375, 256, 3, 1, 1), "conv2": (64, 750, 128, 4, 2, 1), "conv1": (30, 1500, 64, 4, 2, 1)}
for name, (cin, lin, cout, ks, sd, pd) in cfgs.items():
    lout = (lin + 2 * pd - ks) // sd + 1
    ldx, ldy = (lin + 3) & ~3, (lout + 3) & ~3
    x = torch.randn(E, cin, ldx, device=dev); w = torch.randn(cout, cin, ks, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    y = torch.empty(E, cout, ldy, device=dev)
    bm = 64 if cout <= 64 else (256 if (cout >= 256 and ks in (1, 3)) else 128)
    nblk = E * ((lout + 127) // 128) * ((cout + bm - 1) // bm)
    st = torch.zeros(nblk * 4, dtype=torch.int64, device=dev)
    for rep in range(2):
        lib.lbm_debug_stamps(st.data_ptr())
        rc = lib.lbm_conv1d_pitched_f32(x.data_ptr(), ldx, w.data_ptr(), b.data_ptr(), y.data_ptr(), ldy, E, cin, lin, cout, ks, sd, pd, 1, None)
        torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(-1, 4).astype(np.float64)
    odd = s[1::2]   # odd blocks carry (memtime T0, realtime R0, realtime R3, memtime T3)
    clk = (odd[:, 3] - odd[:, 0]) / np.maximum(odd[:, 2] - odd[:, 1], 1) * 100e6
    print(f"{name}: in-kernel clock median {np.median(clk) / 1e9:.3f} GHz (p10 {np.percentile(clk, 10) / 1e9:.3f}, p90 {np.percentile(clk, 90) / 1e9:.3f})")
    s = s[0::2]
    d = np.diff(s, axis=1)
    ok = (d > 0).all(axis=1) & (d < 1e7).all(axis=1)
    d = d[ok]
    span = (s[ok, 3].max() - s[ok, 0].min())
    print(f"{name}: blocks {len(d)}  prologue {d[:,0].mean():.0f}  loop {d[:,1].mean():.0f}  epilogue {d[:,2].mean():.0f}  total {d.sum(1).mean():.0f} ticks; "
          f"p50/p90 prologue {np.percentile(d[:,0],50):.0f}/{np.percentile(d[:,0],90):.0f} epilogue {np.percentile(d[:,2],50):.0f}/{np.percentile(d[:,2],90):.0f}; kernel span {span:.0f} ticks")
