"""Debug: phase times of conv_tile_kernel's chunk loop for the encoder's first two layers (needs tools/probe/liblbm_stamps_tile.so)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LBM_AMD_LIB"] = os.path.join(ROOT, "tools", "probe", "liblbm_stamps_tile.so")
from lbm_amd import _lib
lib = _lib.load()
lib.lbm_debug_tile.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
E = 4096
for name, cin, lin, ldx, cout, lout, ldy, bm in (("conv1 30->64 k4 s2", 30, 1500, 1500, 64, 750, 752, 64), ("conv2 64->128 k4 s2", 64, 750, 752, 128, 375, 376, 128)):
    x = torch.randn(E, cin, ldx, device=dev); w = torch.randn(cout, cin, 4, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    y = torch.empty(E, cout, ldy, device=dev)
    ptiles = (lout + 127) // 128
    nblk = E * ptiles * ((cout + bm - 1) // bm)
    nch = (cin + 7) // 8
    st = torch.zeros(nblk * 10, dtype=torch.int64, device=dev)
    for rep in range(2):
        lib.lbm_debug_tile(st.data_ptr())
        rc = lib.lbm_conv1d_pitched_f32(x.data_ptr(), ldx, w.data_ptr(), b.data_ptr(), y.data_ptr(), ldy, E, cin, lin, cout, 4, 2, 1, 1, None)
        assert rc == 0
        torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(-1, 10).astype(np.float64)
    d = s[:, :8]
    ok = (d >= 0).all(axis=1) & (d < 1e7).all(axis=1) & (s[:, 8] > 0)
    d = d[ok]
    n1 = nch - 1
    names = ["prologue (loads of chunk 0, LDS, barrier, loads of chunk 1)", f"mfma half 1 (x{n1})", f"store_chunk: wait for loads + LDS writes (x{n1})",
             f"load_chunk: issue (x{n1})", f"mfma half 2 (x{n1})", f"barrier (x{n1})", "last chunk (all MFMAs)", "epilogue (bias, ReLU, stores, drain)"]
    tot = d.sum(1).mean()
    # MFMA work of a tile at the pipe's rate: per wave KC * 2 * CN MFMAs of 64 cycles per chunk; two workgroup-waves share a SIMD where 2+ WGs are resident
    cn = bm // 64
    mf = nch * 32 * 2 * cn * 64 / 2
    print(f"{name}: blocks {len(d)}, ticks per tile {tot:.0f} (its {nch * 32 * cn} MFMAs per wave = {mf * 2:.0f} cycles of one wave's matrix time); kernel span {s[ok, 9].max() - s[ok, 8].min():.0f}")
    for n, v in zip(names, d.mean(0)):
        print(f"  {n:62s} {v:9.0f}  {100 * v / tot:5.1f} %")
