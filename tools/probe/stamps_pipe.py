"""Debug: per-iteration times of vq_screen_pipe_kernel (needs tools/probe/liblbm_stamps_pipe.so)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LBM_AMD_LIB"] = os.path.join(ROOT, "tools", "probe", "liblbm_stamps_pipe%s.so" % (("_" + os.environ["VARIANT"]) if os.environ.get("VARIANT") else ""))
import lbm_amd
from lbm_amd import _lib
dev = torch.device("cuda", 0)
lib, p, st = _lib.load(), _lib.ptr, _lib.stream_ptr()
lib.lbm_debug_pipe.argtypes = [ctypes.c_void_p]
g = torch.Generator(device=dev).manual_seed(1)
E, S, PS, K = 16384, 375, 376, int(os.environ.get("K", "512"))
ROWS = os.environ.get("LAYOUT", "cm") == "rows"
z = torch.randn(E * S, 256, device=dev, generator=g) if ROWS else torch.randn(E, 256, PS, device=dev, generator=g)
cb = torch.randn(K, 256, device=dev, generator=g).contiguous()
ee = lbm_amd.quantizer.rownorm2(cb)
tab = lbm_amd.vq_split_tables(cb)
ws = torch.empty(lib.lbm_vq_screen_workspace_bytes(E * S, K), dtype=torch.uint8, device=dev)
tok = torch.empty(E * S, dtype=torch.int64, device=dev)
stamps = torch.zeros(256 * 384, dtype=torch.int64, device=dev)
for _ in range(2):
    lib.lbm_debug_pipe(stamps.data_ptr())
    rc = lib.lbm_vq_argmin_screened_f32(p(z), 256 if ROWS else PS, _lib.VQ_ROWS if ROWS else _lib.VQ_CHANNEL_MAJOR, S, p(cb), p(tab), p(ee), p(tok), None, p(ws), E * S, 256, K, st)
    assert rc == 0
    torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(256, 2, 64, 3).astype(np.float64)
P = (K + 63) // 64 + 2
for blk in (0, 100):
    print(f"workgroup {blk}, P = {P}: iteration | g0: work, dma wait, barrier | g1: same (ticks of s_memtime, 100 MHz)")
    for i in range(0, 3 * P + 2):
        row = []
        for gg in range(2):
            t0, t1, t2 = s[blk, gg, i]
            nxt = s[blk, gg, i + 1, 0]
            row.append(f"{t1 - t0:7.0f} {t2 - t1:6.0f} {nxt - t2:6.0f}")
        ph0 = i % P
        ph1 = (i - P // 2) % P if i >= P // 2 else -1
        print(f"  i={i:3d} ph0={ph0:2d} ph1={ph1:2d} | {row[0]} | {row[1]}")
it = s[:, 0, 1:61, 0] - s[:, 0, 0:60, 0]
print("mean ticks per iteration by i mod P (group 0's phase):")
for ph in range(P):
    sel = [i for i in range(P, 60) if i % P == ph]
    print(f"  ph0={ph:2d}: {it[:, sel].mean():8.1f}")
print("total per period", sum(it[:, [i for i in range(P, 60) if i % P == ph]].mean() for ph in range(P)))
