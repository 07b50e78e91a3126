"""Phase breakdown of the bf16x6 VQ kernel (debug build from tools/probe/build_variant.sh with s_memtime stamps)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LBM_AMD_LIB"] = os.path.join(ROOT, "tools", "probe", "liblbm_ph.so")
sys.path.insert(0, ROOT)
import lbm_amd
from lbm_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
E, S, PS = 4096, 375, 376
z = torch.randn(E, 256, PS, device=dev)
cb = torch.randn(512, 256, device=dev)
ee = (cb * cb).sum(1).contiguous()
tab = lbm_amd.vq_split_tables(cb)
tok = torch.empty(E * S, dtype=torch.int64, device=dev)
raw = ctypes.CDLL(os.environ["LBM_AMD_LIB"])
out = (ctypes.c_ulonglong * 8)()
for rep in range(2):
    raw.lbm_debug_phases(out, 1)
    lib.lbm_vq_argmin_bf16x6_f32(z.data_ptr(), PS, 1, S, tab.data_ptr(), ee.data_ptr(), tok.data_ptr(), None, None, E * S, 256, 512, None)
    torch.cuda.synchronize()
    raw.lbm_debug_phases(out, 0)
v = list(out)
waves = v[7]
names = ["top: pins + issue loads", "operand reads + MFMA first half", "split + LDS image writes", "MFMA second half + |e|^2 write",
         "epilogue (amortised)", "vmcnt(0)", "barrier"]
tot = sum(v[:7])
print(f"waves {waves}, stamped ticks per wave {tot / waves:.0f} (s_memtime ticks, 100 MHz)")
for n_, t in zip(names, v[:7]):
    print(f"  {n_:36s} {100.0 * t / tot:5.1f} %")
