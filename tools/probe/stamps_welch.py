"""Debug: phase stamps of welch_fold_kernel (needs tools/probe/liblbm_stamps_w.so)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LBM_AMD_LIB"] = os.path.join(ROOT, "tools", "probe", "liblbm_stamps_w.so")
import lbm_amd
from lbm_amd import _lib
from lbm_amd.psd import welch_bins, welch_tables
from lbm_amd.synth import synth_epochs
lib = _lib.load()
lib.lbm_debug_wstamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
E = 4096
x = synth_epochs(E, 30, 1500, seed=3, device=dev, chunk=2048)
_, lo, hi = welch_bins(500.0, 500, 4.0, 30.0)
tab = welch_tables(dev, 500, 500, "hann", lo, hi)
psd = torch.empty(E, 30, hi - lo + 1, device=dev)
st = torch.zeros(1024 * 5, dtype=torch.int64, device=dev)
for rep in range(2):
    lib.lbm_debug_wstamps(st.data_ptr())
    lib.lbm_welch_f32(x.data_ptr(), psd.data_ptr(), None, E * 30, 1500, 500.0, 500, 500, 250, tab.data_ptr(), lo, hi, 0, None)
    torch.cuda.synchronize()
s = st.cpu().numpy().reshape(-1, 5).astype(np.float64)
d = np.diff(s, axis=1)
ok = (d > 0).all(axis=1) & (d < 1e7).all(axis=1)
d = d[ok]
print(f"groups {len(d)}: stage {d[:,0].mean():.0f}  means {d[:,1].mean():.0f}  fold {d[:,2].mean():.0f}  average {d[:,3].mean():.0f}  total {d.sum(1).mean():.0f} ticks")
