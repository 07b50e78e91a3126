"""Debug: phase stamps of welch_team_kernel (needs tools/probe/liblbm_stamps_w.so)."""
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
shape = sys.argv[1] if len(sys.argv) > 1 else "S"
if shape == "S":
    E, C, T, fs, nfft, hop = 16384, 30, 1500, 500.0, 500, 250
else:
    E, C, T, fs, nfft, hop = 1024, 32, 10000, 1000.0, 1000, 500
x = synth_epochs(E, C, T, seed=3, device=dev, chunk=1024)
_, lo, hi = welch_bins(fs, nfft, 4.0, 30.0)
tab = welch_tables(dev, nfft, nfft, "hann", lo, hi)
psd = torch.empty(E, C, hi - lo + 1, device=dev)
st = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
for rep in range(2):
    lib.lbm_debug_wstamps(st.data_ptr())
    lib.lbm_welch_f32(x.data_ptr(), psd.data_ptr(), None, E * C, T, fs, nfft, nfft, hop, tab.data_ptr(), lo, hi, 0, None)
    torch.cuda.synchronize()
s = st.cpu().numpy().reshape(-1, 8).astype(np.float64)
s = s[s[:, 0] > 0]
d = np.diff(s, axis=1)
ok = (d >= 0).all(axis=1) & (d < 1e7).all(axis=1)
d = d[ok]
names = ["vmwait", "barB", "fold", "barC", "part+barD", "sum|dma+barE", "avg+store"]
print(f"blocks {len(d)}: " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, d.mean(0))) + f"  total {d.sum(1).mean():.0f} ticks")
