"""Debug: phase stamps of fir_fft_kernel (needs tools/probe/liblbm_stamps_f.so from build_stamps_fir.sh)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["LBM_AMD_LIB"] = os.path.join(ROOT, "tools", "probe", "liblbm_stamps_f.so")
import lbm_amd
from lbm_amd import _lib
from lbm_amd.synth import synth_epochs
lib = _lib.load()
lib.lbm_debug_fstamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
E, C, T, fs = 16384, 30, 1500, 500.0
x = synth_epochs(E, C, T, seed=3, device=dev, chunk=1024)
taps = lbm_amd.create_filter(fs, 4.0, 30.0)
y = torch.empty_like(x)
st = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
from lbm_amd import filter as F
tp = torch.as_tensor(np.asarray(taps, dtype=np.float32), device=dev)
tab = F.fir_fft_tables(tp, T)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    lib.lbm_debug_fstamps(st.data_ptr())
    e0.record()
    assert lib.lbm_fir_fft_f32(x.data_ptr(), y.data_ptr(), E * C, T, tab.data_ptr(), len(taps), 0, None) == 0
    e1.record()
    torch.cuda.synchronize()
print("launch %.3f ms" % e0.elapsed_time(e1))
s = st.cpu().numpy().reshape(-1, 16)[:, :15].astype(np.float64)
s = s[s[:, 0] > 0]
d = np.diff(s, axis=1)
ok = (d >= 0).all(axis=1) & (d < 1e7).all(axis=1)
d = d[ok]
names = ["stage+pad", "inread+bar", "F1 valu+wr", "bar", "F2 rd+bar", "F2 valu+wr", "bar", "F3 all", "bar", "I2 rd+bar", "I2 valu+wr", "bar", "I1 rd+valu", "stores"]
print(f"workgroups {len(d)}")
for n, v, sd in zip(names, d.mean(0), d.std(0)):
    print(f"  {n:12s} {v:8.0f}  +-{sd:6.0f}")
print(f"  total        {d.sum(1).mean():8.0f} ticks per iteration (2 items)")
