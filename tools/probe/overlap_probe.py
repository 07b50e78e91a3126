"""Do the HBM-bound DSP kernels overlap with the MFMA-bound encoder kernels when issued on two streams?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lbm_amd
from lbm_amd import _lib
from lbm_amd.synth import synth_epochs, synth_ica
dev = torch.device("cuda", 0)
lib, p = _lib.load(), _lib.ptr
E, C, T, S, PS = 16384, 30, 1500, 375, 376
with np.load(os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")) as f:
    w = {k: torch.from_numpy(f[k]).to(dev) for k in f.files}
x = synth_epochs(E, C, T, seed=3, device=dev, chunk=2048)
y = torch.empty_like(x); y2 = torch.empty_like(x); keep = torch.empty(E, dtype=torch.uint8, device=dev)
A, b = lbm_amd.ica_projection(**synth_ica(C))
A = torch.as_tensor(A).to(dev, torch.float32).contiguous(); b = torch.as_tensor(b).to(dev, torch.float32).contiguous()
taps = torch.as_tensor(lbm_amd.create_filter(500.0, 4.0, 30.0)).to(dev, torch.float32)
tab = lbm_amd.filter.fir_fft_tables(taps, T)
h3 = torch.randn(E, 256, PS, device=dev); z = torch.empty(E, 256, PS, device=dev)
cb = w["vq_layer.embedding.weight"].contiguous(); ee = (cb * cb).sum(1).contiguous()
tok = torch.empty(E * S, dtype=torch.int64, device=dev)
from lbm_amd.psd import welch_bins, welch_tables
_, lo, hi = welch_bins(500.0, 500, 4.0, 30.0)
wtab = welch_tables(dev, 500, 500, "hann", lo, hi); psd = torch.empty(E, C, hi - lo + 1, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def mfma(st):
    lib.lbm_conv1d_pitched_f32(p(h3), PS, p(w["encoder.6.weight"]), p(w["encoder.6.bias"]), p(z), PS, E, 256, S, 256, 1, 1, 0, 0, st)
    lib.lbm_vq_argmin_f32(p(z), PS, 1, S, p(cb), p(ee), p(tok), None, None, E * S, 256, 512, st)
def dsp(st, which):
    if "fir" in which: lib.lbm_fir_fft_f32(p(x), p(y), E * C, T, p(tab), taps.numel(), 0, st)
    if "affine" in which: lib.lbm_affine_ch_f32(p(x), p(y), E, C, C, T, p(A), p(b), st)
    if "post" in which: lib.lbm_epoch_post_f32(p(y), p(y), p(y2), p(keep), E, C, T, 251, 150e-6, 1, 1e-6, st)
    if "welch" in which: lib.lbm_welch_f32(p(y), p(psd), None, E * C, T, 500.0, 500, 500, 250, p(wtab), lo, hi, 0, st)
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for which in (["fir"], ["affine", "post"], ["welch"], ["fir", "affine", "post", "welch"]):
    for _ in range(2):
        a = timed(lambda: mfma(s1.cuda_stream)); d = timed(lambda: dsp(s2.cuda_stream, which))
        both = timed(lambda: (mfma(s1.cuda_stream), dsp(s2.cuda_stream, which)))
    print(f"{'+'.join(which):24s} mfma {a:6.2f} ms  dsp {d:6.2f} ms  sum {a + d:6.2f}  two streams {both:6.2f} ms")
