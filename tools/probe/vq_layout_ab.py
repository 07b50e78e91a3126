"""Debug: the screened search (screen + fp32 pass) on the same 6.1 M rows given channel-major (the encoder's [E, 256, S] activations, as the
pipeline hands them over) and row-major ([N, 256]): what a position-major third layer output would buy the quantizer."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lbm_amd
from lbm_amd import _lib
from lbm_amd.quantizer import dedup_codebook, fold_projection, vq_split_tables
dev = torch.device("cuda", 0)
lib, p = _lib.load(), _lib.ptr
E, S, P2, D = 16384, 375, 376, 256
with np.load(os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")) as f:
    W = {k: torch.from_numpy(f[k]).to(dev) for k in f.files}
cbf, eef = fold_projection(W["vq_layer.embedding.weight"], W["encoder.6.weight"], W["encoder.6.bias"])
cb, ee, cmap, kc = dedup_codebook(cbf, eef)
tab = vq_split_tables(cb)
K = cb.shape[0]
# activations that look like the third layer's: run the encoder's first three layers on synthetic epochs
from lbm_amd.synth import synth_epochs
x = torch.empty(E, 30, 1500, device=dev)
synth_epochs(E, 30, 1500, sfreq=500.0, seed=5, device=dev, chunk=2048, out=x)
x = (x - x.mean(-1, keepdim=True)) / (x.std(-1, keepdim=True) + 1e-8)
h = x
for i, (st, pd) in zip((0, 2, 4), ((2, 1), (2, 1), (1, 1))):
    h = lbm_amd.conv1d(h, W[f"encoder.{i}.weight"], W[f"encoder.{i}.bias"], st, pd, True, pitched_out=True)
del x
h3 = h                                               # [E, 256, P2] pitched
rows = h3[:, :, :S].permute(0, 2, 1).contiguous().view(E * S, D)
N = E * S
ws = torch.empty(int(lib.lbm_vq_screen_workspace_bytes(N, K)), dtype=torch.uint8, device=dev)
tok_c = torch.empty(N, dtype=torch.int64, device=dev); tok_r = torch.empty(N, dtype=torch.int64, device=dev)
st = _lib.stream_ptr()
def cm(): return lib.lbm_vq_argmin_screened_f32(p(h3), P2, _lib.VQ_CHANNEL_MAJOR | _lib.VQ_NO_XNORM, S, p(cb), p(tab), p(ee), p(tok_c), None, p(ws), N, D, K, st)
def rm(): return lib.lbm_vq_argmin_screened_f32(p(rows), D, _lib.VQ_ROWS | _lib.VQ_NO_XNORM, S, p(cb), p(tab), p(ee), p(tok_r), None, p(ws), N, D, K, st)
res = {"channel-major": [], "row-major": []}
for rnd in range(3):
    for name, fn in (("channel-major", cm), ("row-major", rm)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for r in range(4):
            e0.record(); assert fn() == 0, _lib.last_error(); e1.record(); torch.cuda.synchronize()
            if rnd and r: res[name].append(e0.elapsed_time(e1))
print("tokens equal:", bool(torch.equal(tok_c, tok_r)), " rows searched in fp32:", int(ws[:4].view(torch.int32).item()))
for k, v in res.items(): print(f"{k:14s} {np.mean(v):.3f} ms (min {min(v):.3f})")
