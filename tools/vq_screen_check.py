#!/usr/bin/env python3
"""Screened codebook search vs the fp32 search: token equality, refined-row fraction and time on one chunk of synthetic epochs
through the shipped encoder (folded and unfolded), plus adversarial inputs (midpoints between codes, duplicated codes)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lbm_amd  # noqa: E402
from lbm_amd import _lib  # noqa: E402
from lbm_amd.synth import synth_epochs, synth_ica  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
with np.load(os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")) as f:
    shipped = {k: f[k] for k in f.files}
taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
A, b = lbm_amd.ica_projection(**synth_ica(30))
x = synth_epochs(E, 30, 1500, seed=11, device=dev, chunk=2048)
ok = True
for fold in (True, False):
    res = {}
    for m in ("fp32", "screen"):
        cfg = lbm_amd.PipelineConfig(sfreq=500.0, n_channels=30, n_times=1500, n_baseline=251, emit_cleaned=False, vq_method=m,
                                     fold_projection=fold)
        pipe = lbm_amd.EpochPipeline(cfg, taps, A, b, shipped, shipped["vq_layer.embedding.weight"])
        pipe.run(x)
        pipe.stage_events = []
        out = pipe.run(x)
        torch.cuda.synchronize()
        ms = {n: s.elapsed_time(e) for n, s, e in pipe.stage_events}
        res[m] = (out["tokens"].clone(), ms["vq"])
        if m == "screen":
            nref = int(pipe._buf["vq_ws"][:4].view(torch.int32).item())
            listed = pipe._buf["vq_ws"][256 + 2048:256 + 2048 + 8 * nref].view(torch.int64).clone()
        del pipe
    same = bool(torch.equal(res["fp32"][0], res["screen"][0]))
    ok &= same
    if not same:
        bad = (res["fp32"][0].flatten() != res["screen"][0].flatten()).nonzero().flatten()
        inl = torch.isin(bad, listed)
        print(f"   mismatching rows in the refine list: {int(inl.sum())} of {bad.numel()}; first rows {bad[:8].tolist()}, "
              f"fp32 {res['fp32'][0].flatten()[bad[:8]].tolist()} screened {res['screen'][0].flatten()[bad[:8]].tolist()}")
    print(f"fold={fold}: tokens equal {same} ({int((res['fp32'][0] != res['screen'][0]).sum())} differ of {res['fp32'][0].numel()}), "
          f"refined {nref} rows = {nref / res['fp32'][0].numel():.4%}, vq {res['fp32'][1]:.3f} ms fp32 vs {res['screen'][1]:.3f} ms screened")

# adversarial rows: exact midpoints of code pairs (+ tiny noise), copies of codes, huge and tiny rows, NaN row
cb = torch.from_numpy(shipped["vq_layer.embedding.weight"]).to(dev)
g = torch.Generator(device=dev).manual_seed(5)
K, D = cb.shape
i0 = torch.randint(0, K, (20000,), device=dev, generator=g)
i1 = torch.randint(0, K, (20000,), device=dev, generator=g)
mid = 0.5 * (cb[i0] + cb[i1])
rows = torch.cat([mid, mid + 1e-6 * torch.randn(mid.shape, device=dev, generator=g), mid * (1 + 1e-7), cb, cb * 1.0000001,
                  1e4 * torch.randn(4096, D, device=dev, generator=g), 1e-4 * torch.randn(4096, D, device=dev, generator=g),
                  torch.zeros(64, D, device=dev)])
rows[7, 3] = float("nan")
cbd = cb.clone()
cbd[100] = cbd[7]            # duplicated codes: the lowest index must win
cbd[300] = cbd[7]
for name, book in (("shipped", cb), ("with duplicates", cbd)):
    tab = lbm_amd.vq_split_tables(book)
    ref = lbm_amd.vq_argmin(rows, book)
    got, nref = lbm_amd.vq_argmin(rows, book, split_tables=tab, screened=True, return_refined=True)
    same = bool(torch.equal(ref, got))
    ok &= same
    print(f"adversarial rows, codebook {name}: equal {same} ({int((ref != got).sum())} differ of {rows.shape[0]}), refined {nref}")
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
