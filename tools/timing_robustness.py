#!/usr/bin/env python3
"""Checksums of the MFMA kernels' outputs on bench-sized inputs.  Run once with the product library and once per timing-perturbed
variant (tools/perturb_build.py, LBM_AMD_LIB=...): the kernels are deterministic, so any differing checksum is a timing-dependent
result, i.e. a race.   python tools/timing_robustness.py [epochs] > sums.json"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lbm_amd  # noqa: E402
from lbm_amd.synth import synth_epochs, synth_ica  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
with np.load(os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")) as f:
    shipped = {k: f[k] for k in f.files}
taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
A, b = lbm_amd.ica_projection(**synth_ica(30))
x = synth_epochs(E, 30, 1500, seed=11, device=dev, chunk=2048)
sums = {}


def digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


for name, kw in (("fp32", dict()), ("fp32_fold", dict(fold_projection=True)), ("screen_fold", dict(vq_method="screen", fold_projection=True)),
                 ("screen", dict(vq_method="screen")), ("bf16x6", dict(vq_method="bf16x6")),
                 ("bf16x6_conv", dict(vq_method="bf16x6", conv_method="bf16x6")),
                 ("bf16x6_conv_fold", dict(vq_method="bf16x6", conv_method="bf16x6", fold_projection=True))):
    cfg = lbm_amd.PipelineConfig(sfreq=500.0, n_channels=30, n_times=1500, n_baseline=251, emit_cleaned=False, **kw)
    pipe = lbm_amd.EpochPipeline(cfg, taps, A, b, shipped, shipped["vq_layer.embedding.weight"])
    for rep in range(2):
        pipe.reset_hist()
        out = pipe.run(x)
        torch.cuda.synchronize()
        for k in ("tokens", "psd", "bandpower", "keep", "standardized") + (("z",) if "z" in out else ()):
            sums[f"{name}.{k}.rep{rep}"] = digest(out[k])
        sums[f"{name}.h3.rep{rep}"] = digest(pipe._buf["h3"][:E])
    del pipe
    torch.cuda.empty_cache()
print(json.dumps(sums, indent=0, sort_keys=True))
