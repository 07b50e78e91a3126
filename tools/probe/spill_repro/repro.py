"""Spilling vs spill-free build of the first screening kernel (commit c1a6835): tokens against the fp32 search, channel-major."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lbm_amd
from lbm_amd.synth import synth_codebook
dev = "cuda"
with np.load("tests/golden/vq_vae_shipped.npz") as f:
    cb = torch.from_numpy(f["vq_layer.embedding.weight"]).to(dev)
tab = lbm_amd.vq_split_tables(cb)
g = torch.Generator(device=dev).manual_seed(3)
for B in (64, 2048, 8192):
    z = torch.randn(B, 256, 376, device=dev, generator=g)[:, :, :375] * 8.0
    ref = lbm_amd.vq_argmin(z, cb, channel_major=True)
    bad = []
    for rep in range(4):
        got = lbm_amd.vq_argmin(z, cb, channel_major=True, split_tables=tab, screened=True)
        d = (got != ref)
        bad.append(int(d.sum().item()))
        if bad[-1]:
            rows = torch.nonzero(d).flatten()[:2000].cpu().numpy()
            runs = np.split(rows, np.nonzero(np.diff(rows) != 1)[0] + 1)
            print("   first runs of differing rows (start, len, start % 128):", [(int(r[0]), len(r), int(r[0]) % 128) for r in runs[:8]])
    print(os.environ.get("LBM_AMD_LIB"), "B", B, "rows", ref.numel(), "differing per repetition", bad, flush=True)
