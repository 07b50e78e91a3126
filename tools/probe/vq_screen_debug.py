import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lbm_amd
from lbm_amd import _lib
from lbm_amd.synth import synth_epochs, synth_ica
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
E = 4096
with np.load(os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")) as f:
    shipped = {k: f[k] for k in f.files}
taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
A, b = lbm_amd.ica_projection(**synth_ica(30))
x = synth_epochs(E, 30, 1500, seed=11, device=dev, chunk=2048)
cfg = lbm_amd.PipelineConfig(sfreq=500.0, n_channels=30, n_times=1500, n_baseline=251, emit_cleaned=False, fold_projection=True)
pipe = lbm_amd.EpochPipeline(cfg, taps, A, b, shipped, shipped["vq_layer.embedding.weight"])
out = pipe.run(x); torch.cuda.synchronize()
h3 = pipe._buf["h3"][:E]            # [E, 256, 376] pitched
S = pipe.S
cb, ee = pipe.cb_fold, pipe.ee_fold
tab = lbm_amd.vq_split_tables(cb)
z = h3[:, :, :S]
ref = lbm_amd.vq_argmin(z, cb, ee, channel_major=True, no_xnorm=True)
print("ref == pipeline tokens:", bool(torch.equal(ref, out["tokens"].flatten())))
for rep in range(3):
    got, n = lbm_amd.vq_argmin(z, cb, ee, channel_major=True, no_xnorm=True, split_tables=tab, screened=True, return_refined=True)
    bad = (got != ref).nonzero().flatten()
    print(f"rep {rep}: refined {n}, mismatches {bad.numel()}", bad[:10].tolist())
if bad.numel():
    r = int(bad[0]); bq, sp = divmod(r, S)
    xr = h3[bq, :, sp].double().cpu().numpy()
    cbd = cb.double().cpu().numpy(); eed = ee.double().cpu().numpy()
    sc = eed - 2 * cbd @ xr
    k_ref, k_got = int(ref[r]), int(got[r])
    print("row", r, "epoch", bq, "pos", sp, "|x|", np.linalg.norm(xr), "max|x|", np.abs(xr).max(), "finite", np.isfinite(xr).all())
    print("ref", k_ref, sc[k_ref], "got", k_got, sc[k_got], "true argmin", int(sc.argmin()), sc.min(), "|e_ref|", np.linalg.norm(cbd[k_ref]), "|e_got|", np.linalg.norm(cbd[k_got]))
    # rows layout on the same data
    zr = z.permute(0, 2, 1).reshape(-1, 256).contiguous()
    got2 = lbm_amd.vq_argmin(zr, cb, ee, no_xnorm=True, split_tables=tab, screened=True)
    print("rows layout mismatches:", int((got2 != ref).sum()))
    keep = out["keep"].cpu().numpy()
    print("epoch kept:", keep[bq], "bad epochs:", sorted(set((bad // S).tolist()))[:10])
