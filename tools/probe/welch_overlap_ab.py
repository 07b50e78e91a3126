"""Debug: one 16 384-epoch chunk of the default pipeline with the PSD branch (Welch + band power) on the launch stream (default) and on a
second stream beside the encoder layers (PipelineConfig.overlap_welch), interleaved in one process; outputs must agree."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lbm_amd
from lbm_amd.synth import synth_epochs, synth_ica
dev = torch.device("cuda", 0)
E = 16384
with np.load(os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")) as f:
    W = {k: f[k] for k in f.files}
taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
A, b = lbm_amd.ica_projection(**synth_ica(30))
x = torch.empty(E, 30, 1500, device=dev)
synth_epochs(E, 30, 1500, sfreq=500.0, seed=5, device=dev, chunk=2048, out=x)
pipes = {m: lbm_amd.EpochPipeline(lbm_amd.PipelineConfig(fold_projection=True, vq_method="screen", overlap_welch=m), taps, A, b, W,
                                  W["vq_layer.embedding.weight"], device=dev) for m in (False, True)}
pipes[True]._buf, pipes[True]._cap = None, 0
pipes[False]._reserve(E)
pipes[True]._buf, pipes[True]._cap = {**pipes[False]._buf}, E
res = {False: [], True: []}
outs = {}
for rnd in range(4):
    for m in (False, True):
        p = pipes[m]
        p.reset_hist()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for r in range(4):
            e0.record(); o = p.run(x); e1.record(); torch.cuda.synchronize()
            if rnd and r: res[m].append(e0.elapsed_time(e1))
        outs[m] = {k: o[k].clone() for k in ("tokens", "psd", "bandpower", "keep")}
print("outputs equal:", all(bool(torch.equal(outs[False][k], outs[True][k])) for k in outs[False]))
for m in (False, True):
    print(f"overlap_welch={m}: chunk {np.mean(res[m]):.3f} ms (min {min(res[m]):.3f})")
