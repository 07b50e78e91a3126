"""Debug: the quantizer stage of the default pipeline (16 384 epochs) with the histogram taken by the remap pass (default) and by the
search (EpochPipeline.count_in_remap = False), interleaved in one process; tokens and histograms must agree."""
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
pipe = lbm_amd.EpochPipeline(lbm_amd.PipelineConfig(fold_projection=True, vq_method="screen"), taps, A, b, W, W["vq_layer.embedding.weight"], device=dev)
res = {True: [], False: []}
outs = {}
for rnd in range(4):
    for mode in (True, False):
        pipe.count_in_remap = mode
        pipe.reset_hist()
        pipe.stage_events = []
        for _ in range(3):
            o = pipe.run(x)
        torch.cuda.synchronize()
        if rnd:
            res[mode] += [s.elapsed_time(e) for n, s, e in pipe.stage_events if n == "vq"]
        outs[mode] = (o["tokens"].clone(), o["hist"].clone())
        pipe.stage_events = None
print("tokens equal:", bool(torch.equal(outs[True][0], outs[False][0])), " hist equal:", bool(torch.equal(outs[True][1], outs[False][1])),
      " hist sum:", int(outs[True][1].sum()), "=", 3 * E * pipe.S)
for mode in (True, False):
    print(f"count_in_remap={mode}: vq stage {np.mean(res[mode]):.4f} ms (min {min(res[mode]):.4f})")
