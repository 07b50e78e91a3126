"""Do two chunk pipelines on two HIP streams (alternate chunks) finish sooner than one pipeline on one stream?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lbm_amd
from lbm_amd.synth import synth_epochs, synth_ica
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
C, T, fs, chunk = 30, 1500, 500.0, int(os.environ.get("CHUNK", "16384"))
taps = lbm_amd.create_filter(fs, 4.0, 30.0)
A, b = lbm_amd.ica_projection(**synth_ica(C))
with np.load(os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")) as f:
    weights = {k: f[k] for k in f.files}
cfg = lbm_amd.PipelineConfig(sfreq=fs, n_channels=C, n_times=T, n_baseline=int(round(0.5 * fs)) + 1, fold_projection=True, vq_method="screen")
pipes = [lbm_amd.EpochPipeline(cfg, taps, A, b, weights, weights["vq_layer.embedding.weight"], device=dev) for _ in range(2)]
for p in pipes:
    p._reserve(chunk)
x = synth_epochs(2 * chunk, C, T, sfreq=fs, seed=1, device=dev, chunk=2048)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
NCH = 16
def one():
    for i in range(NCH):
        pipes[0].run(x[(i & 1) * chunk:(i & 1) * chunk + chunk])
def two():
    for i in range(NCH):
        with torch.cuda.stream(streams[i & 1]):
            pipes[i & 1].run(x[(i & 1) * chunk:(i & 1) * chunk + chunk])
def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / NCH
ref = {k: v.clone() for k, v in pipes[0].run(x[:chunk]).items()}
torch.cuda.synchronize()
t1 = timed(one); t2 = timed(two); t1b = timed(one); t2b = timed(two)
print(f"chunk {chunk}: one stream {t1:.3f} / {t1b:.3f} ms per chunk, two streams {t2:.3f} / {t2b:.3f} ms per chunk")
with torch.cuda.stream(streams[1]):
    got = pipes[1].run(x[:chunk])
torch.cuda.synchronize()
print("outputs equal:", {k: bool(torch.equal(v, got[k])) for k, v in ref.items()})
