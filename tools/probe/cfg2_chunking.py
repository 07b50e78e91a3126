import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
import lbm_amd, bench
from lbm_amd.synth import synth_epochs
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
for E2, ch2 in ((100_000, 16384), (100_000, 100_000), (100_000, 50_000)):
    cfg2 = lbm_amd.PipelineConfig(sfreq=500.0, n_channels=30, n_times=1500, n_baseline=251, do_tokenize=False, do_filter=False, do_ica=False, emit_cleaned=False)
    p2 = lbm_amd.EpochPipeline(cfg2, None, None, None, None, None, device=dev)
    x2 = synth_epochs(ch2, 30, 1500, seed=2, device=dev, chunk=2048)
    p2.stage_events = []
    ms2 = bench._timed_chunks(p2, x2, E2, ch2, reps=3)
    acc = {}
    for name, s_, e_ in p2.stage_events: acc.setdefault(name, []).append(s_.elapsed_time(e_))
    by = 183_264.0 * E2
    print(ch2, round(ms2, 3), "ms", round(E2 / ms2 / 1e3, 2), "M epochs/s", round(by / (ms2 * 1e-3) / 8e12, 4), {k: round(float(np.mean(v)), 4) for k, v in acc.items()}, torch.cuda.max_memory_allocated() / 1e9, flush=True)
    del p2, x2; torch.cuda.empty_cache()
