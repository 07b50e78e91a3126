"""Child process of tests/test_gpu_rccl_world1.py (not collected by pytest): brings up a world-size-1 RCCL group on cuda:0 and
drives every collective of the path through it -- the code that runs at N > 1 (BASELINE config 4), on the one GPU a test box has.
Prints one JSON object {check name: bool}.  Order to reproduce: np.concatenate of per-batch lists, training/model.py:423-424."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import numpy as np
    import torch
    import torch.distributed as dist

    import bench
    import lbm_amd
    from lbm_amd.epochs import Scaler
    from lbm_amd.pipeline import allreduce_hist, gather_epoch_outputs, gather_pipeline_outputs
    from lbm_amd.synth import synth_epochs, synth_ica

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    res = {"backend_is_nccl": dist.get_backend() == "nccl"}

    # ---- a real job: 96 epochs in 6 chunks of 16 through the pipeline, outputs exchanged in ragged groups while later chunks compute
    with np.load(os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")) as f:
        shipped = {k: f[k] for k in f.files}
    taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
    A, b = lbm_amd.ica_projection(**synth_ica(30))
    cfg = lbm_amd.PipelineConfig(sfreq=500.0, n_channels=30, n_times=1500, n_baseline=251, emit_cleaned=False, fold_projection=True,
                                 vq_method="screen")
    pipe = lbm_amd.EpochPipeline(cfg, taps, A, b, shipped, shipped["vq_layer.embedding.weight"], device=dev)
    E, chunk = 96, 16
    x = synth_epochs(E, 30, 1500, seed=41, device=dev)
    x *= torch.linspace(0.5, 24.0, E, device=dev).view(E, 1, 1)         # a range of amplitudes: some epochs fail the 150 uV test
    tokens = torch.empty((E, pipe.S), dtype=torch.int16, device=dev)
    psd = torch.empty((E, 30, pipe.F), dtype=torch.float32, device=dev)
    bp = torch.empty((E, 6), dtype=torch.float32, device=dev)
    keep = torch.empty((E,), dtype=torch.uint8, device=dev)
    xch = bench.Exchange(dist, "nccl", 1, {"tokens": tokens, "psd": psd, "bandpower": bp, "keep": keep}, dev)
    group_end = {0: None, 3: None, 4: None, 5: None}                    # groups of 1, 3, 1 and 1 chunks
    pipe.reset_hist()
    g_lo = 0
    for i in range(E // chunk):
        lo = i * chunk
        out = pipe.run(x[lo:lo + chunk])
        tokens[lo:lo + chunk].copy_(out["tokens"])
        psd[lo:lo + chunk].copy_(out["psd"])
        bp[lo:lo + chunk].copy_(out["bandpower"])
        keep[lo:lo + chunk].copy_(out["keep"])
        if i in group_end:
            xch.gather_group(g_lo, lo + chunk)
            g_lo = lo + chunk
    hist = pipe._buf["hist"]
    hist_before = hist.clone()
    xch.finish(hist)
    torch.cuda.synchronize()
    res["exchange_order"] = all(bool(torch.equal(xch.gathered(k), v)) and xch.gathered(k).dtype == v.dtype for k, v in xch.local.items())
    res["exchange_hist"] = bool(torch.equal(hist, hist_before)) and int(hist.sum().item()) == E * pipe.S
    res["exchange_tokens_are_histogrammed"] = bool(torch.equal(torch.bincount(tokens.reshape(-1).long(), minlength=512), hist))
    n_kept = int(keep.sum().item())
    res["some_epochs_rejected"] = 0 < n_kept < E

    # ---- the module-level helpers on the same outputs (the short cut at world size 1 switched off)
    outs = {"tokens": tokens.long(), "psd": psd, "bandpower": bp, "keep": keep}
    eq = gather_pipeline_outputs(outs, force=True)                      # one all_gather_into_tensor per array
    res["gather_equal"] = all(bool(torch.equal(eq[k], outs[k])) for k in outs)
    pad = gather_pipeline_outputs(outs, force="padded")                 # counts first, then padded blocks
    res["gather_padded"] = all(bool(torch.equal(pad[k], outs[k])) for k in outs)
    sel = keep.to(torch.bool)
    dropped = gather_pipeline_outputs(outs, drop_rejected=True, force=True)   # epochs.drop_bad: shards thinned by the keep mask
    res["gather_drop_rejected"] = all(bool(torch.equal(dropped[k], outs[k][sel])) for k in outs) and dropped["tokens"].shape[0] == n_kept
    res["gather_empty_shard"] = gather_epoch_outputs(tokens[:0].long(), force=True).shape == (0, pipe.S)
    h2 = allreduce_hist(hist.clone(), force=True)
    res["allreduce_hist"] = bool(torch.equal(h2, hist))

    # ---- Scaler statistics all-reduced (multi-train/withpsd.py:267-270 over a sharded file set)
    xs = synth_epochs(32, 30, 1500, seed=43, device=dev)
    a, b_ = Scaler().fit(xs), Scaler().fit(xs, distributed=True, force=True)
    res["scaler_allreduce"] = bool(torch.equal(a.mean_, b_.mean_) and torch.equal(a.scale_, b_.scale_))

    # ---- EMA training step with the (counts, dw) all-reduce (training/model.py:103-126)
    from lbm_amd.quantizer import VectorQuantizerEMA
    g = torch.Generator().manual_seed(7)
    w0 = torch.randn(64, 32, generator=g)
    zin = torch.randn(4, 50, 32, generator=g).to(dev)
    outs_ema = []
    for grp in (None, dist.group.WORLD):
        vq = VectorQuantizerEMA(64, 32, 0.25, 0.99).to(dev)
        with torch.no_grad():
            vq.embedding.weight.copy_(w0)
            vq.ema_w.copy_(w0)
        if grp is not None:
            vq.ema_group = grp
        vq.train()
        loss, q, ppl, idx = vq(zin)
        outs_ema.append((vq.embedding.weight.detach().clone(), vq.ema_cluster_size.clone(), idx.clone(), float(loss)))
    res["ema_allreduce"] = all(bool(torch.equal(u, v)) for u, v in zip(outs_ema[0][:3], outs_ema[1][:3])) and outs_ema[0][3] == outs_ema[1][3]

    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
