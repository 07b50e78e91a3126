#!/usr/bin/env python3
"""Benchmark of the batched EEG epoch pipeline on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (FIR band-pass -> ICA affine -> baseline/reject/z-score ->
Welch PSD -> band powers -> VQ-VAE encoder -> codebook argmin + usage histogram) over BASELINE.json
config 3: 1 000 000 synthetic 30-channel x 500 Hz x 3 s fp32 epochs per GPU, RESIDENT in HBM (180 GB) before
the timed region, walked in chunks of 16 384 epochs (the pipeline's stage buffers are sized for one chunk).
Per-epoch outputs (tokens, PSD, band features, keep mask) are kept for the whole job.
For N > 1 (torchrun, one rank per GPU) every rank processes its own 1 M epochs (weak scaling: config 4, epochs
shard trivially); the only collectives of the path -- all-gather of the token array and all-reduce of the
codebook usage histogram -- run once per step inside the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import lbm_amd  # noqa: E402
from lbm_amd.synth import synth_codebook, synth_epochs, synth_ica  # noqa: E402

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
PEAK_HBM_GBS = 8000.0
KERNEL_OF_STAGE = {"fir": "fir_fft_kernel", "affine": "affine_kernel", "post": "post_kernel", "welch": "welch_fold_kernel",
                   "bandpower": "bandpower_kernel", "conv1": "conv_tile_kernel<4,2,64>", "conv2": "conv_tile_kernel<4,2,128>",
                   "conv3": "conv_wino3_kernel", "conv4": "conv_tile_kernel<1,1,256>", "vq": "vq_argmin_dma_kernel"}


def stage_model(E, C, T, ntaps, K, D, S, L1, F, fft_n=2048):
    """Algorithmic flops / compulsory HBM bytes per launch of each stage over E epochs (SURVEY.md 8d figures)."""
    ep = C * T * 4
    fft_flops = 2 * 5.0 * fft_n * np.log2(fft_n) + 6.0 * fft_n       # forward + inverse complex FFT + spectrum product, per row
    return {
        "fir": dict(flops=fft_flops * C * E, direct_flops=2.0 * ntaps * C * T * E, bytes=2.0 * ep * E, bound="hbm"),
        "affine": dict(flops=2.0 * C * C * T * E, bytes=2.0 * ep * E, bound="hbm"),
        "post": dict(flops=8.0 * C * T * E, bytes=3.0 * ep * E, bound="hbm"),
        "welch": dict(flops=2.0e6 * E * (C * T / 45000.0), bytes=(ep + C * F * 4) * E, bound="hbm"),
        "bandpower": dict(flops=0.0, bytes=(C * F * 4 + 24) * E, bound="hbm"),
        "conv1": dict(flops=2.0 * 64 * C * 4 * L1 * E, bytes=(ep + 64 * L1 * 4) * E, bound="mfma"),
        "conv2": dict(flops=2.0 * 128 * 64 * 4 * S * E, bytes=(64 * L1 * 4 + 128 * S * 4) * E, bound="mfma"),
        "conv3": dict(flops=2.0 * 256 * 128 * 3 * S * E, bytes=(128 * S * 4 + 256 * S * 4) * E, bound="mfma"),
        "conv4": dict(flops=2.0 * D * 256 * S * E, bytes=(256 * S * 4 + D * S * 4) * E, bound="mfma"),
        "vq": dict(flops=2.0 * S * K * D * E, bytes=(D * S * 4 + S * 8) * E, bound="mfma"),
    }


def cpu_baseline(taps, A, b, weights, sample_epochs):
    """Times the CPU oracle (float64 SciPy DSP + torch-CPU fp32 encoder/quantizer) on a bounded sample."""
    from oracle import eeg_oracle as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    x = synth_epochs(sample_epochs, 30, 1500, seed=99).numpy()
    t0 = time.perf_counter()
    for lo in range(0, sample_epochs, 256):
        orc.pipeline(x[lo:lo + 256], 500.0, taps, A, b, 251, 150e-6, weights, fft_filter=True)
    dt = time.perf_counter() - t0
    # per-stage rates of the same restatement on one 256-epoch block (beside the GPU stage numbers)
    n = min(256, sample_epochs)
    xs = np.asarray(x[:n], dtype=np.float64)
    per = {}

    def timed(name, fn):
        t = time.perf_counter()
        r = fn()
        per[name] = round(n / (time.perf_counter() - t), 1)
        return r
    y = timed("fir", lambda: orc.fir_zero_phase(xs, taps, fft=True))
    y = timed("affine", lambda: np.einsum("oc,ect->eot", np.asarray(A, dtype=np.float64), y) + np.asarray(b, dtype=np.float64)[None, :, None])
    std = timed("post", lambda: (orc.ptp_keep_mask(orc.baseline_correct(y, 251), 150e-6), orc.zscore_epoch_channel(y, 1e-6))[1])
    timed("welch", lambda: orc.welch_psd(y, 500.0, 4.0, 30.0, n_fft=500, n_overlap=250, window="hann"))
    h = timed("encoder", lambda: orc.encoder_forward(std.astype(np.float32), weights))
    flat = np.ascontiguousarray(np.transpose(np.asarray(h), (0, 2, 1))).reshape(-1, np.asarray(h).shape[1])
    timed("vq", lambda: orc.vq_argmin(flat, weights["vq_layer.embedding.weight"]))
    return {"value": round(sample_epochs / dt, 1), "unit": "epochs/s", "cores": cores, "kind": "port",
            "sample": f"{sample_epochs} synthetic 30x1500 epochs through oracle/eeg_oracle.pipeline "
                      f"(float64 scipy FIR(oaconvolve)/ICA/Welch + torch-CPU fp32 encoder+VQ), {dt:.1f} s",
            "stage_epochs_per_s": per}


def alloc_resident(E, C, T, dev, reserve_bytes):
    """[R, C, T] input with R = E if it fits next to `reserve_bytes` of other buffers, else the largest halving that does."""
    free, _ = torch.cuda.mem_get_info(dev)
    R = E
    while R > 1 and R * C * T * 4 + reserve_bytes > free:
        R = (R + 1) // 2
    return torch.empty((R, C, T), dtype=torch.float32, device=dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--epochs", type=int, default=1_000_000, help="epochs per step per GPU (BASELINE config 3: 1M)")
    ap.add_argument("--chunk", type=int, default=16384, help="epochs per pipeline launch")
    ap.add_argument("--codebook", type=int, default=512)
    ap.add_argument("--cpu-sample", type=int, default=2048, help="epochs for the CPU baseline (0 = skip)")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--from-host", action="store_true",
                    help="non-default: every chunk arrives from pinned host memory (PCIe-inclusive rate; H2D copies of chunk i+1 "
                         "on a copy stream overlap the kernels of chunk i)")
    ap.add_argument("--emit-cleaned", action="store_true",
                    help="also write the baseline-corrected epochs (+180 000 B/epoch; not among SURVEY.md 8(d)'s compulsory outputs)")
    ap.add_argument("--vq", choices=["fp32", "bf16x6"], default="fp32",
                    help="codebook argmin kernel: fp32 MFMA (default) or fp32 products from three bf16 pieces on the bf16 MFMA")
    ap.add_argument("--conv", choices=["fp32", "bf16x6"], default="fp32",
                    help="256-channel encoder layers: fp32 MFMA kernels (default) or the three-bf16-piece implicit GEMM")
    ap.add_argument("--fold-proj", action="store_true",
                    help="non-default variant: tokens from the 3rd encoder layer with the last 1x1 conv folded into the codebook "
                         "(PipelineConfig.fold_projection; no z tensor; equal tokens except fp32 near-ties)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL over xGMI (the measured path); gloo = rehearsal of the N > 1 control flow on a box "
                         "with fewer GPUs than ranks (ranks share devices, collectives staged through host memory)")
    ap.add_argument("--shape", choices=["S", "L"], default="S",
                    help="S = 30ch x 500Hz x 3s, K=512 (configs 3/4, the metric's workload); L = 64ch x 1kHz x 10s, "
                         "K=8192, 16384 epochs (config 5, for measurement only)")
    args = ap.parse_args()
    if args.shape == "L":
        if args.epochs == 1_000_000:
            args.epochs = 16384
        if args.codebook == 512:
            args.codebook = 8192
        args.chunk = min(args.chunk, 1024)
        args.cpu_sample = 0

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    C, T, fs, K, D = (30, 1500, 500.0, args.codebook, 256) if args.shape == "S" else (64, 10000, 1000.0, args.codebook, 256)
    E, chunk = args.epochs, min(args.chunk, args.epochs)
    taps = lbm_amd.create_filter(fs, 4.0, 30.0)
    A, b = lbm_amd.ica_projection(**synth_ica(C))
    if args.shape == "S":
        with np.load(os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")) as f:
            weights = {k: f[k] for k in f.files}          # shipped encoder (C = 30)
    else:
        from lbm_amd.synth import synth_encoder
        weights = synth_encoder(C)
    codebook = weights["vq_layer.embedding.weight"] if (K == 512 and args.shape == "S") else synth_codebook(K, D)
    cfg = lbm_amd.PipelineConfig(sfreq=fs, n_channels=C, n_times=T, n_baseline=int(round(0.5 * fs)) + 1,
                                 fold_projection=args.fold_proj, vq_method=args.vq, conv_method=args.conv,
                                 emit_cleaned=args.emit_cleaned)
    pipe = lbm_amd.EpochPipeline(cfg, taps, A, b, weights, codebook, device=dev)
    pipe._reserve(chunk)
    S, F = pipe.S, pipe.F

    # job-level outputs and the resident input
    tok_dtype = torch.int16 if K < 32768 else torch.int32     # wire/storage width of the tokens; int64 at the API
    tokens = torch.empty((E, S), dtype=tok_dtype, device=dev)
    psd = torch.empty((E, C, F), dtype=torch.float32, device=dev)
    bp = torch.empty((E, 6), dtype=torch.float32, device=dev)
    keep = torch.empty((E,), dtype=torch.uint8, device=dev)
    gathered = torch.empty((world * E, S), dtype=tok_dtype, device=dev) if (world > 1 and not args.no_gather) else None
    x = alloc_resident(E, C, T, dev, reserve_bytes=6 << 30)
    R = x.shape[0]
    synth_epochs(R, C, T, sfreq=fs, seed=lbm_amd.synth.SEED + rank, device=dev, chunk=2048, out=x)
    torch.cuda.synchronize()

    host_ring = dev_ring = copy_stream = None
    if args.from_host:
        copy_stream = torch.cuda.Stream()
        host_ring = [x[:chunk].cpu().pin_memory() for _ in range(2)]
        dev_ring = [torch.empty((chunk, C, T), dtype=torch.float32, device=dev) for _ in range(2)]

    def step_from_host():
        pipe.reset_hist()
        nchunks = (E + chunk - 1) // chunk
        ready = [torch.cuda.Event() for _ in range(2)]
        freed = [torch.cuda.Event() for _ in range(2)]
        with torch.cuda.stream(copy_stream):
            dev_ring[0].copy_(host_ring[0], non_blocking=True)
            ready[0].record(copy_stream)
        for i in range(nchunks):
            cur = i & 1
            if i + 1 < nchunks:
                with torch.cuda.stream(copy_stream):
                    if i >= 1:
                        copy_stream.wait_event(freed[cur ^ 1])          # the kernels of chunk i-1 have consumed that buffer
                    dev_ring[cur ^ 1].copy_(host_ring[cur ^ 1], non_blocking=True)
                    ready[cur ^ 1].record(copy_stream)
            torch.cuda.current_stream().wait_event(ready[cur])
            lo = i * chunk
            n = min(chunk, E - lo)
            out = pipe.run(dev_ring[cur][:n])
            freed[cur].record()
            tokens[lo:lo + n].copy_(out["tokens"])
            psd[lo:lo + n].copy_(out["psd"])
            bp[lo:lo + n].copy_(out["bandpower"])
            keep[lo:lo + n].copy_(out["keep"])
        return pipe._buf["hist"]

    def step():
        if args.from_host:
            return step_from_host()
        pipe.reset_hist()
        for lo in range(0, E, chunk):
            n = min(chunk, E - lo)
            src = lo % R
            if src + n > R:
                src = R - n
            out = pipe.run(x[src:src + n])
            tokens[lo:lo + n].copy_(out["tokens"])
            psd[lo:lo + n].copy_(out["psd"])
            bp[lo:lo + n].copy_(out["bandpower"])
            keep[lo:lo + n].copy_(out["keep"])
        hist = pipe._buf["hist"]
        if gathered is not None and args.backend == "nccl":
            dist.all_gather_into_tensor(gathered.view(torch.uint8), tokens.view(torch.uint8))   # RCCL has no int16: bytes
            dist.all_reduce(hist)
        elif gathered is not None:                      # gloo rehearsal: same calls on host copies
            g_host = torch.empty(gathered.shape, dtype=gathered.dtype).view(torch.uint8)
            dist.all_gather_into_tensor(g_host, tokens.cpu().view(torch.uint8))
            gathered.view(torch.uint8).copy_(g_host)
            h_host = hist.cpu()
            dist.all_reduce(h_host)
            hist.copy_(h_host)
        return hist

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # the exchange step alone ("report gather time separately"), outside the timed region
    gather_ms = None
    if gathered is not None and args.backend == "nccl":
        torch.cuda.synchronize()
        dist.barrier()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record()
        dist.all_gather_into_tensor(gathered.view(torch.uint8), tokens.view(torch.uint8))
        dist.all_reduce(pipe._buf["hist"])
        g1.record()
        torch.cuda.synchronize()
        gather_ms = round(g0.elapsed_time(g1), 3)

    # per-kernel durations from HIP events recorded on the launch stream (one chunk, outside the timed region)
    stages = {}
    if rank == 0:
        reps = 5
        pipe.stage_events = []
        for _ in range(reps):
            pipe.reset_hist()
            pipe.run(x[:chunk])
        torch.cuda.synchronize()
        acc = {}
        for name, s_, e_ in pipe.stage_events:
            acc.setdefault(name, []).append(s_.elapsed_time(e_))
        pipe.stage_events = None
        fft_n = 2048 if args.shape == "S" else 3 * 4096
        model = stage_model(chunk, C, T, len(taps), K, D, S, pipe.L1, F, fft_n)
        if not args.emit_cleaned:
            model["post"]["bytes"] = 2.0 * C * T * 4 * chunk      # one read, one write (standardized)
        for name, ms in acc.items():
            m = float(np.mean(ms))
            stages[name] = {"ms": round(m, 4), "bound": model[name]["bound"],
                            "tflops": round(model[name]["flops"] / (m * 1e-3) / 1e12, 3),
                            "gbs": round(model[name]["bytes"] / (m * 1e-3) / 1e9, 1)}

    # non-default variants of the tokenizer tail, same chunk, outside the timed region (reported, never `value`)
    variants = {}
    if rank == 0 and args.shape == "S" and not args.fold_proj and args.vq == "fp32" and args.conv == "fp32":
        base_ms = sum(v["ms"] for v in stages.values())
        for name, kw in (("vq_bf16x6", dict(vq_method="bf16x6")),
                         ("vq_bf16x6+conv_bf16x6", dict(vq_method="bf16x6", conv_method="bf16x6")),
                         ("vq_bf16x6+fold_projection", dict(vq_method="bf16x6", fold_projection=True)),
                         ("vq_bf16x6+conv_bf16x6+fold_projection", dict(vq_method="bf16x6", conv_method="bf16x6", fold_projection=True))):
            vcfg = lbm_amd.PipelineConfig(sfreq=fs, n_channels=C, n_times=T, n_baseline=int(round(0.5 * fs)) + 1, emit_cleaned=args.emit_cleaned, **kw)
            vp = lbm_amd.EpochPipeline(vcfg, taps, A, b, weights, codebook, device=dev)
            vp._buf, vp._cap = {**pipe._buf}, pipe._cap          # share the stage buffers (z is unused when folded)
            vp.run(x[:chunk])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                vp.run(x[:chunk])
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            variants[name] = {"ms_per_chunk": round(ms, 3), "epochs_per_s": round(chunk / (ms * 1e-3), 1),
                              "default_ms_per_chunk": round(base_ms, 3)}

    if rank == 0:
        dom = max(stages, key=lambda k: stages[k]["ms"])      # the dominant single kernel of a chunk
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # HBM bytes per launch from rocprofv3 --pmc passes
        if os.path.exists(tpath) and args.shape == "S" and chunk == 16384 and K == 512:   # counters were taken on that launch
            with open(tpath) as f:
                traffic = json.load(f).get(dom, {}).get("hbm_bytes_per_launch")
        m = model[dom]
        if m["bound"] == "mfma":
            roof = {"kernel": KERNEL_OF_STAGE[dom], "bound": "mfma", "achieved": stages[dom]["tflops"], "peak": PEAK_FP32_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(stages[dom]["tflops"] / PEAK_FP32_TFLOPS, 4), "traffic": traffic}
        else:
            roof = {"kernel": KERNEL_OF_STAGE[dom], "bound": "hbm", "achieved": stages[dom]["gbs"], "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": round(stages[dom]["gbs"] / PEAK_HBM_GBS, 4), "traffic": traffic}
        roof["launch"] = f"{chunk} epochs, {stages[dom]['ms']} ms"
        cpu = None
        if world == 1 and args.cpu_sample > 0:
            cpu = cpu_baseline(taps, A, b, weights, args.cpu_sample)
        line = {
            "metric": "EEG epochs/sec (filter+ICA+PSD+VQ tokenize)",
            "value": round(world * E * args.steps / dt, 1),
            "unit": "epochs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" + (" (chunks streamed from pinned host memory)" if args.from_host else ""),
            "config": {"workload": "%s: full pipeline (%d-tap FIR + ICA affine + baseline/reject/z-score + Welch PSD "
                                   "+ band power + VQ-VAE encoder + K=%d codebook argmin), %dch x %gHz x %gs fp32 epochs, "
                                   "%d epochs per step per GPU resident in HBM, walked in chunks of %d"
                                   % ("config3" if args.shape == "S" else "config5", len(taps), K, C, fs, T / fs, E, chunk),
                       "epochs_per_step_per_gpu": E, "resident_epochs_per_gpu": R, "chunk": chunk, "channels": C, "samples": T,
                       "codebook": K, "fold_projection": bool(args.fold_proj), "vq_method": args.vq, "conv_method": args.conv, "emit_cleaned": bool(args.emit_cleaned), "input": "host (PCIe-inclusive)" if args.from_host else "resident in HBM",
                       "outputs": "tokens + PSD + band features + keep mask" + (" + cleaned epochs" if args.emit_cleaned else ""),
                       "collectives": "all_gather(tokens,int16)+all_reduce(hist)" if gathered is not None else "none"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "stages": stages,
            "gather_ms_per_step": gather_ms,
            "variants": variants,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()   # rank 0 measured the per-stage events and variants alone: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
