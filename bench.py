#!/usr/bin/env python3
"""Benchmark of the batched EEG epoch pipeline on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (FIR band-pass -> ICA affine -> baseline/reject/z-score ->
Welch PSD -> band powers -> VQ-VAE encoder -> codebook argmin + usage histogram) over one resident chunk
of synthetic 30-channel x 500 Hz x 3 s fp32 epochs (BASELINE.json config 3, streamed in chunks).
For N > 1 (torchrun, one rank per GPU) every rank processes its own chunk (weak scaling, epochs shard
trivially); the only collectives of the path -- all-gather of the token array and all-reduce of the
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
from lbm_amd.synth import synth_codebook, synth_encoder, synth_epochs, synth_ica  # noqa: E402

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
PEAK_HBM_GBS = 8000.0


def stage_model(E, C, T, ntaps, K, D, S, L1, F):
    """Algorithmic flops / bytes per step of each stage (SURVEY.md 8d figures x epochs in the chunk)."""
    ep = C * T * 4
    enc_flops = 2 * (64 * C * 4 * L1 + 128 * 64 * 4 * S + 256 * 128 * 3 * S + D * 256 * S)
    return {
        "fir": dict(flops=2.0 * ntaps * C * T * E, bytes=2.0 * ep * E, bound="mfma"),
        "affine": dict(flops=2.0 * C * C * T * E, bytes=2.0 * ep * E, bound="hbm"),
        "post": dict(flops=8.0 * C * T * E, bytes=3.0 * ep * E, bound="hbm"),
        "welch": dict(flops=2.0e6 * E * (C * T / 45000.0), bytes=(ep + C * F * 4) * E, bound="hbm"),
        "bandpower": dict(flops=0.0, bytes=(C * F * 4 + 24) * E, bound="hbm"),
        "encoder": dict(flops=float(enc_flops) * E, bytes=(ep + D * S * 4) * E, bound="mfma"),
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
    orc.pipeline(x, 500.0, taps, A, b, 251, 150e-6, weights, fft_filter=True)
    dt = time.perf_counter() - t0
    return {"value": sample_epochs / dt, "unit": "epochs/s", "cores": cores, "kind": "port",
            "sample": f"{sample_epochs} synthetic 30x1500 epochs through oracle/eeg_oracle.pipeline "
                      f"(float64 scipy FIR(oaconvolve)/ICA/Welch + torch-CPU fp32 encoder+VQ), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chunk", type=int, default=16384, help="epochs per step per GPU")
    ap.add_argument("--codebook", type=int, default=512)
    ap.add_argument("--cpu-sample", type=int, default=256, help="epochs for the CPU baseline (0 = skip)")
    ap.add_argument("--no-gather", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    C, T, fs, K, D = 30, 1500, 500.0, args.codebook, 256
    taps = lbm_amd.create_filter(fs, 4.0, 30.0)
    A, b = lbm_amd.ica_projection(**synth_ica(C))
    shipped_path = os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")
    with np.load(shipped_path) as f:
        weights = {k: f[k] for k in f.files}          # shipped encoder (C = 30)
    codebook = weights["vq_layer.embedding.weight"] if K == 512 else synth_codebook(K, D)
    cfg = lbm_amd.PipelineConfig(sfreq=fs, n_channels=C, n_times=T, n_baseline=251)
    pipe = lbm_amd.EpochPipeline(cfg, taps, A, b, weights, codebook, device=dev)

    E = args.chunk
    x = synth_epochs(E, C, T, sfreq=fs, seed=lbm_amd.synth.SEED + rank, device=dev, chunk=2048)
    tok_all = torch.empty((world * E, pipe.S), dtype=torch.int64, device=dev) if world > 1 else None

    def step():
        pipe.reset_hist()
        out = pipe.run(x)
        if world > 1 and not args.no_gather:
            dist.all_gather_into_tensor(tok_all, out["tokens"])
            dist.all_reduce(out["hist"])
        return out

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
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # per-stage durations from HIP events recorded on the launch stream (outside the timed region)
    stages = {}
    if rank == 0:
        reps = max(3, min(args.steps, 10))
        pipe.stage_events = []
        for _ in range(reps):
            pipe.reset_hist()
            pipe.run(x)
        torch.cuda.synchronize()
        acc = {}
        for name, s_, e_ in pipe.stage_events:
            acc.setdefault(name, []).append(s_.elapsed_time(e_))
        pipe.stage_events = None
        model = stage_model(E, C, T, len(taps), K, D, pipe.S, pipe.L1, pipe.F)
        for name, ms in acc.items():
            m = float(np.mean(ms))
            st = {"ms": round(m, 4), "bound": model[name]["bound"],
                  "tflops": round(model[name]["flops"] / (m * 1e-3) / 1e12, 3),
                  "gbs": round(model[name]["bytes"] / (m * 1e-3) / 1e9, 1)}
            stages[name] = st

    if rank == 0:
        dom = max(stages, key=lambda k: stages[k]["ms"])
        m = stage_model(E, C, T, len(taps), K, D, pipe.S, pipe.L1, pipe.F)[dom]
        if m["bound"] == "mfma":
            roof = {"kernel": dom, "bound": "mfma", "achieved": stages[dom]["tflops"], "peak": PEAK_FP32_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(stages[dom]["tflops"] / PEAK_FP32_TFLOPS, 4), "traffic": None}
        else:
            roof = {"kernel": dom, "bound": "hbm", "achieved": stages[dom]["gbs"], "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": round(stages[dom]["gbs"] / PEAK_HBM_GBS, 4), "traffic": None}
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
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config3: full pipeline (825-tap FIR + ICA affine + baseline/reject/z-score + Welch PSD "
                                   "+ band power + VQ-VAE encoder + K=%d codebook argmin), 30ch x 500Hz x 3s fp32 epochs, "
                                   "%d epochs per step per GPU, inputs resident in HBM" % (K, E),
                       "epochs_per_step_per_gpu": E, "channels": C, "samples": T, "codebook": K,
                       "collectives": "all_gather(tokens)+all_reduce(hist)" if (world > 1 and not args.no_gather) else "none"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "stages": stages,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
