#!/usr/bin/env python3
"""Run ONE stage of the pipeline repeatedly on resident synthetic data (for rocprofv3 --pmc / --kernel-trace).

    python tools/kbench.py --stage vq|conv1|conv2|conv3|conv4|fir|fir_direct|affine|post|welch --epochs 16384 --reps 5
    python tools/kbench.py --stage ica_stats|ica_fit --cin 30 --samples 1800000     (one hour of 500 Hz EEG)
Prints the HIP-event average per launch.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lbm_amd  # noqa: E402
from lbm_amd import _lib  # noqa: E402
from lbm_amd.synth import synth_epochs, synth_ica  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", required=True)
    ap.add_argument("--epochs", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dense", action="store_true", help="conv stages: dense rows (pitch = length) instead of 16-byte rows")
    ap.add_argument("--samples", type=int, default=1_800_000, help="ica stages: recording length")
    ap.add_argument("--split", action="store_true", help="conv stages: the bf16x6 kernel over pre-split weight tables")
    ap.add_argument("--direct", action="store_true", help="conv1 / conv2: the direct-form kernel instead of the F(3,2) one")
    ap.add_argument("--cin", type=int, default=0, help="conv stages: override the input channel count (random weights)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib, p, st = _lib.load(), _lib.ptr, _lib.stream_ptr()
    E, C, T = args.epochs, 30, 1500
    with np.load(os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")) as f:
        w = {k: torch.from_numpy(f[k]).to(dev) for k in f.files}
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    L1, S = 750, 375
    s = args.stage
    if s in ("ica_stats", "ica_fit"):
        import time
        from lbm_amd import fastica
        n, Tr = args.cin or 30, args.samples
        rs = np.random.RandomState(0)
        src = torch.distributions.Laplace(0.0, 1.0).sample((n, Tr)).to(dev)
        rec = (torch.from_numpy(rs.normal(size=(n, n))).to(dev, torch.float32) @ src) * 1e-5
        if s == "ica_fit":
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = lbm_amd.ica_fit(rec)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"ica_fit: {dt * 1e3:.1f} ms for [{n} x {Tr}] ({got['n_iter']} iterations, n_components={got['n_components']})")
            if Tr <= 400_000:
                from oracle import eeg_oracle as orc
                xh = rec.cpu().numpy().astype(np.float64)
                t0 = time.perf_counter()
                ref = orc.ica_fit(xh)
                print(f"cpu oracle (numpy SVD + sklearn FastICA): {(time.perf_counter() - t0) * 1e3:.1f} ms ({ref['n_iter']} iterations)")
            return
        lib2 = _lib.load()
        Wd = torch.from_numpy(rs.normal(size=(n, n)) / np.sqrt(n)).to(dev, torch.float32).contiguous()
        out = torch.empty(n * n + n, dtype=torch.float64, device=dev)
        ws = torch.empty(lib2.lbm_ica_stats_workspace_bytes(Tr, n), dtype=torch.uint8, device=dev)
        zz = src.contiguous()
        fn = lambda: lib2.lbm_ica_stats_f32(p(zz), Tr, Tr, n, p(Wd), 0, 1.0, p(out), p(out[n * n:]), p(ws), st)  # noqa: E731
    elif s in ("fir", "fir_direct", "affine", "post", "welch", "affine_post", "welch_scaled"):
        x = synth_epochs(E, C, T, seed=3, device=dev, chunk=2048)
        y = torch.empty_like(x)
    if s in ("ica_stats",):
        pass
    elif s in ("vq", "vqb", "vqs"):
        PS = (S + 3) & ~3
        z = rnd(E, 256, PS)
        cb = w["vq_layer.embedding.weight"].contiguous()
        ee = torch.empty(512, device=dev)
        lib.lbm_rownorm2_f32(p(cb), p(ee), 512, 256, st)
        tok = torch.empty(E, S, dtype=torch.int64, device=dev)
        fn = lambda: lib.lbm_vq_argmin_f32(p(z), PS, _lib.VQ_CHANNEL_MAJOR, S, p(cb), p(ee), p(tok), None, None, E * S, 256, 512, st)  # noqa: E731
        if s == "vqs":      # screened search of the folded codebook on the encoder's own activations (random rows would all tie)
            taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
            A, b = lbm_amd.ica_projection(**synth_ica(C))
            with np.load(os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")) as f:
                shipped = {k: f[k] for k in f.files}
            cfgp = lbm_amd.PipelineConfig(sfreq=500.0, n_channels=C, n_times=T, n_baseline=251, emit_cleaned=False, fold_projection=True,
                                          vq_method="screen")
            pipe = lbm_amd.EpochPipeline(cfgp, taps, A, b, shipped, shipped["vq_layer.embedding.weight"], device=dev)
            xe = synth_epochs(E, C, T, seed=3, device=dev, chunk=2048)
            pipe.run(xe)
            torch.cuda.synchronize()
            del xe
            h3, ws, tok = pipe._buf["h3"], pipe._buf["vq_ws"], pipe._buf["tokens"]
            # (what the pipeline searches: the folded codebook without its repeated rows, PipelineConfig.dedup_codes)
            fn = lambda: lib.lbm_vq_argmin_screened_f32(p(h3), pipe.P2, _lib.VQ_CHANNEL_MAJOR | _lib.VQ_NO_XNORM, S, p(pipe.cb_used), p(pipe.vq_tab),  # noqa: E731
                                                        p(pipe.ee_used), p(tok), None, p(ws), E * S, 256, pipe.K_used, st)
        if s == "vqb":
            tab = lbm_amd.vq_split_tables(cb)
            fn = lambda: lib.lbm_vq_argmin_bf16x6_f32(p(z), PS, _lib.VQ_CHANNEL_MAJOR, S, p(tab), p(ee), p(tok), None, None, E * S, 256, 512, st)  # noqa: E731
    elif s.startswith("conv"):
        i = int(s[4]) - 1
        cin, lin, cout, ks, sd, pd, relu = [(30, 1500, 64, 4, 2, 1, 1), (64, 750, 128, 4, 2, 1, 1), (128, 375, 256, 3, 1, 1, 1),
                                            (256, 375, 256, 1, 1, 0, 0)][i]
        wt, bs = w[f"encoder.{2 * i}.weight"].contiguous(), w[f"encoder.{2 * i}.bias"].contiguous()
        if args.cin:
            cin = args.cin
            wt = rnd(cout, cin, ks) * 0.05
        lout = (lin + 2 * pd - ks) // sd + 1
        ldx, ldy = ((lin + 3) & ~3, (lout + 3) & ~3) if not args.dense else (lin, lout)
        xin = rnd(E, cin, ldx)
        yo = torch.empty(E, cout, ldy, device=dev)
        fn = lambda: lib.lbm_conv1d_pitched_f32(p(xin), ldx, p(wt), p(bs), p(yo), ldy, E, cin, lin, cout, ks, sd, pd, relu, st)  # noqa: E731
        if ks == 4 and not args.split and not args.dense and not args.direct and lib.lbm_conv1d_k4s2_wino_supported(cin, cout):
            # layers 1 and 2 as the pipeline runs them: two 2-tap phases in F(3,2) (--direct: lbm_conv1d_pitched_f32)
            fn = lambda: lib.lbm_conv1d_k4s2_wino_f32(p(xin), ldx, p(wt), p(bs), p(yo), ldy, E, cin, lin, cout, relu, st)  # noqa: E731
        if s == "conv3" and not args.split and not args.cin and not args.dense:   # as the pipeline runs it: transformed weights from a table
            wtab = lbm_amd.conv_wino3_tables(wt)
            fn = lambda: lib.lbm_conv1d_wino3_f32(p(xin), ldx, p(wtab), p(bs), p(yo), ldy, E, cin, lin, cout, relu, st)  # noqa: E731
        if args.split:
            ctab = lbm_amd.conv_split_tables(wt)
            fn = lambda: lib.lbm_conv1d_bf16x6_f32(p(xin), ldx, p(ctab), p(bs), p(yo), ldy, E, cin, lin, cout, ks, sd, pd, relu, st)  # noqa: E731
    elif s in ("fir", "fir_direct"):
        taps = torch.as_tensor(lbm_amd.create_filter(500.0, 4.0, 30.0)).to(dev, torch.float32)
        if s == "fir":
            tab = lbm_amd.filter.fir_fft_tables(taps, T)
            fn = lambda: lib.lbm_fir_fft_f32(p(x), p(y), E * C, T, p(tab), taps.numel(), 0, st)  # noqa: E731
        else:
            fn = lambda: lib.lbm_fir_f32(p(x), p(y), E * C, T, p(taps), taps.numel(), 0, st)  # noqa: E731
    elif s == "affine":
        A, b = lbm_amd.ica_projection(**synth_ica(C))
        A = torch.as_tensor(A).to(dev, torch.float32).contiguous()
        b = torch.as_tensor(b).to(dev, torch.float32).contiguous()
        fn = lambda: lib.lbm_affine_ch_f32(p(x), p(y), E, C, C, T, p(A), p(b), st)  # noqa: E731
    elif s == "affine_post":
        A, b = lbm_amd.ica_projection(**synth_ica(C))
        A = torch.as_tensor(A).to(dev, torch.float32).contiguous()
        b = torch.as_tensor(b).to(dev, torch.float32).contiguous()
        keep = torch.empty(E, dtype=torch.uint8, device=dev)
        rs_ = torch.empty(E, C, device=dev)
        fn = lambda: lib.lbm_affine_post_f32(p(x), p(A), p(b), None, p(y), p(rs_), p(keep), E, C, C, T, 251, 150e-6, 1, 1e-6, st)  # noqa: E731
    elif s == "welch_scaled":
        from lbm_amd.psd import welch_bins, welch_tables
        _, lo, hi = welch_bins(500.0, 500, 4.0, 30.0)
        tab = welch_tables(dev, 500, 500, "hann", lo, hi)
        psd = torch.empty(E, C, hi - lo + 1, device=dev)
        rs_ = torch.rand(E, C, device=dev) + 0.5
        fn = lambda: lib.lbm_welch_scaled_f32(p(x), p(rs_), p(psd), None, E * C, T, 500.0, 500, 500, 250, p(tab), lo, hi, 0, st)  # noqa: E731
    elif s == "post":
        y2 = torch.empty_like(x)
        keep = torch.empty(E, dtype=torch.uint8, device=dev)
        fn = lambda: lib.lbm_epoch_post_f32(p(x), p(y), p(y2), p(keep), E, C, T, 251, 150e-6, 1, 1e-6, st)  # noqa: E731
    elif s == "welch":
        from lbm_amd.psd import welch_bins, welch_tables
        _, lo, hi = welch_bins(500.0, 500, 4.0, 30.0)
        tab = welch_tables(dev, 500, 500, "hann", lo, hi)
        psd = torch.empty(E, C, hi - lo + 1, device=dev)
        fn = lambda: lib.lbm_welch_f32(p(x), p(psd), None, E * C, T, 500.0, 500, 500, 250, p(tab), lo, hi, 0, st)  # noqa: E731
    else:
        raise SystemExit(f"unknown stage {s}")
    rc = fn()
    assert rc == 0, _lib.load().lbm_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    extra = ""
    if s in ("vq", "vqb", "vqs"):
        extra = f", {2.0 * 512 * 256 * 375 * E / (ms * 1e-3) / 1e12:.1f} fp32-equivalent TFLOP/s"
    if s.startswith("conv"):
        extra = f", {2.0 * cout * cin * ks * lout * E / (ms * 1e-3) / 1e12:.1f} TFLOP/s"
    if s == "ica_stats":
        extra = f", {args.samples * (args.cin or 30) * 4 / (ms * 1e-3) / 1e12:.2f} TB/s of the recording"
    print(f"{s}: {ms:.4f} ms per launch ({E} epochs{extra})")


if __name__ == "__main__":
    main()
