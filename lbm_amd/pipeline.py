"""Fused driver of the batched EEG epoch pipeline on one GPU, plus the multi-GPU shard/gather helpers.

filter (a1) -> ICA affine (a3) -> baseline / peak-to-peak keep mask / z-score (a4-a6) -> Welch PSD on the
cleaned epochs (a7) -> band powers (a8) -> VQ-VAE encoder (a10) -> codebook argmin + usage histogram (a12).
Order and parameters follow training/preprocess.py:77-287 and multi-train/withpsd.py:118-270, applied per
epoch.  All buffers are allocated once per chunk size; a chunk run only enqueues
kernels on the current stream (no host sync, no allocation).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .constants import ZSCORE_EPSILON
from .psd import welch_bins, welch_tables


@dataclass
class PipelineConfig:
    sfreq: float = 500.0
    n_channels: int = 30
    n_times: int = 1500
    n_baseline: int = 251
    reject: float = 150e-6
    n_fft: Optional[int] = None        # default int(sfreq)  (multi-train/withpsd.py:231)
    n_overlap: Optional[int] = None    # default n_fft // 2  (multi-train/withpsd.py:243)
    window: str = "hann"
    fmin: float = 4.0
    fmax: float = 30.0
    zscore_eps: float = ZSCORE_EPSILON
    do_tokenize: bool = True
    do_filter: bool = True
    do_ica: bool = True
    fir_method: str = "fft"            # 'fft' (fast convolution) or 'direct' (LDS direct form)
    emit_cleaned: bool = True          # write the baseline-corrected epochs (output "cleaned").  No other output depends
                                       # on the baseline offset: Welch removes each segment's mean, peak-to-peak and the
                                       # z-score are offset-invariant -- so False saves one 180 000 B/epoch write
                                       # (the 186 264 B/epoch compulsory figure excludes cleaned epochs)
    vq_method: str = "fp32"            # 'fp32': exact fmaf chain on the fp32 MFMA; 'screen': the SAME tokens, bit for bit, from a
                                       # bf16 screening pass with a rigorous error bound + the fp32 kernel on the rows it cannot
                                       # settle (lbm_vq_argmin_screened_f32); 'bf16x6': fp32 products rebuilt from three bf16
                                       # pieces on the bf16 MFMA (same accuracy vs float64, ~1.6x faster; near-tie rule)
    conv_method: str = "fp32"          # 'bf16x6': the encoder's 256-channel layers (3rd conv, 1x1 projection) on the bf16 MFMA
                                       # with three-piece operands (conv_bf16.hip; same accuracy vs float64 as the fp32 chain)
    conv_k4s2: str = "wino"            # the two 4-tap stride-2 layers: 'wino' = two 2-tap phases in the minimal form F(3,2)
                                       # (lbm_conv1d_k4s2_wino_f32: 2/3 of the products, same accuracy vs float64 as the direct
                                       # chain, not bit-identical to it) wherever rows are 16-byte pitched; 'direct' = lbm_conv1d_pitched_f32
    fuse_post: bool = True             # ICA affine + baseline / reject / z-score in one kernel (lbm_affine_post_f32) and Welch on the
                                       # z-scored epochs with the per-row scale (lbm_welch_scaled_f32): 900 000 instead of
                                       # 1 260 000 B of HBM traffic per 30x1500 epoch in the DSP front; False = the separate kernels
    overlap_welch: bool = False        # Welch + band power on a second HIP stream beside the encoder's first layers (they need the z-scored
                                       # rows only; joined before run() returns); ignored while stage events are being recorded.  Measured:
                                       # -0.05 ms per chunk in a back-to-back walk (profiles/r03_experiments.md), hence off
    dedup_codes: bool = True           # search the codebook without the rows that repeat an earlier row bit for bit (a first-minimum
                                       # argmin can never return them; quantizer.dedup_codebook) and map the tokens back: identical
                                       # tokens and histogram, fewer code tiles (the shipped codebook: 413 of 512 rows, 7 tiles of 8)
    fold_projection: bool = False      # tokens from the 3rd encoder layer with the last 1x1 conv folded into the codebook
                                       # (quantizer.fold_projection; no z output; equal tokens except fp32 near-ties)


class EpochPipeline:
    def __init__(self, cfg: PipelineConfig, taps, A, b, encoder: Optional[dict], codebook, device=None):
        dev = torch.device("cuda") if device is None else torch.device(device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.dev = dev
        with torch.cuda.device(dev):       # tables are built on, and launches go to, the pipeline's own device
            self._init(cfg, taps, A, b, encoder, codebook)

    def _init(self, cfg, taps, A, b, encoder, codebook):
        self.cfg = cfg
        self.lib = _lib.load()
        f32 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64)).to(self.dev, torch.float32).contiguous()  # noqa: E731
        self.taps = f32(taps) if taps is not None else None
        self.fir_tab = None
        if self.taps is not None and cfg.do_filter and cfg.fir_method == "fft":
            from .filter import fir_fft_tables
            self.fir_tab = fir_fft_tables(self.taps, cfg.n_times)   # None -> direct form
        self.A = f32(A) if A is not None else None
        self.b = f32(b) if b is not None else None
        self.n_fft = int(cfg.sfreq) if cfg.n_fft is None else int(cfg.n_fft)
        self.n_overlap = self.n_fft // 2 if cfg.n_overlap is None else int(cfg.n_overlap)
        self.freqs, self.bin_lo, self.bin_hi = welch_bins(cfg.sfreq, self.n_fft, cfg.fmin, cfg.fmax)
        self.F = self.bin_hi - self.bin_lo + 1
        self.freqs_dev = torch.as_tensor(self.freqs).to(self.dev, torch.float32)
        self.tables = welch_tables(self.dev, self.n_fft, self.n_fft, cfg.window, self.bin_lo, self.bin_hi)
        self.window_id = {"boxcar": _lib.WIN_BOXCAR, "hann": _lib.WIN_HANN, "hamming": _lib.WIN_HAMMING}[cfg.window]
        self.enc = None
        if cfg.do_tokenize:
            self.enc = {k: torch.as_tensor(np.asarray(v)).to(self.dev, torch.float32).contiguous()
                        for k, v in encoder.items() if k.startswith("encoder.")}
            self.codebook = torch.as_tensor(np.asarray(codebook)).to(self.dev, torch.float32).contiguous()
            self.K, self.D = self.codebook.shape
            self.ee = torch.empty((self.K,), dtype=torch.float32, device=self.dev)
            _lib.check(self.lib.lbm_rownorm2_f32(_lib.ptr(self.codebook), _lib.ptr(self.ee), self.K, self.D, _lib.stream_ptr()))
            self.cb_fold = self.ee_fold = None
            if cfg.fold_projection:
                from .quantizer import fold_projection
                self.cb_fold, self.ee_fold = fold_projection(self.codebook, self.enc["encoder.6.weight"], self.enc["encoder.6.bias"])
            if cfg.vq_method not in ("fp32", "bf16x6", "screen"):
                raise ValueError(f"vq_method {cfg.vq_method!r}")
            # what the search runs on: the (folded) codebook, without repeated rows if asked
            self.cb_used, self.ee_used = (self.cb_fold, self.ee_fold) if cfg.fold_projection else (self.codebook, self.ee)
            self.K_used, self.K_real, self.code_map = self.K, self.K, None
            if cfg.dedup_codes:
                from .quantizer import dedup_codebook
                r = dedup_codebook(self.cb_used, self.ee_used)
                if r is not None:
                    self.cb_used, self.ee_used, self.code_map, self.K_real = r
                    self.K_used = int(self.cb_used.shape[0])
            self.count_in_remap = True      # (False: the search counts over searched rows, the remap pass moves the counts)
            self.vq_tab = None
            if cfg.vq_method in ("bf16x6", "screen"):
                from .quantizer import vq_split_tables
                self.vq_tab = vq_split_tables(self.cb_used)
            if cfg.conv_method not in ("fp32", "bf16x6"):
                raise ValueError(f"conv_method {cfg.conv_method!r}")
            if cfg.conv_k4s2 not in ("wino", "direct"):
                raise ValueError(f"conv_k4s2 {cfg.conv_k4s2!r}")
            self.conv_tab = {}
            if cfg.conv_method == "bf16x6":
                from .quantizer import conv_split_tables
                for i in (4,) if cfg.fold_projection else (4, 6):
                    self.conv_tab[i] = conv_split_tables(self.enc[f"encoder.{i}.weight"])
            # the third layer's Winograd-transformed weights, built once (same results as transforming them in the kernel)
            self.wino_tab = None
            w3 = self.enc["encoder.4.weight"]
            if 4 not in self.conv_tab and w3.shape[2] == 3 and self.lib.lbm_conv1d_wino3_tables_bytes(w3.shape[1], w3.shape[0]):
                from .quantizer import conv_wino3_tables
                self.wino_tab = conv_wino3_tables(w3)
            T = cfg.n_times
            self.L1 = (T + 2 - 4) // 2 + 1
            self.L2 = (self.L1 + 2 - 4) // 2 + 1
            self.S = self.L2
            self.P1, self.P2 = (self.L1 + 3) & ~3, (self.L2 + 3) & ~3   # row pitches of the activation buffers
        # the fused ICA + post kernel needs the ICA map and a shape it supports; otherwise the separate kernels run
        self.fused = bool(cfg.fuse_post and cfg.do_ica and self.A is not None and
                          self.lib.lbm_affine_post_supported(cfg.n_channels, cfg.n_channels, cfg.n_times))
        self._cap = 0
        self._buf: Dict[str, torch.Tensor] = {}
        self.stage_events = None   # set to [] to record (name, start, end) HIP events per stage on the current stream

    # ------------------------------------------------------------------ buffers
    def _reserve(self, E: int):
        if E <= self._cap:
            return
        c, d = self.cfg, self.dev
        mk = lambda *s, dt=torch.float32: torch.empty(s, dtype=dt, device=d)  # noqa: E731
        B = {"keep": mk(E, dt=torch.uint8), "psd": mk(E, c.n_channels, self.F), "bp": mk(E, 6)}
        if c.do_filter:
            B["filt"] = mk(E, c.n_channels, c.n_times)
        if c.emit_cleaned or (c.do_ica and not self.fused) or (self.fused and not c.do_tokenize):
            B["clean"] = mk(E, c.n_channels, c.n_times)
        if c.do_tokenize:
            B["std"] = mk(E, c.n_channels, c.n_times)
        if self.fused and c.do_tokenize:
            B["rscale"] = mk(E, c.n_channels)
        if c.do_tokenize:
            # encoder activations: rows padded to a multiple of 4 floats, so every conv access is a 16-byte one
            B.update(h1=mk(E, 64, self.P1), h2=mk(E, 128, self.P2), h3=mk(E, 256, self.P2),
                     tokens=mk(E, self.S, dt=torch.int64), hist=torch.zeros((self.K,), dtype=torch.int64, device=d))
            if self.code_map is not None:
                B.update(hist_c=torch.zeros((self.K_used,), dtype=torch.int64, device=d))
            if not c.fold_projection:
                B.update(z=mk(E, self.D, self.P2))
            if c.vq_method == "screen":
                B.update(vq_ws=mk(int(self.lib.lbm_vq_screen_workspace_bytes(E * self.S, self.K_used)), dt=torch.uint8))
        self._buf, self._cap = B, E

    def reset_hist(self):
        if "hist" in self._buf:
            self._buf["hist"].zero_()

    # ------------------------------------------------------------------ one chunk
    # caller-owned outputs run() can write in place of its own buffers: name -> (buffer key, dtype, trailing shape)
    def _out_specs(self):
        c = self.cfg
        specs = {"psd": ("psd", torch.float32, (c.n_channels, self.F)), "bandpower": ("bp", torch.float32, (6,)),
                 "keep": ("keep", torch.uint8, ())}
        if c.do_tokenize:
            specs["tokens"] = ("tokens", torch.int64, (self.S,))
        return specs

    def run(self, x: torch.Tensor, out: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """x: [E, C, T] float32 CUDA tensor on the pipeline's device.  Returns views into the pipeline's buffers (valid
        until the next run).  Kernels are enqueued on that device's current stream.

        ``out``: optional caller-owned destinations for "tokens" [E, S] int64, "psd" [E, C, F] float32, "bandpower" [E, 6]
        float32 and / or "keep" [E] uint8 -- contiguous tensors on the pipeline's device, e.g. the slice of a job-wide array
        this chunk belongs to.  The kernels then write there directly (no copy afterwards) and the returned dict holds
        those tensors."""
        if x.is_cuda and x.device != self.dev:
            raise ValueError(f"x lives on {x.device}, the pipeline on {self.dev}")
        with torch.cuda.device(self.dev):
            return self._run(x, out)

    def _run(self, x: torch.Tensor, out_bufs: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        c, lib, p, st = self.cfg, self.lib, _lib.ptr, _lib.stream_ptr()
        ev = self.stage_events

        def mark(name, start):
            if ev is None:
                return None
            e_ = torch.cuda.Event(enable_timing=True)
            e_.record()
            if start is None:
                return e_
            ev.append((name, start, e_))
            return e_
        if x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous():
            raise ValueError("x must be a contiguous float32 CUDA tensor")
        E, C, T = x.shape
        if (C, T) != (c.n_channels, c.n_times):
            raise ValueError(f"expected [E, {c.n_channels}, {c.n_times}], got {tuple(x.shape)}")
        self._reserve(E)
        B = {k: (v[:E] if k not in ("hist", "hist_c", "vq_ws") else v) for k, v in self._buf.items()}
        if out_bufs:
            specs = self._out_specs()
            for name, t in out_bufs.items():
                if name not in specs:
                    raise ValueError(f"out[{name!r}]: this pipeline can write {sorted(specs)} into caller-owned tensors")
                key, dt, tail = specs[name]
                if (not isinstance(t, torch.Tensor) or t.device != self.dev or t.dtype != dt or tuple(t.shape) != (E,) + tail
                        or not t.is_contiguous()):
                    raise ValueError(f"out[{name!r}] must be a contiguous {dt} tensor of shape {(E,) + tail} on {self.dev}")
                B[key] = t
        cur = x
        side_done = None
        t0 = mark("", None)
        if c.do_filter:
            if self.fir_tab is not None:
                _lib.check(lib.lbm_fir_fft_f32(p(cur), p(B["filt"]), E * C, T, p(self.fir_tab), self.taps.numel(),
                                               _lib.PAD_REFLECT_LIMITED, st), "fir_fft")
            else:
                _lib.check(lib.lbm_fir_f32(p(cur), p(B["filt"]), E * C, T, p(self.taps), self.taps.numel(),
                                           _lib.PAD_REFLECT_LIMITED, st), "fir")
            cur = B["filt"]
            t0 = mark("fir", t0)
        std_mode = _lib.STD_EPOCH_CHANNEL if c.do_tokenize else _lib.STD_NONE
        if self.fused:
            # one pass: channel map + baseline + keep mask + z-score; the cleaned epochs are written only if someone wants them
            base = B["clean"] if (c.emit_cleaned or not c.do_tokenize) else None
            _lib.check(lib.lbm_affine_post_f32(p(cur), p(self.A), p(self.b), p(base), p(B["std"]) if c.do_tokenize else None,
                                               p(B["rscale"]) if c.do_tokenize else None, p(B["keep"]), E, C, C, T, c.n_baseline,
                                               float(c.reject), std_mode, c.zscore_eps, st), "affine_post")
            t0 = mark("affine_post", t0)
            if c.do_tokenize and c.overlap_welch and ev is None:
                # the PSD branch (Welch + band power: latency-bound, 12 waves per CU) on a second stream: it depends on the z-scored rows
                # only, and its workgroups fit beside the persistent workgroups of the encoder's layers
                if getattr(self, "_side", None) is None:
                    self._side = torch.cuda.Stream(device=self.dev)
                ready = torch.cuda.Event()
                ready.record()
                self._side.wait_event(ready)
                sst = self._side.cuda_stream
                _lib.check(lib.lbm_welch_scaled_f32(p(B["std"]), p(B["rscale"]), p(B["psd"]), None, E * C, T, float(c.sfreq), self.n_fft,
                                                    self.n_fft, self.n_overlap, p(self.tables), self.bin_lo, self.bin_hi,
                                                    _lib.AVG_MEAN, sst), "welch")
                _lib.check(lib.lbm_bandpower_f32(p(B["psd"]), p(self.freqs_dev), p(B["bp"]), E, C, self.F, 0, 0, sst), "bandpower")
                side_done = torch.cuda.Event()
                side_done.record(self._side)
            elif c.do_tokenize:      # PSD of the cleaned epochs from the z-scored ones: PSD(s x) = s^2 PSD(x), segment means removed
                _lib.check(lib.lbm_welch_scaled_f32(p(B["std"]), p(B["rscale"]), p(B["psd"]), None, E * C, T, float(c.sfreq), self.n_fft,
                                                    self.n_fft, self.n_overlap, p(self.tables), self.bin_lo, self.bin_hi,
                                                    _lib.AVG_MEAN, st), "welch")
            else:
                _lib.check(lib.lbm_welch_f32(p(base), p(B["psd"]), None, E * C, T, float(c.sfreq), self.n_fft, self.n_fft,
                                             self.n_overlap, p(self.tables), self.bin_lo, self.bin_hi, _lib.AVG_MEAN, st), "welch")
            t0 = mark("welch", t0)
        else:
            if c.do_ica:
                _lib.check(lib.lbm_affine_ch_f32(p(cur), p(B["clean"]), E, C, C, T, p(self.A), p(self.b), st), "affine")
                cur = B["clean"]
                t0 = mark("affine", t0)
            base = B["clean"] if c.emit_cleaned else cur      # the PSD is the same with or without the baseline offset
            if not c.emit_cleaned and not c.do_tokenize:
                # features only: the post stage would write nothing but the keep mask -- the Welch kernel takes the reject test along
                _lib.check(lib.lbm_welch_keep_f32(p(cur), p(B["psd"]), p(B["keep"]), E, C, T, float(c.sfreq), self.n_fft, self.n_fft,
                                                  self.n_overlap, p(self.tables), self.bin_lo, self.bin_hi, float(c.reject), st), "welch+keep")
                t0 = mark("welch", t0)
                _lib.check(lib.lbm_bandpower_f32(p(B["psd"]), p(self.freqs_dev), p(B["bp"]), E, C, self.F, 0, 0, st), "bandpower")
                t0 = mark("bandpower", t0)
                return {"psd": B["psd"], "bandpower": B["bp"], "keep": B["keep"]}
            _lib.check(lib.lbm_epoch_post_f32(p(cur), p(base) if c.emit_cleaned else None, p(B["std"]) if c.do_tokenize else None,
                                              p(B["keep"]), E, C, T, c.n_baseline, float(c.reject), std_mode, c.zscore_eps, st), "post")
            t0 = mark("post", t0)
            _lib.check(lib.lbm_welch_f32(p(base), p(B["psd"]), None, E * C, T, float(c.sfreq), self.n_fft, self.n_fft,
                                         self.n_overlap, p(self.tables), self.bin_lo, self.bin_hi, _lib.AVG_MEAN, st), "welch")
            t0 = mark("welch", t0)
        if side_done is None:
            _lib.check(lib.lbm_bandpower_f32(p(B["psd"]), p(self.freqs_dev), p(B["bp"]), E, C, self.F, 0, 0, st), "bandpower")
        t0 = mark("bandpower", t0)
        out = {"psd": B["psd"], "bandpower": B["bp"], "keep": B["keep"]}
        if c.emit_cleaned:
            out["cleaned"] = base
        if c.do_tokenize:
            e = self.enc

            def conv(xi, ldx, yo, ldy, i, cin, lin, cout, ks, s_, pd, relu):
                tab = self.conv_tab.get(i)
                if i == 4 and tab is None and self.wino_tab is not None and (ks, s_, pd) == (3, 1, 1):
                    _lib.check(lib.lbm_conv1d_wino3_f32(p(xi), ldx, p(self.wino_tab), p(e[f"encoder.{i}.bias"]), p(yo), ldy, E, cin, lin,
                                                        cout, relu, st), "conv(wino3)")
                elif (tab is None and c.conv_k4s2 == "wino" and (ks, s_, pd) == (4, 2, 1) and ldx % 4 == 0 and ldy % 4 == 0 and
                      lib.lbm_conv1d_k4s2_wino_supported(cin, cout)):
                    _lib.check(lib.lbm_conv1d_k4s2_wino_f32(p(xi), ldx, p(e[f"encoder.{i}.weight"]), p(e[f"encoder.{i}.bias"]), p(yo), ldy, E,
                                                            cin, lin, cout, relu, st), "conv(k4s2 wino)")
                elif tab is not None:
                    _lib.check(lib.lbm_conv1d_bf16x6_f32(p(xi), ldx, p(tab), p(e[f"encoder.{i}.bias"]), p(yo), ldy, E, cin, lin, cout,
                                                         ks, s_, pd, relu, st), "conv(bf16x6)")
                else:
                    _lib.check(lib.lbm_conv1d_pitched_f32(p(xi), ldx, p(e[f"encoder.{i}.weight"]), p(e[f"encoder.{i}.bias"]), p(yo), ldy,
                                                          E, cin, lin, cout, ks, s_, pd, relu, st), "conv")
            conv(B["std"], T, B["h1"], self.P1, 0, C, T, 64, 4, 2, 1, 1)
            t0 = mark("conv1", t0)
            conv(B["h1"], self.P1, B["h2"], self.P2, 2, 64, self.L1, 128, 4, 2, 1, 1)
            t0 = mark("conv2", t0)
            conv(B["h2"], self.P2, B["h3"], self.P2, 4, 128, self.L2, 256, 3, 1, 1, 1)
            t0 = mark("conv3", t0)
            vq = lib.lbm_vq_argmin_f32 if self.vq_tab is None else lib.lbm_vq_argmin_bf16x6_f32
            if c.vq_method == "screen":      # same call shape as the other two; needs the fp32 codebook AND its piece table
                def vq(x_, ldx_, layout_, S_, _table, ee_, tok_, _mind, hist_, N_, D_, K_, st_):
                    return lib.lbm_vq_argmin_screened_f32(x_, ldx_, layout_, S_, p(self.cb_used), p(self.vq_tab), ee_, tok_, hist_,
                                                          p(B["vq_ws"]), N_, D_, K_, st_)
            # with a searched sub-codebook the histogram is taken by the remap pass (one read of the tokens less); beyond its bin limit the
            # search counts over searched rows into hist_c and the remap pass moves the counts
            count_in_remap = self.code_map is not None and self.K_used <= 12288 and self.count_in_remap
            hist_k = None if count_in_remap else (B["hist_c"] if self.code_map is not None else B["hist"])

            def remap():     # searched rows -> the caller's code numbers (tokens in place, counts into the full histogram)
                if self.code_map is not None:
                    # (over all K_used searched rows: a padding row can only "win" where every real score is +inf, and maps to code 0 --
                    # what the full search returns there)
                    _lib.check(lib.lbm_vq_remap_codes_i64(p(B["tokens"]), E * self.S, p(self.code_map), self.K_used, p(hist_k), p(B["hist"]),
                                                          st), "vq remap")
            if c.fold_projection:
                _lib.check(vq(p(B["h3"]), self.P2, _lib.VQ_CHANNEL_MAJOR | _lib.VQ_NO_XNORM, self.S,
                              p(self.cb_used if self.vq_tab is None else self.vq_tab), p(self.ee_used), p(B["tokens"]), None,
                              p(hist_k), E * self.S, self.cb_used.shape[1], self.K_used, st), "vq(folded)")
                remap()
                if side_done is not None:
                    torch.cuda.current_stream().wait_event(side_done)
                t0 = mark("vq", t0)
                out.update(tokens=B["tokens"], hist=B["hist"], standardized=B["std"])
                return out
            conv(B["h3"], self.P2, B["z"], self.P2, 6, 256, self.L2, self.D, 1, 1, 0, 0)
            t0 = mark("conv4", t0)
            _lib.check(vq(p(B["z"]), self.P2, _lib.VQ_CHANNEL_MAJOR, self.S, p(self.cb_used if self.vq_tab is None else self.vq_tab),
                          p(self.ee_used), p(B["tokens"]), None, p(hist_k), E * self.S, self.D, self.K_used, st), "vq")
            remap()
            t0 = mark("vq", t0)
            out.update(tokens=B["tokens"], hist=B["hist"], standardized=B["std"], z=B["z"][:, :, :self.S])
        if side_done is not None:
            torch.cuda.current_stream().wait_event(side_done)
        return out


# ---------------------------------------------------------------------- multi-GPU helpers (one process per GPU)
def shard_range(E: int, rank: int, world: int):
    """Contiguous block of epochs owned by `rank`; concatenating shards in rank order reproduces
    np.concatenate order (training/model.py:423)."""
    base, rem = divmod(E, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _collective_needed(group, force: bool) -> bool:
    """A collective runs when a process group exists and has more than one rank -- or, with ``force``, whenever a group
    exists: a world-size-1 group still goes through RCCL's communicator, streams and buffers, which is how the exchange step
    is exercised on a one-GPU box (tests/test_gpu_rccl_world1.py, ``bench.py`` at N = 1)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        if force:
            raise RuntimeError("force=True needs an initialised torch.distributed process group")
        return False
    return force or dist.get_world_size(group) > 1


def gather_epoch_outputs(local: torch.Tensor, group=None, force: bool = False) -> torch.Tensor:
    """all_gather of a per-epoch array [E_local, ...] whose E_local may differ between ranks.
    Equal shards use one all_gather_into_tensor; unequal shards gather counts first, then padded blocks.
    ``force`` runs the collectives even in a group of one rank (see ``_collective_needed``); ``force="padded"`` also takes
    the padded-blocks route that unequal shards take."""
    import torch.distributed as dist
    if not _collective_needed(group, bool(force)):
        return local
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    local = local.contiguous()
    if min(counts) == mx and force != "padded":
        out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    blocks = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(blocks, pad, group=group)
    return torch.cat([blk[:c] for blk, c in zip(blocks, counts)], dim=0)


def gather_pipeline_outputs(out: Dict[str, torch.Tensor], group=None, names: Sequence[str] = ("tokens", "psd", "bandpower", "keep"),
                            drop_rejected: bool = False, force=False) -> Dict[str, torch.Tensor]:
    """The exchange step of the path: every per-epoch array of a rank's outputs all-gathered in rank order,
    i.e. the order np.concatenate gives the reference's per-batch lists (training/model.py:423, multi-train/withpsd.py:394-414).
    ``drop_rejected`` first removes the epochs whose keep flag is 0 (epochs.drop_bad, multi-train/withpsd.py:219): shards then
    differ in length and the counts travel first (gather_epoch_outputs).  Every rank of the group must call this.
    ``force``: as in ``gather_epoch_outputs``."""
    sel = None
    if drop_rejected:
        sel = out["keep"].to(torch.bool)
    res = {}
    for k in names:
        if k in out:
            v = out[k] if sel is None else out[k][sel]
            res[k] = gather_epoch_outputs(v, group, force)
    return res


def allreduce_hist(hist: torch.Tensor, group=None, force: bool = False) -> torch.Tensor:
    """Sum of the K-entry codebook usage histogram over ranks (the only reduction of the path).  ``force``: run the
    all-reduce in a group of one rank too."""
    import torch.distributed as dist
    if _collective_needed(group, force):
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
    return hist


def perplexity_from_counts(hist: torch.Tensor) -> float:
    """Host-side perplexity of a (globally reduced) histogram, exp(-sum p log(p+1e-10)) in fp32 like
    training/model.py:143-144.  Device tensors should use quantizer.perplexity_from_hist instead."""
    h = hist.detach().to("cpu", torch.float32)
    pr = h / h.sum()
    return float(torch.exp(-torch.sum(pr * torch.log(pr + 1e-10))))
