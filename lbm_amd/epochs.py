"""Epoch post-processing over ``lbm_epoch_post_f32`` / ``lbm_channel_moments_f32`` / ``lbm_channel_scale_f32``.

Mirrors ``mne.Epochs(..., baseline=(None, 0), reject=dict(eeg=150e-6))`` + ``drop_bad``
(training/preprocess.py:262-271, multi-train/withpsd.py:213-219), the per-epoch per-channel z-score
(training/preprocess.py:279-287) and ``mne.decoding.Scaler(scalings='mean')`` (multi-train/withpsd.py:267-270).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import _lib
from .constants import ZSCORE_EPSILON


def n_baseline_samples(tmin: float, sfreq: float, b_end: float = 0.0) -> int:
    """Number of samples with tmin <= t <= b_end, i.e. the window of ``baseline=(None, b_end)``."""
    return int(round((b_end - tmin) * sfreq)) + 1


@_lib.on_tensor_device
def make_epochs(raw: torch.Tensor, event_samples, tmin: float, tmax: float, sfreq: float, first_samp: int = 0):
    """Window extraction of ``mne.Epochs(raw, events, tmin=tmin, tmax=tmax)`` (training/preprocess.py:262-271).

    raw: [C, T_rec] (CUDA or array-like); event_samples: event sample indices (column 0 of the MNE events
    array).  Returns (x[E_kept, C, Tw] float32 CUDA, kept: bool numpy mask over the events); an event whose
    window [s + round(tmin*sfreq), s + round(tmax*sfreq)] leaves the recording is dropped, as MNE does.
    Baseline correction / rejection / z-score are ``epoch_post`` on the result."""
    import numpy as np
    raw = _lib.dev_f32(raw)
    if raw.dim() != 2:
        raise ValueError("raw must be [channels, samples]")
    if raw.stride(1) != 1:
        raw = raw.contiguous()
    Cc, Trec = raw.shape
    ev = np.asarray(event_samples, dtype=np.int64).reshape(-1) - int(first_samp)
    lo, hi = int(round(tmin * sfreq)), int(round(tmax * sfreq))
    Tw = hi - lo + 1
    kept = (ev + lo >= 0) & (ev + hi < Trec)
    starts = torch.from_numpy(ev[kept] + lo).to(raw.device)
    E = int(kept.sum())
    out = torch.empty((E, Cc, Tw), dtype=torch.float32, device=raw.device)
    _lib.check(_lib.load().lbm_epochs_gather_f32(_lib.ptr(raw), raw.stride(0) if Cc > 1 else Trec, _lib.ptr(starts), _lib.ptr(out),
                                                 E, Cc, Tw, Trec, _lib.stream_ptr()), "lbm_epochs_gather_f32")
    return out, kept


def save_preprocess_outputs(prefix: str, standardized_epochs, labels) -> Tuple[str, str]:
    """training/preprocess.py:35-36,303-304: ``<prefix>_epochs_ica_std.npy`` (float64, as MNE/NumPy produce)
    and ``<prefix>_labels.npy``."""
    import numpy as np
    a, b = f"{prefix}_epochs_ica_std.npy", f"{prefix}_labels.npy"
    x = standardized_epochs.detach().cpu().numpy() if isinstance(standardized_epochs, torch.Tensor) else np.asarray(standardized_epochs)
    np.save(a, x.astype(np.float64))
    np.save(b, np.asarray(labels))
    return a, b


def save_withpsd_outputs(prefix: str, time_epochs, psd_epochs, psd_freqs, labels):
    """multi-train/withpsd.py:61-66,394-414: ``*_time_epochs.npy``, ``*_psd_epochs.npy``, ``*_psd_freqs.npy``,
    ``*_labels.npy`` (float64 arrays on disk like the reference's)."""
    import numpy as np
    host = lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)  # noqa: E731
    paths = [f"{prefix}_time_epochs.npy", f"{prefix}_psd_epochs.npy", f"{prefix}_psd_freqs.npy", f"{prefix}_labels.npy"]
    np.save(paths[0], host(time_epochs).astype(np.float64))
    np.save(paths[1], host(psd_epochs).astype(np.float64))
    np.save(paths[2], host(psd_freqs).astype(np.float64))
    np.save(paths[3], host(labels))
    return paths


@_lib.on_tensor_device
def epoch_post(x: torch.Tensor, *, n_baseline: int = 0, reject: Optional[float] = None, standardize: bool = False,
               eps: float = ZSCORE_EPSILON, inplace_baseline: bool = False
               ) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
    """One pass over x[E, C, T]: returns (baseline-corrected, standardized or None, keep mask or None)."""
    x = _lib.dev_f32(x)
    E, Cc, T = x.shape
    y_base = x if inplace_baseline else torch.empty_like(x)
    y_std = torch.empty_like(x) if standardize else None
    keep = torch.empty((E,), dtype=torch.uint8, device=x.device) if reject is not None else None
    _lib.check(_lib.load().lbm_epoch_post_f32(
        _lib.ptr(x), _lib.ptr(y_base), _lib.ptr(y_std), _lib.ptr(keep), E, Cc, T, int(n_baseline),
        float(reject) if reject is not None else math.inf,
        _lib.STD_EPOCH_CHANNEL if standardize else _lib.STD_NONE, float(eps), _lib.stream_ptr()), "lbm_epoch_post_f32")
    return y_base, y_std, (keep.bool() if keep is not None else None)


def affine_post_supported(Cin: int, Cout: int, T: int) -> bool:
    return bool(_lib.load().lbm_affine_post_supported(int(Cin), int(Cout), int(T)))


@_lib.on_tensor_device
def affine_post(x: torch.Tensor, A, b=None, *, n_baseline: int = 0, reject: Optional[float] = None, standardize: bool = True,
                eps: float = ZSCORE_EPSILON, want_base: bool = True):
    """ICA channel map + baseline + peak-to-peak keep mask + z-score in ONE pass over x[E, Cin, T] (``lbm_affine_post_f32``):
    what ``ica_apply`` followed by ``epoch_post`` computes (training/preprocess.py:144,262-287; multi-train/withpsd.py:213-219),
    with one read and one write per requested output.  Returns a dict: ``base`` (baseline-corrected, if ``want_base``), ``std``
    and ``row_scale`` = std_t + eps per (epoch, channel) (if ``standardize``), ``keep`` (bool, if ``reject`` is given)."""
    x = _lib.dev_f32(x)
    A = _lib.dev_f32(A, x.device)
    b = _lib.dev_f32(b, x.device) if b is not None else None
    E, Cin, T = x.shape
    Cout = A.shape[0]
    if A.shape[1] != Cin:
        raise ValueError(f"A is {tuple(A.shape)}, x has {Cin} channels")
    mk = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)  # noqa: E731
    y_base = mk(E, Cout, T) if want_base else None
    y_std = mk(E, Cout, T) if standardize else None
    rs = mk(E, Cout) if standardize else None
    keep = torch.empty((E,), dtype=torch.uint8, device=x.device) if reject is not None else None
    _lib.check(_lib.load().lbm_affine_post_f32(
        _lib.ptr(x), _lib.ptr(A), _lib.ptr(b), _lib.ptr(y_base), _lib.ptr(y_std), _lib.ptr(rs), _lib.ptr(keep), E, Cin, Cout, T,
        int(n_baseline), float(reject) if reject is not None else math.inf,
        _lib.STD_EPOCH_CHANNEL if standardize else _lib.STD_NONE, float(eps), _lib.stream_ptr()), "lbm_affine_post_f32")
    out = {}
    if want_base:
        out["base"] = y_base
    if standardize:
        out["std"], out["row_scale"] = y_std, rs
    if keep is not None:
        out["keep"] = keep.bool()
    return out


@_lib.on_tensor_device
def standardize_epochs(x: torch.Tensor, eps: float = ZSCORE_EPSILON) -> torch.Tensor:
    """(x - mean_t) / (std_t + eps) per epoch and channel (training/preprocess.py:282-287)."""
    x = _lib.dev_f32(x)
    E, Cc, T = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.load().lbm_epoch_post_f32(_lib.ptr(x), None, _lib.ptr(y), None, E, Cc, T, 0, math.inf,
                                              _lib.STD_EPOCH_CHANNEL, float(eps), _lib.stream_ptr()), "lbm_epoch_post_f32")
    return y


class Scaler:
    """``mne.decoding.Scaler(scalings='mean', with_mean=True, with_std=True)``: per-channel z-score over
    all epochs x time (sklearn StandardScaler semantics: population std, std == 0 -> 1)."""

    def __init__(self, scalings: str = "mean", with_mean: bool = True, with_std: bool = True):
        if scalings != "mean":
            raise NotImplementedError("only scalings='mean' (the reference's setting)")
        self.with_mean, self.with_std = with_mean, with_std
        self.mean_ = self.scale_ = self.moments_ = None

    @_lib.on_tensor_device
    def moments(self, x: torch.Tensor, keep: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                static_scratch: bool = False) -> torch.Tensor:
        """[C, 3] float64 {sum, sum of squares, count}: the quantity to all-reduce across ranks.
        ``static_scratch=True`` (with a caller-provided ``out`` and a uint8 ``keep``) goes through
        ``lbm_channel_moments_static_f32``: no workspace, no allocation -- for callers under graph capture."""
        x = _lib.dev_f32(x)
        E, Cc, T = x.shape
        if static_scratch:
            if out is None or out.dtype != torch.float64 or tuple(out.shape) != (Cc, 3) or not out.is_contiguous():
                raise ValueError("static_scratch needs out = contiguous float64 [C, 3] on the device")
            if keep is not None and (keep.dtype != torch.uint8 or not keep.is_contiguous()):
                raise ValueError("static_scratch needs keep as a contiguous uint8 tensor")
            _lib.check(_lib.load().lbm_channel_moments_static_f32(_lib.ptr(x), _lib.ptr(keep), _lib.ptr(out), E, Cc, T,
                                                                  _lib.stream_ptr()), "lbm_channel_moments_static_f32")
            return out
        ws = torch.empty((max(E * Cc, 1), 2), dtype=torch.float64, device=x.device)
        mom = torch.empty((Cc, 3), dtype=torch.float64, device=x.device) if out is None else out
        k = keep.to(torch.uint8).contiguous() if keep is not None else None
        _lib.check(_lib.load().lbm_channel_moments_f32(_lib.ptr(x), _lib.ptr(k), _lib.ptr(ws), _lib.ptr(mom), E, Cc, T,
                                                       _lib.stream_ptr()), "lbm_channel_moments_f32")
        return mom

    def fit_from_moments(self, mom: torch.Tensor):
        n = mom[:, 2]
        mean = mom[:, 0] / n
        var = torch.clamp(mom[:, 1] / n - mean * mean, min=0.0)
        std = torch.sqrt(var)
        std = torch.where(std == 0, torch.ones_like(std), std)
        self.moments_ = mom
        self.mean_ = (mean if self.with_mean else torch.zeros_like(mean)).to(torch.float32)
        self.scale_ = (std if self.with_std else torch.ones_like(std)).to(torch.float32)
        return self

    def fit(self, x: torch.Tensor, y=None, keep: Optional[torch.Tensor] = None, group=None, distributed: bool = False,
            force: bool = False):
        """Per-channel statistics of the epochs handed in.  LOCAL by default: the reference fits one Scaler per file
        (multi-train/withpsd.py:267-270), and ranks that walk different files must not meet in a collective.
        With ``distributed=True`` (default group) or an explicit ``group`` the {sum, sum of squares, count} block is
        all-reduced first, so every rank fits the statistics of the whole job -- EVERY rank of the group
        must then make the call, ranks with 0 epochs included (their moments are zeros).  ``force``: run the all-reduce in
        a group of one rank too (exercises RCCL on a one-GPU box)."""
        mom = self.moments(x, keep)
        if distributed or group is not None:
            mom = self.reduce_moments(mom, group, force)
        return self.fit_from_moments(mom)

    @staticmethod
    def reduce_moments(mom: torch.Tensor, group=None, force: bool = False) -> torch.Tensor:
        """Explicit all-reduce(SUM) of a [C, 3] moments block over ``group`` (no-op without an initialised process group, or
        in a group of one rank unless ``force``)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
            dist.all_reduce(mom, op=dist.ReduceOp.SUM, group=group)
        return mom

    @_lib.on_tensor_device
    def transform(self, x: torch.Tensor) -> torch.Tensor:
        x = _lib.dev_f32(x)
        E, Cc, T = x.shape
        out = torch.empty_like(x)
        _lib.check(_lib.load().lbm_channel_scale_f32(_lib.ptr(x), _lib.ptr(out), _lib.ptr(self.mean_), _lib.ptr(self.scale_),
                                                     E, Cc, T, _lib.stream_ptr()), "lbm_channel_scale_f32")
        return out

    def fit_transform(self, x: torch.Tensor, y=None, **fit_kw) -> torch.Tensor:
        return self.fit(x, **fit_kw).transform(x)
