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

    def moments(self, x: torch.Tensor, keep: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[C, 3] float64 {sum, sum of squares, count}: the quantity to all-reduce across ranks."""
        x = _lib.dev_f32(x)
        E, Cc, T = x.shape
        ws = torch.empty((max(E * Cc, 1), 2), dtype=torch.float64, device=x.device)
        mom = torch.empty((Cc, 3), dtype=torch.float64, device=x.device)
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

    def fit(self, x: torch.Tensor, y=None):
        return self.fit_from_moments(self.moments(x))

    def transform(self, x: torch.Tensor) -> torch.Tensor:
        x = _lib.dev_f32(x)
        E, Cc, T = x.shape
        out = torch.empty_like(x)
        _lib.check(_lib.load().lbm_channel_scale_f32(_lib.ptr(x), _lib.ptr(out), _lib.ptr(self.mean_), _lib.ptr(self.scale_),
                                                     E, Cc, T, _lib.stream_ptr()), "lbm_channel_scale_f32")
        return out

    def fit_transform(self, x: torch.Tensor, y=None) -> torch.Tensor:
        return self.fit(x).transform(x)
