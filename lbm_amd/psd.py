"""Welch PSD and band-power features over ``lbm_welch_f32`` / ``lbm_bandpower_f32``.

Mirrors ``epochs.compute_psd(method='welch', fmin, fmax, n_fft, window, n_overlap, average)``
(multi-train/withpsd.py:236-250), ``raw.compute_psd(fmin=4., fmax=30., n_fft=512)``
(preprocessing_unsupervised.py:43) and the band-power / ratio / rule code of
preprocessing_unsupervised.py:45-84 and failed/label.py:13-34.
"""
from __future__ import annotations

import warnings
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib

_WINDOWS = {"boxcar": _lib.WIN_BOXCAR, "hann": _lib.WIN_HANN, "hamming": _lib.WIN_HAMMING}
_AVERAGE = {"mean": _lib.AVG_MEAN, "median": _lib.AVG_MEDIAN, None: _lib.AVG_NONE}
_tables: Dict[tuple, torch.Tensor] = {}


def welch_tables(device, n_fft: int, n_per_seg: int, window: str, bin_lo: int, bin_hi: int) -> torch.Tensor:
    key = (str(device), n_fft, n_per_seg, window, bin_lo, bin_hi)
    t = _tables.get(key)
    if t is None:
        lib = _lib.load()
        nbytes = lib.lbm_welch_tables_bytes(n_fft, n_per_seg, bin_lo, bin_hi)
        t = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=device)
        _lib.check(lib.lbm_welch_tables_init(_lib.ptr(t), n_fft, n_per_seg, _WINDOWS[window], bin_lo, bin_hi,
                                             _lib.stream_ptr()), "lbm_welch_tables_init")
        torch.cuda.current_stream().synchronize()   # one-off: the cached table is reused from other streams without an event
        _tables[key] = t
    return t


def welch_bins(sfreq: float, n_fft: int, fmin: float, fmax: float) -> Tuple[np.ndarray, int, int]:
    freqs = np.fft.rfftfreq(n_fft, 1.0 / sfreq)
    sel = np.nonzero((freqs >= fmin) & (freqs <= fmax))[0]
    if sel.size == 0:
        raise ValueError("no frequency bin inside [fmin, fmax]")
    return freqs[sel], int(sel[0]), int(sel[-1])


@_lib.on_tensor_device
def psd_array_welch(data, sfreq, fmin=0.0, fmax=np.inf, n_fft=256, n_overlap=0, n_per_seg=None,
                    window="hamming", average="mean", row_scale=None):
    """``mne.time_frequency.psd_array_welch`` equivalent.  Returns (psds[..., F] (or [..., F, nseg]), freqs[F]).
    ``row_scale`` (one float per row of ``data``, not an MNE argument): the PSD of ``row_scale[r] * data[r]``, i.e. of the
    un-standardised epochs when ``data`` holds z-scored ones and ``row_scale`` their ``std + eps`` (``affine_post``)."""
    if average is False:  # multi-train/withpsd.py:244 passes average=False; its comments ask for per-epoch PSD
        warnings.warn("average=False is not a valid MNE value; using 'mean' (per-epoch PSD)")
        average = "mean"
    if window not in _WINDOWS:
        raise NotImplementedError(f"window {window!r} (supported: {sorted(_WINDOWS)})")
    if average not in _AVERAGE:
        raise ValueError("average must be 'mean', 'median' or None")
    was_numpy = not isinstance(data, torch.Tensor)
    x = _lib.dev_f32(data)
    T = x.shape[-1]
    n_fft = int(n_fft)
    n_per_seg = n_fft if n_per_seg is None else int(n_per_seg)
    if n_fft > T:
        raise ValueError(f"n_fft ({n_fft}) is larger than the number of samples ({T})")
    if n_per_seg > n_fft:
        raise ValueError("n_per_seg cannot exceed n_fft")
    if n_overlap >= n_per_seg:
        raise ValueError("n_overlap must be smaller than n_per_seg")
    freqs, lo, hi = welch_bins(float(sfreq), n_fft, fmin, fmax)
    F = hi - lo + 1
    rows = x.numel() // T
    nseg = (T - n_overlap) // (n_per_seg - n_overlap)
    tables = welch_tables(x.device, n_fft, n_per_seg, window, lo, hi)
    lead = tuple(x.shape[:-1])
    psd = torch.empty(lead + (F,), dtype=torch.float32, device=x.device) if average is not None else None
    seg = torch.empty(lead + (F, nseg), dtype=torch.float32, device=x.device) if average is None else None
    if row_scale is not None:
        rs = _lib.dev_f32(row_scale, x.device).reshape(-1)
        if rs.numel() != rows:
            raise ValueError(f"row_scale has {rs.numel()} entries for {rows} rows")
        _lib.check(_lib.load().lbm_welch_scaled_f32(_lib.ptr(x), _lib.ptr(rs), _lib.ptr(psd), _lib.ptr(seg), rows, T, float(sfreq),
                                                    n_fft, n_per_seg, int(n_overlap), _lib.ptr(tables), lo, hi, _AVERAGE[average],
                                                    _lib.stream_ptr()), "lbm_welch_scaled_f32")
    else:
        _lib.check(_lib.load().lbm_welch_f32(_lib.ptr(x), _lib.ptr(psd), _lib.ptr(seg), rows, T, float(sfreq), n_fft, n_per_seg,
                                             int(n_overlap), _lib.ptr(tables), lo, hi, _AVERAGE[average], _lib.stream_ptr()),
                   "lbm_welch_f32")
    out = psd if average is not None else seg
    return (out.cpu().numpy().astype(np.float64) if was_numpy else out), freqs


@_lib.on_tensor_device
def band_powers(psds, freqs, per_channel: bool = False, already_db: bool = False):
    """psds[E, C, F] -> [E, 6] (or [E, C, 6]) = theta, alpha, beta, beta/theta, alpha/beta, theta/beta
    (dB means and ratios of dB means, preprocessing_unsupervised.py:45-61)."""
    was_numpy = not isinstance(psds, torch.Tensor)
    p = _lib.dev_f32(psds)
    squeeze = p.dim() == 2
    if squeeze:
        p = p.unsqueeze(0)
    E, Cc, F = p.shape
    f = torch.as_tensor(np.asarray(freqs, dtype=np.float64)).to(device=p.device, dtype=torch.float32)
    if f.numel() != F:
        raise ValueError("freqs does not match the last axis of psds")
    out = torch.empty((E, Cc, 6) if per_channel else (E, 6), dtype=torch.float32, device=p.device)
    _lib.check(_lib.load().lbm_bandpower_f32(_lib.ptr(p), _lib.ptr(f), _lib.ptr(out), E, Cc, F, int(per_channel),
                                             int(already_db), _lib.stream_ptr()), "lbm_bandpower_f32")
    if squeeze:
        out = out[0]
    return out.cpu().numpy().astype(np.float64) if was_numpy else out


def classify_eeg_state(theta_power, alpha_power, beta_power, drowsy_threshold: float) -> int:
    """Rule cascade of preprocessing_unsupervised.py:70-84 (host side: three floats -> label)."""
    beta_theta = beta_power / theta_power
    alpha_beta = alpha_power / beta_power
    theta_beta = theta_power / beta_power
    if beta_theta < 1.0:
        return 0
    elif beta_theta > 2.0:
        return 1
    elif alpha_beta > 1.2:
        return 2
    elif theta_beta > drowsy_threshold:
        return 3
    return 4


def classify_eeg_state_variant(theta_power, alpha_power, beta_power) -> int:
    """failed/label.py:13-34 (last test compares theta_power with beta_power)."""
    if beta_power / theta_power < 1.0:
        return 0
    elif beta_power / theta_power > 2.0:
        return 1
    elif alpha_power / beta_power > 1.2:
        return 2
    elif theta_power > beta_power:
        return 3
    return 4
