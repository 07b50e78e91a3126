"""Zero-phase FIR filtering: MNE-shaped functional surface over ``lbm_fir_f32``.

Mirrors ``raw.filter(l_freq, h_freq, fir_design='firwin')`` (training/preprocess.py:77,
multi-train/withpsd.py:118, preprocessing_unsupervised.py:36) and
``raw.notch_filter(freqs, fir_design='firwin')`` (training/preprocess.py:82, multi-train/withpsd.py:119).
The filter is designed on the host in float64 (a few hundred taps, once per configuration); the
application over ``[..., T]`` runs on the GPU.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib

_LENGTH_FACTORS = {"hamming": 3.3, "hann": 3.1, "blackman": 5.0}


def _sym_window(name: str, n: int) -> np.ndarray:
    if n == 1:
        return np.ones(1)
    k = np.arange(n, dtype=np.float64)
    c1 = np.cos(2.0 * np.pi * k / (n - 1))
    if name == "hamming":
        return 0.54 - 0.46 * c1
    if name == "hann":
        return 0.5 - 0.5 * c1
    if name == "blackman":
        return 0.42 - 0.5 * c1 + 0.08 * np.cos(4.0 * np.pi * k / (n - 1))
    raise ValueError(f"unsupported fir_window {name!r}")


def _lowpass_sinc(numtaps: int, cutoff: float, window: str) -> np.ndarray:
    """Windowed-sinc low-pass, cutoff normalised to Nyquist, unit DC gain (firwin semantics)."""
    m = np.arange(numtaps, dtype=np.float64) - 0.5 * (numtaps - 1)
    h = cutoff * np.sinc(cutoff * m)
    h *= _sym_window(window, numtaps)
    return h / h.sum()


def _compose(n_taps: int, freq: Sequence[float], gain: Sequence[int], window: str) -> np.ndarray:
    h = np.zeros(n_taps, dtype=np.float64)
    if gain[-1] == 1:
        h[n_taps // 2] = 1.0
    prev_f, prev_g = freq[-1], gain[-1]
    for f, g in zip(freq[-2::-1], gain[-2::-1]):
        if g != prev_g:
            trans = (prev_f - f) / 2.0
            n = int(round(_LENGTH_FACTORS[window] / trans))
            n += 1 - n % 2
            if n > n_taps:
                raise ValueError("filter_length is too short for the requested transition band")
            lp = _lowpass_sinc(n, (prev_f + f) / 2.0, window)
            off = (n_taps - n) // 2
            if g == 0:
                h[off:n_taps - off] -= lp
            else:
                h[off:n_taps - off] += lp
        prev_f, prev_g = f, g
    return h


def create_filter(sfreq: float, l_freq: Optional[float], h_freq: Optional[float],
                  l_trans_bandwidth="auto", h_trans_bandwidth="auto", fir_window: str = "hamming",
                  fir_design: str = "firwin", phase: str = "zero") -> np.ndarray:
    """FIR design with MNE's defaults (float64 taps, odd length, linear phase)."""
    if fir_design != "firwin" or phase != "zero":
        raise NotImplementedError("only fir_design='firwin', phase='zero' (the reference's settings)")
    sfreq = float(sfreq)
    nyq = sfreq / 2.0
    l_tr = None if l_freq is None else (min(max(0.25 * l_freq, 2.0), l_freq)
                                        if isinstance(l_trans_bandwidth, str) else float(l_trans_bandwidth))
    h_tr = None if h_freq is None else (min(max(0.25 * h_freq, 2.0), nyq - h_freq)
                                        if isinstance(h_trans_bandwidth, str) else float(h_trans_bandwidth))
    if l_freq is not None and h_freq is not None and l_freq > h_freq:      # band-stop
        freq = [0.0, h_freq, h_freq + h_tr, l_freq - l_tr, l_freq, nyq]
        gain = [1, 1, 0, 0, 1, 1]
        mt = min(l_tr, h_tr)
    elif l_freq is not None and h_freq is not None:                        # band-pass
        freq = [0.0, l_freq - l_tr, l_freq, h_freq, h_freq + h_tr, nyq]
        gain = [0, 0, 1, 1, 0, 0]
        if freq[1] <= 0:
            freq, gain = [0.0] + freq[2:], gain[1:]
        if freq[-2] >= nyq:
            freq, gain = freq[:-1], gain[:-1]
        mt = min(l_tr, h_tr)
    elif h_freq is not None:                                               # low-pass
        freq, gain, mt = [0.0, h_freq, h_freq + h_tr, nyq], [1, 1, 0, 0], h_tr
    elif l_freq is not None:                                               # high-pass
        freq, gain, mt = [0.0, l_freq - l_tr, l_freq, nyq], [0, 0, 1, 1], l_tr
    else:
        raise ValueError("l_freq and h_freq cannot both be None")
    n_taps = int(round(_LENGTH_FACTORS[fir_window] * sfreq / mt))
    n_taps += 1 - n_taps % 2
    fn = [f / nyq for f in freq]
    fn[-1] = 1.0
    return _compose(n_taps, fn, gain, fir_window)


FFT_MIN_TAPS = 65      # below this the direct LDS form (lbm_fir_f32) does less work than a 1024-point transform


def fir_fft_tables(taps_dev: torch.Tensor, T: int) -> Optional[torch.Tensor]:
    """Device table buffer for ``lbm_fir_fft_f32`` (DFT of the taps + twiddles), or None if the
    (T, ntaps) pair is outside what the fast-convolution kernel supports."""
    lib = _lib.load()
    n = int(taps_dev.numel())
    nbytes = int(lib.lbm_fir_fft_tables_bytes(int(T), n))
    if nbytes == 0:
        return None
    tab = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=taps_dev.device)
    _lib.check(lib.lbm_fir_fft_tables_init(_lib.ptr(tab), _lib.ptr(taps_dev), int(T), n, _lib.stream_ptr()),
               "lbm_fir_fft_tables_init")
    return tab


@_lib.on_tensor_device
def fir_apply(data, taps, pad: str = "reflect_limited", method: str = "auto"):
    """Apply symmetric odd-length FIR taps along the last axis of a CUDA/array tensor (fp32 on device).

    method: 'fft' = fast convolution (``lbm_fir_fft_f32``, what MNE's overlap-add does), 'direct' = LDS-staged
    direct form (``lbm_fir_f32``), 'auto' = fft for >= FFT_MIN_TAPS taps when supported, else direct."""
    if method not in ("auto", "fft", "direct"):
        raise ValueError(f"method must be 'auto', 'fft' or 'direct', got {method!r}")
    pad_mode = {"reflect_limited": _lib.PAD_REFLECT_LIMITED, "zero": _lib.PAD_ZERO}[pad]
    was_numpy = not isinstance(data, torch.Tensor)
    x = _lib.dev_f32(data)
    T = x.shape[-1] if x.dim() else 0
    rows = x.numel() // T if T else 0
    t = torch.as_tensor(np.asarray(taps, dtype=np.float64), dtype=torch.float64).to(x.device).to(torch.float32) \
        if not (isinstance(taps, torch.Tensor) and taps.is_cuda) else taps.to(torch.float32).contiguous()
    y = torch.empty_like(x)
    lib = _lib.load()
    tab = None
    if rows and method != "direct" and (method == "fft" or t.numel() >= FFT_MIN_TAPS):
        tab = fir_fft_tables(t, T)
        if tab is None and method == "fft":
            raise NotImplementedError(f"fast-convolution FIR does not support ntaps={t.numel()} (use method='direct')")
    if tab is not None:
        _lib.check(lib.lbm_fir_fft_f32(_lib.ptr(x), _lib.ptr(y), rows, T, _lib.ptr(tab), t.numel(), pad_mode,
                                       _lib.stream_ptr()), "lbm_fir_fft_f32")
    else:
        _lib.check(lib.lbm_fir_f32(_lib.ptr(x), _lib.ptr(y), rows, T, _lib.ptr(t), t.numel(), pad_mode,
                                   _lib.stream_ptr()), "lbm_fir_f32")
    return y.cpu().numpy().astype(np.float64) if was_numpy else y


@_lib.on_tensor_device
def filter_data(data, sfreq, l_freq, h_freq, *, l_trans_bandwidth="auto", h_trans_bandwidth="auto",
                fir_design="firwin", phase="zero", fir_window="hamming", pad="reflect_limited"):
    """``mne.filter.filter_data`` equivalent (array in, array out) on the GPU."""
    taps = create_filter(sfreq, l_freq, h_freq, l_trans_bandwidth, h_trans_bandwidth, fir_window, fir_design, phase)
    return fir_apply(data, taps, pad)


def notch_taps(sfreq, freqs, notch_widths=None, trans_bandwidth: float = 1.0, fir_window="hamming"):
    """The band-stop design of ``mne.filter.notch_filter(method='fir')``: ONE multi-band FIR for all notch frequencies
    (``filter_data(x, sfreq, l_freq=highs, h_freq=lows, tb_2, tb_2)`` -> ``create_filter`` with array edges: pass edges
    f -/+ (w/2 + tb/2), stop edges f -/+ w/2, all edges sorted into one gain template).  Returns a one-element list (kept a list:
    ``notch_filter`` applies every design in it)."""
    sfreq = float(sfreq)
    nyq = sfreq / 2.0
    freqs = np.atleast_1d(np.asarray(freqs, dtype=np.float64))
    widths = freqs / 200.0 if notch_widths is None else np.broadcast_to(np.asarray(notch_widths, dtype=np.float64), freqs.shape)
    tb_2 = trans_bandwidth / 2.0
    f_p1, f_p2 = freqs - widths / 2.0 - tb_2, freqs + widths / 2.0 + tb_2      # pass edges below / above each notch
    f_s1, f_s2 = f_p1 + tb_2, f_p2 - tb_2                                        # stop edges
    edges = np.r_[f_p1, f_s1, f_s2, f_p2]
    gains = np.r_[np.ones_like(f_p1), np.zeros_like(f_s1), np.zeros_like(f_s2), np.ones_like(f_p2)]
    if edges.min() <= 0 or edges.max() >= nyq:
        raise ValueError("notch edges must lie inside (0, sfreq / 2)")
    order = np.argsort(edges, kind="stable")
    freq = np.r_[0.0, edges[order], nyq]
    gain = np.r_[1.0, gains[order], 1.0]
    if np.any(np.abs(np.diff(gain, 2)) > 1):
        raise ValueError("Stop bands are not sufficiently separated.")
    n_taps = int(round(_LENGTH_FACTORS[fir_window] * sfreq / tb_2))
    n_taps += 1 - n_taps % 2
    fn = [f / nyq for f in freq]
    fn[-1] = 1.0
    return [_compose(n_taps, fn, [int(g) for g in gain], fir_window)]


@_lib.on_tensor_device
def notch_filter(data, sfreq, freqs, *, notch_widths=None, trans_bandwidth=1.0, fir_design="firwin",
                 fir_window="hamming", pad="reflect_limited"):
    """``mne.filter.notch_filter`` equivalent (``raw.notch_filter(freqs=60.0, fir_design='firwin')``, training/preprocess.py:82,
    multi-train/withpsd.py:119): one multi-band band-stop FIR over all notch frequencies (``notch_taps``)."""
    if fir_design != "firwin":
        raise NotImplementedError("only fir_design='firwin'")
    was_numpy = not isinstance(data, torch.Tensor)
    y = _lib.dev_f32(data)
    for taps in notch_taps(sfreq, freqs, notch_widths, trans_bandwidth, fir_window):
        y = fir_apply(y, taps, pad)
    return y.cpu().numpy().astype(np.float64) if was_numpy else y


@_lib.on_tensor_device
def resample(data, up: int = 1, down: int = 1):
    """``mne.filter.resample(data, up, down, npad='auto', window='boxcar', pad='reflect_limited')`` along the last axis
    (``raw.resample(250)`` at preprocessing_unsupervised.py:37 is ``resample(data, 250, 500)`` after reduction).
    Any ratio: power-of-two resampled lengths by plain FFTs, every other length by a chirp-z inverse transform
    (see ``lbm_resample_fft_f32``)."""
    from math import gcd
    up, down = int(up), int(down)
    g = gcd(up, down)
    up, down = up // g, down // g
    was_numpy = not isinstance(data, torch.Tensor)
    x = _lib.dev_f32(data)
    T = x.shape[-1]
    rows = x.numel() // max(T, 1)
    lib = _lib.load()
    n_out = lib.lbm_resample_length(T, up, down)
    nbytes = lib.lbm_resample_workspace_bytes(min(rows, 64), T, up, down) if rows else 0
    if T and (n_out == 0 or nbytes == 0):
        raise NotImplementedError(f"resample: ratio {up}/{down} of {T} samples is not supported (row too short or too long)")
    x2 = x.reshape(rows, T)
    y = torch.empty((rows, n_out), dtype=torch.float32, device=x.device)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    for lo in range(0, rows, 64):                                   # 64 rows per call bound the workspace (1 GiB at 2^21 points)
        n = min(64, rows - lo)
        _lib.check(lib.lbm_resample_fft_f32(_lib.ptr(x2[lo:]), T, _lib.ptr(y[lo:]), n_out, n, T, up, down, _lib.ptr(ws),
                                            _lib.stream_ptr()), "lbm_resample_fft_f32")
    y = y.reshape(*x.shape[:-1], n_out)
    return y.cpu().numpy().astype(np.float64) if was_numpy else y
