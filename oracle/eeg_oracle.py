"""CPU oracle for the batched EEG epoch pipeline -- TEST INFRASTRUCTURE ONLY.

This module is a CPU restatement of the reference's arithmetic for the hot path
(filter -> ICA apply -> epoch post-processing -> Welch PSD -> band powers ->
VQ-VAE encoder -> nearest-codebook tokenisation).  It is the *checker*:

  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
    ``bench.py`` may import it;
  * nothing under ``lbm_amd/`` (the product) imports it, and the product has no
    CPU fallback: it raises when the HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):

  * quantizer / encoder (a10-a14): PINNED.  The formulas below are checked
    against the reference's own ``training/model.py`` classes, imported in the
    build container with the shipped checkpoint ``training/vq_vae_model.pth``
    (``oracle/gen_golden.py`` -> ``tests/golden/vq_*.npz``).
  * band power + rule (a8/a9): PINNED by the reference's shipped arrays
    ``X_psd_multi.npy`` -> ``y_label_multi.npy`` (62/62) and the ``failed/label.py``
    variant -> ``unsupervised/y_label_multi.npy`` (62/62).
  * FIR design/application, ICA apply, epoch post-processing, Welch PSD values
    (a1-a7): **parity unpinned at the MNE boundary**.  The arithmetic lives in
    the un-vendored dependency ``mne==1.5.1`` (reference ``requirements.txt:5``),
    which is not installed and cannot be fetched.  What follows restates MNE's
    published algorithm on top of the same SciPy
    primitives MNE itself calls (``scipy.signal.firwin``, ``scipy.signal.spectrogram``)
    and is anchored on the reference's call sites.

All DSP stages compute in float64 (as MNE does); the encoder and quantizer
compute in float32 on torch-CPU (as the reference does).
"""
from __future__ import annotations

import math
from typing import Iterable, Optional, Sequence, Tuple

import numpy as np
from scipy import signal as _sig

# --------------------------------------------------------------------------------------
# a1 / a2: FIR design (MNE ``create_filter`` with fir_design='firwin') and zero-phase apply
# reference call sites: training/preprocess.py:77,82 ; multi-train/withpsd.py:118-119 ;
#                       preprocessing_unsupervised.py:36
# --------------------------------------------------------------------------------------

_LENGTH_FACTORS = {"hamming": 3.3, "hann": 3.1, "blackman": 5.0}


def _auto_trans(l_freq, h_freq, sfreq):
    """MNE 'auto' transition bandwidths."""
    l_trans = None if l_freq is None else min(max(0.25 * l_freq, 2.0), l_freq)
    h_trans = None if h_freq is None else min(max(0.25 * h_freq, 2.0), sfreq / 2.0 - h_freq)
    return l_trans, h_trans


def _firwin_design(n_taps: int, freq: Sequence[float], gain: Sequence[float], window: str) -> np.ndarray:
    """MNE ``_firwin_design``: build a type-I FIR as a sum/difference of windowed-sinc low-passes.

    ``freq`` is normalised to Nyquist (last entry == 1).  Walks the band edges from the top
    down; every gain change contributes one ``scipy.signal.firwin`` low-pass whose own length
    is set by *its* transition width.
    """
    assert freq[0] == 0 and n_taps % 2 == 1
    h = np.zeros(n_taps, dtype=np.float64)
    prev_freq, prev_gain = freq[-1], gain[-1]
    if gain[-1] == 1:
        h[n_taps // 2] = 1.0
    length_factor = _LENGTH_FACTORS[window]
    for this_freq, this_gain in zip(freq[::-1][1:], gain[::-1][1:]):
        assert this_gain in (0, 1)
        if this_gain != prev_gain:
            transition = (prev_freq - this_freq) / 2.0
            this_n = int(round(length_factor / transition))
            this_n += 1 - this_n % 2  # make odd
            if this_n > n_taps:
                raise ValueError("filter_length too short for the requested transition band")
            this_h = _sig.firwin(this_n, (prev_freq + this_freq) / 2.0, window=window,
                                 pass_zero=True, fs=freq[-1] * 2)
            offset = (n_taps - this_n) // 2
            if this_gain == 0:
                h[offset:n_taps - offset] -= this_h
            else:
                h[offset:n_taps - offset] += this_h
        prev_gain, prev_freq = this_gain, this_freq
    return h


def design_fir(sfreq: float, l_freq: Optional[float], h_freq: Optional[float],
               l_trans_bandwidth="auto", h_trans_bandwidth="auto",
               fir_window: str = "hamming") -> np.ndarray:
    """Restatement of ``mne.filter.create_filter(..., method='fir', phase='zero', fir_design='firwin')``.

    band-pass (l<h), band-stop (l>h), low-pass (l None), high-pass (h None).
    Returns float64 taps (odd length, symmetric).
    """
    sfreq = float(sfreq)
    nyq = sfreq / 2.0
    lt_auto, ht_auto = _auto_trans(l_freq, h_freq, sfreq)
    l_trans = lt_auto if isinstance(l_trans_bandwidth, str) else float(l_trans_bandwidth)
    h_trans = ht_auto if isinstance(h_trans_bandwidth, str) else float(h_trans_bandwidth)
    if l_freq is not None and h_freq is not None and l_freq > h_freq:
        # band-stop: l_freq is the upper pass edge, h_freq the lower one
        l_stop, h_stop = l_freq - l_trans, h_freq + h_trans
        freq = [0.0, h_freq, h_stop, l_stop, l_freq, nyq]
        gain = [1, 1, 0, 0, 1, 1]
        min_trans = min(l_trans, h_trans)
    elif l_freq is not None and h_freq is not None:
        freq = [0.0, l_freq - l_trans, l_freq, h_freq, h_freq + h_trans, nyq]
        gain = [0, 0, 1, 1, 0, 0]
        if freq[1] <= 0:      # l_trans == l_freq -> stop edge at DC
            freq, gain = freq[1:], gain[1:]
            freq[0] = 0.0
        if freq[-2] >= nyq:
            freq, gain = freq[:-1], gain[:-1]
        min_trans = min(l_trans, h_trans)
    elif h_freq is not None:  # low-pass
        freq = [0.0, h_freq, h_freq + h_trans, nyq]
        gain = [1, 1, 0, 0]
        min_trans = h_trans
    elif l_freq is not None:  # high-pass
        freq = [0.0, l_freq - l_trans, l_freq, nyq]
        gain = [0, 0, 1, 1]
        min_trans = l_trans
    else:
        raise ValueError("l_freq and h_freq cannot both be None")
    n_taps = int(round(_LENGTH_FACTORS[fir_window] * sfreq / min_trans))
    n_taps += 1 - n_taps % 2
    freq_n = [f / nyq for f in freq]
    freq_n[-1] = 1.0
    return _firwin_design(n_taps, freq_n, gain, fir_window)


def design_notch(sfreq: float, freqs: Iterable[float] | float, notch_widths=None,
                 trans_bandwidth: float = 1.0, fir_window: str = "hamming"):
    """``raw.notch_filter(freqs, fir_design='firwin')`` designs (training/preprocess.py:82).

    MNE calls ``filter_data(x, sfreq, l_freq=highs, h_freq=lows, tb_2, tb_2)`` with ARRAY edges: ``create_filter`` sorts the pass
    and stop edges of all notches into one template and designs ONE multi-band band-stop filter.  Returned as a one-element list.
    LOWEST-CONFIDENCE item of SURVEY.md App. A1; unverifiable without MNE.
    """
    sfreq = float(sfreq)
    nyq = sfreq / 2.0
    freqs = np.atleast_1d(np.asarray(freqs, dtype=np.float64))
    widths = freqs / 200.0 if notch_widths is None else np.broadcast_to(
        np.asarray(notch_widths, dtype=np.float64), freqs.shape)
    tb_2 = trans_bandwidth / 2.0
    lows = np.array([f - w / 2.0 - tb_2 for f, w in zip(freqs, widths)])
    highs = np.array([f + w / 2.0 + tb_2 for f, w in zip(freqs, widths)])
    f_p1, f_s1, f_s2, f_p2 = lows, lows + tb_2, highs - tb_2, highs
    freq = np.r_[f_p1, f_s1, f_s2, f_p2]
    gain = np.r_[np.ones_like(f_p1), np.zeros_like(f_s1), np.zeros_like(f_s2), np.ones_like(f_p2)]
    order = np.argsort(freq, kind="stable")
    freq = np.r_[0.0, freq[order]]
    gain = np.r_[1.0, gain[order]]
    if freq[-1] != nyq:
        freq = np.r_[freq, nyq]
        gain = np.r_[gain, 1.0]
    if np.any(np.abs(np.diff(gain, 2)) > 1):
        raise ValueError("Stop bands are not sufficiently separated.")
    n_taps = int(round(_LENGTH_FACTORS[fir_window] * sfreq / tb_2))
    n_taps += 1 - n_taps % 2
    fn = [f / nyq for f in freq]
    fn[-1] = 1.0
    return [_firwin_design(n_taps, fn, [int(g) for g in gain], fir_window)]


def _smart_pad(x: np.ndarray, n_pad: int) -> np.ndarray:
    """MNE ``_smart_pad(x, (n_pad, n_pad), 'reflect_limited')`` for one 1-D float64 row."""
    if n_pad == 0:
        return x
    n = len(x)
    z = np.zeros(max(n_pad - n + 1, 0), dtype=x.dtype)
    left = 2 * x[0] - x[n_pad:0:-1]
    right = 2 * x[-1] - x[-2:-n_pad - 2:-1]
    return np.concatenate([z, left, x, right, z])


def fir_zero_phase(x: np.ndarray, h: np.ndarray, fft: bool = False) -> np.ndarray:
    """Zero-phase application of a symmetric odd-length FIR along the last axis (float64).

    Equivalent to MNE ``_overlap_add_filter(x, h, phase='zero', pad='reflect_limited')``:
    odd-reflect pad (zero-filled beyond the available samples), convolve, remove the
    ``(N-1)/2`` group delay.  ``fft=True`` uses ``scipy.signal.oaconvolve`` (what MNE's
    overlap-add does, used for the timed CPU baseline); ``fft=False`` is a direct
    float64 convolution (used as the exact checker).
    """
    x = np.asarray(x, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64)
    n_taps = len(h)
    assert n_taps % 2 == 1
    T = x.shape[-1]
    n_edge = max(min(n_taps, T) - 1, 0)
    shift = (n_taps - 1) // 2 + n_edge
    rows = x.reshape(-1, T)
    out = np.empty_like(rows)
    for i, r in enumerate(rows):
        xe = _smart_pad(r, n_edge)
        if fft:
            full = _sig.oaconvolve(xe, h, mode="full")
        else:
            full = np.convolve(xe, h, mode="full")
        out[i] = full[shift:shift + T]
    return out.reshape(x.shape)


def filter_data(x, sfreq, l_freq, h_freq, fft: bool = False, **kw):
    return fir_zero_phase(x, design_fir(sfreq, l_freq, h_freq, **kw), fft=fft)


def notch_filter(x, sfreq, freqs, fft: bool = False, **kw):
    y = np.asarray(x, dtype=np.float64)
    for h in design_notch(sfreq, freqs, **kw):
        y = fir_zero_phase(y, h, fft=fft)
    return y


def resample(x, up=1.0, down=1.0):
    """``mne.filter.resample(x, up, down, npad='auto', window='boxcar', pad='reflect_limited')`` along the last axis
    (``raw.resample(250)``, preprocessing_unsupervised.py:37), restated from MNE 1.5 (parity unpinned)."""
    x = np.asarray(x, dtype=np.float64)
    T = x.shape[-1]
    ratio = float(up) / float(down)
    final_len = int(round(ratio * T))
    min_add = min(T // 8, 100) * 2
    npad = 2 ** int(np.ceil(np.log2(T + min_add))) - T
    npads = np.array([npad // 2, npad - npad // 2])
    orig_len = T + npad
    new_len = max(int(round(ratio * orig_len)), 1)
    rem0 = int(round(ratio * npads[0]))
    rem1 = new_len - final_len - rem0
    shorter = new_len < orig_len
    use_len = new_len if shorter else orig_len
    W = np.ones(use_len) * (float(new_len) / float(orig_len))     # boxcar window, ifftshift is the identity on ones
    W = W[: use_len // 2 + 1]
    out = np.empty(x.shape[:-1] + (final_len,))
    for idx in np.ndindex(*x.shape[:-1]):
        row = x[idx]
        pad = np.concatenate([_reflect_limited_side(row, npads[0], left=True), row, _reflect_limited_side(row, npads[1], left=False)])
        xf = np.fft.rfft(pad)[: use_len // 2 + 1].copy()
        if use_len % 2 == 0:
            xf[use_len // 2] *= 2 if shorter else 0.5
        xf *= W
        y = np.fft.irfft(xf, new_len)
        out[idx] = y[rem0:new_len - rem1]
    return out


def _reflect_limited_side(x, n_pad, left):
    """One side of MNE ``_smart_pad(..., 'reflect_limited')``: odd reflection while samples last, zeros beyond."""
    n = len(x)
    k = min(n_pad, n - 1)
    if left:
        ref = 2 * x[0] - x[k:0:-1]
        return np.concatenate([np.zeros(n_pad - k), ref])
    ref = 2 * x[-1] - x[-2:-k - 2:-1]
    return np.concatenate([ref, np.zeros(n_pad - k)])


# --------------------------------------------------------------------------------------
# a3: ICA apply / sources (MNE ``ICA.apply`` / ``get_sources``)
# reference call sites: training/preprocess.py:144 ; multi-train/withpsd.py:152 ;
#                       preprocessing_unsupervised.py:41
# --------------------------------------------------------------------------------------

def ica_projection(unmixing, mixing, pca_components, pca_mean, pre_whitener,
                   n_components: int, exclude: Sequence[int] = ()):
    """Fold ``ICA.apply`` into one affine map ``out = A @ x + b`` (float64).

    out = s * (P @ (x / s - mu) + mu),  P = M[:, keep] @ U[keep, :]
    """
    unmixing = np.asarray(unmixing, dtype=np.float64)
    mixing = np.asarray(mixing, dtype=np.float64)
    pca_components = np.asarray(pca_components, dtype=np.float64)
    n_pca = pca_components.shape[0]
    C = pca_components.shape[1]
    mu = np.zeros(C) if pca_mean is None else np.asarray(pca_mean, dtype=np.float64).reshape(C)
    s = np.broadcast_to(np.asarray(pre_whitener, dtype=np.float64).reshape(-1), (C,)) \
        if np.ndim(pre_whitener) else np.full(C, float(pre_whitener))
    U = np.eye(n_pca)
    U[:n_components, :n_components] = unmixing
    U = U @ pca_components[:n_pca]
    M = np.eye(n_pca)
    M[:n_components, :n_components] = mixing
    M = pca_components[:n_pca].T @ M
    keep = np.concatenate([np.setdiff1d(np.arange(n_components), np.asarray(list(exclude), dtype=int)),
                           np.arange(n_components, n_pca)]).astype(int)
    P = M[:, keep] @ U[keep, :]
    A = (s[:, None] * P) / s[None, :]
    b = s * (mu - P @ mu)
    return A, b


def ica_apply(x, unmixing, mixing, pca_components, pca_mean, pre_whitener,
              n_components: int, exclude: Sequence[int] = ()):
    """Literal step-by-step ``ICA.apply`` on ``x[..., C, T]`` (float64), NOT via the folded A, b."""
    x = np.asarray(x, dtype=np.float64)
    pca_components = np.asarray(pca_components, dtype=np.float64)
    n_pca, C = pca_components.shape
    mu = np.zeros(C) if pca_mean is None else np.asarray(pca_mean, dtype=np.float64).reshape(C)
    s = np.broadcast_to(np.asarray(pre_whitener, dtype=np.float64).reshape(-1, 1), (C, 1)) \
        if np.ndim(pre_whitener) else np.full((C, 1), float(pre_whitener))
    data = x / s
    data = data - mu[:, None]
    U = np.eye(n_pca)
    U[:n_components, :n_components] = np.asarray(unmixing, dtype=np.float64)
    U = U @ pca_components[:n_pca]
    M = np.eye(n_pca)
    M[:n_components, :n_components] = np.asarray(mixing, dtype=np.float64)
    M = pca_components[:n_pca].T @ M
    keep = np.concatenate([np.setdiff1d(np.arange(n_components), np.asarray(list(exclude), dtype=int)),
                           np.arange(n_components, n_pca)]).astype(int)
    P = M[:, keep] @ U[keep, :]
    out = P @ data
    out = out + mu[:, None]
    return out * s


def ica_sources(x, unmixing, pca_components, pca_mean, pre_whitener, n_components: int):
    """``ICA.get_sources``: S = unmixing @ (pca_components[:n_comp] @ (x / s - mu))."""
    x = np.asarray(x, dtype=np.float64)
    pca_components = np.asarray(pca_components, dtype=np.float64)
    C = pca_components.shape[1]
    mu = np.zeros(C) if pca_mean is None else np.asarray(pca_mean, dtype=np.float64).reshape(C)
    s = np.broadcast_to(np.asarray(pre_whitener, dtype=np.float64).reshape(-1, 1), (C, 1)) \
        if np.ndim(pre_whitener) else np.full((C, 1), float(pre_whitener))
    data = x / s - mu[:, None]
    return np.asarray(unmixing, dtype=np.float64) @ (pca_components[:n_components] @ data)


def average_reference_matrix(C: int) -> np.ndarray:
    """``raw.set_eeg_reference('average')`` (preprocessing_unsupervised.py:35) as I - 11^T/C."""
    return np.eye(C) - np.full((C, C), 1.0 / C)


# --------------------------------------------------------------------------------------
# 8(f)-3: ICA fit (MNE ICA._fit glue restated; the FastICA loop is sklearn's, the back-end MNE calls)
# --------------------------------------------------------------------------------------

def _exp_var_ncomp(var, n):
    cvar = np.asarray(var, dtype=np.float64).cumsum()
    cvar /= cvar[-1]
    k = min(int((cvar <= n).sum()) + 1, len(cvar))
    return k, float(cvar[k - 1])


def ica_fit(data, n_components=None, random_state=97, max_iter=1000, tol=1e-4, fun="logcosh") -> dict:
    """``ICA(n_components, method='fastica', random_state=97, max_iter='auto').fit`` on a ``[C, T]`` float64
    array (training/preprocess.py:91-98, multi-train/withpsd.py:127-130): scalar pre-whitener, PCA by SVD with
    ``svd_flip`` and whitening, ``sklearn.decomposition.FastICA(whiten=False)``, un-whitening of the unmixing
    matrix, ``pinv`` mixing, components sorted by explained variance.  MNE-side glue: parity unpinned."""
    from sklearn.decomposition import FastICA
    x = np.asarray(data, dtype=np.float64)
    C, T = x.shape
    pre = float(np.std(x))
    pre = pre if pre != 0 else 1.0
    xw = x / pre
    mean = xw.mean(axis=1)
    X = (xw - mean[:, None]).T                                    # [T, C]
    U, S, V = np.linalg.svd(X, full_matrices=False)
    mx = np.argmax(np.abs(U), axis=0)                             # svd_flip, u-based
    signs = np.sign(U[mx, np.arange(U.shape[1])])
    signs[signs == 0] = 1.0
    U, V = U * signs, V * signs[:, None]
    ev = S ** 2 / (T - 1)
    ratio = ev / ev.sum()
    if n_components is None:
        n = _exp_var_ncomp(ratio, 0.999999)[0]
    elif isinstance(n_components, float):
        n = _exp_var_ncomp(ratio, n_components)[0]
    else:
        n = int(n_components)
    Z = U[:, :n] * math.sqrt(T - 1)                               # whitened data [T, n]
    ica = FastICA(whiten=False, random_state=np.random.RandomState(random_state), max_iter=max_iter, tol=tol, fun=fun)
    ica.fit(Z)
    W = ica.components_
    norms = np.sqrt(ev[:n])
    norms[norms == 0] = 1.0
    unmixing = W / norms[None, :]
    mixing = np.linalg.pinv(unmixing)
    src = unmixing @ (V[:n] @ X.T)
    var = (mixing ** 2).sum(axis=0) * (src ** 2).sum(axis=1) / (n * T - 1)
    srt = np.argsort(var)[::-1]
    return dict(unmixing=unmixing[srt, :], mixing=mixing[:, srt], pca_components=V, pca_mean=mean,
                pre_whitener=np.full((C, 1), pre), n_components=n, pca_explained_variance=ev, n_iter=int(ica.n_iter_),
                exclude=[], W=W[srt, :], Z=Z)


def fastica_stats(z, W, fun="logcosh", alpha=1.0):
    """One ``_ica_par`` pass in float64: (sum_t g(y) z^T, sum_t g'(y)), y = W z, z [n, T]."""
    z = np.asarray(z, dtype=np.float64)
    y = np.asarray(W, dtype=np.float64) @ z
    if fun == "logcosh":
        g = np.tanh(alpha * y)
        gp = alpha * (1.0 - g ** 2)
    elif fun == "exp":
        e = np.exp(-0.5 * y ** 2)
        g, gp = y * e, (1.0 - y ** 2) * e
    elif fun == "cube":
        g, gp = y ** 3, 3.0 * y ** 2
    else:
        g, gp = y, y
    return g @ z.T, gp.sum(axis=1)


def find_outliers(X, threshold=3.0, max_iter=2):
    """mne.preprocessing.bads._find_outliers (tail=0): iterative |zscore| > threshold on the unmasked entries."""
    from scipy.stats import zscore
    X = np.asarray(X, dtype=np.float64)
    mask = np.zeros(len(X), dtype=bool)
    for _ in range(max_iter):
        Xm = np.ma.masked_array(X, mask)
        this_z = np.abs(zscore(Xm))
        local_bad = np.asarray(np.ma.filled(this_z > threshold, False))
        mask = np.max([mask, local_bad], 0)
        if not np.any(local_bad):
            break
    return np.where(mask)[0]


def find_bads_eog(ica: dict, data, eog, sfreq, threshold=3.0, l_freq=1.0, h_freq=10.0):
    """``ica.find_bads_eog`` (training/preprocess.py:108): Pearson correlation of the 1-10 Hz sources with the 1-10 Hz
    EOG signal, outliers by iterative z-score, ordered by |score|."""
    src = ica_sources(data, ica["unmixing"], ica["pca_components"], ica["pca_mean"], ica["pre_whitener"], ica["n_components"])
    src = filter_data(src, sfreq, l_freq, h_freq)
    tgt = filter_data(np.asarray(eog, dtype=np.float64)[None, :], sfreq, l_freq, h_freq)[0]
    scores = np.array([np.corrcoef(s, tgt)[0, 1] for s in src])
    bad = find_outliers(scores, threshold)
    bad = bad[np.argsort(np.abs(scores[bad]))[::-1]]
    return [int(i) for i in bad], scores


def find_bads_ecg(ica: dict, data, ecg, sfreq, threshold="auto", l_freq=8.0, h_freq=16.0):
    """``ica.find_bads_ecg(raw, ch_name, method='correlation', threshold='auto')`` (multi-train/withpsd.py:145): MNE 1.5 routes
    the correlation method through the same ``_find_bads_ch`` as the EOG scoring, with the 8-16 Hz band and, for
    threshold='auto' with the z-score measure, a threshold of 3.0.  Restated from MNE (parity unpinned)."""
    thr = 3.0 if threshold == "auto" else float(threshold)
    return find_bads_eog(ica, data, ecg, sfreq, threshold=thr, l_freq=l_freq, h_freq=h_freq)


# --------------------------------------------------------------------------------------
# 8(f)-4: annotations -> trials -> vigilance labels (training/preprocess.py:156-259), line-by-line semantics
# --------------------------------------------------------------------------------------

def vigilance_trials(annotations, sfreq, deviation_codes=(251, 252), response_code=253, rt_filter=0.1, percentile=5,
                     lock="response"):
    """Returns (events [E, 3] int, labels [E] str, rts [E], alert_rt_baseline) exactly as the script builds them.
    multi-train/withpsd.py:159-206 is the same logic with ``percentile=10`` and stimulus-locked events (``lock='deviation'``)."""
    ev = []
    for onset, desc in annotations:                                     # :157-165
        try:
            code = int(float(str(desc).strip()))
        except (ValueError, TypeError):
            continue
        if code in deviation_codes or code == response_code:
            ev.append((float(onset), code))
    ev.sort(key=lambda e: e[0])                                          # :173 (stable)
    trials, last = [], None
    for t, code in ev:                                                   # :180-196
        if code in deviation_codes:
            last = (t, code)
        elif code == response_code:
            if last is not None:
                rt = t - last[0]
                if rt >= 0:
                    trials.append((int(round((t if lock == "response" else last[0]) * sfreq)), rt))
                last = None
    trials = [tr for tr in trials if tr[1] >= rt_filter]                 # :204
    rts = np.asarray([tr[1] for tr in trials], dtype=np.float64)
    if rts.size == 0:
        return np.zeros((0, 3), dtype=np.int64), np.asarray([], dtype="<U10"), rts, float("nan")
    base = float(np.median(rts)) if rts.size < 5 else float(np.percentile(rts, percentile))      # :216-225
    labels = []
    for rt in rts:                                                       # :233-236
        labels.append("alert" if rt < 1.5 * base else ("drowsy" if rt > 2.5 * base else "transition"))
    ids = {"alert": 1, "transition": 2, "drowsy": 3}
    events = np.asarray([[tr[0], 0, ids[lab]] for tr, lab in zip(trials, labels)], dtype=np.int64).reshape(-1, 3)
    return events, np.asarray(labels, dtype="<U10"), rts, base


# --------------------------------------------------------------------------------------
# a4-a6: epoch post-processing
# --------------------------------------------------------------------------------------

def n_baseline_samples(tmin: float, sfreq: float, b_end: float = 0.0) -> int:
    """Samples with tmin <= t <= b_end for ``baseline=(None, b_end)`` (training/preprocess.py:262-271)."""
    return int(round((b_end - tmin) * sfreq)) + 1


def baseline_correct(x, n_baseline: int):
    """``mne.Epochs(..., baseline=(None, 0))``: subtract the mean of the first n_baseline samples."""
    x = np.asarray(x, dtype=np.float64)
    if n_baseline <= 0:
        return x.copy()
    return x - x[..., :n_baseline].mean(axis=-1, keepdims=True)


def ptp_keep_mask(x, thresh: float):
    """``reject=dict(eeg=thresh)`` + ``drop_bad`` (multi-train/withpsd.py:44-46,213-219).

    An epoch is dropped when any channel's peak-to-peak exceeds ``thresh`` (strict >).
    """
    x = np.asarray(x, dtype=np.float64)
    ptp = x.max(axis=-1) - x.min(axis=-1)
    return ~(ptp > thresh).any(axis=-1)


def zscore_epoch_channel(x, eps: float = 1e-6):
    """training/preprocess.py:282-287: (x - mean_t) / (std_t + eps), population std."""
    x = np.asarray(x, dtype=np.float64)
    mean = x.mean(axis=-1, keepdims=True)
    std = x.std(axis=-1, keepdims=True)
    return (x - mean) / (std + eps)


def scaler_mean(x):
    """``mne.decoding.Scaler(scalings='mean')`` (multi-train/withpsd.py:267-270).

    sklearn ``StandardScaler`` per channel over all epochs x time: (x - mean) / std,
    population std, std == 0 -> 1.
    """
    x = np.asarray(x, dtype=np.float64)
    mean = x.mean(axis=(0, 2), keepdims=True)
    std = x.std(axis=(0, 2), keepdims=True)
    std = np.where(std == 0.0, 1.0, std)
    return (x - mean) / std


# --------------------------------------------------------------------------------------
# a7: Welch PSD (MNE ``psd_array_welch`` -> scipy.signal.spectrogram)
# reference call sites: multi-train/withpsd.py:236-250 ; preprocessing_unsupervised.py:43
# --------------------------------------------------------------------------------------

def welch_psd(x, sfreq: float, fmin: float = 0.0, fmax: float = np.inf, n_fft: int = 256,
              n_overlap: int = 0, n_per_seg: Optional[int] = None, window="hamming",
              average: Optional[str] = "mean") -> Tuple[np.ndarray, np.ndarray]:
    x = np.asarray(x, dtype=np.float64)
    T = x.shape[-1]
    n_per_seg = n_fft if n_per_seg is None else n_per_seg
    if n_fft > T:
        raise ValueError("n_fft is larger than the number of samples")
    if n_per_seg > n_fft or n_overlap >= n_per_seg:
        raise ValueError("bad n_per_seg / n_overlap")
    freqs = np.fft.rfftfreq(n_fft, 1.0 / sfreq)
    mask = (freqs >= fmin) & (freqs <= fmax)
    _, _, spect = _sig.spectrogram(x, fs=sfreq, window=window, nperseg=n_per_seg,
                                   noverlap=n_overlap, nfft=n_fft, detrend="constant",
                                   scaling="density", mode="psd", return_onesided=True)
    spect = spect[..., mask, :]
    if average == "mean":
        psd = np.nanmean(spect, axis=-1)
    elif average == "median":
        n = spect.shape[-1]
        ii = np.arange(1, n + 1)
        bias = 1.0 + np.sum((-1.0) ** (ii[1:] + 1) / ii[1:]) if n > 1 else 1.0  # scipy _median_bias
        psd = np.nanmedian(spect, axis=-1) / bias
    elif average is None:
        psd = spect
    else:
        raise ValueError("average must be 'mean', 'median' or None")
    return psd, freqs[mask]


# --------------------------------------------------------------------------------------
# a8 / a9: band power, ratios, rule labels (preprocessing_unsupervised.py:45-84)
# --------------------------------------------------------------------------------------

def band_powers(psd, freqs, already_db: bool = False):
    """psd[..., C, F] -> (theta, alpha, beta, beta/theta, alpha/beta, theta/beta).

    Power = mean over ALL channels and band bins of the dB values; ratios are ratios of
    those dB means (sic, preprocessing_unsupervised.py:59-61).
    """
    psd = np.asarray(psd, dtype=np.float64)
    freqs = np.asarray(freqs, dtype=np.float64)
    log_psd = psd if already_db else 10.0 * np.log10(psd)
    th = (freqs >= 4) & (freqs < 8)
    al = (freqs >= 8) & (freqs < 13)
    be = (freqs >= 13) & (freqs < 30)
    theta = log_psd[..., :, th].mean(axis=(-2, -1))
    alpha = log_psd[..., :, al].mean(axis=(-2, -1))
    beta = log_psd[..., :, be].mean(axis=(-2, -1))
    return theta, alpha, beta, beta / theta, alpha / beta, theta / beta


def classify_eeg_state(theta, alpha, beta, drowsy_threshold: float) -> int:
    """preprocessing_unsupervised.py:70-84."""
    beta_theta = beta / theta
    alpha_beta = alpha / beta
    theta_beta = theta / beta
    if beta_theta < 1.0:
        return 0
    elif beta_theta > 2.0:
        return 1
    elif alpha_beta > 1.2:
        return 2
    elif theta_beta > drowsy_threshold:
        return 3
    return 4


def classify_eeg_state_variant(theta, alpha, beta) -> int:
    """failed/label.py:13-34 (last test is theta_power > beta_power)."""
    beta_theta = beta / theta
    alpha_beta = alpha / beta
    if beta_theta < 1.0:
        return 0
    elif beta_theta > 2.0:
        return 1
    elif alpha_beta > 1.2:
        return 2
    elif theta > beta:
        return 3
    return 4


# --------------------------------------------------------------------------------------
# a10-a14: VQ-VAE encoder + quantizer (training/model.py:84-158, 169-177, 206-214), torch-CPU fp32
# --------------------------------------------------------------------------------------

def encoder_forward(x, weights: dict):
    """``VQVAE.encoder`` (training/model.py:169-177).  weights: encoder.{0,2,4,6}.{weight,bias}."""
    import torch
    import torch.nn.functional as F
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)  # noqa: E731
    with torch.no_grad():
        z = torch.as_tensor(np.asarray(x), dtype=torch.float32)
        z = F.relu(F.conv1d(z, t(weights["encoder.0.weight"]), t(weights["encoder.0.bias"]), stride=2, padding=1))
        z = F.relu(F.conv1d(z, t(weights["encoder.2.weight"]), t(weights["encoder.2.bias"]), stride=2, padding=1))
        z = F.relu(F.conv1d(z, t(weights["encoder.4.weight"]), t(weights["encoder.4.bias"]), stride=1, padding=1))
        z = F.conv1d(z, t(weights["encoder.6.weight"]), t(weights["encoder.6.bias"]), stride=1)
    return z.numpy()


def vq_distances(flat_input, codebook):
    """training/model.py:90-92 / 154-156, in torch-CPU fp32, same expression order."""
    import torch
    with torch.no_grad():
        fi = torch.as_tensor(np.asarray(flat_input), dtype=torch.float32)
        w = torch.as_tensor(np.asarray(codebook), dtype=torch.float32)
        d = (torch.sum(fi ** 2, dim=1, keepdim=True) + torch.sum(w ** 2, dim=1)
             - 2 * torch.matmul(fi, w.t()))
    return d.numpy()


def vq_argmin(flat_input, codebook) -> np.ndarray:
    """``get_code_indices`` (training/model.py:152-158): first minimum wins."""
    import torch
    d = torch.from_numpy(vq_distances(flat_input, codebook))
    return torch.argmin(d, dim=1).numpy().astype(np.int64)


def vq_distances_f64(flat_input, codebook) -> np.ndarray:
    x = np.asarray(flat_input, dtype=np.float64)
    e = np.asarray(codebook, dtype=np.float64)
    return (x * x).sum(1, keepdims=True) + (e * e).sum(1)[None, :] - 2.0 * (x @ e.T)


def vq_near_tie_ok(flat_input, codebook, idx_build, idx_ref) -> np.ndarray:
    """SURVEY.md 8(a) near-tie rule.  True where a row is either equal or a documented near-tie.

    A mismatch at row i is a near-tie iff |d64[i,k_build] - d64[i,k_ref]| <=
    8 * eps_fp32 * (|x_i|^2 + |e|^2 + 2|x_i . e|) for both codes.  Exact duplicates of a
    codebook row are NOT near-ties: the lowest index is required.
    """
    x = np.asarray(flat_input, dtype=np.float64)
    e = np.asarray(codebook, dtype=np.float64)
    idx_build = np.asarray(idx_build).reshape(-1)
    idx_ref = np.asarray(idx_ref).reshape(-1)
    ok = idx_build == idx_ref
    bad = np.nonzero(~ok)[0]
    eps = 2.0 ** -23
    for i in bad:
        kb, kr = int(idx_build[i]), int(idx_ref[i])
        if np.array_equal(e[kb], e[kr]):
            ok[i] = False  # exact duplicate rows: lowest index required
            continue
        xx = float(x[i] @ x[i])
        db = xx + e[kb] @ e[kb] - 2.0 * (x[i] @ e[kb])
        dr = xx + e[kr] @ e[kr] - 2.0 * (x[i] @ e[kr])
        tol_b = 8 * eps * (xx + e[kb] @ e[kb] + 2.0 * abs(x[i] @ e[kb]))
        tol_r = 8 * eps * (xx + e[kr] @ e[kr] + 2.0 * abs(x[i] @ e[kr]))
        ok[i] = abs(db - dr) <= min(tol_b, tol_r)
    return ok


def vq_tokens_explained(z_ref, z_build, codebook, tok_build, tok_ref) -> np.ndarray:
    """Reference-side form of the near-tie rule, for comparing a build's tokens with the REFERENCE's tokens when the two
    quantizers saw slightly different inputs (the build's encoder output z_build differs from the reference's z_ref by fp32
    rounding of the stages in front).  True where a row's tokens are equal or the mismatch is explained:

    the reference chose b on z_ref, the build chose a on z_build = z_ref + dz.  Exactly,
        d(z_build, a) - d(z_build, b) = [d(z_ref, a) - d(z_ref, b)] - 2 dz.(e_a - e_b),
    so both choices are consistent iff the reference-side gap g = d64(z_ref, a) - d64(z_ref, b) satisfies
        |g| <= tol_fp32 + 2 |dz.(e_a - e_b)|,
    tol_fp32 = SURVEY.md 8(a)'s 8 eps (|z|^2 + |e|^2 + 2|z.e|) (the smaller of the two codes' values).  With z_build = None
    (the build exposes no z) dz = 0: the plain near-tie rule on the reference's z.  Exact duplicate code rows are never
    explained: the lowest index is required."""
    zr = np.asarray(z_ref, dtype=np.float64)
    zb = zr if z_build is None else np.asarray(z_build, dtype=np.float64)
    e = np.asarray(codebook, dtype=np.float64)
    a_all = np.asarray(tok_build).reshape(-1)
    b_all = np.asarray(tok_ref).reshape(-1)
    ok = a_all == b_all
    eps = 2.0 ** -23
    for i in np.nonzero(~ok)[0]:
        a, b = int(a_all[i]), int(b_all[i])
        if np.array_equal(e[a], e[b]):
            continue
        zz = float(zr[i] @ zr[i])
        da = zz + e[a] @ e[a] - 2.0 * (zr[i] @ e[a])
        db = zz + e[b] @ e[b] - 2.0 * (zr[i] @ e[b])
        tol = 8 * eps * min(zz + e[a] @ e[a] + 2.0 * abs(zr[i] @ e[a]), zz + e[b] @ e[b] + 2.0 * abs(zr[i] @ e[b]))
        ok[i] = abs(da - db) <= tol + 2.0 * abs((zb[i] - zr[i]) @ (e[a] - e[b]))
    return ok


def vq_forward(inputs, codebook, commitment_cost: float = 0.25):
    """``VectorQuantizerEMA.forward`` in eval mode (training/model.py:84-149).

    inputs [B,S,D] -> (loss, quantized [B,S,D], perplexity, indices int64 [B,S], counts [K]).
    """
    import torch
    import torch.nn.functional as F
    with torch.no_grad():
        inp = torch.as_tensor(np.asarray(inputs), dtype=torch.float32)
        w = torch.as_tensor(np.asarray(codebook), dtype=torch.float32)
        K, D = w.shape
        flat = inp.reshape(-1, D)
        d = (torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(w ** 2, dim=1)
             - 2 * torch.matmul(flat, w.t()))
        idx = torch.argmin(d, dim=1).unsqueeze(1)
        enc = torch.zeros(idx.shape[0], K)
        enc.scatter_(1, idx, 1)
        quant = torch.matmul(enc, w).view(inp.shape)
        e_latent = F.mse_loss(quant, inp)
        q_latent = F.mse_loss(quant, inp)
        loss = q_latent + commitment_cost * e_latent
        quant = inp + (quant - inp)  # straight-through estimator arithmetic (training/model.py:139)
        avg_probs = torch.mean(enc, dim=0)
        perplexity = torch.exp(-torch.sum(avg_probs * torch.log(avg_probs + 1e-10)))
        counts = enc.sum(0).to(torch.int64)
    return (float(loss), quant.numpy(), float(perplexity),
            idx.view(inp.shape[0], -1).numpy().astype(np.int64), counts.numpy())


def perplexity_from_counts(counts) -> float:
    """training/model.py:143-144 evaluated from integer usage counts (fp32 arithmetic as torch)."""
    c = np.asarray(counts, dtype=np.float64)
    p = (c / c.sum()).astype(np.float32)
    return float(np.exp(-np.sum(p * np.log(p + np.float32(1e-10)), dtype=np.float32)))


def vqvae_encode(x, weights: dict):
    """``VQVAE.encode`` (training/model.py:206-214): encoder -> permute -> flatten -> argmin -> [B, L//4]."""
    z = encoder_forward(x, weights)                       # [B, D, S]
    B, D, S = z.shape
    flat = np.ascontiguousarray(z.transpose(0, 2, 1)).reshape(-1, D)
    idx = vq_argmin(flat, weights["vq_layer.embedding.weight"])
    return idx.reshape(B, S), z


def decoder_forward(q, weights: dict):
    """``VQVAE.decoder`` (training/model.py:184-192): four ConvTranspose1d layers, ReLU after the first three.
    weights: decoder.{0,2,4,6}.{weight,bias}.  q [B, D, S] -> [B, C, 4 S + 3]."""
    import torch
    import torch.nn.functional as F
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)  # noqa: E731
    w = lambda i, n: t(weights[f"decoder.{i}.{n}"])  # noqa: E731
    with torch.no_grad():
        h = torch.as_tensor(np.asarray(q), dtype=torch.float32)
        h = F.relu(F.conv_transpose1d(h, w(0, "weight"), w(0, "bias"), stride=1))
        h = F.relu(F.conv_transpose1d(h, w(2, "weight"), w(2, "bias"), stride=1, padding=1))
        h = F.relu(F.conv_transpose1d(h, w(4, "weight"), w(4, "bias"), stride=2, padding=1, output_padding=1))
        h = F.conv_transpose1d(h, w(6, "weight"), w(6, "bias"), stride=2, padding=1, output_padding=1)
    return h.numpy()


def vqvae_forward(x, weights: dict, commitment_cost: float = 0.25):
    """``VQVAE.forward`` in eval mode (training/model.py:195-203): encoder -> permute -> quantizer forward -> permute ->
    decoder.  Returns (vq_loss, x_recon [B, C, 4 S + 3], perplexity, indices int64 [B, S])."""
    z = encoder_forward(x, weights)
    zp = np.ascontiguousarray(z.transpose(0, 2, 1))
    loss, quant, perp, idx, _ = vq_forward(zp, weights["vq_layer.embedding.weight"], commitment_cost)
    x_recon = decoder_forward(np.ascontiguousarray(quant.transpose(0, 2, 1)), weights)
    return loss, x_recon, perp, idx


def ema_update(codebook_state: dict, flat_input, idx, decay: float = 0.99, epsilon: float = 1e-5):
    """Training-mode EMA update (training/model.py:103-126), fp32 torch arithmetic.

    codebook_state: {'ema_cluster_size' [K], 'ema_w' [K,D]} -> new (ema_cluster_size, ema_w, weight).
    """
    import torch
    with torch.no_grad():
        cs = torch.as_tensor(np.asarray(codebook_state["ema_cluster_size"]), dtype=torch.float32)
        ew = torch.as_tensor(np.asarray(codebook_state["ema_w"]), dtype=torch.float32)
        flat = torch.as_tensor(np.asarray(flat_input), dtype=torch.float32)
        K = cs.shape[0]
        enc = torch.zeros(flat.shape[0], K)
        enc.scatter_(1, torch.as_tensor(np.asarray(idx), dtype=torch.int64).view(-1, 1), 1)
        cs = cs * decay + (1 - decay) * torch.sum(enc, 0)
        n = torch.sum(cs)
        cs = (cs + epsilon) / (n + K * epsilon) * n
        dw = torch.matmul(enc.t(), flat)
        ew = ew * decay + (1 - decay) * dw
        w = ew / cs.unsqueeze(1)
    return cs.numpy(), ew.numpy(), w.numpy()


# --------------------------------------------------------------------------------------
# whole pipeline (per-epoch formulation)
# --------------------------------------------------------------------------------------

def pipeline(x, sfreq: float, taps, A, b, n_baseline: int, ptp_thresh: float,
             weights: dict, n_fft: Optional[int] = None, fft_filter: bool = False,
             zscore_eps: float = 1e-6):
    """filter -> ICA affine -> baseline -> ptp keep -> Welch (on cleaned, un-standardised epochs,
    multi-train/withpsd.py:236) -> band powers -> z-score (training/preprocess.py:282-287)
    -> VQ-VAE encode.  Returns a dict of float64 / int64 arrays."""
    x = np.asarray(x, dtype=np.float64)
    y = fir_zero_phase(x, taps, fft=fft_filter)
    y = np.einsum("oc,ect->eot", np.asarray(A, dtype=np.float64), y) + np.asarray(b, dtype=np.float64)[None, :, None]
    y = baseline_correct(y, n_baseline)
    keep = ptp_keep_mask(y, ptp_thresh)
    n_fft = int(sfreq) if n_fft is None else n_fft
    psd, freqs = welch_psd(y, sfreq, 4.0, 30.0, n_fft=n_fft, n_overlap=n_fft // 2, window="hann")
    bp = np.stack(band_powers(psd, freqs), axis=-1)
    std = zscore_epoch_channel(y, zscore_eps)
    tokens, z = vqvae_encode(std.astype(np.float32), weights)
    return dict(filtered=y, keep=keep, psd=psd, freqs=freqs, bandpower=bp, standardized=std,
                tokens=tokens, z=z)
