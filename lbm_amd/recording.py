"""Recording-level edges of the hot path: EEGLAB ``.set``/``.fdt`` reader, annotation -> trial ->
vigilance-label logic, and the per-recording driver that strings the GPU stages together in the reference's order.

Mirrors ``training/preprocess.py``: ``read_raw_eeglab`` (:48), event extraction (:156-169), deviation -> response
pairing and reaction times (:171-199), the RT filter (:201-207), 5th-percentile baseline and alert / transition /
drowsy thresholds (:213-237), MNE event array (:241-259), then filter -> notch -> ICA fit -> ICA apply -> epochs ->
baseline -> per-epoch z-score (:77-82, :91-146, :262-287).  Host logic here is plain NumPy; every pass over samples
runs through the C ABI kernels (``filter_data``, ``notch_filter``, ``ica_fit``, ``ica_apply``, ``make_epochs``,
``epoch_post``).
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .constants import (BASELINE, DEVIATION_CODES, FILTER_HFREQ, FILTER_LFREQ, ICA_METHOD, ICA_N_COMPONENTS, ICA_RANDOM_STATE,
                        NOTCH_FREQ, RESPONSE_CODE, RT_FILTER_THRESHOLD, TMAX, TMIN, ZSCORE_EPSILON)

VIGILANCE_EVENT_IDS = {"alert": 1, "transition": 2, "drowsy": 3}     # training/preprocess.py:244


# ------------------------------------------------------------------------------------------------ EEGLAB reader
def _field(obj, name, default=None):
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, name, default)


def read_raw_eeglab(path: str) -> dict:
    """``mne.io.read_raw_eeglab(path, preload=True)`` for MATLAB v5-v7 ``.set`` files (training/preprocess.py:48).

    Returns ``dict(data [C, T] float64 volts, sfreq, ch_names, annotations=[(onset_s, description), ...])``.
    EEGLAB stores microvolts (MNE scales by 1e-6) and 1-based event latencies (onset = (latency - 1) / srate).
    The samples come from the ``.set`` itself or from the float32 ``.fdt`` file it names (channels fastest).
    v7.3 (HDF5) files are not supported here.
    """
    from scipy.io import loadmat
    mat = loadmat(path, squeeze_me=True, struct_as_record=False)
    eeg = mat["EEG"] if "EEG" in mat else mat
    nbchan, pnts = int(_field(eeg, "nbchan")), int(_field(eeg, "pnts"))
    trials = int(_field(eeg, "trials", 1) or 1)
    if trials != 1:
        raise ValueError("read_raw_eeglab: epoched .set files are not continuous recordings")
    sfreq = float(_field(eeg, "srate"))
    data = _field(eeg, "data")
    if isinstance(data, str):
        fdt = os.path.join(os.path.dirname(os.path.abspath(path)), os.path.basename(data))
        data = np.fromfile(fdt, dtype="<f4").reshape(pnts, nbchan).T
    data = np.asarray(data, dtype=np.float64).reshape(nbchan, pnts) * 1e-6
    chanlocs = np.atleast_1d(_field(eeg, "chanlocs", []))
    ch_names = [str(_field(c, "labels", f"EEG {i:03d}")) for i, c in enumerate(chanlocs)] or [f"EEG {i:03d}" for i in range(nbchan)]
    annotations = []
    for ev in np.atleast_1d(_field(eeg, "event", [])):
        lat, typ = _field(ev, "latency"), _field(ev, "type")
        if lat is None or typ is None:
            continue
        annotations.append(((float(lat) - 1.0) / sfreq, str(typ)))
    return dict(data=data, sfreq=sfreq, ch_names=ch_names, annotations=annotations)


def pick_eeg_channels(ch_names: Sequence[str]) -> List[int]:
    """Channels kept as EEG: everything but the vehicle position / deviation traces (training/preprocess.py:54)."""
    return [i for i, ch in enumerate(ch_names) if "position" not in ch.lower() and "vehicle" not in ch.lower()]


# ------------------------------------------------------------------------------------------------ events -> labels
def extract_events(annotations: Iterable[Tuple[float, str]], deviation_codes=DEVIATION_CODES, response_code=RESPONSE_CODE):
    """(times_s, codes) of the deviation / response annotations, time-sorted (stable), training/preprocess.py:156-173."""
    t, c = [], []
    for onset, desc in annotations:
        try:
            code = int(float(desc))
        except (ValueError, TypeError):
            continue
        if code in deviation_codes or code == response_code:
            t.append(float(onset))
            c.append(code)
    t, c = np.asarray(t, dtype=np.float64), np.asarray(c, dtype=np.int64)
    order = np.argsort(t, kind="stable")
    return t[order], c[order]


def pair_trials(times, codes, sfreq: float, deviation_codes=DEVIATION_CODES, response_code=RESPONSE_CODE) -> Dict[str, np.ndarray]:
    """Each response is paired with the last unanswered deviation before it (training/preprocess.py:177-196)."""
    times, codes = np.asarray(times, dtype=np.float64), np.asarray(codes, dtype=np.int64)
    samples = np.round(times * sfreq).astype(np.int64)                               # :175
    out = {k: [] for k in ("dev_code", "dev_time", "dev_sample", "response_time", "response_sample", "local_rt_sec")}
    last = -1
    for i in range(len(times)):
        if codes[i] in deviation_codes:
            last = i
        elif codes[i] == response_code:
            if last >= 0:
                rt = times[i] - times[last]
                if rt >= 0:
                    for k, v in (("dev_code", codes[last]), ("dev_time", times[last]), ("dev_sample", samples[last]),
                                 ("response_time", times[i]), ("response_sample", samples[i]), ("local_rt_sec", rt)):
                        out[k].append(v)
                last = -1
    return {k: np.asarray(v, dtype=np.float64 if "time" in k or "rt" in k else np.int64) for k, v in out.items()}


def vigilance_labels(local_rt_sec, rt_filter: float = RT_FILTER_THRESHOLD, percentile: float = 5.0):
    """RT filter, alert baseline and labels (training/preprocess.py:201-237; multi-train/withpsd.py:184-194 uses the
    10th percentile: ``percentile=10``).

    Returns ``(keep mask over the trials, labels of the kept trials, alert_rt_baseline)``: baseline = 5th percentile of
    the kept RTs (median when fewer than 5), alert < 1.5 * baseline, drowsy > 2.5 * baseline, else transition.
    """
    rt = np.asarray(local_rt_sec, dtype=np.float64)
    keep = rt >= rt_filter
    kept = rt[keep]
    if kept.size == 0:
        return keep, np.asarray([], dtype="<U10"), float("nan")
    base = float(np.median(kept)) if kept.size < 5 else float(np.percentile(kept, percentile))
    lab = np.where(kept < 1.5 * base, "alert", np.where(kept > 2.5 * base, "drowsy", "transition"))
    return keep, lab.astype("<U10"), base


def mne_events(trials: Dict[str, np.ndarray], keep: np.ndarray, labels: np.ndarray, lock: str = "response") -> np.ndarray:
    """``[sample, 0, id]`` rows: response-locked (training/preprocess.py:251-259) or, ``lock='deviation'``,
    stimulus-locked (multi-train/withpsd.py:204)."""
    ids = np.asarray([VIGILANCE_EVENT_IDS[str(lab)] for lab in labels], dtype=np.int64)
    col = trials["response_sample" if lock == "response" else "dev_sample"][keep]
    return np.stack([col, np.zeros_like(ids), ids], axis=1) if ids.size else np.zeros((0, 3), dtype=np.int64)


# ------------------------------------------------------------------------------------------------ driver
def preprocess_recording(data, sfreq: float, annotations, *, l_freq: float = FILTER_LFREQ, h_freq: float = FILTER_HFREQ,
                         notch: Optional[float] = NOTCH_FREQ, ica: bool = True, n_components=ICA_N_COMPONENTS,
                         random_state=ICA_RANDOM_STATE, ica_fit_params=None, eog=None, exclude: Sequence[int] = (), tmin: float = TMIN, tmax: float = TMAX,
                         rt_filter: float = RT_FILTER_THRESHOLD) -> dict:
    """The flow of training/preprocess.py on one continuous ``[C, T]`` recording, every sample pass on the GPU.

    ``eog``: optional ``[T]`` EOG signal for ``find_bads_eog`` (the script's automatic detection, :104-117; without an
    EOG channel MNE raises and the script continues with no automatic exclusions); ``exclude`` adds manual indices.
    Returns ``dict(epochs [E, C, Tw] standardized float32 CUDA, labels [E], events [E, 3], trials, ica, alert_rt_baseline)``.
    """
    import torch
    from . import _lib
    from .epochs import epoch_post, make_epochs, n_baseline_samples
    from .fastica import find_bads_eog, ica_fit
    from .filter import filter_data, notch_filter
    from .ica import ica_apply

    x = _lib.dev_f32(data)
    if x.dim() != 2:
        raise ValueError("preprocess_recording: data must be [channels, samples]")
    x = filter_data(x, sfreq, l_freq, h_freq)                                          # :77
    if notch is not None and notch < sfreq / 2.0:
        x = notch_filter(x, sfreq, notch)                                              # :82
    state = None
    if ica:
        state = ica_fit(x, n_components=n_components, method=ICA_METHOD, random_state=random_state,
                        fit_params=ica_fit_params)                                         # :91-98
        bad = [int(i) for i in exclude]
        if eog is not None:
            idx, _ = find_bads_eog(state, x, eog, sfreq, threshold=3.0)                # :108
            bad.extend(idx)
        state["exclude"] = sorted(set(bad))
        keys = ("unmixing", "mixing", "pca_components", "pca_mean", "pre_whitener", "n_components")
        x = ica_apply(x.unsqueeze(0), exclude=state["exclude"], **{k: state[k] for k in keys})[0]       # :144
    times, codes = extract_events(annotations)
    trials = pair_trials(times, codes, sfreq)
    keep, labels, base = vigilance_labels(trials["local_rt_sec"], rt_filter)
    events = mne_events(trials, keep, labels)
    ep, kept = make_epochs(x, events[:, 0], tmin, tmax, sfreq)                          # :262-271 (windows inside the recording)
    labels, events = labels[kept], events[kept]
    nb = n_baseline_samples(tmin, sfreq, 0.0 if BASELINE[1] is None else float(BASELINE[1]))
    _, std, _ = epoch_post(ep, n_baseline=nb, reject=None, standardize=True, eps=ZSCORE_EPSILON)        # :262 baseline, :279-287
    return dict(epochs=std, labels=labels, events=events, trials=trials, ica=state, alert_rt_baseline=base)


def preprocess_recording_withpsd(data, sfreq: float, annotations, *, l_freq: float = FILTER_LFREQ, h_freq: float = FILTER_HFREQ,
                                 notch: Optional[float] = NOTCH_FREQ, ica: bool = True, n_components=ICA_N_COMPONENTS,
                                 random_state=ICA_RANDOM_STATE, ica_fit_params=None, eog=None, ecg=None, exclude: Sequence[int] = (),
                                 tmin: float = TMIN, tmax: float = TMAX, rt_filter: float = RT_FILTER_THRESHOLD,
                                 reject: Optional[float] = 150e-6, psd_fmin: float = FILTER_LFREQ, psd_fmax: float = FILTER_HFREQ,
                                 psd_window: str = "hann") -> dict:
    """The per-file body of multi-train/withpsd.py (:96-283) on one continuous ``[C, T]`` recording: filter -> notch ->
    ICA fit, EOG scoring (``eog``: the [T] signal of an EOG / frontal channel, or [n, T] for several: the union of what each
    finds, :136-139) and ECG scoring (``ecg``: the [T] signal of the ECG channel, ``find_bads_ecg(method='correlation',
    threshold='auto')``, :145) -- both channels pass the same band-pass / notch first, as the script filters
    ``picks=['eeg', 'eog', 'ecg']`` (:118-119) --, apply -> trials (10th-percentile baseline, :192) -> stimulus-locked epochs (:204) with baseline and
    peak-to-peak rejection (:213-219) -> Welch PSD per epoch (:236-250, n_fft = int(sfreq), half overlap, Hann) ->
    ``Scaler(scalings='mean')`` over the kept epochs (:267-270).

    Returns ``dict(time_epochs [E, C, Tw], psd_epochs [E, C, F], psd_freqs [F], labels [E], events, keep, ica)``;
    ``save_withpsd_outputs`` writes them under the reference's file names.
    """
    import torch
    from . import _lib
    from .epochs import Scaler, epoch_post, make_epochs, n_baseline_samples
    from .fastica import find_bads_ecg, find_bads_eog, ica_fit
    from .filter import filter_data, notch_filter
    from .ica import ica_apply
    from .psd import psd_array_welch

    x = _lib.dev_f32(data)
    if x.dim() != 2:
        raise ValueError("preprocess_recording_withpsd: data must be [channels, samples]")
    n_eeg = x.shape[0]
    aux = [_lib.dev_f32(a, x.device).reshape(-1, x.shape[1]) for a in (eog, ecg) if a is not None]
    n_eog = aux[0].shape[0] if eog is not None else 0
    if aux and ica:                                                                    # picks=['eeg', 'eog', 'ecg']: one filter pass
        x = torch.cat([x] + aux, dim=0)
    x = filter_data(x, sfreq, l_freq, h_freq)                                          # :118
    if notch is not None and notch < sfreq / 2.0:
        x = notch_filter(x, sfreq, notch)                                              # :119
    x, aux_f = x[:n_eeg], x[n_eeg:]
    state = None
    if ica:
        state = ica_fit(x, n_components=n_components, method=ICA_METHOD, random_state=random_state, fit_params=ica_fit_params)
        bad = [int(i) for i in exclude]
        for ch in range(n_eog):
            idx, _ = find_bads_eog(state, x, aux_f[ch], sfreq)                         # :136-139 (default threshold)
            bad.extend(idx)
        if ecg is not None:
            idx, _ = find_bads_ecg(state, x, aux_f[n_eog], sfreq, method="correlation", threshold="auto")  # :145
            bad.extend(idx)
        state["exclude"] = sorted(set(bad))
        keys = ("unmixing", "mixing", "pca_components", "pca_mean", "pre_whitener", "n_components")
        x = ica_apply(x.unsqueeze(0), exclude=state["exclude"], **{k: state[k] for k in keys})[0]       # :152
    times, codes = extract_events([(t, str(d).strip()) for t, d in annotations])       # :161 strips the description
    trials = pair_trials(times, codes, sfreq)
    keep_rt, labels, base = vigilance_labels(trials["local_rt_sec"], rt_filter, percentile=10.0)
    events = mne_events(trials, keep_rt, labels, lock="deviation")
    ep, inside = make_epochs(x, events[:, 0], tmin, tmax, sfreq)
    labels, events = labels[inside], events[inside]
    nb = n_baseline_samples(tmin, sfreq, 0.0 if BASELINE[1] is None else float(BASELINE[1]))
    base_ep, _, keep = epoch_post(ep, n_baseline=nb, reject=reject, standardize=False)
    if keep is not None:
        base_ep, labels, events = base_ep[keep], labels[keep.cpu().numpy()], events[keep.cpu().numpy()]
    n_fft = int(sfreq * 1.0)                                                            # :231
    psd, freqs = psd_array_welch(base_ep, sfreq, psd_fmin, psd_fmax, n_fft=n_fft, n_overlap=n_fft // 2, window=psd_window)
    std = Scaler(scalings="mean").fit_transform(base_ep) if base_ep.shape[0] else base_ep
    return dict(time_epochs=std, psd_epochs=psd, psd_freqs=freqs, labels=labels, events=events, keep=keep, ica=state,
                alert_rt_baseline=base)


def process_unsupervised_recording(data, sfreq: float, *, new_sfreq: float = 250.0, l_freq: float = 4.0, h_freq: float = 30.0,
                                   n_components=20, random_state=42, ica_fit_params=None, n_fft: int = 512) -> dict:
    """The per-file body of preprocessing_unsupervised.py:34-61 on one continuous ``[C, T]`` recording:
    average reference (:35) -> 4-30 Hz FIR (:36) -> resample to 250 Hz (:37) -> ICA(n_components=20, random_state=42)
    fit + apply with nothing excluded (:39-41) -> Welch PSD of the whole recording (``raw.compute_psd(fmin=4, fmax=30,
    n_fft=512)``: Hamming, no overlap, :43) -> theta / alpha / beta dB means and their ratios (:45-61).

    Returns ``dict(theta, alpha, beta, beta_theta, alpha_beta, theta_beta, psds [C, F], freqs [F], ica)``; the label of
    the file is ``classify_eeg_state(theta, alpha, beta, median of theta_beta over the data set)`` (:66-84).
    """
    import torch
    from . import _lib
    from .fastica import ica_fit
    from .filter import filter_data, resample
    from .ica import affine_channels, average_reference_matrix, ica_apply
    from .psd import band_powers, psd_array_welch

    x = _lib.dev_f32(data)
    if x.dim() != 2:
        raise ValueError("process_unsupervised_recording: data must be [channels, samples]")
    C = x.shape[0]
    x = affine_channels(x, average_reference_matrix(C))                                # :35
    x = filter_data(x, sfreq, l_freq, h_freq)                                          # :36
    if float(new_sfreq) != float(sfreq):
        x = resample(x, int(round(new_sfreq)), int(round(sfreq)))                      # :37
    state = ica_fit(x, n_components=n_components, random_state=random_state, fit_params=ica_fit_params)   # :39-40
    keys = ("unmixing", "mixing", "pca_components", "pca_mean", "pre_whitener", "n_components")
    x = ica_apply(x.unsqueeze(0), exclude=[], **{k: state[k] for k in keys})[0]        # :41
    # Welch with no overlap = mean of the per-segment periodograms: segments as rows, averaged in float64
    nseg = x.shape[1] // n_fft
    if nseg < 1:
        raise ValueError("process_unsupervised_recording: recording shorter than n_fft")
    seg = x[:, : nseg * n_fft].reshape(C * nseg, n_fft)
    p, freqs = psd_array_welch(seg, new_sfreq, l_freq, h_freq, n_fft=n_fft, n_overlap=0, window="hamming")
    psds = p.reshape(C, nseg, -1).to(torch.float64).mean(dim=1).to(torch.float32)      # [C, F]
    bp = band_powers(psds.unsqueeze(0), freqs)[0].cpu().numpy().astype(np.float64)     # :45-61
    return dict(theta=bp[0], alpha=bp[1], beta=bp[2], beta_theta=bp[3], alpha_beta=bp[4], theta_beta=bp[5], psds=psds,
                freqs=np.asarray(freqs), ica=state)
