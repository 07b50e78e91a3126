"""Shared helpers of the data-set driver tests (not collected): synthetic EEGLAB ``.set`` files written with scipy.io.savemat and
the ORACLE's per-file bodies (float64 CPU restatements of multi-train/withpsd.py:92-305 and preprocessing_unsupervised.py:34-61)."""
import os

import numpy as np

from oracle import eeg_oracle as orc


def write_set(path, data_uv, sfreq, ch_names, events):
    """A MATLAB v5 EEGLAB file: data in microvolts [C, T], events = [(latency_1based, type_str), ...]."""
    from scipy.io import savemat
    C, T = data_uv.shape
    eeg = dict(nbchan=C, pnts=T, trials=1, srate=float(sfreq), data=np.asarray(data_uv, dtype=np.float32),
               chanlocs=np.array([(n,) for n in ch_names], dtype=[("labels", "O")]))
    if events:
        eeg["event"] = np.array([(float(lat), str(typ)) for lat, typ in events], dtype=[("latency", "O"), ("type", "O")])
    savemat(path, {"EEG": eeg})


def synth_recording(seed, C, T, sfreq, scale_uv=0.4):
    """Mixed band-limited sources + noise, microvolts, with deviation / response events every few seconds."""
    rs = np.random.RandomState(seed)
    t = np.arange(T) / sfreq
    src = np.stack([np.sin(2 * np.pi * f * t + rs.uniform(0, 6.28)) * (1 + 0.5 * np.sin(2 * np.pi * 0.1 * (i + 1) * t))
                    for i, f in enumerate(rs.uniform(5, 28, C))])
    x = (rs.randn(C, C) @ src + 0.3 * rs.randn(C, T)) * scale_uv
    events, tt = [], 3.0
    while tt < T / sfreq - 4.0:
        events.append((tt * sfreq + 1, str(rs.choice([251, 252]))))
        events.append(((tt + rs.choice([0.3, 0.45, 0.7, 1.2, 0.05])) * sfreq + 1, "253"))
        tt += rs.uniform(2.5, 4.0)
    return x, events


def make_withpsd_dataset(tmp, n_files=5, C=4, T=12000, sfreq=250.0):
    """n_files recordings + one unreadable file + one without annotations; returns the sorted list of paths."""
    os.makedirs(tmp, exist_ok=True)
    names = [f"E{i}" for i in range(C)]
    for i in range(n_files):
        x, ev = synth_recording(100 + i, C, T + 500 * i, sfreq)
        if i == 1:
            x[2, 5000:5030] += 400.0                 # an artefact: some epochs of this file fail the 150 uV test
        write_set(os.path.join(tmp, f"rec{i:02d}.set"), x, sfreq, names, ev)
    with open(os.path.join(tmp, "rec_bad.set"), "wb") as f:
        f.write(b"not a MATLAB file")
    x, _ = synth_recording(7, C, T, sfreq)
    write_set(os.path.join(tmp, "rec_noann.set"), x, sfreq, names, [])
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith(".set"))


def oracle_withpsd_file(path, **kw):
    """The per-file body of multi-train/withpsd.py in the oracle's float64 (ICA off: the fit is covered per recording)."""
    import lbm_amd
    raw = lbm_amd.read_raw_eeglab(path)
    if not raw["annotations"]:
        raise lbm_amd.SkipFile("no annotations found")
    picks = lbm_amd.classify_channels(raw["ch_names"])
    x, fs = raw["data"][picks["eeg"]], raw["sfreq"]
    y = orc.notch_filter(orc.filter_data(x, fs, 4.0, 30.0), fs, 60.0)
    events, labels, _, _ = orc.vigilance_trials([(t, str(d).strip()) for t, d in raw["annotations"]], fs, percentile=10, lock="deviation")
    lo, hi = int(round(-0.5 * fs)), int(round(1.5 * fs))
    ok = (events[:, 0] + lo >= 0) & (events[:, 0] + hi < x.shape[1])
    events, labels = events[ok], labels[ok]
    ep = orc.baseline_correct(np.stack([y[:, s + lo:s + hi + 1] for s in events[:, 0]]), orc.n_baseline_samples(-0.5, fs))
    keep = orc.ptp_keep_mask(ep, 150e-6)
    ep, labels = ep[keep], labels[keep]
    if len(labels) == 0:
        raise lbm_amd.SkipFile("no epochs remaining after rejection")
    n_fft = int(fs)
    psd, freqs = orc.welch_psd(ep, fs, 4.0, 30.0, n_fft=n_fft, n_overlap=n_fft // 2, window="hann")
    return dict(time_epochs=orc.scaler_mean(ep), psd_epochs=psd, psd_freqs=freqs, labels=labels)


def oracle_unsupervised_file(path, **kw):
    """preprocessing_unsupervised.py:34-61 in the oracle's float64, without the ICA round trip (n_pca = C: identity up to
    rounding) -- the CPU stand-in of the orchestration test."""
    import lbm_amd
    raw = lbm_amd.read_raw_eeglab(path)
    x, fs = raw["data"], raw["sfreq"]
    y = orc.average_reference_matrix(x.shape[0]) @ x
    y = orc.filter_data(y, fs, 4.0, 30.0)
    if fs != 250.0:
        y = orc.resample(y, 250, int(fs))
    psd, freqs = orc.welch_psd(y, 250.0, 4.0, 30.0, n_fft=512, n_overlap=0, window="hamming")
    th, al, be, _, _, _ = orc.band_powers(psd, freqs)
    return dict(theta=float(th), alpha=float(al), beta=float(be), log_psds=10.0 * np.log10(psd))
