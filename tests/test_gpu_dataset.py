"""Data-set drivers with the PRODUCT's per-file bodies (every sample pass on the GPU) against the oracle's float64 flow, on
synthetic EEGLAB files: multi-train/withpsd.py:70-414 and preprocessing_unsupervised.py:30-123."""
import os

import numpy as np
import pytest

import lbm_amd
from oracle import eeg_oracle as orc
from tests import dataset_util as du

pytestmark = pytest.mark.gpu


def test_withpsd_dataset_matches_oracle_flow(tmp_path):
    files = du.make_withpsd_dataset(str(tmp_path / "data"))
    got = lbm_amd.preprocess_dataset_withpsd(files=files, output_dir=str(tmp_path / "out"), ica=False)
    want = lbm_amd.preprocess_dataset_withpsd(files=files, process_file=du.oracle_withpsd_file)
    assert [(os.path.basename(p), s if not isinstance(s, str) else "skipped") for p, s in got["files"]] == \
        [(os.path.basename(p), s if not isinstance(s, str) else "skipped") for p, s in want["files"]]
    assert got["labels"].tolist() == want["labels"].tolist()
    np.testing.assert_array_equal(got["psd_freqs"], want["psd_freqs"])
    assert got["time_epochs"].shape == want["time_epochs"].shape and got["time_epochs"].dtype == np.float64
    assert np.abs(got["time_epochs"] - want["time_epochs"]).max() <= 2e-3                       # Scaler units (z-scores)
    assert np.abs(got["psd_epochs"] - want["psd_epochs"]).max() <= 2e-4 * want["psd_epochs"].max()
    np.testing.assert_array_equal(np.load(got["paths"][1]), got["psd_epochs"])
    assert np.load(got["paths"][3]).tolist() == want["labels"].tolist()


def test_withpsd_dataset_with_ica_and_aux_channels(tmp_path):
    """The full per-file body (ICA fit, EOG + ECG scoring on the typed channels, misc channels left out) through the loop."""
    fs, T, C = 250.0, 15000, 8
    names = [f"E{i}" for i in range(C)] + ["VEOG", "ECG", "vehicle position", "A1"]
    d = tmp_path / "data"
    os.makedirs(d)
    for i in range(2):
        x, ev = du.synth_recording(50 + i, C, T, fs)
        rs = np.random.RandomState(i)
        t = np.arange(T) / fs
        blink = (np.sin(2 * np.pi * 0.3 * t) > 0.995) * 60.0
        heart = (np.sin(2 * np.pi * 1.2 * t) > 0.99) * 40.0 * np.sin(2 * np.pi * 12 * t)
        x = x + np.outer(rs.uniform(0.5, 1.0, C), blink) + np.outer(rs.uniform(0.2, 0.5, C), heart)
        full = np.concatenate([x, blink[None] + 2 * rs.randn(1, T), heart[None] + 2 * rs.randn(1, T), 1e3 * rs.randn(2, T)])
        du.write_set(str(d / f"r{i}.set"), full, fs, names, ev)
    out = lbm_amd.preprocess_dataset_withpsd(data_dir=str(d))
    assert all(not isinstance(s, str) and s > 0 for _, s in out["files"]), out["files"]
    E = sum(s for _, s in out["files"])
    assert out["time_epochs"].shape == (E, C, 501) and out["psd_epochs"].shape == (E, C, 27)   # EEG channels only
    assert np.isfinite(out["time_epochs"]).all() and (out["psd_epochs"] > 0).all()
    # per channel over one file's epochs x time the Scaler output has mean 0 / std 1
    n0 = out["files"][0][1]
    first = out["time_epochs"][:n0]
    assert np.abs(first.mean(axis=(0, 2))).max() <= 1e-4 and np.abs(first.std(axis=(0, 2)) - 1).max() <= 1e-3


def test_unsupervised_dataset_matches_oracle_flow(tmp_path):
    fs, T, C = 500.0, 60000, 22
    d = tmp_path / "data"
    os.makedirs(d)
    names = [f"E{i}" for i in range(C)]
    for i in range(3):
        x, ev = du.synth_recording(70 + i, C, T, fs, scale_uv=1.0 + i)
        du.write_set(str(d / f"u{i}.set"), x, fs, names, ev)
    with open(d / "broken.set", "wb") as f:
        f.write(b"\0" * 64)
    got = lbm_amd.process_unsupervised_dataset(dataset_path=str(d), output_path=str(tmp_path / "out"), ica_fit_params=dict(tol=1e-9))
    ok = [p for p, s in got["files"] if s == "ok"]
    assert len(ok) == 3 and len(got["files"]) == 4
    assert got["X"].shape == (3, C, 53) and got["X"].dtype == np.float64                      # the grid of the shipped X_psd_multi.npy
    keys = ("unmixing", "mixing", "pca_components", "pca_mean", "pre_whitener", "n_components")
    th, be, al = [], [], []
    for j, p in enumerate(ok):                                                              # the oracle's flow on the same files
        raw = lbm_amd.read_raw_eeglab(p)
        y = orc.average_reference_matrix(C) @ raw["data"]
        y = orc.resample(orc.filter_data(y, fs, 4.0, 30.0), 250, 500)
        ref = orc.ica_fit(y, n_components=20, random_state=42, tol=1e-9)
        y = orc.ica_apply(y, *(ref[k] for k in keys), exclude=[])
        psd, freqs = orc.welch_psd(y, 250.0, 4.0, 30.0, n_fft=512, n_overlap=0, window="hamming")
        assert np.abs(got["X"][j] - 10 * np.log10(psd)).max() <= 5e-3                        # dB
        t_, a_, b_, _, _, _ = orc.band_powers(psd, freqs)
        th.append(t_); al.append(a_); be.append(b_)
    np.testing.assert_allclose(got["theta"], th, rtol=2e-4)
    np.testing.assert_allclose(got["beta"], be, rtol=2e-4)
    thr = float(np.median(np.asarray(th) / np.asarray(be)))
    assert abs(got["drowsy_threshold"] - thr) <= 2e-4 * abs(thr)
    assert got["y"].tolist() == [lbm_amd.classify_eeg_state(t_, a_, b_, got["drowsy_threshold"])
                                 for t_, a_, b_ in zip(got["theta"], got["alpha"], got["beta"])]
    np.testing.assert_array_equal(np.load(got["paths"][0]), got["X"])
    np.testing.assert_array_equal(np.load(got["paths"][1]), got["y"])
