"""Host-side recording edges: EEGLAB reader, annotation -> trial -> vigilance-label logic vs the
oracle's line-by-line restatement of training/preprocess.py:156-259 (CPU only)."""
import numpy as np
import pytest

import lbm_amd
from oracle import eeg_oracle as orc


def _annotations(rs, n, sfreq=500.0):
    t, out = 5.0, []
    for _ in range(n):
        kind = rs.rand()
        t += rs.uniform(0.05, 4.0)
        if kind < 0.45:
            out.append((t, str(rs.choice([251, 252]))))
        elif kind < 0.85:
            out.append((t, "253"))
        elif kind < 0.92:
            out.append((t, "253.0"))                # float-formatted code (int(float(desc)))
        elif kind < 0.96:
            out.append((t, "boundary"))             # non-numeric annotation
        else:
            out.append((t, "254"))                  # irrelevant code
    rs.shuffle(out)                                  # the script sorts by time itself
    return out


@pytest.mark.parametrize("n", [0, 3, 12, 60, 400])
def test_trials_and_labels_match_reference_logic(n):
    rs = np.random.RandomState(n)
    ann = _annotations(rs, n)
    want_ev, want_lab, want_rt, want_base = orc.vigilance_trials(ann, 500.0)
    times, codes = lbm_amd.extract_events(ann)
    trials = lbm_amd.pair_trials(times, codes, 500.0)
    keep, labels, base = lbm_amd.vigilance_labels(trials["local_rt_sec"])
    events = lbm_amd.mne_events(trials, keep, labels)
    assert np.array_equal(events, want_ev)
    assert labels.tolist() == want_lab.tolist()
    np.testing.assert_array_equal(trials["local_rt_sec"][keep], want_rt)
    assert (np.isnan(base) and np.isnan(want_base)) or base == want_base


def test_label_rules_and_small_sample_fallback():
    keep, lab, base = lbm_amd.vigilance_labels([0.05, 0.5, 0.6, 2.0])        # < 5 kept trials: median baseline
    assert keep.tolist() == [False, True, True, True] and base == 0.6
    assert lab.tolist() == ["alert", "alert", "drowsy"]
    rts = np.concatenate([np.full(30, 0.4), [0.61, 1.01, 0.99, 0.62]])
    keep, lab, base = lbm_amd.vigilance_labels(rts)
    assert base == pytest.approx(0.4) and keep.all()
    assert lab[-4:].tolist() == ["transition", "drowsy", "transition", "transition"]
    # a response with no preceding deviation, and two deviations in a row (the later one is paired)
    t, c = lbm_amd.extract_events([(1.0, "253"), (2.0, "251"), (3.0, "252"), (3.5, "253"), (4.0, "253")])
    tr = lbm_amd.pair_trials(t, c, 500.0)
    assert tr["dev_time"].tolist() == [3.0] and tr["local_rt_sec"].tolist() == [0.5] and tr["response_sample"].tolist() == [1750]


@pytest.mark.parametrize("external_fdt", [False, True])
def test_read_raw_eeglab_round_trip(tmp_path, external_fdt):
    from scipy.io import savemat
    rs = np.random.RandomState(1)
    C, T, fs = 5, 1000, 250.0
    data_uv = (rs.randn(C, T) * 20).astype(np.float32)
    names = ["Fz", "Cz", "Pz", "vehicle position", "Oz"]
    eeg = dict(nbchan=C, pnts=T, trials=1, srate=fs,
               chanlocs=np.array([(n,) for n in names], dtype=[("labels", "O")]),
               event=np.array([(101.0, "251"), (251.0, "253"), (400.0, "boundary")], dtype=[("latency", "O"), ("type", "O")]))
    if external_fdt:
        data_uv.T.tofile(tmp_path / "rec.fdt")       # channels fastest
        eeg["data"] = "rec.fdt"
    else:
        eeg["data"] = data_uv
    savemat(tmp_path / "rec.set", {"EEG": eeg})
    raw = lbm_amd.read_raw_eeglab(str(tmp_path / "rec.set"))
    assert raw["sfreq"] == fs and raw["ch_names"] == names
    np.testing.assert_allclose(raw["data"], data_uv.astype(np.float64) * 1e-6, rtol=0, atol=0)
    assert raw["annotations"] == [(100.0 / fs, "251"), (250.0 / fs, "253"), (399.0 / fs, "boundary")]
    assert lbm_amd.pick_eeg_channels(raw["ch_names"]) == [0, 1, 2, 4]


def test_withpsd_variant_options():
    rs = np.random.RandomState(4)
    ann = _annotations(rs, 200)
    want_ev, want_lab, want_rt, want_base = orc.vigilance_trials(ann, 500.0, percentile=10, lock="deviation")
    times, codes = lbm_amd.extract_events(ann)
    trials = lbm_amd.pair_trials(times, codes, 500.0)
    keep, labels, base = lbm_amd.vigilance_labels(trials["local_rt_sec"], percentile=10.0)
    assert np.array_equal(lbm_amd.mne_events(trials, keep, labels, lock="deviation"), want_ev)
    assert labels.tolist() == want_lab.tolist() and base == want_base
