"""GPU parity tests (run with -m gpu on an MI355X): HIP kernels through the C ABI vs the CPU oracle.

Tolerances.  Integer / index outputs: bit-exact, except VQ indices where SURVEY.md 8(a)'s near-tie rule
applies (and exact duplicate codes must resolve to the lowest index).  Floating point (north_star:
"within rtol 1e-5 (fp32)"): the oracle computes in float64; FIR/affine/post outputs are compared with
atol = 1e-5 * rms(row) (per-sample relative error is unbounded at zero crossings), PSD with rtol 1e-5
plus atol = 1e-5 * max bin of the row.
"""
import math
import os

import numpy as np
import pytest
import torch

import lbm_amd
from lbm_amd import _lib
from lbm_amd.synth import synth_codebook, synth_encoder, synth_epochs, synth_ica
from oracle import eeg_oracle as orc

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _lib.load()


MIN_TOKEN_AGREEMENT = 0.999


def _assert_tokens(tok, ref, z_ref, cb, z_build=None, min_agree=MIN_TOKEN_AGREEMENT):
    """Tokens against the reference's / oracle's: every mismatch must be explained on the REFERENCE-side z (oracle
    vq_tokens_explained: fp32 near-tie rule + the exact effect of the build's own z differing from the reference's), and the
    agreement must be what is measured (>= 0.999), not merely 'most'.  z arrays are [B, D, S] (channel-major) or flat [N, D]."""
    flat = lambda z: None if z is None else (np.transpose(np.asarray(z), (0, 2, 1)).reshape(-1, np.asarray(z).shape[1])  # noqa: E731
                                             if np.asarray(z).ndim == 3 else np.asarray(z))
    tok, ref = np.asarray(tok).reshape(-1), np.asarray(ref).reshape(-1)
    ok = orc.vq_tokens_explained(flat(z_ref), flat(z_build), cb, tok, ref)
    agree = float((tok == ref).mean())
    assert ok.all(), f"{int((~ok).sum())} unexplained token mismatches of {tok.size} (agreement {agree:.5f})"
    assert agree >= min_agree, f"token agreement {agree:.5f} < {min_agree}"


def _rms_ok(got, want, tol=1e-5):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    rms = np.sqrt((want ** 2).mean(axis=-1, keepdims=True)) + 1e-300
    err = np.abs(got - want) / rms
    return float(err.max())


# ------------------------------------------------------------------------------------------------ a1 / a2 FIR
@pytest.mark.parametrize("method", ["fft", "direct"])
@pytest.mark.parametrize("T", [1500, 1001, 300, 37, 1, 4100])
def test_fir_bandpass_matches_oracle(T, method):
    x = synth_epochs(3, 5, T, seed=1).numpy() if T > 8 else np.random.RandomState(0).randn(3, 5, T).astype(np.float32)
    taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
    y = lbm_amd.fir_apply(torch.from_numpy(x).to(DEV), taps, method=method).cpu().numpy()
    want = orc.fir_zero_phase(x.astype(np.float64), taps)
    scale = np.sqrt((x.astype(np.float64) ** 2).mean()) + 1e-300
    # short rows are dominated by the padding transient: compare against the input scale there
    err = _rms_ok(y, want) if T >= 1001 else float(np.abs(y - want).max() / scale)
    assert err <= 1e-5, err


@pytest.mark.parametrize("fs,T,lo,hi", [(1000.0, 10000, 4.0, 30.0),    # config L: 1651 taps, 4096-point transform, 3 block pairs
                                        (250.0, 2000, 4.0, 30.0),      # preprocessing_unsupervised.py after resample: 413 taps
                                        (500.0, 1001, 1.0, 40.0),      # multi-train/CLEAN.py:59: 1651 taps
                                        (100.0, 555, 8.0, None),       # high-pass, short filter -> 1024-point transform
                                        (100.0, 64, None, 20.0)])      # T shorter than the filter
def test_fir_fft_plans_match_oracle_and_direct(fs, T, lo, hi):
    """Every transform size (1024 / 2048 / 4096) and multi-pair rows of the fast-convolution kernel."""
    x = synth_epochs(2, 3, T, sfreq=fs, seed=21).numpy()
    taps = lbm_amd.create_filter(fs, lo, hi)
    xd = torch.from_numpy(x).to(DEV)
    want = orc.fir_zero_phase(x.astype(np.float64), taps)
    scale = np.sqrt((x.astype(np.float64) ** 2).mean()) + 1e-300
    for method in ("fft", "direct"):
        y = lbm_amd.fir_apply(xd, taps, method=method).cpu().numpy()
        assert float(np.abs(y - want).max() / scale) <= 1e-5, (method, len(taps))
    yz = lbm_amd.fir_apply(xd, taps, pad="zero", method="fft").cpu().numpy()
    yzd = lbm_amd.fir_apply(xd, taps, pad="zero", method="direct").cpu().numpy()
    assert float(np.abs(yz - yzd).max() / scale) <= 1e-5


def test_fir_notch_and_zero_pad_and_via_filter_data():
    x = synth_epochs(2, 4, 1500, seed=2)
    y = lbm_amd.notch_filter(x.to(DEV), 500.0, 60.0).cpu().numpy()
    want = orc.notch_filter(x.numpy().astype(np.float64), 500.0, 60.0)
    assert _rms_ok(y, want) <= 1e-5
    # numpy in -> float64 numpy out (drop-in for the scripts' arrays)
    yn = lbm_amd.filter_data(x.numpy(), 500.0, 4.0, 30.0)
    assert isinstance(yn, np.ndarray) and yn.dtype == np.float64
    assert _rms_ok(yn, orc.filter_data(x.numpy().astype(np.float64), 500.0, 4.0, 30.0)) <= 1e-5
    taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
    yz = lbm_amd.fir_apply(x.to(DEV), taps, pad="zero").cpu().numpy()
    xz = np.pad(x.numpy().astype(np.float64), [(0, 0), (0, 0), (412, 412)])
    wz = np.stack([[np.convolve(r, taps, mode="valid") for r in e] for e in xz])
    assert _rms_ok(yz, wz) <= 1e-5
    # empty input
    assert lbm_amd.fir_apply(torch.empty(0, 30, 1500, device=DEV), taps).shape == (0, 30, 1500)


def test_fir_linearity_at_full_size():
    """Size-independent property on a BASELINE-size chunk: FIR(a*x1 + x2) == a*FIR(x1) + FIR(x2)."""
    E = 2048
    x1 = synth_epochs(E, 30, 1500, seed=3, device=DEV)
    x2 = synth_epochs(E, 30, 1500, seed=4, device=DEV)
    taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
    y1, y2 = lbm_amd.fir_apply(x1, taps), lbm_amd.fir_apply(x2, taps)
    y12 = lbm_amd.fir_apply(2.0 * x1 + x2, taps)
    err = (y12 - (2.0 * y1 + y2)).abs().max() / y12.pow(2).mean().sqrt()
    assert float(err) < 2e-5
    # and a sample of rows against the oracle
    rows = [(0, 0), (1023, 17), (2047, 29)]
    for e, c in rows:
        want = orc.fir_zero_phase(x1[e, c].cpu().numpy().astype(np.float64), taps)
        assert _rms_ok(y1[e, c].cpu().numpy(), want) <= 1e-5


def test_continuous_recording_order_filter_ica_then_epochs(tmp_path):
    """The reference's exact order (training/preprocess.py:77 -> 144 -> 262-287): filter and clean the CONTINUOUS
    recording, then cut epochs, baseline-correct, z-score, save with the reference's file names."""
    C, Trec, fs = 6, 60000, 500.0
    raw = synth_epochs(1, C, Trec, sfreq=fs, seed=31)[0]
    taps = lbm_amd.create_filter(fs, 4.0, 30.0)
    kw = synth_ica(C, exclude=(1,))
    events = np.array([100, 700, 5000, 5001, 30000, 59200, 59900])       # first and last windows leave the recording
    rawd = raw.to(DEV)
    filt = lbm_amd.fir_apply(rawd, taps)                                  # long rows: many block pairs per row
    want_f = orc.fir_zero_phase(raw.numpy().astype(np.float64), taps)
    assert _rms_ok(filt.cpu().numpy(), want_f) <= 1e-5
    clean = lbm_amd.ica_apply(filt[None], **kw)[0]
    x, kept = lbm_amd.make_epochs(clean, events, lbm_amd.TMIN, lbm_amd.TMAX, fs)
    assert kept.tolist() == [False, True, True, True, True, True, False] and x.shape == (5, C, 1001)
    want_c = orc.ica_apply(want_f[None], **kw)[0]
    want_x = np.stack([want_c[:, s - 250:s + 751] for s in events[kept]])
    assert _rms_ok(x.cpu().numpy(), want_x) <= 2e-5
    base, std, _ = lbm_amd.epoch_post(x, n_baseline=lbm_amd.n_baseline_samples(lbm_amd.TMIN, fs), standardize=True)
    wb = orc.baseline_correct(want_x, 251)
    assert float(np.abs(std.cpu().numpy() - orc.zscore_epoch_channel(wb)).max()) <= 1e-4
    a, b = lbm_amd.save_preprocess_outputs(str(tmp_path / "s02_051115m"), std, np.array(["alert"] * 5))
    assert a.endswith("s02_051115m_epochs_ica_std.npy") and np.load(a).dtype == np.float64 and np.load(a).shape == (5, C, 1001)
    assert np.load(b).shape == (5,)


# ------------------------------------------------------------------------------------------------ a3 ICA
@pytest.mark.parametrize("C,T", [(30, 1500), (30, 1001), (64, 1000), (5, 7)])
def test_ica_apply_matches_oracle(C, T):
    x = synth_epochs(3, C, T, seed=5).numpy() if T > 8 else np.random.RandomState(1).randn(3, C, T).astype(np.float32) * 1e-5
    kw = synth_ica(C, exclude=(0, 3))
    y = lbm_amd.ica_apply(torch.from_numpy(x).to(DEV), **kw).cpu().numpy()
    want = orc.ica_apply(x.astype(np.float64), **kw)
    assert _rms_ok(y, want) <= 1e-5
    # rectangular map: sources of a 20-component decomposition
    rs = np.random.RandomState(2)
    ncomp = min(20, C)
    un = rs.randn(ncomp, ncomp)
    pca = np.linalg.qr(rs.randn(C, C))[0]
    s = lbm_amd.ica_get_sources(torch.from_numpy(x).to(DEV), unmixing=un, pca_components=pca, pca_mean=rs.randn(C) * 0.01,
                                pre_whitener=1e-5, n_components=ncomp).cpu().numpy()
    rs = np.random.RandomState(2); un = rs.randn(ncomp, ncomp); pca = np.linalg.qr(rs.randn(C, C))[0]; mu = rs.randn(C) * 0.01
    assert s.shape == (3, ncomp, T)
    assert _rms_ok(s, orc.ica_sources(x.astype(np.float64), un, pca, mu, 1e-5, ncomp)) <= 1e-5


# ------------------------------------------------------------------------------------------------ a4-a6 post
@pytest.mark.parametrize("T,nb", [(1001, 251), (1500, 251), (10000, 501), (64, 0), (3000, 100)])
def test_epoch_post_matches_oracle(T, nb):
    x = synth_epochs(6, 7, T, seed=6) * 3.0 + 2e-5
    x[2, 3, T // 2] += 400e-6        # a clear artefact: epoch 2 must be rejected
    xd = x.to(DEV)
    base, std, keep = lbm_amd.epoch_post(xd, n_baseline=nb, reject=150e-6, standardize=True)
    x64 = x.numpy().astype(np.float64)
    wb = orc.baseline_correct(x64, nb)
    assert _rms_ok(base.cpu().numpy(), wb) <= 1e-5
    ptp = x64.max(-1) - x64.min(-1)
    assert not np.any(np.abs(ptp - 150e-6) < 1e-9)   # no epoch sits on the threshold
    assert np.array_equal(keep.cpu().numpy(), orc.ptp_keep_mask(wb, 150e-6))
    assert not keep[2]
    assert float(np.abs(std.cpu().numpy() - orc.zscore_epoch_channel(wb)).max()) <= 2e-5   # unit-variance output
    z2 = lbm_amd.standardize_epochs(xd).cpu().numpy()
    assert float(np.abs(z2 - orc.zscore_epoch_channel(x64)).max()) <= 2e-5


def test_scaler_matches_oracle():
    x = synth_epochs(9, 6, 1001, seed=7) + 1e-5
    sc = lbm_amd.Scaler(scalings="mean")
    y = sc.fit_transform(x.to(DEV)).cpu().numpy()
    assert float(np.abs(y - orc.scaler_mean(x.numpy().astype(np.float64))).max()) <= 2e-5
    mom = sc.moments_.cpu().numpy()
    x64 = x.numpy().astype(np.float64)
    np.testing.assert_allclose(mom[:, 0], x64.sum(axis=(0, 2)), rtol=1e-12)
    np.testing.assert_allclose(mom[:, 1], (x64 ** 2).sum(axis=(0, 2)), rtol=1e-12)
    assert (mom[:, 2] == 9 * 1001).all()


# ------------------------------------------------------------------------------------------------ a7 Welch
# End-to-end PSD tolerance, in units of the stage tolerance (rtol 1e-5 + atol 1e-5 * row max).  The Welch STAGE meets 1.0 on exact
# inputs (test_welch_matches_oracle).  At the end of the chain its input already carries the fp32 rounding of the 825-tap FIR
# (~1e-6 of the row rms per sample) and of the 30-channel ICA map (~1e-6): a PSD bin is quadratic in the signal, so those input
# errors arrive doubled, and out-of-band-attenuated bins near 4 and 30 Hz sit 20-30 dB under the row maximum.  Measured on the
# bench's config-1 sample: 0.05 of the stage tolerance; 1.0 -- the stage tolerance itself, i.e. north_star's rtol 1e-5 -- is the
# bound asserted (round 1 allowed 4.0, round 2 2.0).
PIPELINE_PSD_TOL = 1.0


def _psd_err(got, want):
    got = np.asarray(got, dtype=np.float64)
    tol = 1e-5 * np.abs(want) + 1e-5 * np.abs(want).max(axis=-1, keepdims=True)
    return float((np.abs(got - want) / tol).max())


@pytest.mark.parametrize("T,fs,nfft,nov,win,fmin,fmax", [
    (1500, 500.0, 500, 250, "hann", 4.0, 30.0),       # config S (multi-train/withpsd.py:236-246)
    (1001, 500.0, 500, 250, "hann", 4.0, 30.0),       # config R
    (10000, 1000.0, 1000, 500, "hann", 4.0, 30.0),    # config L
    (2048, 250.0, 512, 0, "hamming", 4.0, 30.0),      # preprocessing_unsupervised.py:43 defaults
    (1001, 500.0, 512, 0, "hamming", 1.0, 40.0),      # multi-train/CLEAN.py:175
    (700, 100.0, 101, 30, "hann", 0.0, 50.0),         # odd nfft, DC and last bin included
    (600, 100.0, 100, 50, "boxcar", 0.0, 50.0),       # even nfft incl. Nyquist
])
def test_welch_matches_oracle(T, fs, nfft, nov, win, fmin, fmax):
    x = synth_epochs(2, 5, T, sfreq=fs, seed=8) + 3e-5    # DC offset exercises the per-segment mean removal
    psd, freqs = lbm_amd.psd_array_welch(x.to(DEV), fs, fmin, fmax, n_fft=nfft, n_overlap=nov, window=win)
    want, wf = orc.welch_psd(x.numpy().astype(np.float64), fs, fmin, fmax, n_fft=nfft, n_overlap=nov, window=win)
    assert np.array_equal(freqs, wf) and psd.shape == want.shape
    assert _psd_err(psd.cpu().numpy(), want) <= 1.0


@pytest.mark.parametrize("rows,T,fs,nfft,fmin", [(7, 1500, 500.0, 500, 4.0), (3, 10000, 1000.0, 1000, 4.0), (5, 1500, 500.0, 500, 0.0),
                                               (4, 1502, 500.0, 500, 4.0), (1, 1001, 500.0, 500, 2.0)])
def test_welch_team_kernel_offsets_and_ragged_groups(rows, T, fs, nfft, fmin):
    """The team kernel (round 2): rows with a DC offset 1000x the signal (the mean-free path subtracts the segment's first sample
    instead of its mean: exact where the window annihilates constants), row counts that leave a ragged last group, bins that
    include DC (per-segment means are computed again), and a row length the team path does not take (T % 4 != 0)."""
    x = synth_epochs(1, rows, T, sfreq=fs, seed=81)[0]
    x = x + torch.linspace(-1.0, 1.0, rows).unsqueeze(1) * 2e-2
    psd, freqs = lbm_amd.psd_array_welch(x.to(DEV), fs, fmin, 30.0, n_fft=nfft, n_overlap=nfft // 2, window="hann")
    want, wf = orc.welch_psd(x.numpy().astype(np.float64), fs, fmin, 30.0, n_fft=nfft, n_overlap=nfft // 2, window="hann")
    assert np.array_equal(freqs, wf) and psd.shape == want.shape
    assert _psd_err(psd.cpu().numpy(), want) <= 1.0


@pytest.mark.parametrize("E,C,T,fs,nfft", [(37, 30, 1500, 500.0, 500), (5, 4, 10000, 1000.0, 1000), (9, 3, 1502, 500.0, 500), (4, 2, 600, 100.0, 100)])
def test_welch_keep_fused_reject_mask(E, C, T, fs, nfft):
    """lbm_welch_keep_f32: the PSD of lbm_welch_f32 bit for bit and the keep mask of lbm_epoch_post_f32 (oracle: ptp_keep_mask), on
    the team kernel's shapes (fused) and on one it does not take (separate pass)."""
    from lbm_amd.psd import welch_bins, welch_tables
    x = synth_epochs(E, C, T, sfreq=fs, seed=E) * 2.0
    x[1, C - 1, T // 3] += 2e-4        # one sample out of range in two epochs
    x[E - 1, 0, T - 1] -= 3e-4
    xd = x.to(DEV)
    _, lo, hi = welch_bins(fs, nfft, 4.0, 30.0)
    tab = welch_tables(DEV, nfft, nfft, "hann", lo, hi)
    lib, p_, st = lbm_amd._lib.load(), lbm_amd._lib.ptr, lbm_amd._lib.stream_ptr()
    psd = torch.empty(E, C, hi - lo + 1, device=DEV)
    ref = torch.empty_like(psd)
    keep = torch.full((E,), 7, dtype=torch.uint8, device=DEV)
    assert lib.lbm_welch_f32(p_(xd), p_(ref), None, E * C, T, fs, nfft, nfft, nfft // 2, p_(tab), lo, hi, 0, st) == 0
    for _ in range(2):
        assert lib.lbm_welch_keep_f32(p_(xd), p_(psd), p_(keep), E, C, T, fs, nfft, nfft, nfft // 2, p_(tab), lo, hi, 150e-6, st) == 0
        assert torch.equal(psd, ref)
        want = orc.ptp_keep_mask(x.numpy().astype(np.float64), 150e-6)
        assert np.array_equal(keep.cpu().numpy().astype(bool), want)
        assert int((~want).sum()) == 2


def test_welch_average_modes_and_errors():
    x = synth_epochs(2, 3, 1500, seed=9)
    xd = x.to(DEV)
    x64 = x.numpy().astype(np.float64)
    for avg in ("median", None):
        got, _ = lbm_amd.psd_array_welch(xd, 500.0, 4.0, 30.0, n_fft=500, n_overlap=250, window="hann", average=avg)
        want, _ = orc.welch_psd(x64, 500.0, 4.0, 30.0, n_fft=500, n_overlap=250, window="hann", average=avg)
        assert got.shape == want.shape
        assert _psd_err(got.cpu().numpy().reshape(2, 3, -1), want.reshape(2, 3, -1)) <= 1.0
    with pytest.warns(UserWarning):   # multi-train/withpsd.py:244 passes average=False
        got, _ = lbm_amd.psd_array_welch(xd, 500.0, 4.0, 30.0, n_fft=500, n_overlap=250, window="hann", average=False)
    assert got.shape == (2, 3, 27)
    with pytest.raises(ValueError):
        lbm_amd.psd_array_welch(xd[..., :100], 500.0, n_fft=500)
    # n_per_seg < n_fft (zero padded segments)
    got, _ = lbm_amd.psd_array_welch(xd, 500.0, 4.0, 30.0, n_fft=500, n_per_seg=400, n_overlap=100, window="hann")
    want, _ = orc.welch_psd(x64, 500.0, 4.0, 30.0, n_fft=500, n_per_seg=400, n_overlap=100, window="hann")
    assert _psd_err(got.cpu().numpy(), want) <= 1.0


def test_welch_full_size_against_torch_fft():
    """Welch at the bench's launch size (16384 epochs x 30 channels x 1500 samples): every row against a float64
    torch.fft restatement of the same definition on the GPU (segments, mean removal, periodic Hann, one-sided density),
    plus homogeneity psd(a x) = a^2 psd(x) bit-for-bit for a power-of-two a."""
    E, C, T, fs, nfft, hop = 16384, 30, 1500, 500.0, 500, 250
    x = synth_epochs(E, C, T, seed=17, device=DEV, chunk=2048) + 2e-5
    psd, freqs = lbm_amd.psd_array_welch(x, fs, 4.0, 30.0, n_fft=nfft, n_overlap=nfft - hop, window="hann")
    assert psd.shape == (E, C, 27)
    n = torch.arange(nfft, device=DEV, dtype=torch.float64)
    w = 0.5 - 0.5 * torch.cos(2.0 * math.pi * n / nfft)
    scale = 2.0 / (fs * float((w * w).sum()))
    worst = 0.0
    for lo in range(0, E, 1024):
        seg = x[lo:lo + 1024].double().unfold(-1, nfft, hop)            # [e, c, 5, 500]
        seg = (seg - seg.mean(-1, keepdim=True)) * w
        p = (torch.fft.rfft(seg, dim=-1)[..., 4:31].abs() ** 2).mean(-2) * scale
        got = psd[lo:lo + 1024].double()
        tol = 1e-5 * p + 1e-5 * p.max(dim=-1, keepdim=True).values
        worst = max(worst, float(((got - p).abs() / tol).max()))
    assert worst <= 1.0
    psd4, _ = lbm_amd.psd_array_welch(x[:2048] * 4.0, fs, 4.0, 30.0, n_fft=nfft, n_overlap=nfft - hop, window="hann")
    assert torch.equal(psd4, psd[:2048] * 16.0)


# ------------------------------------------------------------------------------------------------ a8 / a9
def test_bandpower_known_answer_labels(golden_dir):
    with np.load(os.path.join(golden_dir, "bandpower_kat.npz")) as f:
        X, y, yv, freqs = f["log_psd"], f["labels"], f["labels_variant"], f["freqs"]
    bp = lbm_amd.band_powers(torch.from_numpy(X).to(DEV), freqs, already_db=True).cpu().numpy().astype(np.float64)
    want = np.stack(orc.band_powers(X, freqs, already_db=True), axis=-1)
    np.testing.assert_allclose(bp, want, rtol=2e-7)
    thr = float(np.median(bp[:, 5]))
    lab = np.array([lbm_amd.classify_eeg_state(t, a, b, thr) for t, a, b in bp[:, :3]])
    assert np.array_equal(lab, y)                                  # 62/62 (preprocessing_unsupervised.py rule)
    labv = np.array([lbm_amd.classify_eeg_state_variant(t, a, b) for t, a, b in bp[:, :3]])
    assert np.array_equal(labv, yv)                                # 62/62 (failed/label.py rule)
    # linear PSD in -> same numbers through the 10*log10 path; per-channel variant
    lin = 10.0 ** (X / 10.0)
    bp2 = lbm_amd.band_powers(torch.from_numpy(lin).to(DEV), freqs).cpu().numpy()
    np.testing.assert_allclose(bp2, want, rtol=1e-5)
    bpc = lbm_amd.band_powers(torch.from_numpy(lin[:4]).to(DEV), freqs, per_channel=True).cpu().numpy()
    wantc = np.stack(orc.band_powers(lin[:4, :, None, :], freqs), axis=-1)
    np.testing.assert_allclose(bpc, wantc, rtol=1e-5)


# ------------------------------------------------------------------------------------------------ a10 encoder
def test_conv_encoder_matches_reference_golden(shipped, vq_golden):
    for name in ("S", "R", "P"):
        x = torch.from_numpy(vq_golden[f"x_{name}"]).to(DEV)
        z = lbm_amd.encoder_forward(x, {k: torch.from_numpy(v).to(DEV) for k, v in shipped.items()})
        ref = vq_golden[f"z_{name}"]
        assert z.shape == ref.shape
        assert float(np.abs(z.cpu().numpy() - ref).max()) <= 1e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("B,Cin,L,Cout,k,s,p", [(3, 30, 1500, 64, 4, 2, 1), (2, 64, 101, 128, 4, 2, 1), (2, 7, 33, 70, 3, 1, 1),
                                                (1, 256, 40, 256, 1, 1, 0), (2, 3, 10, 5, 5, 3, 2)])
def test_conv1d_matches_torch_cpu(B, Cin, L, Cout, k, s, p):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, L, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) / math.sqrt(Cin * k)
    b = torch.randn(Cout, generator=g)
    for relu in (False, True):
        y = lbm_amd.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), s, p, relu).cpu()
        want = torch.nn.functional.conv1d(x.double(), w.double(), b.double(), stride=s, padding=p)
        want = torch.relu(want) if relu else want
        assert y.shape == want.shape
        assert float((y.double() - want).abs().max()) <= 1e-5


@pytest.mark.parametrize("B,Cin,L,Cout", [(3, 128, 375, 256), (2, 64, 2500, 256), (1, 8, 5, 64), (2, 24, 130, 200), (1, 16, 1, 96)])
def test_conv1d_wino3_table_is_bitwise_the_default_kernel(B, Cin, L, Cout):
    """lbm_conv1d_wino3_f32 (Winograd-transformed weights from a caller-owned table) = lbm_conv1d_pitched_f32 on the same shape, bit
    for bit, for dense and pitched rows, with and without ReLU / bias."""
    from lbm_amd.quantizer import pitched_empty
    g = torch.Generator().manual_seed(B * 1000 + L)
    x = torch.randn(B, Cin, L, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 3, generator=g) / math.sqrt(Cin * 3)).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    tab = lbm_amd.conv_wino3_tables(w)
    for relu in (False, True):
        for bias in (b, None):
            want = lbm_amd.conv1d(x, w, bias, 1, 1, relu)
            assert torch.equal(lbm_amd.conv1d(x, w, bias, 1, 1, relu, wino_tables=tab), want)
    xp = pitched_empty(B, Cin, L, DEV)
    xp.copy_(x)
    got = lbm_amd.conv1d(xp, w, b, 1, 1, True, pitched_out=True, wino_tables=tab)
    assert got.stride(1) % 4 == 0 and torch.equal(got, lbm_amd.conv1d(x, w, b, 1, 1, True))
    ref = torch.nn.functional.conv1d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1).relu()
    assert float((got.double().cpu() - ref).abs().max()) <= 1e-5
    with pytest.raises(NotImplementedError):
        lbm_amd.conv_wino3_tables(torch.zeros(64, 12, 3))
    with pytest.raises(ValueError):
        lbm_amd.conv1d(x, w, b, 2, 1, True, wino_tables=tab)


@pytest.mark.parametrize("B,L,ld,Cout", [(3, 375, 376, 256), (100, 375, 376, 256), (77, 250, 252, 256), (2, 100, 100, 256), (2, 128, 128, 64),
                                         (3, 129, 132, 192), (2, 136, 140, 256), (2, 2500, 2500, 256), (1, 3, 4, 64), (5, 1001, 1004, 128)])
def test_conv1d_wino3_ws_is_the_tile_kernel_bit_for_bit(B, L, ld, Cout):
    """Cin = 128 with 16-byte rows: lbm_conv1d_wino3_f32 runs the weights-stationary kernel (csrc/conv_wino_ws.hip: persistent workgroups,
    a transform and 32 channels per wave, windows by LDS-DMA, exchange through LDS).  Same products in the same order as the tile kernel
    (lbm_conv1d_pitched_f32), so: equal bit for bit -- one tile, many tiles per workgroup (B = 100: 300 tiles over 64 streams), rows that
    end inside a quad, NaN in the pitch padding, with / without bias and ReLU, on a side stream -- and within 1e-5 of float64."""
    lib, p = _lib.load(), _lib.ptr
    g = torch.Generator().manual_seed(B * 7 + L)
    x = torch.randn(B, 128, ld, generator=g).to(DEV)
    x[..., L:] = float("nan")
    w = (torch.randn(Cout, 128, 3, generator=g) / math.sqrt(384)).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    tab = lbm_amd.conv_wino3_tables(w)
    side = torch.cuda.Stream()
    for relu, bias in ((1, b), (0, None)):
        y0 = torch.full((B, Cout, ld), -7.0, device=DEV)
        y1 = torch.full((B, Cout, ld), -7.0, device=DEV)
        bp = p(bias) if bias is not None else None
        assert lib.lbm_conv1d_pitched_f32(p(x), ld, p(w), bp, p(y0), ld, B, 128, L, Cout, 3, 1, 1, relu, None) == 0, _lib.last_error()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            assert lib.lbm_conv1d_wino3_f32(p(x), ld, p(tab), bp, p(y1), ld, B, 128, L, Cout, relu, _lib.stream_ptr()) == 0, _lib.last_error()
        torch.cuda.synchronize()
        assert torch.equal(y0[..., :L], y1[..., :L])
        assert not torch.isnan(y1[..., :L]).any()
        if ld - (L + 3) // 4 * 4 > 0:      # columns past the last quad of a row are never written
            assert bool((y1[..., (L + 3) // 4 * 4:] == -7.0).all())
        ref = torch.nn.functional.conv1d(x[..., :L].double().cpu(), w.double().cpu(), bias.double().cpu() if bias is not None else None, padding=1)
        ref = ref.relu() if relu else ref
        assert float((y1[..., :L].double().cpu() - ref).abs().max()) <= 1e-5


@pytest.mark.parametrize("B,Cin,L,Cout", [(3, 30, 1500, 64), (100, 30, 1500, 64), (2, 30, 1001, 64), (2, 64, 750, 128), (300, 64, 750, 128),
                                          (3, 64, 501, 128), (1, 30, 2, 64), (2, 30, 10, 64), (2, 64, 5000, 128)])
def test_conv1d_k4s2_ws_is_the_tile_kernel_bit_for_bit(B, Cin, L, Cout):
    """The encoder's 4-tap stride-2 layers on 16-byte rows run the weights-stationary kernels (csrc/conv_k4s2_ws.hip: persistent
    workgroups, a wave's weights in registers, windows by LDS-DMA, one barrier per tile).  Same products in the same order as the tile
    kernel -- which the same entry point still runs on rows with an odd pitch --, so: equal bit for bit, one tile and many tiles per
    workgroup, odd and even lengths, NaN in the pitch padding, with / without bias and ReLU, on a side stream; within 1e-5 of float64."""
    lib, p = _lib.load(), _lib.ptr
    g = torch.Generator().manual_seed(B * 7 + L)
    Lout = (L - 2) // 2 + 1
    ld, ldo = (L + 3) & ~3, (Lout + 3) & ~3
    x = torch.randn(B, Cin, ld, generator=g).to(DEV)
    x[..., L:] = float("nan")
    xo = torch.full((B, Cin, L + 1 + (L & 1)), float("nan"), device=DEV)      # odd pitch: not taken by the weights-stationary form
    xo[..., :L] = x[..., :L]
    w = (torch.randn(Cout, Cin, 4, generator=g) / math.sqrt(Cin * 4)).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    side = torch.cuda.Stream()
    for relu, bias in ((1, b), (0, None)):
        y0 = torch.full((B, Cout, Lout + 1 + (Lout & 1)), -7.0, device=DEV)
        y1 = torch.full((B, Cout, ldo), -7.0, device=DEV)
        bp = p(bias) if bias is not None else None
        assert lib.lbm_conv1d_pitched_f32(p(xo), xo.shape[2], p(w), bp, p(y0), y0.shape[2], B, Cin, L, Cout, 4, 2, 1, relu, None) == 0, _lib.last_error()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            assert lib.lbm_conv1d_pitched_f32(p(x), ld, p(w), bp, p(y1), ldo, B, Cin, L, Cout, 4, 2, 1, relu, _lib.stream_ptr()) == 0, _lib.last_error()
        torch.cuda.synchronize()
        assert torch.equal(y0[..., :Lout], y1[..., :Lout])
        assert not torch.isnan(y1[..., :Lout]).any()
        ref = torch.nn.functional.conv1d(x[..., :L].double().cpu(), w.double().cpu(), bias.double().cpu() if bias is not None else None,
                                         stride=2, padding=1)
        ref = ref.relu() if relu else ref
        assert float((y1[..., :Lout].double().cpu() - ref).abs().max()) <= 1e-5


@pytest.mark.parametrize("B,Cin,L,Cout", [(3, 64, 750, 128), (300, 64, 750, 128), (2, 30, 1500, 64), (100, 30, 1500, 64), (2, 64, 1001, 64),
                                          (1, 30, 39, 64), (4, 17, 200, 192), (2, 64, 2, 128), (3, 64, 5, 64), (2, 48, 3000, 128),
                                          (1, 64, 383, 128), (1, 64, 385, 128), (2, 1, 64, 64)])
def test_conv1d_k4s2_wino_matches_float64(B, Cin, L, Cout):
    """lbm_conv1d_k4s2_wino_f32 (csrc/conv_k4s2_wino_ws.hip): the 4-tap stride-2 layers (training/model.py:169-172) as two 2-tap phases
    in the minimal form F(3,2).  Not the direct kernels' bits (the transforms round differently), so the bar is float64: within 4e-6 of
    the output scale -- the direct kernel's own distance on the same inputs is asserted beside it, and the two agree to 5e-6 -- over one
    tile and many tiles per workgroup, odd and even lengths, rows that end inside a quad, fewer than 32 / 64 input channels, NaN in the
    pitch padding of x, with / without bias and ReLU, on a side stream; columns past a row's last quad are never written."""
    lib, p = _lib.load(), _lib.ptr
    g = torch.Generator().manual_seed(B * 11 + L)
    Lout = (L - 2) // 2 + 1
    ld, ldo = (L + 3) & ~3, ((Lout + 3) & ~3) + 4
    x = torch.randn(B, Cin, ld, generator=g).to(DEV)
    x[..., L:] = float("nan")
    w = (torch.randn(Cout, Cin, 4, generator=g) / math.sqrt(Cin * 4)).to(DEV)
    b = (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    assert lib.lbm_conv1d_k4s2_wino_supported(Cin, Cout) == 1
    side = torch.cuda.Stream()
    for relu, bias in ((1, b), (0, None)):
        y0 = torch.full((B, Cout, ldo), -7.0, device=DEV)
        y1 = torch.full((B, Cout, ldo), -7.0, device=DEV)
        bp = p(bias) if bias is not None else None
        assert lib.lbm_conv1d_pitched_f32(p(x), ld, p(w), bp, p(y0), ldo, B, Cin, L, Cout, 4, 2, 1, relu, None) == 0, _lib.last_error()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            assert lib.lbm_conv1d_k4s2_wino_f32(p(x), ld, p(w), bp, p(y1), ldo, B, Cin, L, Cout, relu, _lib.stream_ptr()) == 0, _lib.last_error()
        torch.cuda.synchronize()
        assert not torch.isnan(y1[..., :Lout]).any()
        assert bool((y1[..., (Lout + 3) // 4 * 4:] == -7.0).all())
        ref = torch.nn.functional.conv1d(x[..., :L].double().cpu(), w.double().cpu(), bias.double().cpu() if bias is not None else None,
                                         stride=2, padding=1)
        ref = ref.relu() if relu else ref
        scale = max(1.0, float(ref.abs().max()))
        assert float((y1[..., :Lout].double().cpu() - ref).abs().max()) <= 4e-6 * scale
        assert float((y0[..., :Lout].double().cpu() - ref).abs().max()) <= 6e-6 * scale
        assert float((y1[..., :Lout] - y0[..., :Lout]).abs().max()) <= 5e-6 * scale


def test_conv1d_k4s2_wino_rejects_what_it_does_not_take():
    lib, p = _lib.load(), _lib.ptr
    assert lib.lbm_conv1d_k4s2_wino_supported(65, 128) == 0 and lib.lbm_conv1d_k4s2_wino_supported(64, 96) == 0
    x = torch.zeros(2, 64, 752, device=DEV)
    w = torch.zeros(128, 64, 4, device=DEV)
    y = torch.zeros(2, 128, 376, device=DEV)
    assert lib.lbm_conv1d_k4s2_wino_f32(p(x), 751, p(w), None, p(y), 376, 2, 64, 750, 128, 1, None) == -2      # odd pitch
    assert lib.lbm_conv1d_k4s2_wino_f32(p(x), 752, p(w), None, p(y), 376, 2, 65, 750, 128, 1, None) == -2      # Cin
    assert lib.lbm_conv1d_k4s2_wino_f32(p(x), 752, p(w), None, p(y), 372, 2, 64, 750, 128, 1, None) == -1            # ldy < Lout
    assert lib.lbm_conv1d_k4s2_wino_f32(p(x), 752, None, None, p(y), 376, 2, 64, 750, 128, 1, None) == -1            # NULL
    assert lib.lbm_conv1d_k4s2_wino_f32(p(x), 752, p(w), None, p(y), 376, 0, 64, 750, 128, 1, None) == 0                      # B = 0


@pytest.mark.parametrize("B,Cin,L,Cout,k,s,p", [(3, 30, 1500, 64, 4, 2, 1), (2, 64, 750, 128, 4, 2, 1), (2, 128, 375, 256, 3, 1, 1),
                                                (2, 256, 375, 256, 1, 1, 0), (2, 12, 130, 40, 3, 1, 1), (1, 20, 9, 7, 1, 1, 0)])
def test_conv1d_pitched_rows_bitwise_equal_dense(B, Cin, L, Cout, k, s, p):
    """Row-padded (16-byte) activation buffers in and out give bit-identical results to dense tensors."""
    from lbm_amd.quantizer import pitched_empty
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, Cin, L, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, k, generator=g) / math.sqrt(Cin * k)).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    dense = lbm_amd.conv1d(x, w, b, s, p, True)
    xp = pitched_empty(B, Cin, L, DEV)
    xp.copy_(x)
    xp.as_strided((B, Cin, xp.stride(1)), xp.stride())[:, :, L:] = float("nan")     # the padding must never be read as data
    yp = lbm_amd.conv1d(xp, w, b, s, p, True, pitched_out=True)
    assert yp.shape == dense.shape and yp.stride(1) % 4 == 0
    assert torch.equal(yp, dense)
    want = torch.relu(torch.nn.functional.conv1d(x.double().cpu(), w.double().cpu(), b.double().cpu(), stride=s, padding=p))
    assert float((dense.double().cpu() - want).abs().max()) <= 1e-5


@pytest.mark.parametrize("B,Cin,L,Cout,k,s,p", [(3, 128, 375, 256, 3, 1, 1), (2, 256, 375, 256, 1, 1, 0), (2, 64, 750, 128, 4, 2, 1),
                                                (3, 30, 1500, 64, 4, 2, 1), (2, 12, 130, 300, 3, 1, 1), (1, 20, 9, 7, 1, 1, 0),
                                                (2, 7, 33, 70, 5, 3, 2)])
def test_conv1d_bf16x6_matches_float64(B, Cin, L, Cout, k, s, p):
    """lbm_conv1d_bf16x6_f32 (fp32 products rebuilt from three bf16 pieces): the fp32 kernels' tolerance against float64,
    dense and pitched rows."""
    from lbm_amd.quantizer import pitched_empty
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, Cin, L, generator=g)
    w = torch.randn(Cout, Cin, k, generator=g) / math.sqrt(Cin * k)
    b = torch.randn(Cout, generator=g)
    tab = lbm_amd.conv_split_tables(w.to(DEV))
    for relu in (False, True):
        y = lbm_amd.conv1d(x.to(DEV), w.to(DEV), b.to(DEV), s, p, relu, split_tables=tab).cpu()
        want = torch.nn.functional.conv1d(x.double(), w.double(), b.double(), stride=s, padding=p)
        want = torch.relu(want) if relu else want
        assert y.shape == want.shape
        assert float((y.double() - want).abs().max()) <= 1e-5
    xp = pitched_empty(B, Cin, L, DEV)
    xp.copy_(x)
    xp.as_strided((B, Cin, xp.stride(1)), xp.stride())[:, :, L:] = float("nan")
    yp = lbm_amd.conv1d(xp, w.to(DEV), b.to(DEV), s, p, True, pitched_out=True, split_tables=tab)
    assert torch.equal(yp.cpu(), y)


# ------------------------------------------------------------------------------------------------ a3 + a4-a6 fused
@pytest.mark.parametrize("E,Cin,Cout,T,nb", [(5, 30, 30, 1500, 251), (4, 30, 30, 1001, 251), (3, 64, 64, 700, 100), (6, 5, 3, 37, 9),
                                             (2, 31, 40, 1534, 0), (3, 1, 2, 130, 1), (2, 33, 33, 1536, 1536), (7, 2, 1, 5, 5)])
def test_affine_post_fused_matches_oracle(E, Cin, Cout, T, nb):
    """lbm_affine_post_f32 (ica.apply + baseline + reject + z-score in one pass) against the oracle's separate steps
    (training/preprocess.py:144,262-287; multi-train/withpsd.py:213-219): ragged channel counts (odd Cin, Cin = 1, Cout > 32),
    T not a multiple of 4, single- and multi-wave rows, baseline windows of 0 / 1 / all samples."""
    rs = np.random.RandomState(E * 1000 + T)
    x = (synth_epochs(E, Cin, T, seed=E + T).numpy() if T >= 64 else rs.randn(E, Cin, T).astype(np.float32) * 1e-5)
    x = x * np.float32(2.0) + np.float32(3e-6) * rs.randn(1, Cin, 1).astype(np.float32)
    x[E // 2, 0, T // 2] += np.float32(4e-4)                              # one artefact -> one epoch may be rejected
    A = rs.randn(Cout, Cin) / np.sqrt(Cin)
    b = rs.randn(Cout) * 1e-5
    out = lbm_amd.affine_post(torch.from_numpy(x).to(DEV), A, b, n_baseline=nb, reject=150e-6, standardize=True)
    y = np.einsum("oc,ect->eot", A, x.astype(np.float64)) + b[None, :, None]
    wb = orc.baseline_correct(y, nb) if nb > 0 else y
    ws = orc.zscore_epoch_channel(wb, 1e-6)
    scale = np.sqrt((y ** 2).mean()) + 1e-300
    assert float(np.abs(out["base"].cpu().numpy() - wb).max() / scale) <= 1e-5
    assert float(np.abs(out["std"].cpu().numpy() - ws).max()) <= 2e-5
    np.testing.assert_allclose(out["row_scale"].cpu().numpy(), wb.std(axis=-1) + 1e-6, rtol=1e-5)
    ptp = y.max(axis=-1) - y.min(axis=-1)
    safe = np.abs(ptp - 150e-6).min(axis=1) > 1e-9                        # epochs whose verdict does not hang on rounding
    assert np.array_equal(out["keep"].cpu().numpy()[safe], orc.ptp_keep_mask(y, 150e-6)[safe])
    # the separate kernels give the same thing (to fp32 rounding), and outputs can be switched off one by one
    xs = torch.from_numpy(x).to(DEV)
    clean = lbm_amd.affine_channels(xs, A, b)
    b2, s2, k2 = lbm_amd.epoch_post(clean, n_baseline=nb, reject=150e-6, standardize=True)
    assert float((out["base"] - b2).abs().max() / scale) <= 1e-5 and float((out["std"] - s2).abs().max()) <= 2e-5
    only_std = lbm_amd.affine_post(xs, A, b, n_baseline=nb, standardize=True, want_base=False)
    assert set(only_std) == {"std", "row_scale"} and torch.equal(only_std["std"], out["std"])
    only_base = lbm_amd.affine_post(xs, A, b, n_baseline=nb, reject=150e-6, standardize=False)
    assert set(only_base) == {"base", "keep"} and torch.equal(only_base["base"], out["base"]) and torch.equal(only_base["keep"], out["keep"])


def test_affine_post_limits_and_welch_from_standardized_rows():
    """Shapes beyond the fused kernel return LBM_EUNSUPPORTED (the separate kernels remain); lbm_welch_scaled_f32 on the z-scored
    epochs with row_scale reproduces the PSD of the cleaned epochs (multi-train/withpsd.py:236-250)."""
    assert lbm_amd.affine_post_supported(30, 30, 1500) and lbm_amd.affine_post_supported(64, 256, 1536)
    assert not lbm_amd.affine_post_supported(30, 30, 1537) and not lbm_amd.affine_post_supported(65, 30, 100)
    with pytest.raises(NotImplementedError):
        lbm_amd.affine_post(torch.zeros(1, 4, 2000, device=DEV), np.eye(4))
    assert lbm_amd.affine_post(torch.zeros(0, 4, 100, device=DEV), np.eye(4))["std"].shape == (0, 4, 100)
    x = synth_epochs(6, 30, 1500, seed=77) * 2.0
    kw = synth_ica(30)
    A, b = lbm_amd.ica_projection(**kw)
    out = lbm_amd.affine_post(x.to(DEV), A, b, n_baseline=251, standardize=True)
    want, _ = orc.welch_psd(orc.baseline_correct(orc.ica_apply(x.numpy().astype(np.float64), **kw), 251), 500.0, 4.0, 30.0, n_fft=500,
                            n_overlap=250, window="hann")
    got, _ = lbm_amd.psd_array_welch(out["std"], 500.0, 4.0, 30.0, n_fft=500, n_overlap=250, window="hann", row_scale=out["row_scale"])
    assert _psd_err(got.cpu().numpy(), want) <= 1.0
    # generic (non-folded) Welch kernel and per-segment output take the scale too
    got2, _ = lbm_amd.psd_array_welch(out["std"], 500.0, 4.0, 30.0, n_fft=500, n_overlap=250, window="hann", average=None,
                                      row_scale=out["row_scale"])
    want2, _ = orc.welch_psd(orc.baseline_correct(orc.ica_apply(x.numpy().astype(np.float64), **kw), 251), 500.0, 4.0, 30.0, n_fft=500,
                             n_overlap=250, window="hann", average=None)
    assert _psd_err(got2.cpu().numpy().reshape(6, 30, -1), want2.reshape(6, 30, -1)) <= 1.0


@pytest.mark.parametrize("fuse", [True, False])
def test_pipeline_fused_and_separate_dsp_fronts_agree(shipped, fuse):
    """PipelineConfig.fuse_post: both DSP fronts against the oracle, and against each other at fp32 rounding."""
    taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
    A, b = lbm_amd.ica_projection(**synth_ica(30))
    pipe = lbm_amd.EpochPipeline(lbm_amd.PipelineConfig(fuse_post=fuse), taps, A, b, shipped, shipped["vq_layer.embedding.weight"])
    assert pipe.fused == fuse
    x = synth_epochs(8, 30, 1500, seed=19) * 2.0
    x[6, 2, 900] -= 1e-2
    out = pipe.run(x.to(DEV))
    want = orc.pipeline(x.numpy(), 500.0, taps, A, b, 251, 150e-6, shipped)
    assert _rms_ok(out["cleaned"].cpu().numpy(), want["filtered"]) <= 2e-5
    assert np.array_equal(out["keep"].cpu().numpy().astype(bool), want["keep"])
    assert _psd_err(out["psd"].cpu().numpy(), want["psd"]) <= PIPELINE_PSD_TOL
    assert float(np.abs(out["standardized"].cpu().numpy() - want["standardized"]).max()) <= 1e-4
    _assert_tokens(out["tokens"].cpu().numpy(), want["tokens"], want["z"], shipped["vq_layer.embedding.weight"],
                   z_build=out["z"].cpu().numpy())


# ------------------------------------------------------------------------------------------------ a12-a14 VQ
def test_vq_golden_get_code_indices(shipped, vq_golden):
    cb = shipped["vq_layer.embedding.weight"]
    idx = lbm_amd.vq_argmin(torch.from_numpy(vq_golden["gci_in"]).to(DEV), torch.from_numpy(cb).to(DEV)).cpu().numpy()
    assert idx.dtype == np.int64
    ok = orc.vq_near_tie_ok(vq_golden["gci_in"], cb, idx, vq_golden["gci_idx"])
    assert ok.all()
    assert (idx[:8] == 406).all()                      # exact ties among the 100 all-zero codes: lowest index
    assert (idx[8:16] == vq_golden["gci_idx"][8:16]).all()
    assert (idx == vq_golden["gci_idx"]).mean() >= MIN_TOKEN_AGREEMENT


def test_vq_duplicate_codes_lowest_index_wins():
    rs = np.random.RandomState(3)
    cb = rs.randn(200, 256).astype(np.float32)
    cb[150] = cb[20]; cb[199] = cb[20]; cb[64] = cb[63]      # duplicates across lane halves, tiles and sub-tiles
    x = cb[[20, 63, 150, 199, 64, 5]] + 0.0
    idx = lbm_amd.vq_argmin(torch.from_numpy(x).to(DEV), torch.from_numpy(cb).to(DEV)).cpu().numpy()
    assert idx.tolist() == [20, 63, 20, 20, 63, 5]


@pytest.mark.parametrize("N,D,K", [(1000, 256, 512), (129, 256, 8192), (77, 64, 100), (300, 128, 65), (5, 72, 7), (128, 256, 1)])
def test_vq_random_codebooks_match_oracle(N, D, K):
    rs = np.random.RandomState(N + K)
    cb = rs.randn(K, D).astype(np.float32)
    x = (cb[rs.randint(0, K, N)] + 0.7 * rs.randn(N, D)).astype(np.float32)
    hist = torch.zeros(K, dtype=torch.int64, device=DEV)
    idx, mind = lbm_amd.vq_argmin(torch.from_numpy(x).to(DEV), torch.from_numpy(cb).to(DEV), return_min=True, hist=hist)
    idx = idx.cpu().numpy()
    ref = orc.vq_argmin(x, cb)
    assert orc.vq_near_tie_ok(x, cb, idx, ref).all()
    assert (idx == ref).mean() >= 0.99
    d64 = orc.vq_distances_f64(x, cb)
    np.testing.assert_allclose(mind.cpu().numpy(), d64[np.arange(N), idx], rtol=1e-4, atol=1e-3)
    assert np.array_equal(hist.cpu().numpy(), np.bincount(idx, minlength=K))
    # strided rows (ldx > D) and channel-major layout give the same indices
    wide = torch.zeros(N, D + 8, device=DEV); wide[:, :D] = torch.from_numpy(x).to(DEV)
    assert np.array_equal(lbm_amd.vq_argmin(wide[:, :D], torch.from_numpy(cb).to(DEV)).cpu().numpy(), idx)
    if N % 5 == 0:
        S = N // 5
        z = torch.from_numpy(x).to(DEV).view(5, S, D).permute(0, 2, 1).contiguous()   # [B, D, S]
        assert np.array_equal(lbm_amd.vq_argmin(z, torch.from_numpy(cb).to(DEV), channel_major=True).cpu().numpy(), idx)
        zp = torch.full((5, D, S + 3), float("nan"), device=DEV)                      # pitched rows, padding never read
        zp[:, :, :S] = z
        assert np.array_equal(lbm_amd.vq_argmin(zp[:, :, :S], torch.from_numpy(cb).to(DEV), channel_major=True).cpu().numpy(), idx)


@pytest.mark.parametrize("N,D,K", [(1000, 256, 512), (129, 256, 8192), (77, 64, 100), (300, 128, 65), (5, 72, 7), (128, 256, 1),
                                   (4000, 250, 300)])
def test_vq_bf16x6_matches_oracle_and_fp32_kernel(N, D, K):
    """lbm_vq_argmin_bf16x6_f32 (fp32 products rebuilt from three bf16 pieces): same near-tie rule, same layouts."""
    rs = np.random.RandomState(N + K)
    cb = rs.randn(K, D).astype(np.float32)
    x = (cb[rs.randint(0, K, N)] + 0.7 * rs.randn(N, D)).astype(np.float32)
    xd, cbd = torch.from_numpy(x).to(DEV), torch.from_numpy(cb).to(DEV)
    tab = lbm_amd.vq_split_tables(cbd)
    hist = torch.zeros(K, dtype=torch.int64, device=DEV)
    idx, mind = lbm_amd.vq_argmin(xd, cbd, return_min=True, hist=hist, split_tables=tab)
    idx = idx.cpu().numpy()
    ref = orc.vq_argmin(x, cb)
    assert orc.vq_near_tie_ok(x, cb, idx, ref).all()
    assert (idx == ref).mean() >= 0.99
    d64 = orc.vq_distances_f64(x, cb)
    np.testing.assert_allclose(mind.cpu().numpy(), d64[np.arange(N), idx], rtol=1e-4, atol=1e-3)
    assert np.array_equal(hist.cpu().numpy(), np.bincount(idx, minlength=K))
    if D % 8 == 0:
        assert (idx == lbm_amd.vq_argmin(xd, cbd).cpu().numpy()).mean() >= 0.999
    wide = torch.full((N, D + 8), float("nan"), device=DEV); wide[:, :D] = xd
    assert np.array_equal(lbm_amd.vq_argmin(wide[:, :D], cbd, split_tables=tab).cpu().numpy(), idx)
    if N % 5 == 0:
        S = N // 5
        zp = torch.full((5, D, S + 3), float("nan"), device=DEV)                      # pitched channel-major rows
        zp[:, :, :S] = xd.view(5, S, D).permute(0, 2, 1)
        assert np.array_equal(lbm_amd.vq_argmin(zp[:, :, :S], cbd, channel_major=True, split_tables=tab).cpu().numpy(), idx)


def test_vq_bf16x6_golden_and_duplicates(shipped, vq_golden):
    cb = shipped["vq_layer.embedding.weight"]
    cbd = torch.from_numpy(cb).to(DEV)
    tab = lbm_amd.vq_split_tables(cbd)
    idx = lbm_amd.vq_argmin(torch.from_numpy(vq_golden["gci_in"]).to(DEV), cbd, split_tables=tab).cpu().numpy()
    assert orc.vq_near_tie_ok(vq_golden["gci_in"], cb, idx, vq_golden["gci_idx"]).all()
    zg = torch.from_numpy(vq_golden["z_S"]).to(DEV)
    tk = lbm_amd.vq_argmin(zg, cbd, channel_major=True, split_tables=tab).cpu().numpy()
    assert orc.vq_near_tie_ok(np.transpose(vq_golden["z_S"], (0, 2, 1)).reshape(-1, 256), cb, tk, vq_golden["tokens_S"].reshape(-1)).all()
    # exact duplicate codes: the lowest index wins (piece products of equal rows are equal)
    rs = np.random.RandomState(2)
    c2 = rs.randn(300, 256).astype(np.float32)
    c2[299] = c2[17]; c2[150] = c2[17]
    x = (c2[[17, 150, 299, 5]] + 1e-3 * rs.randn(4, 256)).astype(np.float32)
    c2d = torch.from_numpy(c2).to(DEV)
    got = lbm_amd.vq_argmin(torch.from_numpy(x).to(DEV), c2d, split_tables=lbm_amd.vq_split_tables(c2d)).cpu().numpy()
    assert got.tolist() == [17, 17, 17, 5]


@pytest.mark.parametrize("N,D,K", [(1000, 256, 512), (129, 256, 8192), (77, 64, 100), (300, 128, 65), (5, 72, 7), (128, 256, 1),
                                   (4000, 248, 300), (20000, 256, 512),
                                   # D = 256 is the persistent kernel: more than 2 x 256 blocks of 128 rows make its groups walk
                                   # several blocks each (unevenly); one, two (ragged) and eight tiles per period
                                   (140000, 256, 512), (70005, 256, 100), (66000, 256, 64)])
def test_vq_screened_is_the_fp32_kernel_bit_for_bit(N, D, K):
    """lbm_vq_argmin_screened_f32 (bf16 screening pass with a rigorous bound + fp32 search of the unsettled rows): EXACTLY the
    indices of lbm_vq_argmin_f32, on spread-out rows, rows at and next to midpoints between codes, copies of codes, tiny and huge
    rows -- in both layouts, with and without the row constant, histogram included."""
    rs = np.random.RandomState(N + K)
    scale = rs.choice([0.3, 1.0, 30.0], size=(K, 1))
    cb = (rs.randn(K, D) * scale).astype(np.float32)                                          # norms over two decades
    x = (cb[rs.randint(0, K, N)] + 0.7 * rs.randn(N, D)).astype(np.float32)
    q = N // 5
    small = np.flatnonzero(scale[:, 0] == 0.3)                 # midpoints of two short codes: those two are the nearest, and tie
    small = small if small.size >= 2 else np.arange(K)
    i0, i1 = small[rs.randint(0, small.size, q)], small[rs.randint(0, small.size, q)]
    x[:q] = 0.5 * (cb[i0] + cb[i1])                                                            # exact midpoints
    x[q:2 * q] = x[:q] * (1 + 1e-7) + 1e-6 * rs.randn(q, D)                                    # next to them
    x[2 * q:2 * q + min(q, K)] = cb[:min(q, K)]                                                # the codes themselves
    if N > 10:
        x[-3] *= 1e4; x[-2] *= 1e-5; x[-1] = 0.0
    xd, cbd = torch.from_numpy(x).to(DEV), torch.from_numpy(cb).to(DEV)
    tab = lbm_amd.vq_split_tables(cbd)
    for no_xnorm in (False, True):
        ee = torch.from_numpy((cb.astype(np.float64) ** 2).sum(1) - (3.0 * cb[:, 0] if no_xnorm else 0.0)).to(DEV, torch.float32)
        hist0 = torch.zeros(K, dtype=torch.int64, device=DEV)
        hist1 = torch.zeros(K, dtype=torch.int64, device=DEV)
        ref = lbm_amd.vq_argmin(xd, cbd, ee, no_xnorm=no_xnorm, hist=hist0)
        got, nref = lbm_amd.vq_argmin(xd, cbd, ee, no_xnorm=no_xnorm, hist=hist1, split_tables=tab, screened=True, return_refined=True)
        assert torch.equal(got, ref)
        assert torch.equal(hist0, hist1)
        assert 0 <= nref <= N
        if K > 1 and N >= 1000 and not no_xnorm:   # (with the shifted constants the midpoints are no ties)
            assert nref >= q // 2            # the midpoints cannot be settled by the screen
            assert nref <= N - N // 4        # and the spread-out rows are
        wide = torch.full((N, D + 8), float("nan"), device=DEV); wide[:, :D] = xd
        assert torch.equal(lbm_amd.vq_argmin(wide[:, :D], cbd, ee, no_xnorm=no_xnorm, split_tables=tab, screened=True), ref)
        if N <= 20000:   # rows that start 4 bytes off a 16-byte boundary: the kernels' 4-byte load paths
            flat = torch.empty(N * D + 1, device=DEV)
            flat[1:] = xd.reshape(-1)
            assert torch.equal(lbm_amd.vq_argmin(flat[1:].view(N, D), cbd, ee, no_xnorm=no_xnorm, split_tables=tab, screened=True), ref)
        if N % 5 == 0:
            S = N // 5
            zp = torch.full((5, D, S + 3), float("nan"), device=DEV)                           # pitched channel-major rows
            zp[:, :, :S] = xd.view(5, S, D).permute(0, 2, 1)
            assert torch.equal(lbm_amd.vq_argmin(zp[:, :, :S], cbd, ee, channel_major=True, no_xnorm=no_xnorm, split_tables=tab,
                                                 screened=True), ref)


@pytest.mark.parametrize("K,fold16", [(128, False), (512, True)])
def test_vq_screened_operands_built_against_the_matrix_cores_adder(K, fold16):
    """The screening bound prices one bf16 MFMA accumulate at 16 * 2^-24 (|C| + sum |a b|) (csrc/vq_screen.hip; measured <= 7.8, modelled
    worst 9.5 on operands built against an adder that aligns all 16 products of a k-step to the largest and drops what falls below its
    guard bits -- tools/probe/mfma_acc_probe.hip).  Here such operands go through the PRODUCT: in every 16-dimension k-step one dimension
    carries a large product and fifteen carry same-sign products 2^-24 .. 2^-25 of it (the cut), rows are copies of a code a_p, and each a_p
    has a twin b_p whose distance from it sweeps from 0.03 to 8 times the bound, so the screen settles some rows and must pass the others
    on.  Whatever it settles has to be the fp32 search's answer, bit for bit."""
    rs = np.random.RandomState(K)
    D, P = 256, K // 2
    big = np.zeros((P, D), np.float32)
    for g in range(D // 16):
        jb = g * 16 + rs.randint(0, 16, P)
        big[np.arange(P), jb] = (8.0 * (1.0 + rs.rand(P))).astype(np.float32)
    bigmax = big.reshape(P, D // 16, 16).max(2).repeat(16, axis=1)                       # the k-step's large value, per dimension
    small = (bigmax * 2.0 ** -12.25 * (0.72 + 0.28 * rs.rand(P, D))).astype(np.float32)  # small x small ~ 2^-24.5 (0.5 .. 1) large x large
    a = np.where(big > 0, big, small).astype(np.float32)
    if fold16:      # every other pair: the large products alternate in sign between k-steps (the running sum stays small, the cut relative to it large)
        sgn = np.where((np.arange(D) // 16) % 2 == 0, 1.0, -1.0).astype(np.float32)
        a[1::2] *= sgn
    t = (10.0 ** rs.uniform(-2.6, -0.35, P)).astype(np.float32)                          # gap a_j^2 t^2 against B ~ 1.5e-4 * 16 a_j^2 * 2
    b = a.copy()
    j0 = np.abs(a).argmax(1)
    b[np.arange(P), j0] *= (1.0 + t)
    cb = np.empty((K, D), np.float32)
    cb[0::2], cb[1::2] = a, b
    N = 64 * P
    rows = np.tile(np.arange(P), 64)
    x = a[rows].copy()
    x[N // 2:] = b[rows[N // 2:]]                                                        # the other half: copies of the twins
    xd, cbd = torch.from_numpy(x).to(DEV), torch.from_numpy(cb).to(DEV)
    ref = lbm_amd.vq_argmin(xd, cbd)
    got, nref = lbm_amd.vq_argmin(xd, cbd, split_tables=lbm_amd.vq_split_tables(cbd), screened=True, return_refined=True)
    assert torch.equal(got, ref)
    want = np.concatenate([2 * rows[:N // 2], 2 * rows[N // 2:] + 1])
    assert np.array_equal(ref.cpu().numpy(), want)                                       # (the fp32 search itself resolves every twin)
    assert N // 20 <= nref <= N - N // 20                                                # both sides of the bound were exercised


def test_vq_screened_golden_duplicates_and_nan(shipped, vq_golden):
    cb = shipped["vq_layer.embedding.weight"]
    cbd = torch.from_numpy(cb).to(DEV)
    tab = lbm_amd.vq_split_tables(cbd)
    gx = torch.from_numpy(vq_golden["gci_in"]).to(DEV)
    assert torch.equal(lbm_amd.vq_argmin(gx, cbd, split_tables=tab, screened=True), lbm_amd.vq_argmin(gx, cbd))
    for name in ("S", "R", "P"):
        zg = torch.from_numpy(vq_golden[f"z_{name}"]).to(DEV)
        tk = lbm_amd.vq_argmin(zg, cbd, channel_major=True, split_tables=tab, screened=True)
        assert torch.equal(tk, lbm_amd.vq_argmin(zg, cbd, channel_major=True))
        zr = np.transpose(vq_golden[f"z_{name}"], (0, 2, 1)).reshape(-1, 256)
        assert orc.vq_near_tie_ok(zr, cb, tk.cpu().numpy(), vq_golden[f"tokens_{name}"].reshape(-1)).all()
    # exact duplicate codes: the lowest index wins, as in the fp32 kernel (such rows always go through the fp32 pass)
    rs = np.random.RandomState(2)
    c2 = rs.randn(300, 256).astype(np.float32)
    c2[299] = c2[17]; c2[150] = c2[17]
    x = (c2[[17, 150, 299, 5]] + 1e-3 * rs.randn(4, 256)).astype(np.float32)
    x = np.concatenate([x, np.full((1, 256), np.nan, np.float32), x[:1] * np.float32(np.inf)])
    c2d, xd = torch.from_numpy(c2).to(DEV), torch.from_numpy(x).to(DEV)
    got, nref = lbm_amd.vq_argmin(xd, c2d, split_tables=lbm_amd.vq_split_tables(c2d), screened=True, return_refined=True)
    assert got.tolist()[:4] == [17, 17, 17, 5]
    assert torch.equal(got, lbm_amd.vq_argmin(xd, c2d))          # NaN / inf rows: whatever the fp32 kernel says
    assert nref >= 5
    with pytest.raises(ValueError):
        lbm_amd.vq_argmin(xd, c2d, screened=True)
    with pytest.raises(ValueError):
        lbm_amd.vq_argmin(xd, c2d, split_tables=lbm_amd.vq_split_tables(c2d), screened=True, return_min=True)


@pytest.mark.parametrize("fold", [False, True])
def test_pipeline_screened_quantizer_is_bit_identical(shipped, fold):
    """vq_method='screen': every output of the pipeline equals the fp32 quantizer's, tokens and histogram bit for bit."""
    taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
    A, b = lbm_amd.ica_projection(**synth_ica(30))
    cb = shipped["vq_layer.embedding.weight"]
    mk = lambda m: lbm_amd.EpochPipeline(lbm_amd.PipelineConfig(fold_projection=fold, vq_method=m, emit_cleaned=False),  # noqa: E731
                                         taps, A, b, shipped, cb)
    x = (synth_epochs(200, 30, 1500, seed=41) * 2.0).to(DEV)
    ref = {k: v.clone() for k, v in mk("fp32").run(x).items()}
    pipe = mk("screen")
    for _ in range(2):          # twice: the workspace and its row counter are reused
        out = pipe.run(x)
        assert torch.equal(out["tokens"], ref["tokens"])
        assert torch.equal(out["psd"], ref["psd"]) and torch.equal(out["keep"], ref["keep"])
    n_ref = int(pipe._buf["vq_ws"][:4].view(torch.int32).item())
    assert 0 < n_ref < 0.1 * ref["tokens"].numel()
    assert torch.equal(out["hist"], 2 * ref["hist"])


def test_vqvae_encode_screened_option(shipped, vq_golden):
    """VQVAE.encode(screened=True) / encode_data_for_transformer(screened=True): the indices of the default search bit for bit, folded
    or not, also after an in-place EMA update of the codebook (the cached piece tables follow it)."""
    m = lbm_amd.VQVAE(30, 256, 512, 0.25, 0.99)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in shipped.items()}, strict=False)
    m = m.to(DEV).eval()
    x = torch.from_numpy(vq_golden["x_S"]).to(DEV)
    def same(fold):      # indices AND histogram (accumulating over two calls) of the screened search == the default search's
        h1 = torch.zeros(512, dtype=torch.int64, device=DEV)
        h2 = torch.zeros(512, dtype=torch.int64, device=DEV)
        for _ in range(2):
            assert torch.equal(m.encode(x, hist=h1, fold_projection=fold, screened=True), m.encode(x, hist=h2, fold_projection=fold))
        assert torch.equal(h1, h2) and int(h1.sum()) == 2 * x.shape[0] * (x.shape[2] // 4)
    for fold in (False, True):
        same(fold)
        # the shipped codebook's 100 all-zero rows are not searched (quantizer.dedup_codebook): 413 distinct rows, padded to 448
        cb, ee, tab, cmap, kc = m._screen_set(fold)
        assert (cb.shape[0], kc) == (448, 413) and cmap is not None and int(cmap[:kc].max()) <= 511
    m.train()
    with torch.no_grad():
        m(torch.from_numpy(vq_golden["x_P"]).to(DEV))          # one EMA step rewrites the codebook in place
    m.eval()
    for fold in (False, True):
        same(fold)
    loader = [(torch.from_numpy(vq_golden["x_S"]), torch.zeros(2, dtype=torch.int64))]
    a, _ = lbm_amd.encode_data_for_transformer(m, loader, DEV, screened=True)
    b, _ = lbm_amd.encode_data_for_transformer(m, loader, DEV)
    assert np.array_equal(a, b)


def test_vqvae_module_encode_and_forward_golden(shipped, vq_golden):
    m = lbm_amd.VQVAE(30, 256, 512, 0.25, 0.99)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in shipped.items()}, strict=False)
    m = m.to(DEV).eval()
    cb = shipped["vq_layer.embedding.weight"]
    for name in ("S", "R", "P"):
        x = torch.from_numpy(vq_golden[f"x_{name}"]).to(DEV)
        tok = m.encode(x).cpu().numpy()
        ref = vq_golden[f"tokens_{name}"]
        assert tok.shape == ref.shape and tok.dtype == np.int64
        z = lbm_amd.encoder_forward(x, m.encoder).permute(0, 2, 1).reshape(-1, 256).cpu().numpy()
        # same-input near-tie rule (the encoder output differs from the CPU one by fp32 rounding)
        assert orc.vq_near_tie_ok(z, cb, tok.reshape(-1), orc.vq_argmin(z, cb)).all()
        # against the reference's own tokens, judged on the reference's own z (golden)
        _assert_tokens(tok, ref, vq_golden[f"z_{name}"], cb, z_build=z)
    # VectorQuantizerEMA.forward in eval mode (training/model.py:84-149)
    with torch.no_grad():
        loss, quant, perp, idx = m.vq_layer(torch.from_numpy(vq_golden["vqfwd_in"]).to(DEV))
    assert orc.vq_near_tie_ok(vq_golden["vqfwd_in"].reshape(-1, 256), cb, idx.cpu().numpy().reshape(-1),
                              vq_golden["vqfwd_indices"].reshape(-1)).all()
    same = idx.cpu().numpy() == vq_golden["vqfwd_indices"]
    assert same.mean() >= MIN_TOKEN_AGREEMENT
    if same.all():
        np.testing.assert_allclose(quant.cpu().numpy(), vq_golden["vqfwd_quantized"], rtol=0, atol=0)
        np.testing.assert_allclose(float(loss), vq_golden["vqfwd_loss"], rtol=1e-5)
        np.testing.assert_allclose(float(perp), vq_golden["vqfwd_perplexity"], rtol=1e-5)
    # get_code_indices keeps the reference signature: flat [N, D] in, int64 [N] out
    gi = m.vq_layer.get_code_indices(torch.from_numpy(vq_golden["gci_in"]).to(DEV))
    assert gi.shape == (1024,) and gi.dtype == torch.int64


def test_vqvae_forward_golden(shipped, vq_golden):
    """VQVAE.forward in eval mode (training/model.py:195-203) against the imported reference's outputs on the shipped checkpoint
    (oracle/gen_golden.py: fwd_vq_loss, fwd_perplexity, fwd_indices, fwd_x_recon).  The quantizer runs on the HIP kernels; the
    encoder / decoder of this training-side entry point run through torch (training is off the hot path)."""
    m = lbm_amd.VQVAE(30, 256, 512, 0.25, 0.99)
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in shipped.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys          # vq_vae_model.pth's keys, all of them
    m = m.to(DEV).eval()
    x = torch.from_numpy(vq_golden["x_P"]).to(DEV)
    with torch.no_grad():
        vq_loss, x_recon, perplexity, indices = m(x)
    assert indices.shape == (4, 9) and indices.dtype == torch.int64 and x_recon.shape == (4, 30, 39)
    idx = indices.cpu().numpy()
    z = lbm_amd.encoder_forward(x, m.encoder).cpu().numpy()
    _assert_tokens(idx, vq_golden["fwd_indices"], vq_golden["z_P"], shipped["vq_layer.embedding.weight"], z_build=z)
    if np.array_equal(idx, vq_golden["fwd_indices"]):
        np.testing.assert_allclose(float(vq_loss), float(vq_golden["fwd_vq_loss"]), rtol=1e-5)
        np.testing.assert_allclose(float(perplexity), float(vq_golden["fwd_perplexity"]), rtol=1e-5)
        ref = vq_golden["fwd_x_recon"]
        assert float(np.abs(x_recon.cpu().numpy() - ref).max()) <= 2e-5 * float(np.abs(ref).max())
    # the same call with gradients enabled (the training loop's forward, training/model.py:349): same numbers, differentiable
    m.train(False)
    xg = x.clone().requires_grad_(True)
    l2, xr2, p2, i2 = m(xg)
    assert torch.equal(i2, indices)
    np.testing.assert_allclose(float(l2.detach()), float(vq_loss), rtol=1e-5)
    assert float((xr2.detach() - x_recon).abs().max()) <= 2e-5 * float(x_recon.abs().max())
    (l2 + xr2.square().mean()).backward()
    assert xg.grad is not None and bool(torch.isfinite(xg.grad).all())


def test_vq_forward_tail_kernels_with_reference_indices(shipped, vq_golden):
    """a13 (training/model.py:100,130-144), independent of argmin near-ties: lbm_vq_gather_loss_f32 fed the REFERENCE's
    indices and lbm_perplexity_u64 fed the REFERENCE's counts must give the reference's quantized / loss / perplexity."""
    cb = shipped["vq_layer.embedding.weight"]
    cbd = torch.from_numpy(cb).to(DEV)
    flat = torch.from_numpy(vq_golden["vqfwd_in"].reshape(-1, 256)).to(DEV)
    ridx = torch.from_numpy(vq_golden["vqfwd_indices"].reshape(-1)).to(DEV)
    q, err = lbm_amd.vq_gather_loss(flat, cbd, ridx)
    np.testing.assert_array_equal(q.cpu().numpy().reshape(vq_golden["vqfwd_quantized"].shape), vq_golden["vqfwd_quantized"])
    mse = float(err[0]) / flat.numel()
    np.testing.assert_allclose((1.0 + 0.25) * mse, float(vq_golden["vqfwd_loss"]), rtol=1e-5)
    counts = torch.bincount(ridx, minlength=512)
    perp = lbm_amd.perplexity_from_hist(counts)
    np.testing.assert_allclose(float(perp), float(vq_golden["vqfwd_perplexity"]), rtol=1e-5)
    # want_quantized=False leaves the loss unchanged; the sum is ADDED to (caller zeroes)
    _, err2 = lbm_amd.vq_gather_loss(flat, cbd, ridx, want_quantized=False)
    assert abs(float(err2[0]) - float(err[0])) <= 1e-12 * float(err[0])      # float64 atomics: order-dependent last bits
    # the EMA-step golden (larger input, 512 rows): loss of the training forward = same formula on the pre-update codebook
    flat_e = torch.from_numpy(vq_golden["ema_in"].reshape(-1, 256)).to(DEV)
    ridx_e = torch.from_numpy(vq_golden["ema_idx"].reshape(-1)).to(DEV)
    _, err_e = lbm_amd.vq_gather_loss(flat_e, cbd, ridx_e, want_quantized=False)
    np.testing.assert_allclose(1.25 * float(err_e[0]) / flat_e.numel(), float(vq_golden["ema_loss"]), rtol=1e-5)
    np.testing.assert_allclose(float(lbm_amd.perplexity_from_hist(torch.bincount(ridx_e, minlength=512))),
                               float(vq_golden["ema_perplexity"]), rtol=1e-5)
    # perplexity against the oracle on ragged K and skewed counts
    rs = np.random.RandomState(4)
    for K in (1, 7, 512, 8192):
        c = (rs.exponential(3.0, K) * (rs.rand(K) > 0.4)).astype(np.int64)
        c[0] += 1
        got = float(lbm_amd.perplexity_from_hist(torch.from_numpy(c).to(DEV)))
        np.testing.assert_allclose(got, orc.perplexity_from_counts(c), rtol=2e-5)


def test_vq_ema_two_training_steps_use_the_updated_codebook(shipped, vq_golden):
    """Two training-mode forwards in a row under no_grad, no optimizer: the second step's argmin must run against
    the codebook the first step's EMA kernel wrote -- with ITS row norms, not the cached ones."""
    m = lbm_amd.VQVAE(30, 256, 512, 0.25, 0.99)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in shipped.items()}, strict=False)
    vq = m.vq_layer.to(DEV).train()
    rs = np.random.RandomState(8)
    x1 = vq_golden["ema_in"]
    x2 = (vq_golden["ema_in"][::-1] * 1.5 + 0.3 * rs.randn(*x1.shape)).astype(np.float32)
    with torch.no_grad():
        vq.decay = 0.5                                  # a large step, so the codebook (and its norms) really move
        _, _, _, idx1 = vq(torch.from_numpy(x1).to(DEV))
        w1 = vq.embedding.weight.detach().cpu().numpy().copy()
        _, _, _, idx2 = vq(torch.from_numpy(x2).to(DEV))
    # oracle: step 1 with the build's own step-1 indices, then argmin of step 2 against the UPDATED codebook
    st = {"ema_cluster_size": shipped["vq_layer.ema_cluster_size"], "ema_w": shipped["vq_layer.ema_w"]}
    cs, ew, w_or = orc.ema_update(st, x1.reshape(-1, 256), idx1.cpu().numpy().reshape(-1), decay=0.5)
    np.testing.assert_allclose(w1, w_or, rtol=1e-5, atol=1e-6)
    assert np.abs(w1 - shipped["vq_layer.embedding.weight"]).max() > 1e-2
    flat2 = x2.reshape(-1, 256)
    ref2 = orc.vq_argmin(flat2, w1)
    got2 = idx2.cpu().numpy().reshape(-1)
    assert orc.vq_near_tie_ok(flat2, w1, got2, ref2).all()
    assert (got2 == ref2).mean() >= MIN_TOKEN_AGREEMENT
    # what the round-1 bug computed: the NEW codebook with the OLD row norms.  It must differ from the right answer by
    # more than near-ties, or this test could not have caught it
    w0 = shipped["vq_layer.embedding.weight"].astype(np.float64)
    d_stale = (w0 * w0).sum(1)[None, :] - 2.0 * (flat2.astype(np.float64) @ w1.astype(np.float64).T)
    stale = d_stale.argmin(axis=1)
    assert (~orc.vq_near_tie_ok(flat2, w1, stale, ref2)).sum() >= 3


def test_vq_histogram_up_to_32768_codes():
    """hist_kernel keeps one LDS bin per code: 128 KiB at K = 32768 (opt-in dynamic LDS).  Counts bit-exact."""
    rs = np.random.RandomState(6)
    for K in (16384, 16392, 32768):
        cb = rs.randn(K, 8).astype(np.float32)
        pick = rs.randint(0, K, 20000)
        x = cb[pick] + 0.01 * rs.randn(20000, 8).astype(np.float32)
        hist = torch.zeros(K, dtype=torch.int64, device=DEV)
        idx = lbm_amd.vq_argmin(torch.from_numpy(x).to(DEV), torch.from_numpy(cb).to(DEV), hist=hist).cpu().numpy()
        assert np.array_equal(hist.cpu().numpy(), np.bincount(idx, minlength=K))
        assert (idx == pick).mean() > 0.99
    with pytest.raises(NotImplementedError):
        K = 32776
        lbm_amd.vq_argmin(torch.zeros(4, 8, device=DEV), torch.zeros(K, 8, device=DEV), hist=torch.zeros(K, dtype=torch.int64, device=DEV))


def test_vqvae_encode_folded_projection_option(shipped, vq_golden):
    """VQVAE.encode(x, fold_projection=True) / encode_data_for_transformer(..., fold_projection=True): the reference's own
    golden tokens except documented near-ties; the folded codebook is cached and follows weight changes."""
    m = lbm_amd.VQVAE(30, 256, 512, 0.25, 0.99)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in shipped.items()}, strict=False)
    m = m.to(DEV).eval()
    cb = shipped["vq_layer.embedding.weight"]
    for name in ("S", "R", "P"):
        x = torch.from_numpy(vq_golden[f"x_{name}"]).to(DEV)
        tok = m.encode(x, fold_projection=True).cpu().numpy()
        ref = vq_golden[f"tokens_{name}"]
        assert tok.shape == ref.shape and tok.dtype == np.int64
        zg = np.transpose(vq_golden[f"z_{name}"], (0, 2, 1)).reshape(-1, 256)
        assert orc.vq_near_tie_ok(zg, cb, tok.reshape(-1), ref.reshape(-1)).all()
        assert (tok == ref).mean() >= MIN_TOKEN_AGREEMENT
    f1 = m._folded_codebook()
    assert m._folded_codebook() is f1                       # cached
    with torch.no_grad():
        m.encoder[6].bias.add_(0.5)                         # a weight change (version bump) invalidates it
    f2 = m._folded_codebook()
    assert f2 is not f1 and not torch.equal(f2[1], f1[1])
    x = torch.from_numpy(vq_golden["x_P"]).to(DEV)
    assert torch.equal(m.encode(x, fold_projection=True), m.encode(x)) or \
        orc.vq_near_tie_ok(lbm_amd.encoder_forward(x, m.encoder).permute(0, 2, 1).reshape(-1, 256).cpu().numpy(), cb,
                           m.encode(x, fold_projection=True).cpu().numpy().reshape(-1), m.encode(x).cpu().numpy().reshape(-1)).all()
    data = torch.randn(70, 30, 39, generator=torch.Generator().manual_seed(2))
    dl = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(data, torch.arange(70) % 3), batch_size=32)
    a, la = lbm_amd.encode_data_for_transformer(m, dl, torch.device(DEV), fold_projection=True)
    b_, lb = lbm_amd.encode_data_for_transformer(m, dl, torch.device(DEV))
    assert a.shape == b_.shape == (70, 9) and np.array_equal(la, lb)
    zd = lbm_amd.encoder_forward(data.to(DEV), m.encoder).cpu().numpy()
    # folded against unfolded on THIS test's perturbed model (every projection bias + 0.5 pushes z away from the live codes, where
    # many codes are nearly equidistant): measured 0.981, every mismatch a documented near-tie of the unfolded z
    _assert_tokens(a, b_, zd, m.vq_layer.embedding.weight.detach().cpu().numpy(), min_agree=0.98)


def test_vq_ema_training_step_golden(shipped, vq_golden):
    """VectorQuantizerEMA.forward in training mode: EMA buffers and codebook after one step (training/model.py:103-126)."""
    m = lbm_amd.VQVAE(30, 256, 512, 0.25, 0.99)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in shipped.items()}, strict=False)
    vq = m.vq_layer.to(DEV).train()
    x = torch.from_numpy(vq_golden["ema_in"]).to(DEV)
    loss, quant, perp, idx = vq(x)
    same = (idx.cpu().numpy() == vq_golden["ema_idx"])
    _assert_tokens(idx.cpu().numpy(), vq_golden["ema_idx"], vq_golden["ema_in"].reshape(-1, 256), shipped["vq_layer.embedding.weight"])
    if same.all():
        np.testing.assert_allclose(vq.ema_cluster_size.cpu().numpy(), vq_golden["ema_cluster_size"], rtol=1e-6)
        np.testing.assert_allclose(vq.ema_w.cpu().numpy(), vq_golden["ema_w"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(vq.embedding.weight.detach().cpu().numpy(), vq_golden["ema_weight"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(float(loss.detach()), vq_golden["ema_loss"], rtol=1e-5)
    # the kernels against the oracle with the reference's own indices (independent of argmin near-ties)
    flat = vq_golden["ema_in"].reshape(-1, 256)
    ridx = vq_golden["ema_idx"].reshape(-1)
    dw, counts = lbm_amd.vq_ema_sums(torch.from_numpy(flat).to(DEV), torch.from_numpy(ridx).to(DEV), 512)
    assert np.array_equal(counts.cpu().numpy(), np.bincount(ridx, minlength=512))
    cs = torch.from_numpy(shipped["vq_layer.ema_cluster_size"]).to(DEV).clone()
    ew = torch.from_numpy(shipped["vq_layer.ema_w"]).to(DEV).clone()
    w = torch.empty_like(ew)
    lbm_amd.vq_ema_apply(cs, ew, w, dw, counts, 0.99, 1e-5)
    np.testing.assert_allclose(cs.cpu().numpy(), vq_golden["ema_cluster_size"], rtol=1e-6)
    np.testing.assert_allclose(ew.cpu().numpy(), vq_golden["ema_w"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(w.cpu().numpy(), vq_golden["ema_weight"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("N,D,K", [(200000, 256, 512), (50000, 64, 8192), (777, 72, 5), (0, 256, 512), (3, 256, 512)])
def test_vq_ema_sums_match_numpy_and_are_deterministic(N, D, K):
    rs = np.random.RandomState(N % 1000 + K)
    x = rs.randn(N, D).astype(np.float32)
    # skewed code usage: a few popular codes (long runs -> several reduction pieces), many empty ones
    idx = np.minimum((rs.exponential(0.08, N) * K).astype(np.int64), K - 1)
    xd, idd = torch.from_numpy(x).to(DEV), torch.from_numpy(idx).to(DEV)
    dw, counts = lbm_amd.vq_ema_sums(xd, idd, K)
    dw2, counts2 = lbm_amd.vq_ema_sums(xd, idd, K)
    assert torch.equal(dw, dw2) and torch.equal(counts, counts2)          # bitwise reproducible
    assert np.array_equal(counts.cpu().numpy(), np.bincount(idx, minlength=K))
    want = np.zeros((K, D))
    np.add.at(want, idx, x.astype(np.float64))
    scale = np.sqrt(np.maximum(np.bincount(idx, minlength=K), 1))[:, None]
    assert float((np.abs(dw.cpu().numpy() - want) / scale).max()) <= 2e-5


def test_encode_data_for_transformer_signature(shipped):
    m = lbm_amd.VQVAE(30, 256, 512, 0.25, 0.99)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in shipped.items()}, strict=False)
    m = m.to(DEV)
    data = torch.randn(150, 30, 39, generator=torch.Generator().manual_seed(1))
    labels = torch.arange(150) % 3
    dl = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(data, labels), batch_size=lbm_amd.BATCH_SIZE)
    idx, lab = lbm_amd.encode_data_for_transformer(m, dl, torch.device(DEV))
    assert idx.shape == (150, 9) and idx.dtype == np.int64 and np.array_equal(lab, labels.numpy())
    ref, z_ref = orc.vqvae_encode(data.numpy(), shipped)
    _assert_tokens(idx, ref, z_ref, shipped["vq_layer.embedding.weight"],
                   z_build=lbm_amd.encoder_forward(data.to(DEV), m.encoder).cpu().numpy())


# ------------------------------------------------------------------------------------------------ whole pipeline
def _make_pipeline(shipped, T=1500, C=30, fs=500.0, nb=251):
    taps = lbm_amd.create_filter(fs, 4.0, 30.0)
    kw = synth_ica(C)
    A, b = lbm_amd.ica_projection(**kw)
    cfg = lbm_amd.PipelineConfig(sfreq=fs, n_channels=C, n_times=T, n_baseline=nb)
    return lbm_amd.EpochPipeline(cfg, taps, A, b, shipped, shipped["vq_layer.embedding.weight"]), taps, A, b


@pytest.mark.parametrize("T", [1500, 1001])
def test_pipeline_matches_oracle(shipped, T):
    pipe, taps, A, b = _make_pipeline(shipped, T=T)
    x = synth_epochs(6, 30, T, seed=10) * 2.0
    x[4, 7, 700] += 1e-2       # artefact -> epoch 4 rejected
    out = pipe.run(x.to(DEV))
    torch.cuda.synchronize()
    want = orc.pipeline(x.numpy(), 500.0, taps, A, b, 251, 150e-6, shipped)
    assert _rms_ok(out["cleaned"].cpu().numpy(), want["filtered"]) <= 2e-5
    assert np.array_equal(out["keep"].cpu().numpy().astype(bool), want["keep"])
    assert _psd_err(out["psd"].cpu().numpy(), want["psd"]) <= PIPELINE_PSD_TOL
    np.testing.assert_allclose(out["bandpower"].cpu().numpy(), want["bandpower"], rtol=2e-6)
    assert float(np.abs(out["standardized"].cpu().numpy() - want["standardized"]).max()) <= 1e-4
    tok = out["tokens"].cpu().numpy()
    assert tok.shape == want["tokens"].shape
    z = out["z"].permute(0, 2, 1).reshape(-1, 256).cpu().numpy()
    cb = shipped["vq_layer.embedding.weight"]
    assert orc.vq_near_tie_ok(z, cb, tok.reshape(-1), orc.vq_argmin(z, cb)).all()
    _assert_tokens(tok, want["tokens"], want["z"], cb, z_build=out["z"].cpu().numpy())
    assert np.array_equal(out["hist"].cpu().numpy(), np.bincount(tok.reshape(-1), minlength=512))


@pytest.mark.parametrize("vq_method", ["fp32", "bf16x6"])
@pytest.mark.parametrize("codebook", ["shipped", "random"])
def test_pipeline_folded_projection_tokens(shipped, vq_golden, codebook, vq_method):
    """the last 1x1 convolution folded into the codebook (fold_projection) gives the tokens of the
    unfolded path; a mismatch must be a documented near-tie of the unfolded path's own z (same 8-eps rule)."""
    cb = shipped["vq_layer.embedding.weight"] if codebook == "shipped" else np.asarray(synth_codebook(512, 256)) * 0.3
    taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
    A, b = lbm_amd.ica_projection(**synth_ica(30))
    mk = lambda fold: lbm_amd.EpochPipeline(lbm_amd.PipelineConfig(fold_projection=fold, vq_method=vq_method if fold else "fp32"),  # noqa: E731
                                            taps, A, b, shipped, cb)
    x = (synth_epochs(96, 30, 1500, seed=21) * 2.0).to(DEV)
    ref = mk(False).run(x)
    tok_ref, z = ref["tokens"].cpu().numpy(), ref["z"].permute(0, 2, 1).reshape(-1, 256).cpu().numpy()
    out = mk(True).run(x)
    assert "z" not in out
    tok = out["tokens"].cpu().numpy()
    assert (tok == tok_ref).mean() >= 0.999
    assert orc.vq_near_tie_ok(z, cb, tok.reshape(-1), tok_ref.reshape(-1)).all()
    assert np.array_equal(out["hist"].cpu().numpy(), np.bincount(tok.reshape(-1), minlength=512))
    if codebook == "shipped":   # the reference's own golden tokens (training/model.py:206-214 on the shipped checkpoint)
        from lbm_amd.quantizer import encoder_forward, fold_projection, vq_argmin
        enc = {k: torch.from_numpy(v).to(DEV) for k, v in shipped.items() if k.startswith("encoder.")}
        xs = torch.from_numpy(vq_golden["x_S"]).to(DEV)
        h3 = xs
        for i, (s_, pd) in zip((0, 2, 4), ((2, 1), (2, 1), (1, 1))):
            h3 = lbm_amd.conv1d(h3, enc[f"encoder.{i}.weight"], enc[f"encoder.{i}.bias"], stride=s_, padding=pd, relu=True)
        cbf, eef = fold_projection(torch.from_numpy(cb).to(DEV), enc["encoder.6.weight"], enc["encoder.6.bias"])
        tk = vq_argmin(h3, cbf, eef, channel_major=True, no_xnorm=True).view(xs.shape[0], -1).cpu().numpy()
        zg = np.transpose(vq_golden["z_S"], (0, 2, 1)).reshape(-1, 256)
        assert orc.vq_near_tie_ok(zg, cb, tk.reshape(-1), vq_golden["tokens_S"].reshape(-1)).all()


def test_pipeline_k4s2_wino_tokens_are_the_direct_layers_tokens(shipped):
    """PipelineConfig.conv_k4s2: 'wino' (default; the first two encoder layers through lbm_conv1d_k4s2_wino_f32) against 'direct'
    (lbm_conv1d_pitched_f32): z within 2e-5 of its scale, tokens equal except documented near-ties of the direct path's own z, PSD /
    band power / keep mask untouched (bit for bit: the DSP front does not change)."""
    cb = shipped["vq_layer.embedding.weight"]
    taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
    A, b = lbm_amd.ica_projection(**synth_ica(30))
    mk = lambda how: lbm_amd.EpochPipeline(lbm_amd.PipelineConfig(conv_k4s2=how), taps, A, b, shipped, cb)  # noqa: E731
    x = (synth_epochs(96, 30, 1500, seed=22) * 2.0).to(DEV)
    ref = {k: v.clone() for k, v in mk("direct").run(x).items()}
    out = mk("wino").run(x)
    for k in ("psd", "bandpower", "keep", "standardized"):
        assert torch.equal(out[k], ref[k]), k
    zr = ref["z"].permute(0, 2, 1).reshape(-1, 256).cpu().numpy()
    assert float((out["z"] - ref["z"]).abs().max()) <= 2e-5 * max(1.0, float(ref["z"].abs().max()))
    tok, tok_ref = out["tokens"].cpu().numpy(), ref["tokens"].cpu().numpy()
    assert (tok == tok_ref).mean() >= 0.999
    assert orc.vq_near_tie_ok(zr, cb, tok.reshape(-1), tok_ref.reshape(-1)).all()
    with pytest.raises(ValueError):
        mk("fast")


@pytest.mark.parametrize("vq_method,fold", [("screen", True), ("fp32", False), ("bf16x6", True)])
def test_pipeline_dedup_codes_changes_nothing(shipped, vq_method, fold):
    """PipelineConfig.dedup_codes (default): the search runs on the codebook without the rows that repeat an earlier row bit for bit --
    100 all-zero rows of the shipped checkpoint's 512 -- and lbm_vq_remap_codes_i64 translates tokens and histogram back.  A
    first-minimum argmin (training/model.py:152-158) can never return a repeated row, so EVERYTHING must equal the full search: tokens
    and histogram bit for bit, on the shipped codebook, on one whose repeated rows are scattered (a late first occurrence, repeats in
    front of and behind other codes), with inputs that sit ON the repeated rows, and with no repeats at all (nothing to do)."""
    from lbm_amd.quantizer import dedup_codebook
    taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
    A, b = lbm_amd.ica_projection(**synth_ica(30))
    rs = np.random.RandomState(5)
    scattered = (np.asarray(synth_codebook(512, 256)) * 0.3).astype(np.float32)
    scattered[[3, 77, 400, 511]] = scattered[200]          # first occurrence of that row: 3
    scattered[[10, 11, 12]] = 0.0                          # a run of zero rows
    scattered[300] = scattered[299]
    plain = (np.asarray(synth_codebook(512, 256)) * 0.3).astype(np.float32)
    x = (synth_epochs(40, 30, 1500, seed=23) * 2.0).to(DEV)
    for name, cb in (("shipped", shipped["vq_layer.embedding.weight"]), ("scattered", scattered), ("plain", plain)):
        mk = lambda d: lbm_amd.EpochPipeline(lbm_amd.PipelineConfig(fold_projection=fold, vq_method=vq_method, dedup_codes=d),  # noqa: E731
                                             taps, A, b, shipped, cb)
        full, ded = mk(False), mk(True)
        assert full.code_map is None
        if name == "plain":
            assert ded.code_map is None and ded.K_used == 512
        else:
            assert ded.code_map is not None and ded.K_used % 64 == 0 and ded.K_real < 512
            assert ded.K_real == {"shipped": 413, "scattered": 512 - 4 - 2 - 1}[name]
        for rep in range(2):      # twice: the searched-row histogram is cleared by the remap, the full one keeps accumulating
            ref = {k: v.clone() for k, v in full.run(x).items()}
            out = ded.run(x)
            assert torch.equal(out["tokens"], ref["tokens"]), name
            assert torch.equal(out["hist"], ref["hist"]), name
            assert int(out["hist"].sum()) == (rep + 1) * x.shape[0] * ded.S
    # rows that ARE repeated codes: the lowest original index wins, with and without the shortcut
    cbd = torch.from_numpy(scattered).to(DEV)
    ee = (cbd.double() ** 2).sum(1).float()
    cbc, eec, cmap, kc = dedup_codebook(cbd, ee)
    xs = cbd[[200, 3, 511, 11, 300, 299, 5]] + 1e-4 * torch.from_numpy(rs.randn(7, 256).astype(np.float32)).to(DEV)
    tok = lbm_amd.vq_argmin(xs, cbc, eec)
    assert _lib.load().lbm_vq_remap_codes_i64(_lib.ptr(tok), tok.numel(), _lib.ptr(cmap), kc, None, None, _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert tok.tolist() == lbm_amd.vq_argmin(xs, cbd, ee).tolist() == [3, 3, 3, 10, 299, 299, 5]
    # a padding row (index >= kc) maps to code 0 when the remap runs over all searched rows, and the counting form agrees
    tok = torch.tensor([kc, 0, kc + 1, 2, 1], dtype=torch.int64, device=DEV)[1:]            # 8-byte aligned only, odd count
    h = torch.zeros(512, dtype=torch.int64, device=DEV)
    lbm_amd.remap_codes(tok, cmap, cbc.shape[0], None, h)
    assert tok.tolist() == [0, 0, int(cmap[2]), int(cmap[1])] and int(h.sum()) == 4 and int(h[0]) == 2


def test_pipeline_psd_branch_on_a_second_stream_changes_nothing(shipped):
    """PipelineConfig.overlap_welch (an option): Welch + band power run on a second HIP stream beside the encoder layers and are joined
    before run() returns.  Same outputs bit for bit as in line, over back-to-back runs on different inputs (the buffers are reused while
    the other stream may still be working), also when the caller's own stream is not the default one and when a consumer on that stream
    reads the outputs right away."""
    taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
    A, b = lbm_amd.ica_projection(**synth_ica(30))
    mk = lambda ov: lbm_amd.EpochPipeline(lbm_amd.PipelineConfig(fold_projection=True, vq_method="screen", overlap_welch=ov), taps, A, b,  # noqa: E731
                                          shipped, shipped["vq_layer.embedding.weight"])
    inline, over = mk(False), mk(True)
    xs = [(synth_epochs(48, 30, 1500, seed=41 + i) * 2.0).to(DEV) for i in range(3)]
    want = []
    for x in xs:
        o = inline.run(x)
        want.append({k: o[k].clone() for k in ("tokens", "psd", "bandpower", "keep")})
    side = torch.cuda.Stream()
    for stream in (torch.cuda.current_stream(), side):
        with torch.cuda.stream(stream):
            got = []
            for x in xs:                      # no synchronisation between the runs: the copies below are ordered by the stream alone
                o = over.run(x)
                got.append({k: o[k].clone() for k in ("tokens", "psd", "bandpower", "keep")})
        torch.cuda.synchronize()
        for g, w in zip(got, want):
            for k in w:
                assert torch.equal(g[k], w[k]), k
    assert getattr(over, "_side", None) is not None and getattr(inline, "_side", None) is None


def test_pipeline_writes_into_caller_owned_outputs(shipped):
    """EpochPipeline.run(x, out=...): tokens / PSD / band features / keep mask land in the caller's tensors (here: odd-offset slices of
    larger arrays, so the token slice is only 8-byte aligned), bit-identical to the pipeline's own buffers; wrong shapes, dtypes, devices
    and names are refused before anything is launched."""
    taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
    A, b = lbm_amd.ica_projection(**synth_ica(30))
    pipe = lbm_amd.EpochPipeline(lbm_amd.PipelineConfig(fold_projection=True, vq_method="screen"), taps, A, b, shipped,
                                 shipped["vq_layer.embedding.weight"])
    assert pipe.code_map is not None          # the remap pass runs on the caller's token slice
    E, lo = 37, 3
    x = (synth_epochs(E, 30, 1500, seed=31) * 2.0).to(DEV)
    x[5, 3] += 400e-6 * torch.sin(2 * math.pi * 10.0 * torch.arange(1500, device=DEV) / 500.0)     # an in-band swing: epoch 5 is rejected
    ref = {k: v.clone() for k, v in pipe.run(x).items()}
    big = {"tokens": torch.full((E + 8, pipe.S), -7, dtype=torch.int64, device=DEV),
           "psd": torch.full((E + 8, 30, pipe.F), -7.0, device=DEV), "bandpower": torch.full((E + 8, 6), -7.0, device=DEV),
           "keep": torch.full((E + 8,), 9, dtype=torch.uint8, device=DEV)}
    dest = {k: v[lo:lo + E] for k, v in big.items()}
    assert dest["tokens"].data_ptr() % 16 == 8
    out = pipe.run(x, out=dest)
    torch.cuda.synchronize()
    for k in dest:
        assert out[k].data_ptr() == dest[k].data_ptr()
        assert torch.equal(dest[k], ref[k]), k
        assert bool((big[k][:lo] == (9 if k == "keep" else -7)).all()) and bool((big[k][lo + E:] == (9 if k == "keep" else -7)).all()), k
    assert int(ref["keep"][5]) == 0 and 0 < int(ref["keep"].sum()) < E
    only = pipe.run(x, out={"psd": dest["psd"]})              # a subset: the rest stays in the pipeline's buffers
    assert only["psd"].data_ptr() == dest["psd"].data_ptr() and torch.equal(only["tokens"], ref["tokens"])
    for bad in ({"tokens": dest["tokens"].to(torch.int32)}, {"psd": big["psd"][lo:lo + E + 1]}, {"psd": dest["psd"].cpu()},
                {"bandpower": big["bandpower"][lo:lo + E, ::2]}, {"standardized": x}):
        with pytest.raises(ValueError):
            pipe.run(x, out=bad)


def test_pipeline_without_cleaned_output(shipped):
    """emit_cleaned=False: no "cleaned" tensor; tokens / keep / standardized identical, PSD equal to fp32 rounding
    (Welch removes each segment's mean, so the baseline offset never reaches it)."""
    pipe, taps, A, b = _make_pipeline(shipped, T=1500)
    pipe2 = lbm_amd.EpochPipeline(lbm_amd.PipelineConfig(emit_cleaned=False), taps, A, b, shipped, shipped["vq_layer.embedding.weight"])
    x = (synth_epochs(40, 30, 1500, seed=12) * 2.0)
    x[3, 5, 100] += 1e-2
    x = x.to(DEV)
    ref = {k: v.clone() for k, v in pipe.run(x).items()}
    out = pipe2.run(x)
    assert "cleaned" not in out
    for k in ("tokens", "keep", "standardized", "hist"):
        assert torch.equal(out[k], ref[k]), k
    assert _psd_err(out["psd"].cpu().numpy(), ref["psd"].cpu().numpy().astype(np.float64)) <= 4.0
    np.testing.assert_allclose(out["bandpower"].cpu().numpy(), ref["bandpower"].cpu().numpy(), rtol=2e-6)


def test_pipeline_features_only_no_tokenizer(shipped):
    """BASELINE config 2's branch (PSD + band-power features only): do_tokenize=False builds no encoder / codebook state,
    returns no tokens, and its PSD / band powers / keep mask equal the full pipeline's bit for bit; with do_filter=False and
    do_ica=False on top it is Welch + band power straight on the input, checked against the oracle."""
    pipe, taps, A, b = _make_pipeline(shipped, T=1500)
    cfg = lbm_amd.PipelineConfig(do_tokenize=False)
    pipe2 = lbm_amd.EpochPipeline(cfg, taps, A, b, None, None)
    x = (synth_epochs(24, 30, 1500, seed=14) * 2.0)
    x[2, 9, 40] -= 1e-2
    xd = x.to(DEV)
    ref = {k: v.clone() for k, v in pipe.run(xd).items()}
    out = pipe2.run(xd)
    assert "tokens" not in out and "hist" not in out and "standardized" not in out
    for k in ("keep", "cleaned"):
        assert torch.equal(out[k], ref[k]), k
    # the full pipeline takes the PSD from the z-scored epochs and the per-row scale, this branch from the cleaned epochs
    assert _psd_err(out["psd"].cpu().numpy(), ref["psd"].cpu().numpy().astype(np.float64)) <= 0.2
    np.testing.assert_allclose(out["bandpower"].cpu().numpy(), ref["bandpower"].cpu().numpy(), rtol=2e-6)
    cfg3 = lbm_amd.PipelineConfig(do_tokenize=False, do_filter=False, do_ica=False, emit_cleaned=False)
    out3 = lbm_amd.EpochPipeline(cfg3, None, None, None, None, None).run(xd)
    assert set(out3) == {"psd", "bandpower", "keep"}
    x64 = x.numpy().astype(np.float64)
    want_psd, freqs = orc.welch_psd(x64, 500.0, 4.0, 30.0, n_fft=500, n_overlap=250, window="hann")
    assert _psd_err(out3["psd"].cpu().numpy(), want_psd) <= 1.0
    np.testing.assert_allclose(out3["bandpower"].cpu().numpy(), np.stack(orc.band_powers(want_psd, freqs), axis=-1), rtol=2e-6)
    assert np.array_equal(out3["keep"].cpu().numpy().astype(bool), orc.ptp_keep_mask(x64, 150e-6))
    assert out3["keep"].cpu().numpy().tolist().count(0) == 1


def test_pipeline_bf16x6_quantizer_tokens(shipped):
    """The pipeline with the bf16-piece argmin kernel: tokens of the fp32-MFMA path except documented near-ties."""
    pipe, taps, A, b = _make_pipeline(shipped, T=1500)
    cfg = lbm_amd.PipelineConfig(vq_method="bf16x6")
    pipe2 = lbm_amd.EpochPipeline(cfg, taps, A, b, shipped, shipped["vq_layer.embedding.weight"])
    x = (synth_epochs(64, 30, 1500, seed=33) * 2.0).to(DEV)
    ref = pipe.run(x)
    tok_ref, z = ref["tokens"].cpu().numpy(), ref["z"].permute(0, 2, 1).reshape(-1, 256).cpu().numpy()
    out = pipe2.run(x)
    tok = out["tokens"].cpu().numpy()
    cb = shipped["vq_layer.embedding.weight"]
    assert (tok == tok_ref).mean() >= 0.999
    assert orc.vq_near_tie_ok(z, cb, tok.reshape(-1), orc.vq_argmin(z, cb)).all()
    assert np.array_equal(out["hist"].cpu().numpy(), np.bincount(tok.reshape(-1), minlength=512))


@pytest.mark.parametrize("fold", [False, True])
def test_pipeline_bf16x6_convs(shipped, fold):
    """conv_method='bf16x6': the 256-channel encoder layers on the bf16 MFMA.  z within fp32 rounding of the fp32-MFMA
    path's z (both are ~1 ulp chains of the same products), tokens equal except documented near-ties."""
    pipe, taps, A, b = _make_pipeline(shipped, T=1500)
    cfg = lbm_amd.PipelineConfig(vq_method="bf16x6", conv_method="bf16x6", fold_projection=fold)
    pipe2 = lbm_amd.EpochPipeline(cfg, taps, A, b, shipped, shipped["vq_layer.embedding.weight"])
    x = (synth_epochs(48, 30, 1500, seed=35) * 2.0).to(DEV)
    ref = pipe.run(x)
    tok_ref, zr = ref["tokens"].cpu().numpy(), ref["z"].permute(0, 2, 1).reshape(-1, 256).cpu().numpy().copy()
    out = pipe2.run(x)
    tok = out["tokens"].cpu().numpy()
    cb = shipped["vq_layer.embedding.weight"]
    assert (tok == tok_ref).mean() >= 0.998
    if not fold:
        z = out["z"].permute(0, 2, 1).reshape(-1, 256).cpu().numpy()
        assert np.abs(z - zr).max() <= 2e-5 * np.abs(zr).max()
        assert orc.vq_near_tie_ok(z, cb, tok.reshape(-1), orc.vq_argmin(z, cb)).all()
    else:
        assert "z" not in out
        assert orc.vq_near_tie_ok(zr, cb, tok.reshape(-1), tok_ref.reshape(-1)).all()
    assert np.array_equal(out["hist"].cpu().numpy(), np.bincount(tok.reshape(-1), minlength=512))


def test_pipeline_config_L_matches_oracle():
    """BASELINE config 5 shape: 64 channels x 1 kHz x 10 s epochs, 1651-tap FIR, n_fft = 1000, 8192-entry codebook."""
    from lbm_amd.synth import synth_encoder
    C, T, fs, K = 64, 10000, 1000.0, 8192
    taps = lbm_amd.create_filter(fs, 4.0, 30.0)
    assert len(taps) == 1651
    A, b = lbm_amd.ica_projection(**synth_ica(C))
    weights = synth_encoder(C)
    weights["vq_layer.embedding.weight"] = synth_codebook(K, 256) * 0.05
    cfg = lbm_amd.PipelineConfig(sfreq=fs, n_channels=C, n_times=T, n_baseline=501)
    pipe = lbm_amd.EpochPipeline(cfg, taps, A, b, weights, weights["vq_layer.embedding.weight"])
    x = synth_epochs(2, C, T, sfreq=fs, seed=12)
    out = pipe.run(x.to(DEV))
    torch.cuda.synchronize()
    want = orc.pipeline(x.numpy(), fs, taps, A, b, 501, 150e-6, weights)
    assert _rms_ok(out["cleaned"].cpu().numpy(), want["filtered"]) <= 2e-5
    assert np.array_equal(out["keep"].cpu().numpy().astype(bool), want["keep"])
    assert out["psd"].shape == (2, C, 27)
    assert _psd_err(out["psd"].cpu().numpy(), want["psd"]) <= PIPELINE_PSD_TOL
    np.testing.assert_allclose(out["bandpower"].cpu().numpy(), want["bandpower"], rtol=2e-6)
    tok = out["tokens"].cpu().numpy()
    assert tok.shape == want["tokens"].shape == (2, 2500)
    z = out["z"].permute(0, 2, 1).reshape(-1, 256).cpu().numpy()
    cb = weights["vq_layer.embedding.weight"]
    assert orc.vq_near_tie_ok(z, cb, tok.reshape(-1), orc.vq_argmin(z, cb)).all()
    _assert_tokens(tok, want["tokens"], want["z"], cb, z_build=out["z"].cpu().numpy())
    assert np.array_equal(out["hist"].cpu().numpy(), np.bincount(tok.reshape(-1), minlength=K))


def test_pipeline_full_size_properties(shipped):
    """BASELINE-size chunk (4096 epochs of 30x1500): determinism, batch-split invariance, histogram checksum."""
    pipe, *_ = _make_pipeline(shipped)
    E = 4096
    x = synth_epochs(E, 30, 1500, seed=11, device=DEV)
    o1 = {k: v.clone() for k, v in pipe.run(x).items()}
    pipe.reset_hist()
    o2 = {k: v.clone() for k, v in pipe.run(x).items()}
    for k in ("psd", "bandpower", "tokens", "keep", "hist"):
        assert torch.equal(o1[k], o2[k]), k                      # bitwise deterministic
    assert int(o1["hist"].sum()) == E * 375
    assert torch.equal(o1["hist"], torch.bincount(o1["tokens"].reshape(-1), minlength=512))
    pipe.reset_hist()
    oa = {k: v.clone() for k, v in pipe.run(x[:1000]).items()}
    ob = pipe.run(x[1000:1777])
    assert torch.equal(torch.cat([oa["tokens"], ob["tokens"]]), o1["tokens"][:1777])   # epochs are independent
    assert torch.equal(torch.cat([oa["psd"], ob["psd"]]), o1["psd"][:1777])
    assert bool(torch.isfinite(o1["psd"]).all()) and bool((o1["psd"] > 0).all())


def test_encoder_and_affine_large_batch_against_torch_float64(shipped):
    """The conv stack and the channel affine map on a 4096-epoch batch (every position tile shape and tile edge many
    times over) against float64 torch on the GPU."""
    from lbm_amd.quantizer import encoder_forward
    E = 4096
    x = synth_epochs(E, 30, 1500, seed=19, device=DEV, chunk=2048) * 1e5      # ~unit scale, as after the z-score
    w = {k: torch.from_numpy(v).to(DEV) for k, v in shipped.items() if k.startswith("encoder.")}
    z = encoder_forward(x, w)
    assert z.shape == (E, 256, 375)
    F_ = torch.nn.functional
    worst = 0.0
    for lo in range(0, E, 512):
        h = x[lo:lo + 512].double()
        h = F_.relu(F_.conv1d(h, w["encoder.0.weight"].double(), w["encoder.0.bias"].double(), stride=2, padding=1))
        h = F_.relu(F_.conv1d(h, w["encoder.2.weight"].double(), w["encoder.2.bias"].double(), stride=2, padding=1))
        h = F_.relu(F_.conv1d(h, w["encoder.4.weight"].double(), w["encoder.4.bias"].double(), stride=1, padding=1))
        h = F_.conv1d(h, w["encoder.6.weight"].double(), w["encoder.6.bias"].double())
        worst = max(worst, float((z[lo:lo + 512].double() - h).abs().max() / h.abs().max()))
    assert worst <= 2e-5
    A, b = lbm_amd.ica_projection(**synth_ica(30))
    At, bt = torch.as_tensor(A).to(DEV), torch.as_tensor(b).to(DEV)
    y = lbm_amd.affine_channels(x, At.float(), bt.float())
    werr = 0.0
    for lo in range(0, E, 1024):
        ref = torch.einsum("oc,ect->eot", At.double(), x[lo:lo + 1024].double()) + bt.double()[None, :, None]
        werr = max(werr, float((y[lo:lo + 1024].double() - ref).abs().max() / ref.abs().max()))
    assert werr <= 1e-6


def test_fir_and_post_large_batch_against_torch_float64():
    """FIR fast convolution and the epoch post-processing on a 4096-epoch batch against float64 torch on the GPU
    (odd-reflection padding + FFT convolution; baseline, peak-to-peak mask, z-score)."""
    E, C, T = 4096, 30, 1500
    x = synth_epochs(E, C, T, seed=23, device=DEV, chunk=2048)
    burst = 2e-4 * torch.sin(2.0 * math.pi * 10.0 * torch.arange(T, device=DEV) / 500.0)   # in-band artefacts -> rejected epochs
    x[17, 3] += burst
    x[4000, 29] -= burst
    taps = lbm_amd.create_filter(500.0, 4.0, 30.0)
    y = lbm_amd.filter_data(x, 500.0, 4.0, 30.0)
    h = torch.as_tensor(np.asarray(taps, dtype=np.float64)).to(DEV)
    half = (h.numel() - 1) // 2
    N = 4096
    H = torch.fft.rfft(h, n=N)
    worst = 0.0
    for lo in range(0, E, 512):
        xr = x[lo:lo + 512].double().reshape(-1, T)
        left = 2.0 * xr[:, :1] - xr[:, 1:half + 1].flip(-1)
        right = 2.0 * xr[:, -1:] - xr[:, -half - 1:-1].flip(-1)
        xp = torch.cat([left, xr, right], dim=-1)                                  # T + 2 half
        full = torch.fft.irfft(torch.fft.rfft(xp, n=N) * H, n=N)                   # linear convolution (T + 2 half + 2 half < N)
        ref = full[:, 2 * half:2 * half + T]
        worst = max(worst, float((y[lo:lo + 512].double().reshape(-1, T) - ref).abs().max() / ref.abs().max()))
    assert worst <= 2e-5
    base, std, keep = lbm_amd.epoch_post(y, n_baseline=251, reject=150e-6, standardize=True)
    yd = y.double()
    bref = yd - yd[..., :251].mean(-1, keepdim=True)
    kref = ((yd.max(-1).values - yd.min(-1).values) <= 150e-6).all(-1)
    sref = (bref - bref.mean(-1, keepdim=True)) / (bref.std(-1, unbiased=False, keepdim=True) + 1e-6)
    assert torch.equal(keep, kref) and int((~kref).sum()) >= 2
    assert float((base.double() - bref).abs().max()) <= 1e-5 * float(bref.abs().max())
    assert float((std.double() - sref).abs().max()) <= 2e-5 * float(sref.abs().max())


def test_vq_full_size_every_row_is_a_minimum(shipped):
    """The quantizer at the bench's launch size (16384 x 375 rows, channel-major, K = 512: the LDS-DMA kernel): every
    returned code attains the row's minimum distance within the near-tie band, recomputed with torch on the GPU;
    rows in ragged tail blocks (N not a multiple of 256) included; the ROWS layout gives the same tokens."""
    E, S, D = 16384, 375, 256
    cb = torch.from_numpy(shipped["vq_layer.embedding.weight"]).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(21)
    z = torch.randn(E, D, 376, device=DEV, generator=g) * 0.05
    tok = lbm_amd.vq_argmin(z[:, :, :S], cb, channel_major=True)
    assert tok.shape == (E * S,) or tok.shape == (E, S)
    tok = tok.reshape(E, S)
    ee = (cb.double() ** 2).sum(1)
    eps = 2.0 ** -23
    bad = 0
    for lo in range(0, E, 1024):
        zc = z[lo:lo + 1024, :, :S].permute(0, 2, 1).reshape(-1, D).double()
        d = (zc * zc).sum(1, keepdim=True) + ee[None, :] - 2.0 * (zc @ cb.double().t())
        dmin = d.min(dim=1).values
        dgot = d.gather(1, tok[lo:lo + 1024].reshape(-1, 1)).squeeze(1)
        tol = 8 * eps * ((zc * zc).sum(1) + ee.max() + 2.0 * (zc.abs() @ cb.double().abs().t()).max(dim=1).values)
        bad += int((dgot - dmin > tol).sum())
    assert bad == 0
    # ragged N and the row-major layout: same codes as the channel-major launch
    flat = z[:3, :, :S].permute(0, 2, 1).reshape(-1, D).contiguous()[:1001]
    t2 = lbm_amd.vq_argmin(flat, cb)
    assert torch.equal(t2.reshape(-1), tok[:3].reshape(-1)[:1001])


# ------------------------------------------------------------------------------------------------ 8(f)-4 resample
@pytest.mark.parametrize("T,up,down,rows", [(5000, 1, 2, 3), (1001, 1, 2, 2), (30000, 250, 1000, 2), (65539, 250, 500, 1), (3000, 2, 1, 2),
                                            (70, 1, 2, 1), (300000, 1, 2, 70),
                                            (2 ** 17 - 300, 1, 2, 2),     # T + min_add <= 2^17 < T + 2 min_add: MNE pads to 2^17
                                            (2 ** 12 - 150, 1, 2, 1),
                                            # resampled padded length not a power of two: chirp-z inverse transform
                                            (30000, 250, 600, 2), (4000, 250, 512, 3), (5000, 256, 500, 1), (3000, 3, 2, 2),
                                            (100000, 250, 1024, 2), (333, 7, 5, 1), (20, 1, 3, 1),
                                            # an hour of 500 Hz EEG through the chirp-z path (2^21-point padded row, 2^22-point convolution)
                                            (1_800_000, 250, 600, 1)])
def test_resample_matches_oracle(T, up, down, rows):
    """raw.resample (preprocessing_unsupervised.py:37): FFT resampling with MNE's auto padding, vs the float64 oracle."""
    rs = np.random.RandomState(T)
    t = np.arange(T) / 500.0
    x = (np.sin(2 * np.pi * 7 * t)[None] * rs.uniform(0.5, 2, (rows, 1)) + 0.2 * rs.randn(rows, T).cumsum(axis=1) / 20 + 0.1 * rs.randn(rows, T))
    x = (x * 1e-5).astype(np.float32)
    got = lbm_amd.resample(torch.from_numpy(x).to(DEV), up, down).cpu().numpy()
    sub = slice(0, min(rows, 4))                                    # the oracle loops over rows in Python
    want = orc.resample(x[sub].astype(np.float64), up, down)
    assert got.shape == (rows, want.shape[1])
    assert np.abs(got[sub] - want).max() <= 2e-5 * np.abs(x).max()
    if rows > 4:                                                    # every launch group computes the same function
        want_last = orc.resample(x[-1:].astype(np.float64), up, down)
        assert np.abs(got[-1:] - want_last).max() <= 2e-5 * np.abs(x).max()


def test_resample_unsupported_ratio_and_empty():
    with pytest.raises(NotImplementedError):
        lbm_amd.resample(torch.zeros(2, 3, device=DEV), 1, 2)            # fewer than 16 padded samples
    assert lbm_amd.resample(torch.zeros(2, 1000, device=DEV), 250, 512).shape == (2, 488)   # round(1000 * 250 / 512)
    lib = _lib.load()
    assert lib.lbm_resample_length(1001, 1, 2) == 500 and lib.lbm_resample_length(1003, 1, 2) == 502      # round-half-even
    assert lib.lbm_resample_fft_f32(None, 0, None, 0, 0, 100, 1, 2, None, None) == 0
