"""CPU tests: the oracle against the reference's golden vectors / known answers (no GPU)."""
import os

import numpy as np
import pytest
from scipy import signal

from oracle import eeg_oracle as orc


# ---------------------------------------------------------------- quantizer / encoder (pinned)
def test_oracle_encode_matches_reference_tokens(shipped, vq_golden):
    for name in ("S", "R", "P"):
        tok, z = orc.vqvae_encode(vq_golden[f"x_{name}"], shipped)
        np.testing.assert_allclose(z, vq_golden[f"z_{name}"], rtol=0, atol=2e-5)
        ref = vq_golden[f"tokens_{name}"]
        assert tok.shape == ref.shape and tok.dtype == np.int64
        flat = np.ascontiguousarray(vq_golden[f"z_{name}"].transpose(0, 2, 1)).reshape(-1, 256)
        ok = orc.vq_near_tie_ok(flat, shipped["vq_layer.embedding.weight"], tok, ref)
        assert ok.all()
        assert (tok == ref).mean() > 0.99


def test_oracle_get_code_indices_golden(shipped, vq_golden):
    cb = shipped["vq_layer.embedding.weight"]
    idx = orc.vq_argmin(vq_golden["gci_in"], cb)
    ok = orc.vq_near_tie_ok(vq_golden["gci_in"], cb, idx, vq_golden["gci_idx"])
    assert ok.all()
    # all-zero rows tie among the 100 all-zero codes: lowest index (406) must win
    zero_rows = np.nonzero((cb == 0).all(1))[0]
    assert len(zero_rows) == 100 and zero_rows[0] == 406
    assert (vq_golden["gci_idx"][:8] == 406).all() and (idx[:8] == 406).all()
    # rows that ARE a codebook row map to that row (or an identical duplicate with a lower index)
    assert (idx[8:16] == vq_golden["gci_pick"][8:16]).all()


def test_oracle_vq_forward_golden(shipped, vq_golden):
    cb = shipped["vq_layer.embedding.weight"]
    loss, quant, perp, idx, counts = orc.vq_forward(vq_golden["vqfwd_in"], cb, 0.25)
    assert np.array_equal(idx, vq_golden["vqfwd_indices"])
    np.testing.assert_allclose(quant, vq_golden["vqfwd_quantized"], rtol=0, atol=0)
    np.testing.assert_allclose(loss, vq_golden["vqfwd_loss"], rtol=1e-6)
    np.testing.assert_allclose(perp, vq_golden["vqfwd_perplexity"], rtol=1e-6)
    np.testing.assert_allclose(orc.perplexity_from_counts(counts), vq_golden["vqfwd_perplexity"], rtol=1e-5)
    assert np.array_equal(vq_golden["fwd_indices"], vq_golden["vqfwd_indices"])


def test_oracle_vqvae_forward_golden(shipped, vq_golden):
    """VQVAE.forward (training/model.py:195-203) restated in the oracle against the imported reference's outputs."""
    loss, x_recon, perp, idx = orc.vqvae_forward(vq_golden["x_P"], shipped, 0.25)
    assert np.array_equal(idx, vq_golden["fwd_indices"])
    np.testing.assert_allclose(loss, vq_golden["fwd_vq_loss"], rtol=1e-6)
    np.testing.assert_allclose(perp, vq_golden["fwd_perplexity"], rtol=1e-6)
    assert x_recon.shape == vq_golden["fwd_x_recon"].shape == (4, 30, 39)
    np.testing.assert_allclose(x_recon, vq_golden["fwd_x_recon"], rtol=0, atol=1e-6 * float(np.abs(vq_golden["fwd_x_recon"]).max()))


def test_oracle_vq_tokens_explained_rule():
    """The reference-side near-tie rule: equal tokens pass; a swap between two far codes fails; a swap that the build's own
    z perturbation explains passes; duplicate code rows never do."""
    rs = np.random.RandomState(0)
    cb = rs.randn(16, 8)
    cb[9] = cb[3]
    z = cb[[1, 2, 3, 4]] + 0.01 * rs.randn(4, 8)
    ref = np.array([1, 2, 3, 4])
    assert orc.vq_tokens_explained(z, None, cb, ref, ref).all()
    assert orc.vq_tokens_explained(z, None, cb, np.array([1, 2, 3, 5]), ref).tolist() == [True, True, True, False]
    assert orc.vq_tokens_explained(z, None, cb, np.array([1, 2, 9, 4]), ref).tolist() == [True, True, False, True]   # duplicate row
    mid = 0.5 * (cb[5] + cb[6])                      # equidistant from codes 5 and 6
    zr = np.stack([mid + 1e-9 * (cb[5] - cb[6])])    # the reference side leans to 5
    zb = np.stack([mid - 1e-6 * (cb[5] - cb[6])])    # the build side to 6, by more than fp32 rounding
    assert orc.vq_tokens_explained(zr, zb, cb, np.array([6]), np.array([5])).all()
    far = np.stack([mid + 1e-3 * (cb[5] - cb[6])])
    assert not orc.vq_tokens_explained(far, far, cb, np.array([6]), np.array([5])).any()


def test_oracle_ema_update_golden(shipped, vq_golden):
    state = {"ema_cluster_size": shipped["vq_layer.ema_cluster_size"], "ema_w": shipped["vq_layer.ema_w"]}
    flat = vq_golden["ema_in"].reshape(-1, 256)
    cs, ew, w = orc.ema_update(state, flat, vq_golden["ema_idx"].reshape(-1))
    np.testing.assert_allclose(cs, vq_golden["ema_cluster_size"], rtol=1e-6)
    np.testing.assert_allclose(ew, vq_golden["ema_w"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(w, vq_golden["ema_weight"], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------- band power KAT (pinned 62/62)
def test_bandpower_rule_reproduces_shipped_labels(golden_dir):
    with np.load(os.path.join(golden_dir, "bandpower_kat.npz")) as f:
        X, y, yv, freqs = f["log_psd"], f["labels"], f["labels_variant"], f["freqs"]
    assert X.shape == (62, 33, 53)
    assert ((freqs >= 4) & (freqs < 8)).sum() == 8
    assert ((freqs >= 8) & (freqs < 13)).sum() == 10
    assert ((freqs >= 13) & (freqs < 30)).sum() == 35
    th, al, be, bt, ab, tb = orc.band_powers(X, freqs, already_db=True)
    thr = float(np.median(tb))
    assert abs(thr - 0.95106419) < 1e-7
    lab = np.array([orc.classify_eeg_state(t, a, b, thr) for t, a, b in zip(th, al, be)])
    assert np.array_equal(lab, y)
    labv = np.array([orc.classify_eeg_state_variant(t, a, b) for t, a, b in zip(th, al, be)])
    assert np.array_equal(labv, yv)


# ---------------------------------------------------------------- FIR (MNE-boundary: unpinned, SciPy identities)
def _resp(h, f, fs):
    n = np.arange(len(h)) - (len(h) - 1) / 2
    return abs(np.sum(h * np.exp(-2j * np.pi * f / fs * n)))


def test_fir_design_lengths_and_response():
    h500 = orc.design_fir(500.0, 4.0, 30.0)
    assert len(h500) == 825 and np.allclose(h500, h500[::-1], atol=0)
    assert len(orc.design_fir(1000.0, 4.0, 30.0)) == 1651
    assert len(orc.design_fir(250.0, 4.0, 30.0)) == 413
    for f, g in ((1, 0.0008), (4, 0.9974), (10, 0.999), (20, 0.9976), (30, 0.9963), (37.5, 0.0026), (45, 0.0011)):
        assert abs(_resp(h500, f, 500.0) - g) < 6e-4, (f, _resp(h500, f, 500.0))
    (hn,) = orc.design_notch(500.0, 60.0)
    assert len(hn) == 3301
    assert _resp(hn, 60.0, 500.0) < 0.01 and abs(_resp(hn, 20.0, 500.0) - 1) < 1e-3
    # the single notch is the plain band-stop design with the same edges
    np.testing.assert_allclose(hn, orc.design_fir(500.0, 60.65, 59.35, l_trans_bandwidth=0.5, h_trans_bandwidth=0.5), atol=1e-15)
    # several frequencies: ONE multi-band filter (mne.filter.notch_filter hands create_filter array edges), a zero at each
    (hm,) = orc.design_notch(500.0, [60.0, 120.0, 180.0])
    assert len(hm) == 3301 and np.allclose(hm, hm[::-1], atol=0)
    for f in (60.0, 120.0, 180.0):
        assert _resp(hm, f, 500.0) < 0.01
    for f in (20.0, 90.0, 150.0, 220.0):
        assert abs(_resp(hm, f, 500.0) - 1) < 2e-3
    with pytest.raises(ValueError):
        orc.design_notch(500.0, [60.0, 60.9])          # stop bands overlap


def test_fir_zero_phase_identities():
    rs = np.random.RandomState(0)
    h = orc.design_fir(500.0, 4.0, 30.0)
    x = rs.randn(3, 1500)
    y = orc.fir_zero_phase(x, h)
    yf = orc.fir_zero_phase(x, h, fft=True)
    np.testing.assert_allclose(y, yf, rtol=0, atol=1e-12)
    # equivalent statement: odd-reflect pad by (N-1)/2 then 'valid' convolution
    p = (len(h) - 1) // 2
    for r in range(3):
        xp = np.pad(x[r], p, mode="reflect", reflect_type="odd")
        np.testing.assert_allclose(y[r], np.convolve(xp, h, mode="valid"), rtol=0, atol=1e-12)
    # zero phase: a 10 Hz sine comes back with gain ~0.999 and no delay (away from the edges)
    t = np.arange(3000) / 500.0
    s = np.sin(2 * np.pi * 10 * t)
    ys = orc.fir_zero_phase(s, h)
    np.testing.assert_allclose(ys[900:2100], 0.999 * s[900:2100], atol=2e-3)
    # short rows: reflect-limited padding zero-fills what reflection cannot supply
    xs = rs.randn(2, 300)
    ysht = orc.fir_zero_phase(xs, h)
    n_edge = 299
    xe = np.concatenate([2 * xs[0, 0] - xs[0, n_edge:0:-1], xs[0], 2 * xs[0, -1] - xs[0, -2:-n_edge - 2:-1]])
    full = np.convolve(xe, h)
    np.testing.assert_allclose(ysht[0], full[412 + n_edge:412 + n_edge + 300], atol=1e-12)


# ---------------------------------------------------------------- ICA
def test_ica_affine_fold_equals_literal_apply():
    rs = np.random.RandomState(97)
    C, T = 30, 200
    q, _ = np.linalg.qr(rs.randn(C, C))
    pca = np.linalg.qr(rs.randn(C, C))[0]
    mu = rs.randn(C) * 0.1
    x = rs.randn(4, C, T) * 1e-5
    kw = dict(unmixing=q[:20, :20] + 0.1 * rs.randn(20, 20), mixing=None, pca_components=pca, pca_mean=mu,
              pre_whitener=1e-5, n_components=20, exclude=[0, 3])
    kw["mixing"] = np.linalg.pinv(kw["unmixing"])
    lit = orc.ica_apply(x, **kw)
    A, b = orc.ica_projection(**kw)
    fold = np.einsum("oc,ect->eot", A, x) + b[None, :, None]
    np.testing.assert_allclose(fold, lit, rtol=1e-10, atol=1e-18)
    # no exclusion and n_pca == C -> identity
    kw["exclude"] = []
    np.testing.assert_allclose(orc.ica_apply(x, **kw), x, rtol=1e-9, atol=1e-18)
    src = orc.ica_sources(x, kw["unmixing"], pca, mu, 1e-5, 20)
    assert src.shape == (4, 20, T)


# ---------------------------------------------------------------- epoch post-processing
def test_epoch_post():
    rs = np.random.RandomState(1)
    x = rs.randn(5, 4, 1001) * 20e-6
    x[2, 1, 500] += 400e-6
    assert orc.n_baseline_samples(-0.5, 500.0) == 251
    bc = orc.baseline_correct(x, 251)
    np.testing.assert_allclose(bc[..., :251].mean(-1), 0, atol=1e-18)
    keep = orc.ptp_keep_mask(bc, 150e-6)
    assert keep.tolist() == [True, True, False, True, True]
    assert np.array_equal(keep, orc.ptp_keep_mask(x, 150e-6))  # ptp is offset invariant
    z = orc.zscore_epoch_channel(bc)
    np.testing.assert_allclose(z.mean(-1), 0, atol=1e-12)
    sc = orc.scaler_mean(bc)
    np.testing.assert_allclose(sc.std(axis=(0, 2)), 1, rtol=1e-12)


# ---------------------------------------------------------------- Welch
def test_welch_manual_restatement():
    rs = np.random.RandomState(2)
    x = rs.randn(2, 3, 1500) + 5.0
    psd, freqs = orc.welch_psd(x, 500.0, 4.0, 30.0, n_fft=500, n_overlap=250, window="hann")
    assert psd.shape == (2, 3, 27) and np.array_equal(freqs, np.arange(4.0, 31.0))
    w = signal.get_window("hann", 500)  # periodic
    acc = np.zeros((2, 3, 251))
    nseg = (1500 - 250) // 250
    assert nseg == 5
    for s in range(nseg):
        seg = x[..., s * 250:s * 250 + 500]
        seg = seg - seg.mean(-1, keepdims=True)
        X = np.fft.rfft(seg * w, n=500)
        p = (X.real ** 2 + X.imag ** 2) / (500.0 * (w ** 2).sum())
        p[..., 1:-1] *= 2
        acc += p
    np.testing.assert_allclose(psd, acc[..., 4:31] / nseg, rtol=1e-12)
    # unsupervised-path defaults: n_fft=512, hamming, no overlap, 250 Hz -> 53 bins
    p2, f2 = orc.welch_psd(rs.randn(2, 2048), 250.0, 4.0, 30.0, n_fft=512)
    assert p2.shape == (2, 53)
    pm, _ = orc.welch_psd(x, 500.0, 4.0, 30.0, n_fft=500, n_overlap=250, window="hann", average="median")
    pn, _ = orc.welch_psd(x, 500.0, 4.0, 30.0, n_fft=500, n_overlap=250, window="hann", average=None)
    assert pn.shape == (2, 3, 27, 5)
    np.testing.assert_allclose(pm, np.median(pn, -1) / (1 + 1 / 3 - 1 / 2 + 1 / 5 - 1 / 4), rtol=1e-12)
    with pytest.raises(ValueError):
        orc.welch_psd(rs.randn(100), 500.0, n_fft=500)


def test_resample_oracle_band_limited_signal_and_lengths():
    """The restated MNE resampler: a band-limited signal decimates to its own samples away from the edges, lengths
    follow round-half-even, and the interior agrees with scipy.signal.resample on the same padded row."""
    import numpy as np
    from oracle import eeg_oracle as orc
    t = np.arange(8000) / 500.0
    x = np.sin(2 * np.pi * 7.3 * t) + 0.5 * np.cos(2 * np.pi * 31.0 * t + 0.4)
    y = orc.resample(x[None], 250, 500)[0]
    assert y.shape == (4000,)
    assert np.abs(y[200:-200] - x[::2][200:-200]).max() < 2e-3
    assert orc.resample(np.zeros((1, 1001)), 1, 2).shape == (1, 500)
    assert orc.resample(np.zeros((1, 1003)), 1, 2).shape == (1, 502)
    up = orc.resample(x[None, :1000], 2, 1)[0]
    assert up.shape == (2000,) and np.abs(up[::2][50:-50] - x[:1000][50:-50]).max() < 2e-3


def test_oracle_resample_core_equals_scipy_signal_resample():
    """The oracle's FFT resampler (MNE's mne.filter.resample restated: npad='auto' padding by odd reflection, spectrum cut with
    the Nyquist-bin rule, trim) against scipy.signal.resample applied to the same padded rows -- pins everything but the padding
    rule itself (parity unpinned at the MNE boundary) for power-of-two and other resampled lengths."""
    from scipy.signal import resample as sresample
    rs = np.random.RandomState(0)
    for T, up, down in [(3000, 250, 600), (4000, 250, 512), (1000, 1, 2), (3000, 3, 2), (2 ** 12 - 150, 1, 2)]:
        x = rs.randn(2, T).cumsum(1)
        y = orc.resample(x, up, down)
        min_add = min(T // 8, 100) * 2
        npad = 2 ** int(np.ceil(np.log2(T + min_add))) - T          # mne/filter.py resample(npad='auto')
        n0, n1 = npad // 2, npad - npad // 2
        ratio = up / down
        for r, got in zip(x, y):
            pad = np.concatenate([orc._reflect_limited_side(r, n0, True), r, orc._reflect_limited_side(r, n1, False)])
            z = sresample(pad, int(round(ratio * len(pad))))
            rem0, fl = int(round(ratio * n0)), int(round(ratio * T))
            assert got.shape == (fl,)
            assert np.abs(got - z[rem0:rem0 + fl]).max() <= 1e-12 * np.abs(z).max()
