"""CPU tests of the host-side logic and of the C-ABI library surface (no GPU, no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import lbm_amd
from lbm_amd import _lib
from lbm_amd.filter import notch_taps
from oracle import eeg_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "lbm_abi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lbm_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 16
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/lbm_abi.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes SIGNATURES and the header disagree"
    assert _lib.load().lbm_version() == 1


def test_fir_design_matches_scipy_based_oracle():
    for fs in (250.0, 500.0, 1000.0):
        np.testing.assert_allclose(lbm_amd.create_filter(fs, 4.0, 30.0), orc.design_fir(fs, 4.0, 30.0), rtol=0, atol=1e-15)
    np.testing.assert_allclose(lbm_amd.create_filter(500.0, 1.0, 40.0), orc.design_fir(500.0, 1.0, 40.0), atol=1e-15)
    np.testing.assert_allclose(lbm_amd.create_filter(500.0, None, 30.0), orc.design_fir(500.0, None, 30.0), atol=1e-15)
    np.testing.assert_allclose(lbm_amd.create_filter(500.0, 4.0, None), orc.design_fir(500.0, 4.0, None), atol=1e-15)
    for fr in ([60.0, 120.0], 60.0, [50.0, 100.0, 150.0, 200.0]):
        got, want = notch_taps(500.0, fr), orc.design_notch(500.0, fr)
        assert len(got) == len(want) == 1                  # one multi-band design, as mne.filter.notch_filter builds
        np.testing.assert_allclose(got[0], want[0], atol=1e-15)
    with pytest.raises(ValueError):
        notch_taps(500.0, [60.0, 60.9])
    with pytest.raises(ValueError):
        lbm_amd.create_filter(500.0, None, None)


def test_ica_projection_matches_oracle():
    rs = np.random.RandomState(3)
    C = 30
    pca = np.linalg.qr(rs.randn(C, C))[0]
    un = rs.randn(20, 20)
    kw = dict(unmixing=un, mixing=np.linalg.inv(un), pca_components=pca, pca_mean=rs.randn(C), pre_whitener=2e-5,
              n_components=20, exclude=[1, 7])
    A, b = lbm_amd.ica_projection(**kw)
    Ao, bo = orc.ica_projection(**kw)
    np.testing.assert_allclose(A, Ao, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(b, bo, rtol=1e-12, atol=1e-18)
    np.testing.assert_allclose(lbm_amd.average_reference_matrix(5), orc.average_reference_matrix(5))


def test_constants_keep_reference_names_and_defaults():
    assert (lbm_amd.FILTER_LFREQ, lbm_amd.FILTER_HFREQ, lbm_amd.NOTCH_FREQ) == (4.0, 30.0, 60.0)
    assert (lbm_amd.TMIN, lbm_amd.TMAX, lbm_amd.BASELINE) == (-0.5, 1.5, (None, 0))
    assert lbm_amd.REJECT_CRITERIA == dict(eeg=150e-6) and lbm_amd.PSD_WINDOW == "hann"
    assert (lbm_amd.VQ_EMBEDDING_DIM, lbm_amd.VQ_NUM_EMBEDDINGS, lbm_amd.VQ_COMMITMENT_COST, lbm_amd.VQ_DECAY) == (256, 512, 0.25, 0.99)
    assert lbm_amd.BATCH_SIZE == 64 and lbm_amd.n_baseline_samples(-0.5, 500.0) == 251


def test_vqvae_loads_shipped_state_dict_unchanged(shipped):
    m = lbm_amd.VQVAE(30, 256, 512, 0.25, 0.99)
    sd = {k: torch.from_numpy(v) for k, v in shipped.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("decoder.") for k in missing)
    assert set(m.state_dict()) >= {"vq_layer.embedding.weight", "vq_layer.ema_cluster_size", "vq_layer.ema_w",
                                   "encoder.0.weight", "encoder.6.bias"}


def test_shard_range_partitions_in_rank_order():
    for E, W in ((10, 3), (8, 8), (5, 8), (1000000, 8), (0, 2)):
        spans = [lbm_amd.shard_range(E, r, W) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == E
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_welch_bins_grid():
    from lbm_amd.psd import welch_bins
    f, lo, hi = welch_bins(500.0, 500, 4.0, 30.0)
    assert (lo, hi) == (4, 30) and np.array_equal(f, np.arange(4.0, 31.0))
    f, lo, hi = welch_bins(250.0, 512, 4.0, 30.0)
    assert len(f) == 53 and (lo, hi) == (9, 61)
    f, lo, hi = welch_bins(500.0, 512, 1.0, 40.0)
    assert len(f) == 39


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback():
    x = np.zeros((2, 3, 64), dtype=np.float32)
    with pytest.raises(Exception):
        lbm_amd.filter_data(x, 500.0, 4.0, 30.0)
    with pytest.raises(Exception):
        lbm_amd.vq_argmin(torch.zeros(4, 256), torch.zeros(8, 256))


def test_argument_validation_without_launch():
    lib = _lib.load()
    # bad arguments are rejected before anything touches a device
    assert lib.lbm_fir_f32(None, None, 4, 100, None, 5, 0, None) == -1
    assert b"NULL" in lib.lbm_last_error()
    assert lib.lbm_fir_f32(None, None, 0, 100, None, 4, 0, None) == 0          # empty input is a no-op
    assert lib.lbm_vq_argmin_f32(None, 256, 0, 0, None, None, None, None, None, 10, 100, 512, None) == -2   # D % 8 != 0
    assert lib.lbm_welch_f32(None, None, None, 10, 100, 500.0, 500, 500, 250, None, 4, 30, 0, None) == -1  # nfft > T
    assert b"larger than the number of samples" in lib.lbm_last_error()
    assert lib.lbm_welch_tables_bytes(500, 500, 4, 30) >= 64 + 4 * 1500
    # the bf16-piece entry points and the resampler
    assert lib.lbm_vq_split_tables_bytes(512, 256) == 16 * 3 * 512 * 32 + 512 * 4 + 512 * 1024 and lib.lbm_vq_split_tables_bytes(512, 300) == 0
    assert lib.lbm_conv1d_split_tables_bytes(128, 256, 3) == 24 * 3 * 256 * 32
    assert lib.lbm_conv1d_split_tables_bytes(30, 64, 4) == 8 * 3 * 256 * 32            # Cin -> 32, Cout -> 256
    assert lib.lbm_conv1d_split_tables_bytes(5000, 64, 4) == 0
    assert lib.lbm_vq_argmin_bf16x6_f32(None, 256, 0, 0, None, None, None, None, None, 10, 300, 512, None) == -2
    assert lib.lbm_vq_argmin_bf16x6_f32(None, 256, 0, 0, None, None, None, None, None, 10, 256, 512, None) == -1
    assert b"NULL" in lib.lbm_last_error()
    assert lib.lbm_conv1d_bf16x6_f32(None, 376, None, None, None, 376, 4, 128, 375, 256, 3, 1, 1, 1, None) == -1
    assert lib.lbm_conv1d_bf16x6_f32(None, 376, None, None, None, 376, 0, 128, 375, 256, 3, 1, 1, 1, None) == 0   # empty batch
    assert lib.lbm_resample_length(1800000, 250, 500) == 900000 and lib.lbm_resample_length(1000, 3, 7) == 429   # any ratio
    assert lib.lbm_resample_length(3, 1, 2) == 0 and lib.lbm_resample_workspace_bytes(2, 1000, 3, 7) > 2 * 2 * 2048 * 8
    assert lib.lbm_resample_fft_f32(None, 1000, None, 500, 2, 1000, 3, 7, None, None) < 0


def test_library_is_built_from_the_sources_in_the_tree():
    """A failed rebuild leaves the previous liblbm_hip.so in place and every test keeps passing against it (round 3: an assembler error in
    a new kernel went unnoticed for three GPU runs).  The library must be newer than every source it is built from, and hold the entry
    points of the kernels added last (host symbols of csrc/conv_wino_ws.hip and csrc/conv_k4s2_ws.hip)."""
    from lbm_amd import build as b
    import shutil
    if not b.up_to_date() and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        b.build()                      # raises if a source no longer compiles
    assert os.path.exists(b.LIB), "liblbm_hip.so is missing: python -m lbm_amd.build"
    assert b.up_to_date(), "liblbm_hip.so is older than its sources: python -m lbm_amd.build (and check its exit code)"
    with open(b.LIB, "rb") as f:
        blob = f.read()
    for sym in (b"launch_wino3_ws", b"launch_k4s2_ws", b"wino3_ws_eligible", b"k4s2_ws_eligible", b"launch_k4s2_wino", b"k4s2_wino_eligible"):
        assert sym in blob, sym


def test_lds_dma_kernels_do_not_spill():
    """lbm_amd/build.py records every kernel's register use; the kernels that move data with LDS-DMA must be spill-free
    (build.py NO_SPILL_KERNELS).  No report = no evidence: where hipcc exists the library is rebuilt to produce it, otherwise the
    test FAILS rather than skips."""
    import json
    import shutil
    from lbm_amd import build as b
    path = os.path.join(os.path.dirname(b.__file__), "csrc", "_obj", "resources.json")
    if not os.path.exists(path):
        assert shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"), \
            "csrc/_obj/resources.json is missing and there is no hipcc to rebuild it: the spill check has no evidence"
        b.build(force=True)
    with open(path) as f:
        rep = json.load(f)
    for name in b.NO_SPILL_KERNELS:                       # every watched kernel must be in the report, with both fields parsed
        hits = [(fname, k, v) for fname, kernels in rep.items() for k, v in kernels.items() if name in k]
        assert hits, f"{name}: not in resources.json"
        for fname, k, v in hits:
            assert "VGPRs Spill" in v and "ScratchSize" in v, (fname, k, v)
            assert v["VGPRs Spill"] == 0 and v["ScratchSize"] == 0, (fname, k, v)


def test_build_spill_check_rejects_an_unparsed_report(monkeypatch):
    """if the compiler's remark text changes, build() must not pass silently on an empty parse."""
    from lbm_amd import build as b
    good = ("f.hip:1:1: remark: Function Name: _Z17welch_team_kernelILi2ELi1EEvv [-Rpass-analysis=kernel-resource-usage]\n"
            "f.hip:1:1: remark:     VGPRs: 100 [-Rpass-analysis=kernel-resource-usage]\n"
            "f.hip:1:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]\n"
            "f.hip:1:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]\n")
    rep = b._resources(good)
    assert rep == {"_Z17welch_team_kernelILi2ELi1EEvv": {"VGPRs": 100, "ScratchSize": 0, "VGPRs Spill": 0}}
    assert b._resources(good.replace("remark:", "note:")) == {}


def test_vq_screen_workspace_and_arguments():
    lib = _lib.load()
    assert lib.lbm_vq_screen_workspace_bytes(0, 512) == 256 + 2048
    assert lib.lbm_vq_screen_workspace_bytes(1000, 300) == 256 + 2048 + 8000
    assert lib.lbm_vq_screen_workspace_bytes(-1, 512) == 0
    assert lib.lbm_vq_split_tables_bytes(512, 256) == 16 * 3 * 512 * 32 + 512 * 4 + 512 * 1024      # pieces + the code-norm tail
    # argument validation happens before any launch
    assert lib.lbm_vq_argmin_screened_f32(None, 256, 0, 0, None, None, None, None, None, None, 0, 256, 512, None) == 0   # N == 0
    assert lib.lbm_vq_argmin_screened_f32(None, 256, 0, 0, None, None, None, None, None, None, 10, 256, 512, None) == -1
    assert lib.lbm_vq_argmin_screened_f32(None, 256, 0, 0, None, None, None, None, None, None, 10, 250, 512, None) == -2
    assert lib.lbm_vq_argmin_screened_f32(None, 256, 7, 0, None, None, None, None, None, None, 10, 256, 512, None) == -1


def test_dedup_codebook_and_remap_arguments():
    """quantizer.dedup_codebook on the host: first occurrences in ascending order, padding that cannot win, the map back; and the
    argument checks of lbm_vq_remap_codes_i64 that need no GPU."""
    import torch
    from lbm_amd.quantizer import dedup_codebook
    rs = np.random.RandomState(0)
    cb = rs.randn(130, 8).astype(np.float32)
    cb[[5, 90]] = cb[40]                  # first occurrence: 5
    cb[[0, 129]] = 0.0                    # first occurrence: 0
    ee = (cb.astype(np.float64) ** 2).sum(1).astype(np.float32)
    assert dedup_codebook(torch.from_numpy(rs.randn(64, 8).astype(np.float32)), torch.ones(64)) is None
    cbc, eec, cmap, kc = dedup_codebook(torch.from_numpy(cb), torch.from_numpy(ee))
    assert kc == 127 and cbc.shape == (128, 8) and eec.shape == (128,) and cmap.shape == (128,) and cmap.dtype == torch.int64
    first = cmap[:kc].numpy()
    assert np.all(np.diff(first) > 0) and set(first) == set(range(130)) - {40, 90, 129}
    assert np.array_equal(cbc[:kc].numpy(), cb[first]) and np.array_equal(eec[:kc].numpy(), ee[first])
    assert float(eec[kc:].min()) > 1e38 and float(cbc[kc:].abs().max()) == 0.0
    ee2 = ee.copy(); ee2[90] += 1.0       # same values, another constant: not a repeat (folded codebooks carry their own constants)
    assert dedup_codebook(torch.from_numpy(cb), torch.from_numpy(ee2))[3] == 128
    lib = _lib.load()
    assert lib.lbm_vq_remap_codes_i64(None, 0, None, 4, None, None, None) == -1          # NULL map
    assert lib.lbm_vq_remap_codes_i64(None, -1, cmap.data_ptr(), 4, None, None, None) == -1
    assert lib.lbm_vq_remap_codes_i64(None, 5, cmap.data_ptr(), 4, None, None, None) == -1   # NULL idx with N > 0
    assert lib.lbm_vq_remap_codes_i64(None, 0, cmap.data_ptr(), 4, cmap.data_ptr(), None, None) == -1   # searched-row counts with nowhere to go
    assert lib.lbm_vq_remap_codes_i64(None, 0, cmap.data_ptr(), 20000, None, cmap.data_ptr(), None) == -2   # counting: Kc <= 12288
    assert lib.lbm_vq_remap_codes_i64(None, 5, cmap.data_ptr(), 4, None, cmap.data_ptr(), None) == -1       # NULL idx with N > 0, counting mode
    assert lib.lbm_vq_remap_codes_i64(None, 0, cmap.data_ptr(), 0, None, None, None) == -1


def test_conv_k4s2_wino_arguments():
    """lbm_conv1d_k4s2_wino_f32 argument checks that need no GPU (include/lbm_abi.h)."""
    lib = _lib.load()
    assert lib.lbm_conv1d_k4s2_wino_supported(64, 128) == 1 and lib.lbm_conv1d_k4s2_wino_supported(30, 64) == 1
    assert lib.lbm_conv1d_k4s2_wino_supported(65, 128) == 0 and lib.lbm_conv1d_k4s2_wino_supported(64, 100) == 0
    assert lib.lbm_conv1d_k4s2_wino_supported(0, 64) == 0 and lib.lbm_conv1d_k4s2_wino_supported(64, 0) == 0
    assert lib.lbm_conv1d_k4s2_wino_f32(None, 752, None, None, None, 376, 0, 64, 750, 128, 1, None) == 0       # B == 0
    assert lib.lbm_conv1d_k4s2_wino_f32(None, 752, None, None, None, 376, 4, 64, 750, 128, 1, None) == -1      # NULL pointers
    assert lib.lbm_conv1d_k4s2_wino_f32(None, 752, None, None, None, 376, 4, 128, 750, 128, 1, None) == -2     # Cin > 64
    assert lib.lbm_conv1d_k4s2_wino_f32(None, 700, None, None, None, 376, 4, 64, 750, 128, 1, None) == -1      # pitch < row
    assert lib.lbm_conv1d_k4s2_wino_f32(None, 752, None, None, None, 376, 4, 64, 1, 128, 1, None) == -1        # Lin < 2


def test_conv_wino3_table_sizes_and_arguments():
    lib = _lib.load()
    assert lib.lbm_conv1d_wino3_tables_bytes(128, 256) == 16 * 2 * 4 * 256 * 16
    assert lib.lbm_conv1d_wino3_tables_bytes(64, 64) == 8 * 1 * 4 * 256 * 16
    assert lib.lbm_conv1d_wino3_tables_bytes(30, 256) == 0 and lib.lbm_conv1d_wino3_tables_bytes(128, 32) == 0
    assert lib.lbm_conv1d_wino3_tables_init(None, None, 128, 256, None) == -1
    assert lib.lbm_conv1d_wino3_f32(None, 376, None, None, None, 376, 0, 128, 375, 256, 1, None) == 0      # B == 0
    assert lib.lbm_conv1d_wino3_f32(None, 376, None, None, None, 376, 4, 128, 375, 256, 1, None) == -1     # NULL pointers
    assert lib.lbm_conv1d_wino3_f32(None, 376, None, None, None, 376, 4, 30, 375, 256, 1, None) == -2      # Cin % 8
    assert lib.lbm_conv1d_wino3_f32(None, 100, None, None, None, 376, 4, 128, 375, 256, 1, None) == -1     # pitch < row


def test_wrappers_run_on_the_device_of_their_tensor(monkeypatch):
    """a tensor on cuda:1 while cuda:0 is current must work as with plain torch ops -- the public wrappers switch to the
    device of their first CUDA tensor argument (no GPU needed here: torch.cuda's device calls are stubbed)."""
    import torch
    from lbm_amd import _lib as L
    state = {"cur": 0, "entered": []}

    class FakeDeviceCtx:
        def __init__(self, dev):
            self.idx = dev.index
        def __enter__(self):
            self.prev, state["cur"] = state["cur"], self.idx
            state["entered"].append(self.idx)
        def __exit__(self, *a):
            state["cur"] = self.prev

    monkeypatch.setattr(torch.cuda, "current_device", lambda: state["cur"])
    monkeypatch.setattr(torch.cuda, "device", FakeDeviceCtx)

    class FakeTensor(torch.Tensor):
        pass

    def fake(idx):
        t = FakeTensor()
        t.__class__ = type("T", (FakeTensor,), {"is_cuda": property(lambda s: True), "device": property(lambda s: torch.device("cuda", idx))})
        return t

    seen = []

    @L.on_tensor_device
    def wrapper(a, b=None, *, k=None):
        seen.append(state["cur"])
        return "r"

    assert wrapper(fake(1)) == "r" and seen == [1] and state["cur"] == 0            # switched for the call, restored after it
    assert wrapper(fake(0)) == "r" and seen == [1, 0] and state["entered"] == [1]      # already current: no switch
    assert wrapper(None, k=fake(2)) == "r" and seen[-1] == 2                           # keyword tensors count
    assert wrapper("array-like") == "r" and seen[-1] == 0                              # no CUDA tensor: the current device
    import lbm_amd
    for fn in (lbm_amd.vq_argmin, lbm_amd.conv1d, lbm_amd.psd_array_welch, lbm_amd.affine_post, lbm_amd.filter_data, lbm_amd.ica_apply,
               lbm_amd.Scaler.moments, lbm_amd.Scaler.transform, lbm_amd.VQVAE.encode, lbm_amd.vq_split_tables):
        assert hasattr(fn, "__wrapped__"), fn
