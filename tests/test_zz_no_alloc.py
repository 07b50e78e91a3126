"""The no-allocation entry point lbm_channel_moments_static_f32 (static device scratch) against lbm_channel_moments_f32.

Called through the C ABI (ctypes) with raw device pointers: two calls in a row, the second with a smaller input, each
compared bit for bit with the workspace-argument entry point on the same input; plus the size limit's error.
Collected last (file name) so that a failure here cannot hide other tests under `pytest -x`."""
import numpy as np
import pytest
import torch

from lbm_amd import _lib

pytestmark = pytest.mark.gpu

MAX_ROWS = 1 << 20


def _both(lib, x, keep, check_driver_memory=False):
    E, C, T = x.shape
    st = _lib.stream_ptr()
    ws = torch.full((max(E * C, 1), 2), float("nan"), dtype=torch.float64, device=x.device)
    want = torch.full((C, 3), float("nan"), dtype=torch.float64, device=x.device)
    got = torch.full((C, 3), float("nan"), dtype=torch.float64, device=x.device)
    kp = _lib.ptr(keep)
    assert lib.lbm_channel_moments_f32(_lib.ptr(x), kp, _lib.ptr(ws), _lib.ptr(want), E, C, T, st) == 0
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    free_before, _ = torch.cuda.mem_get_info()
    assert lib.lbm_channel_moments_static_f32(_lib.ptr(x), kp, _lib.ptr(got), E, C, T, st) == 0, lib.lbm_last_error()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    if check_driver_memory:
        # torch's counter cannot see a hipMalloc made inside the library; the driver's free-memory figure can.  Asked
        # only of a call AFTER the first on this device (the first resolves the symbol address and may load the module)
        free_after, _ = torch.cuda.mem_get_info()
        assert free_after >= free_before, (free_before, free_after)
    return got.cpu().numpy(), want.cpu().numpy()


def test_static_scratch_moments_match_workspace_entry_point_twice():
    assert torch.cuda.is_available()
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(3)
    # call 1: a large input (4096 epochs x 30 channels = 122 880 rows of the scratch), some epochs rejected
    x1 = torch.randn((4096, 30, 1500), generator=g, device="cuda") * 1e-5 + 3e-6
    keep1 = (torch.rand((4096,), generator=g, device="cuda") > 0.25).to(torch.uint8)
    got1, want1 = _both(lib, x1, keep1)
    assert np.array_equal(got1, want1) and np.isfinite(got1).all()
    assert got1[0, 2] == float(int(keep1.sum()) * 1500)
    ref = x1[keep1.bool()].double()
    np.testing.assert_allclose(got1[:, 0], ref.sum(dim=(0, 2)).cpu().numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(got1[:, 1], (ref * ref).sum(dim=(0, 2)).cpu().numpy(), rtol=1e-9)
    # call 2, right after, SMALLER: the scratch still holds call 1's partial sums beyond row 7*5
    x2 = torch.randn((7, 5, 333), generator=g, device="cuda")
    got2, want2 = _both(lib, x2, None, check_driver_memory=True)
    assert np.array_equal(got2, want2)
    np.testing.assert_allclose(got2[:, 0], x2.double().sum(dim=(0, 2)).cpu().numpy(), rtol=1e-9, atol=1e-9)
    assert (got2[:, 2] == 7 * 333).all()
    # empty input: moments are zeros, nothing is launched over the scratch
    got3, want3 = _both(lib, torch.empty((0, 4, 100), device="cuda"), None)
    assert np.array_equal(got3, want3) and (got3 == 0).all()


def test_static_scratch_rejects_sizes_beyond_its_maximum():
    lib = _lib.load()
    mom = torch.zeros((2, 3), dtype=torch.float64, device="cuda")
    x = torch.zeros((8,), device="cuda")                     # never read: the size check comes first
    rc = lib.lbm_channel_moments_static_f32(_lib.ptr(x), None, _lib.ptr(mom), MAX_ROWS // 2 + 1, 2, 1, _lib.stream_ptr())
    assert rc == -2 and b"static scratch" in lib.lbm_last_error()
    # the Python surface: Scaler.moments(static_scratch=True)
    from lbm_amd.epochs import Scaler
    xs = torch.randn((64, 3, 500), device="cuda")
    out = torch.empty((3, 3), dtype=torch.float64, device="cuda")
    a = Scaler().moments(xs, out=out, static_scratch=True)
    b = Scaler().moments(xs)
    assert a is out and torch.equal(a, b)
