"""Screened search at D < 256 (vq_screen_kernel, the non-persistent form): time per call."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lbm_amd
from lbm_amd import _lib
dev = torch.device("cuda", 0)
lib, p, st = _lib.load(), _lib.ptr, _lib.stream_ptr()
g = torch.Generator(device=dev).manual_seed(1)
N = 4096 * 375
for D, K in ((128, 512), (64, 512), (128, 2048)):
    x = torch.randn(N, D, device=dev, generator=g)
    cb = torch.randn(K, D, device=dev, generator=g).contiguous()
    ee = lbm_amd.quantizer.rownorm2(cb)
    tab = lbm_amd.vq_split_tables(cb)
    ws = torch.empty(lib.lbm_vq_screen_workspace_bytes(N, K), dtype=torch.uint8, device=dev)
    tok = torch.empty(N, dtype=torch.int64, device=dev)
    ref = torch.empty(N, dtype=torch.int64, device=dev)
    assert lib.lbm_vq_argmin_f32(p(x), D, _lib.VQ_ROWS, 0, p(cb), p(ee), p(ref), None, None, N, D, K, st) == 0
    def run():
        assert lib.lbm_vq_argmin_screened_f32(p(x), D, _lib.VQ_ROWS, 0, p(cb), p(tab), p(ee), p(tok), None, p(ws), N, D, K, st) == 0
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize()
    print(f"D={D} K={K}: {(time.perf_counter() - t0) * 100:.3f} ms per call, equal fp32 search: {bool((tok == ref).all())}, refined {int(ws[:4].view(torch.int32).item())}")
