"""Screen kernel alone at several K (rocprofv3 --kernel-trace --stats gives the per-kernel times)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lbm_amd
from lbm_amd import _lib
dev = torch.device("cuda", 0)
lib, p, st = _lib.load(), _lib.ptr, _lib.stream_ptr()
g = torch.Generator(device=dev).manual_seed(1)
E, S, PS = 4096, 375, 376
z = torch.randn(E, 256, PS, device=dev, generator=g)
for K in [int(k) for k in os.environ.get("KS", "512,2048,8192").split(",")]:
    cb = torch.randn(K, 256, device=dev, generator=g).contiguous()
    ee = lbm_amd.quantizer.rownorm2(cb)
    tab = lbm_amd.vq_split_tables(cb)
    ws = torch.empty(lib.lbm_vq_screen_workspace_bytes(E * S, K), dtype=torch.uint8, device=dev)
    tok = torch.empty(E * S, dtype=torch.int64, device=dev)
    for _ in range(2):
        rc = lib.lbm_vq_argmin_screened_f32(p(z), PS, _lib.VQ_CHANNEL_MAJOR, S, p(cb), p(tab), p(ee), p(tok), None, p(ws), E * S, 256, K, st)
        assert rc == 0
    torch.cuda.synchronize()
    print("K", K, "refined", int(ws[:4].view(torch.int32).item()))
