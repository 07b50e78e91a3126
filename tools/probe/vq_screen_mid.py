import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lbm_amd
from lbm_amd import _lib
dev = torch.device("cuda", 0)
N, D, K = 1000, 256, 512
rs = np.random.RandomState(N + K)
scale = rs.choice([0.3, 1.0, 30.0], size=(K, 1))
cb = (rs.randn(K, D) * scale).astype(np.float32)
x = (cb[rs.randint(0, K, N)] + 0.7 * rs.randn(N, D)).astype(np.float32)
q = N // 5
small = np.flatnonzero(scale[:, 0] == 0.3)
print("small codes", small.size)
i0, i1 = small[rs.randint(0, small.size, q)], small[rs.randint(0, small.size, q)]
x[:q] = 0.5 * (cb[i0] + cb[i1])
xd, cbd = torch.from_numpy(x).to(dev), torch.from_numpy(cb).to(dev)
tab = lbm_amd.vq_split_tables(cbd)
lib = _lib.load()
ee = lbm_amd.quantizer.rownorm2(cbd)
ws = torch.zeros(lib.lbm_vq_screen_workspace_bytes(N, K), dtype=torch.uint8, device=dev)
idx = torch.empty(N, dtype=torch.int64, device=dev)
rc = lib.lbm_vq_argmin_screened_f32(_lib.ptr(xd), D, 0, 0, _lib.ptr(cbd), _lib.ptr(tab), _lib.ptr(ee), _lib.ptr(idx), None, _lib.ptr(ws), N, D, K, _lib.stream_ptr())
torch.cuda.synchronize()
n = int(ws[:4].view(torch.int32).item())
lst = ws[256 + 2048:256 + 2048 + 8 * n].view(torch.int64).cpu().numpy()
print("rc", rc, "refined", n, "listed among midpoints", np.sum(lst < q), "others", np.sum(lst >= q))
ref = lbm_amd.vq_argmin(xd, cbd)
print("equal", bool(torch.equal(ref, idx)), "ref tokens of first midpoints", ref[:6].tolist(), "pairs", list(zip(i0[:6].tolist(), i1[:6].tolist())))
d = ((x[:q, None, :].astype(np.float64) - cb[None].astype(np.float64)) ** 2).sum(-1)
o = np.sort(d, axis=1)
print("midpoint gap quantiles", np.quantile(o[:, 1] - o[:, 0], [0, 0.5, 1]), "best dist", np.median(o[:, 0]))
