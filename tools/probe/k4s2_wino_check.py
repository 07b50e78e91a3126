"""Debug: lbm_conv1d_k4s2_wino_f32 against a float64 convolution (and the direct kernel beside it), several shapes, NaN pitch padding,
then its time on the encoder's layers 1 and 2 against lbm_conv1d_pitched_f32."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lbm_amd
from lbm_amd import _lib
lib, p = _lib.load(), _lib.ptr
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)

def pitched(B, C, L, fill):
    P = (L + 3) & ~3
    buf = torch.full((B, C, P), fill, device=dev)
    return buf, P

def run(fn, x, w, bias, relu, wino):
    B, Cin, Lin = x.shape
    Cout = w.shape[0]
    Lout = (Lin - 2) // 2 + 1
    xb, ldx = pitched(B, Cin, Lin, float("nan"))
    xb[:, :, :Lin] = x
    yb, ldy = pitched(B, Cout, Lout, float("nan"))
    if wino:
        rc = lib.lbm_conv1d_k4s2_wino_f32(p(xb), ldx, p(w), p(bias), p(yb), ldy, B, Cin, Lin, Cout, int(relu), None)
    else:
        rc = lib.lbm_conv1d_pitched_f32(p(xb), ldx, p(w), p(bias), p(yb), ldy, B, Cin, Lin, Cout, 4, 2, 1, int(relu), None)
    assert rc == 0, _lib.last_error() if hasattr(_lib, "last_error") else rc
    torch.cuda.synchronize()
    return yb[:, :, :Lout], yb

bad = 0
shapes = [(3, 64, 750, 128, 1), (2, 30, 1500, 64, 1), (5, 64, 750, 128, 0), (2, 64, 1001, 64, 1), (1, 30, 39, 64, 1), (4, 17, 200, 192, 0),
          (2, 64, 2, 128, 1), (3, 64, 5, 64, 1), (2, 48, 3000, 128, 1), (1, 64, 383, 128, 1), (1, 64, 385, 128, 1), (300, 64, 750, 128, 1)]
for B, Cin, Lin, Cout, relu in shapes:
    x = torch.randn(B, Cin, Lin, device=dev, generator=g)
    if relu: x = x.clamp_min(0)     # the layers' own inputs are post-ReLU / z-scored
    w = torch.randn(Cout, Cin, 4, device=dev, generator=g) * (1.0 / np.sqrt(4 * Cin))
    bias = torch.randn(Cout, device=dev, generator=g) * 0.1
    want = F.conv1d(x.double(), w.double(), bias.double(), stride=2, padding=1)
    if relu: want = want.clamp_min(0)
    yw, ybuf = run(None, x, w, bias, relu, True)
    yd, _ = run(None, x, w, bias, relu, False)
    ew, ed = (yw.double() - want).abs().max().item(), (yd.double() - want).abs().max().item()
    sc = want.abs().max().item()
    Lout = want.shape[2]
    nanpad = bool(torch.isnan(ybuf[:, :, :Lout]).any())
    ok = ew <= 4e-6 * max(sc, 1.0) and not nanpad
    bad += not ok
    print(f"B {B} Cin {Cin} Lin {Lin} Cout {Cout} relu {relu}: wino err {ew:.3g}  direct err {ed:.3g}  scale {sc:.3g}  {'ok' if ok else 'BAD'}", flush=True)

with np.load(os.path.join(ROOT, "tests", "golden", "vq_vae_shipped.npz")) as f:
    W = {k: torch.from_numpy(f[k]).to(dev) for k in f.files}
E = 16384
for name, Cin, Lin, Cout, wk in (("conv1", 30, 1500, 64, "encoder.0"), ("conv2", 64, 750, 128, "encoder.2")):
    PX = (Lin + 3) & ~3
    x = torch.randn(E, Cin, PX, device=dev, generator=g).clamp_min(0)
    w, bias = W[wk + ".weight"], W[wk + ".bias"]
    Lout = Lin // 2
    P = (Lout + 3) & ~3
    yb = torch.empty(E, Cout, P, device=dev)
    for label, wino in (("direct (ws)", False), ("F(3,2) phases", True)):
        ts = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for r in range(6):
            e0.record()
            if wino:
                rc = lib.lbm_conv1d_k4s2_wino_f32(p(x), PX, p(w), p(bias), p(yb), P, E, Cin, Lin, Cout, 1, None)
            else:
                rc = lib.lbm_conv1d_pitched_f32(p(x), PX, p(w), p(bias), p(yb), P, E, Cin, Lin, Cout, 4, 2, 1, 1, None)
            e1.record(); torch.cuda.synchronize()
            assert rc == 0
            ts.append(e0.elapsed_time(e1))
        print(f"{name} {label}: {np.mean(ts[1:]):.3f} ms per 16384 epochs (min {min(ts):.3f})", flush=True)
sys.exit(1 if bad else 0)
