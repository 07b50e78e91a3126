"""Weights-stationary 4-tap stride-2 kernels (conv_k4s2_ws.hip, reached through lbm_conv1d_pitched_f32 on 16-byte rows) against the tile
kernel (the same entry point on rows with an odd pitch, which the weights-stationary form does not take): bitwise + timing."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lbm_amd import _lib
lib, p = _lib.load(), _lib.ptr
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)

def run(E, Cin, L, Cout, relu=1, bias=True, time_it=False):
    Lout = (L - 2) // 2 + 1
    ld, ldo = (L + 3) & ~3, (Lout + 3) & ~3
    x = torch.randn(E, Cin, ld, device=dev, generator=g)
    x[..., L:] = float("nan")
    xo = torch.empty(E, Cin, L + 1 + (L & 1), device=dev); xo[..., :L] = x[..., :L]; xo[..., L:] = float("nan")   # odd pitch
    w = torch.randn(Cout, Cin, 4, device=dev, generator=g) * 0.05
    b = torch.randn(Cout, device=dev, generator=g) if bias else None
    y0 = torch.full((E, Cout, Lout + 1 + (Lout & 1)), -1.0, device=dev); y1 = torch.full((E, Cout, ldo), -1.0, device=dev)
    bp = p(b) if bias else None
    rc = lib.lbm_conv1d_pitched_f32(p(xo), xo.shape[2], p(w), bp, p(y0), y0.shape[2], E, Cin, L, Cout, 4, 2, 1, relu, None); assert rc == 0, _lib.last_error()
    rc = lib.lbm_conv1d_pitched_f32(p(x), ld, p(w), bp, p(y1), ldo, E, Cin, L, Cout, 4, 2, 1, relu, None); assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    same = torch.equal(y0[..., :Lout], y1[..., :Lout])
    ref = torch.nn.functional.conv1d(x[..., :L].double(), w.double(), b.double() if bias else None, stride=2, padding=1)
    if relu: ref = ref.clamp_min(0)
    err = (y1[..., :Lout].double() - ref).abs().max().item()
    print(f"E={E} Cin={Cin} L={L} Cout={Cout} relu={relu} bias={bias}: bitwise equal to the tile kernel: {same}; max |err| vs float64 {err:.2e}", flush=True)
    if time_it:
        for name, fn in (("tile", lambda: lib.lbm_conv1d_pitched_f32(p(xo), xo.shape[2], p(w), bp, p(y0), y0.shape[2], E, Cin, L, Cout, 4, 2, 1, relu, None)),
                         ("ws", lambda: lib.lbm_conv1d_pitched_f32(p(x), ld, p(w), bp, p(y1), ldo, E, Cin, L, Cout, 4, 2, 1, relu, None))):
            for _ in range(2): fn()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(5): fn()
            t1.record(); torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / 5
            print(f"   {name}: {ms:.3f} ms per launch = {2 * E * Lout * Cout * Cin * 4 / ms / 1e9:.1f} TFLOP/s", flush=True)
    return same and err < 1e-4

ok = True
for a in ((3, 30, 1500, 64), (100, 30, 1500, 64), (2, 30, 1001, 64), (2, 64, 750, 128), (300, 64, 750, 128), (2, 64, 500, 128), (1, 30, 2, 64),
          (2, 30, 10, 64), (2, 64, 5000, 128), (5, 30, 1500, 64, 0, False), (5, 64, 750, 128, 0, False)):
    ok &= run(*a)
if len(sys.argv) > 1:
    ok &= run(int(sys.argv[1]), 30, 1500, 64, time_it=True)
    ok &= run(int(sys.argv[1]), 64, 750, 128, time_it=True)
print("ALL BITWISE EQUAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
