"""Weights-stationary Winograd kernel (lbm_conv1d_wino3_f32, Cin = 128) against the tile kernel (lbm_conv1d_pitched_f32): bitwise + timing."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lbm_amd import _lib
lib, p = _lib.load(), _lib.ptr
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)

def run(E, L, Cout, relu=1, bias=True, pitch=None, time_it=False):
    ld = pitch or ((L + 3) & ~3)
    x = torch.randn(E, 128, ld, device=dev, generator=g)
    x[..., L:] = 7.0e3   # the pitch padding must never reach a result
    w = torch.randn(Cout, 128, 3, device=dev, generator=g) * 0.05
    b = torch.randn(Cout, device=dev, generator=g) if bias else None
    tab = torch.empty(lib.lbm_conv1d_wino3_tables_bytes(128, Cout), dtype=torch.uint8, device=dev)
    assert lib.lbm_conv1d_wino3_tables_init(p(tab), p(w), 128, Cout, None) == 0
    y0 = torch.full((E, Cout, ld), -1.0, device=dev); y1 = torch.full((E, Cout, ld), -1.0, device=dev)
    bp = p(b) if bias else None
    rc = lib.lbm_conv1d_pitched_f32(p(x), ld, p(w), bp, p(y0), ld, E, 128, L, Cout, 3, 1, 1, relu, None); assert rc == 0, _lib.last_error()
    rc = lib.lbm_conv1d_wino3_f32(p(x), ld, p(tab), bp, p(y1), ld, E, 128, L, Cout, relu, None); assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    same = torch.equal(y0[..., :L], y1[..., :L])
    ref = torch.nn.functional.conv1d(x[..., :L].double(), w.double(), b.double() if bias else None, padding=1)
    if relu: ref = ref.clamp_min(0)
    err = (y1[..., :L].double() - ref).abs().max().item()
    print(f"E={E} L={L} ld={ld} Cout={Cout} relu={relu} bias={bias}: bitwise equal to the tile kernel: {same}; max |err| vs float64 {err:.2e}", flush=True)
    if time_it:
        for name, fn in (("tile", lambda: lib.lbm_conv1d_pitched_f32(p(x), ld, p(w), bp, p(y0), ld, E, 128, L, Cout, 3, 1, 1, relu, None)),
                         ("ws", lambda: lib.lbm_conv1d_wino3_f32(p(x), ld, p(tab), bp, p(y1), ld, E, 128, L, Cout, relu, None))):
            for _ in range(2): fn()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(5): fn()
            t1.record(); torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / 5
            print(f"   {name}: {ms:.3f} ms per launch = {2 * E * L * Cout * 128 * 3 / ms / 1e9:.1f} TFLOP/s direct-equivalent", flush=True)
    return same

ok = True
for a in ((3, 375, 256), (2, 100, 256), (2, 128, 64), (2, 129, 192), (1, 250, 256), (2, 2500, 256), (5, 375, 256, 0, False), (300, 375, 256), (2, 136, 256, 1, True, 140)):
    ok &= run(*a)
if len(sys.argv) > 1:
    ok &= run(int(sys.argv[1]), 375, 256, time_it=True)
print("ALL BITWISE EQUAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
