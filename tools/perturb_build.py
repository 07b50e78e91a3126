#!/usr/bin/env python3
"""Timing-perturbed builds of one HIP source: a correctness tool, not a product path.

A kernel whose results depend on instruction timing has a race (or leans on an undocumented hardware timing).  This builds
lbm_amd/csrc/<file>.hip to device assembly with the product's flags, inserts delays at chosen places, assembles it again and links
a variant library next to the product's objects; run any test or tool against it with LBM_AMD_LIB=<variant .so>.

    python tools/perturb_build.py vq_screen.hip --after v_mfma --nops 2 --out /path/lib_variant.so [--kernel substring]
    python tools/perturb_build.py vq_screen.hip --before ds_read --nops 4 ...
    python tools/perturb_build.py vq_screen.hip --wait-after ds_read ...        (s_waitcnt lgkmcnt(0) after each match)

(round 3: this is how the 'spilling build returns wrong tokens' event of round 2 was traced -- the spill-free build of the same
kernel fails the same way with 16 idle cycles after each MFMA)
"""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LL = "/opt/rocm/lib/llvm/bin"


def run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise SystemExit(f"failed: {' '.join(cmd)}\n{r.stdout}")
    return r.stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--kernel", default="", help="only inside kernels whose (mangled) name contains this")
    ap.add_argument("--after", default=None, help="insert s_nop 7 x NOPS after each instruction starting with this")
    ap.add_argument("--before", default=None, help="insert s_nop 7 x NOPS before each instruction starting with this")
    ap.add_argument("--wait-after", default=None, help="insert s_waitcnt lgkmcnt(0) after each instruction starting with this")
    ap.add_argument("--nops", type=int, default=2)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    from lbm_amd import build as B
    src = os.path.join(B.CSRC, a.source)
    base = os.path.basename(src)
    flags = ["-O3", "-std=c++17", f"--offload-arch={B.ARCH}", "-fPIC", "-munsafe-fp-atomics"]
    if base in {"fir_fft.hip", "welch.hip", "resample.hip", "vq_screen.hip", "conv_k4s2_wino_ws.hip", "affine_post.hip"}:
        flags += ["-fno-slp-vectorize"]
    if base == "vq.hip":
        flags += ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
    if not B.up_to_date():
        B.build()
    with tempfile.TemporaryDirectory() as t:
        s = os.path.join(t, "dev.s")
        run([B._hipcc(), *flags, "--cuda-device-only", "-S", src, "-o", s])
        out, inside, n = [], False, 0
        for ln in open(s).read().split("\n"):
            tt = ln.strip()
            if ln and not ln[0].isspace() and tt.endswith(":") and not tt.startswith("."):      # a function label
                inside = a.kernel in tt
            if tt.startswith("s_endpgm"):
                out.append(ln)
                inside = False
                continue
            hit = lambda pat: inside and pat is not None and tt.startswith(pat)  # noqa: E731
            if hit(a.before):
                out += ["\ts_nop 7"] * a.nops
                n += 1
            out.append(ln)
            if hit(a.after):
                out += ["\ts_nop 7"] * a.nops
                n += 1
            if hit(a.wait_after):
                out.append("\ts_waitcnt lgkmcnt(0)")
                n += 1
        open(s, "w").write("\n".join(out))
        run([f"{LL}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={B.ARCH}", "-c", s, "-o", os.path.join(t, "dev.o")])
        run([f"{LL}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", os.path.join(t, "dev.out"), os.path.join(t, "dev.o")])
        run([f"{LL}/clang-offload-bundler", "-type=o", "-bundle-align=4096",
             f"-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--{B.ARCH}", "-input=/dev/null",
             f"-input={os.path.join(t, 'dev.out')}", f"-output={os.path.join(t, 'dev.hipfb')}"])
        obj = os.path.join(t, "variant.o")
        run([B._hipcc(), *flags, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", os.path.join(t, "dev.hipfb"), "-c", src, "-o", obj])
        objs = [os.path.join(B.CSRC, "_obj", os.path.basename(p)[:-4] + ".o") for p in B.sources() if os.path.basename(p) != base]
        run([B._hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", a.out, *objs, obj])
    print(f"{a.out}: {n} insertions in {base}" + (f" (kernels matching {a.kernel!r})" if a.kernel else ""))


if __name__ == "__main__":
    main()
