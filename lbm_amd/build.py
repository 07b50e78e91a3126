"""Build liblbm_hip.so (gfx950) in-tree with hipcc.  `python -m lbm_amd.build [--force]`."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblbm_hip.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def up_to_date() -> bool:
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "lbm_abi.h")]
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and up_to_date():
        return LIB
    objs = []
    obj_dir = os.path.join(HERE, "csrc", "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    flags = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]
    # The FFT kernels keep complex values as split re/im registers; the SLP vectorizer pairs them into packed-fp32 ops
    # and pays for it with register moves (fir_fft_kernel: 384 v_mov per item, 16 B of scratch; 3.9 -> 3.2 ms without).
    no_slp = {"fir_fft.hip", "welch.hip", "resample.hip"}
    procs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        extra = ["-fno-slp-vectorize"] if os.path.basename(src) in no_slp else []
        if os.path.basename(src) == "vq.hip":   # 12.23 -> 12.04 ms on the quantizer (A/B x3); neutral on the conv kernels
            extra += ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
        cmd = [_hipcc(), *flags, *extra, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
