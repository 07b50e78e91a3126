"""Build liblbm_hip.so (gfx950) in-tree with hipcc.  `python -m lbm_amd.build [--force]`."""
from __future__ import annotations

import glob
import json
import os
import re
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


# Kernels that move data with LDS-DMA (global_load_lds) must not spill registers: a build of vq_screen_kernel with 13 spilled
# VGPRs returned wrong, run-to-run different tokens for whole half-waves of rows (round 2; the spill-free build of the same
# source is bit-exact).  The build records every kernel's registers / scratch / spills in csrc/_obj/resources.json and refuses
# a spilling kernel in these files.
NO_SPILL_FILES = {"vq.hip", "vq_bf16.hip", "vq_screen.hip", "conv_bf16.hip", "welch.hip"}
NO_SPILL_KERNELS = ("vq_argmin_dma_kernel", "vq_bf16x6_kernel", "vq_screen_kernel", "vq_screen_pipe_kernel", "conv_bf16x6_kernel", "welch_team_kernel")


def _resources(remarks: str) -> dict:
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    return out


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
        cmd = [_hipcc(), *flags, *extra, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    report = {}
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        res = _resources(out)
        report[os.path.basename(src)] = res
        if os.path.basename(src) in NO_SPILL_FILES:
            bad = {k: v for k, v in res.items() if any(n in k for n in NO_SPILL_KERNELS) and (v.get("VGPRs Spill", 0) or v.get("ScratchSize", 0))}
            if bad:
                raise RuntimeError(f"{os.path.basename(src)}: LDS-DMA kernels must not spill (see lbm_amd/build.py): {bad}")
        other = "\n".join(l for l in out.splitlines() if "kernel-resource-usage" not in l and not re.match(r"^\s*(\d+ \||\||\^)", l) and l.strip())
        if verbose and other.strip():
            print(other)
    with open(os.path.join(obj_dir, "resources.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
