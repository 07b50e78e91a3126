#!/usr/bin/env python3
"""Scan the device assembly of csrc/*.hip for MFMA operands that are formed in ONE register right in front of each MFMA
(`v_add v153 ...; s_nop 1; v_mfma ... v153 ...` again and again): every such VALU write waits for the previous MFMA to have read the
register (round 3: that pattern cost conv_wino3_ws_kernel 2 % and welch_team_kernel 5 %).   python tools/asm_war_scan.py [file.hip ...]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lbm_amd import build as B

def scan(src):
    base = os.path.basename(src)
    flags = ["-O3", "-std=c++17", f"--offload-arch={B.ARCH}", "-fPIC", "-munsafe-fp-atomics"]
    if base in {"fir_fft.hip", "welch.hip", "resample.hip", "vq_screen.hip", "conv_k4s2_wino_ws.hip", "affine_post.hip"}: flags += ["-fno-slp-vectorize"]
    if base == "vq.hip": flags += ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
    with tempfile.TemporaryDirectory() as t:
        s = os.path.join(t, "dev.s")
        subprocess.run([B._hipcc(), *flags, "--cuda-device-only", "-S", src, "-o", s], check=True, stderr=subprocess.DEVNULL)
        text = open(s).read()
    out = []
    for m in re.finditer(r"\n(_Z\S+):.*?\n(.*?)\n\s*s_endpgm", text, re.S):
        name, body = m.group(1), [l.strip() for l in m.group(2).split("\n") if l.strip() and not l.strip().startswith(";")]
        mf = [i for i, l in enumerate(body) if l.startswith("v_mfma")]
        if not mf: continue
        hits, prev_ops = 0, set()
        for i in mf:
            ops = set(re.findall(r"\bv(\d+)\b(?!:)", body[i].split(",", 1)[1]))      # scalar-register A / B operands
            # was one of them written by a VALU instruction in the two instructions before, and also read by the previous MFMA?
            recent = set()
            for l in body[max(0, i - 3):i]:
                if l.startswith("v_") and not l.startswith("v_mfma"):
                    d = re.match(r"v_\S+\s+v(\d+)\b", l)
                    if d: recent.add(d.group(1))
            if ops & recent & prev_ops: hits += 1
            prev_ops = ops
        out.append((name, len(mf), hits))
    return out

files = sys.argv[1:] or sorted(glob.glob(os.path.join(B.CSRC, "*.hip")))
for f in files:
    for name, n, hits in scan(f if os.path.isabs(f) else os.path.join(B.CSRC, f)):
        if hits: print(f"{os.path.basename(f):18s} {hits:5d} of {n:5d} MFMAs take an operand re-formed in the previous MFMA's register   {name[:90]}")
