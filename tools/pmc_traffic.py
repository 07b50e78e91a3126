#!/usr/bin/env python3
"""HBM bytes per launch of the hot kernels from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
tools/profile_round.sh  ->  profiles/pmc_traffic.json.   usage: tools/pmc_traffic.py gpurun_out/prof_<tag>"""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
E, C, T, S = 16384, 30, 1500, 375
KERNEL = {"vq": "vq_argmin_dma_kernel", "vqs": "vq_screen_pipe_kernel", "vqb": "vq_bf16x6_kernel", "conv1": "conv_k4s2_wino_ws_kernel<32", "conv2": "conv_k4s2_wino_ws_kernel<64", "conv3": "conv_wino3_ws_kernel", "conv4": "conv_tile_kernel<1, 1, 256",
          "fir": "fir_fft_kernel", "welch_scaled": "welch_team_kernel", "affine_post": "affine_post_kernel"}
OUTKEY = {"vqs": "vq_screen", "vqb": "vq_bf16x6"}     # bench.py looks the quantizer up as vq_<method>
ALGO = {"vq": (256 * S * 4 + S * 8) * E, "vqs": (256 * S * 4 + S * 8) * E, "vqb": (256 * S * 4 + S * 8) * E, "conv1": (C * T * 4 + 64 * 750 * 4) * E, "conv2": (64 * 750 * 4 + 128 * S * 4) * E, "conv3": (128 * S * 4 + 256 * S * 4) * E,
        "conv4": (256 * S * 4 + 256 * S * 4) * E, "fir": 2 * C * T * 4 * E, "welch_scaled": (C * T * 4 + C * 27 * 4 + C * 4) * E,
        "affine_post": (2 * C * T * 4 + C * 4 + 1) * E}


def mean_counter(d, kernel, counter):
    vals = {}
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                key = (fn, r["Dispatch_Id"])   # one file per profiled process: dispatch ids repeat across files
                vals.setdefault(key, 0.0)
                vals[key] += float(r["Counter_Value"])
    if not vals:
        return None
    return sum(vals.values()) / len(vals)


out = {}
for st, kern in KERNEL.items():
    f = mean_counter(os.path.join(root, f"fetch_{st}"), kern, "FETCH_SIZE")
    w = mean_counter(os.path.join(root, f"write_{st}"), kern, "WRITE_SIZE")
    if f is None or w is None:
        continue
    out[OUTKEY.get(st, st)] = {"launch": f"16384 epochs (tools/kbench.py --stage {st})", "kernel": kern, "FETCH_SIZE_KiB_raw": round(f, 1),
               "WRITE_SIZE_KiB_raw": round(w, 1), "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024),
               "algorithmic_bytes_per_launch": ALGO[st],
               "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; KiB units"}
json.dump(out, sys.stdout, indent=1)
print()
