#!/usr/bin/env python3
"""Per-kernel ISA statistics of a hipcc -S listing: loads, full vmcnt drains, branches, MFMAs.
Many `vmcnt(0)` + branches next to loads = the compiler serialised "load or zero" selects."""
import re
import subprocess
import sys

PAT = {"gload": r"global_load", "vm0": r"vmcnt\(0\)", "br": r"s_cbranch", "mfma": r"v_mfma", "gstore": r"global_store",
       "scratch": r"scratch_"}
for fn in sys.argv[1:]:
    txt = open(fn).read()
    for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", txt, re.S | re.M):
        body, name = m.group(0), m.group(1)
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = dn.replace("(anonymous namespace)::", "").split("(")[0][-60:]
        stats = " ".join(f"{k}={len(re.findall(p, body)):4d}" for k, p in PAT.items())
        print(f"{dn:60s} lines={body.count(chr(10)):5d} {stats}")
