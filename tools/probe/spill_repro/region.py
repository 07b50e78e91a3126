import re, sys
src, out, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
L = open(src).read().split("\n")
res = []
for i, ln in enumerate(L, 1):
    res.append(ln)
    t = ln.strip()
    if lo <= i <= hi and t and not t.startswith((";", ".", "s_waitcnt", "s_cbranch", "s_branch", "s_endpgm", "s_barrier")) and not t.endswith(":") and not t.startswith(";;#"):
        res.append("\ts_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)")
open(out, "w").write("\n".join(res))
