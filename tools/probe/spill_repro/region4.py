import sys
src, out, mode = sys.argv[1], sys.argv[2], sys.argv[3]
L = open(src).read().split("\n")
res, nwait = [], 0
for i, ln in enumerate(L, 1):
    res.append(ln)
    t = ln.strip()
    if 4669 <= i <= 4716 and t.startswith("s_waitcnt lgkmcnt(0)"):
        nwait += 1
        if mode == "u1" or (mode == "u2" and nwait == 1) or (mode == "u3" and nwait > 1):
            res += ["\ts_nop 7", "\ts_nop 7"]
open(out, "w").write("\n".join(res))
