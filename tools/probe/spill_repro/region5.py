import sys
src, out, ks = sys.argv[1], sys.argv[2], set(int(k) for k in sys.argv[3].split(","))
L = open(src).read().split("\n")
res, n = [], 0
for i, ln in enumerate(L, 1):
    res.append(ln)
    if 4669 <= i <= 4716 and ln.strip().startswith("ds_read"):
        n += 1
        if n in ks:
            res.append("\ts_waitcnt lgkmcnt(0)")
open(out, "w").write("\n".join(res))
