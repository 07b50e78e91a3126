import sys
src, out, mode = sys.argv[1], sys.argv[2], sys.argv[3]
L = open(src).read().split("\n")
res = []
NOP64 = ["\ts_nop 7"] * 8
in_k = False
for i, ln in enumerate(L, 1):
    t = ln.strip()
    if 4669 <= i <= 4716 and mode == "t3" and t.startswith("ds_read") and res[-1].strip().startswith("v_mfma"):
        res += NOP64
    if i == 4391 and t.startswith(".LBB1_30"):
        res.append(ln)
        if mode == "t1":
            res += NOP64
        if mode == "t2":
            res += ["\tv_mov_b32_e32 v3, v4", "\tv_mov_b32_e32 v3, v20"]
        continue
    res.append(ln)
open(out, "w").write("\n".join(res))
