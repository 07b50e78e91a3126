import sys
src, out, lo, hi, mode = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
L = open(src).read().split("\n")
res = []
for i, ln in enumerate(L, 1):
    t = ln.strip()
    if mode == "vm_once" and i == lo:
        res.append("\ts_waitcnt vmcnt(0)")
    res.append(ln)
    if lo <= i <= hi:
        if mode == "after_ds" and t.startswith("ds_read"):
            res.append("\ts_waitcnt lgkmcnt(0)")
        if mode == "after_mfma_wait" and t.startswith("v_mfma"):
            res.append("\ts_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)")
        if mode == "after_mfma_nop" and t.startswith("v_mfma"):
            res += ["\ts_nop 7", "\ts_nop 7"]
open(out, "w").write("\n".join(res))
