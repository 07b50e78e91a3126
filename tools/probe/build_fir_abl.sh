#!/bin/bash
# Experiment build: tools/probe/liblbm_fir_<NAME>.so = the in-tree library with fir_fft.hip patched for an ablation
# (comma-separated list: nostore,nopad,nostage,notw,noxch,noin).  Results are garbage by design; only the launch time is read.
set -e
NAME=$1
ROOT=$(cd $(dirname $0)/../.. && pwd)
W=/tmp/lbm_fir_$NAME; rm -rf $W; mkdir -p $W/include; cp -r $ROOT/lbm_amd/csrc $W/; cp $ROOT/include/lbm_abi.h $W/include/
sed -i 's#../../include/lbm_abi.h#../include/lbm_abi.h#' $W/csrc/common.h
python3 - "$W/csrc/fir_fft.hip" "$NAME" <<'PY'
import re, sys
p, what = sys.argv[1], set(sys.argv[2].split(","))
s = open(p).read()
def rep(a, b, n=1):
    global s
    assert s.count(a) == n, (s.count(a), a)
    s = s.replace(a, b)
s = s.replace("    const int half = (ntaps - 1) >> 1;\n", "    const int half = (ntaps - 1) >> 1;\n    const bool never = ntaps == -5;\n", 1)
if "nostore" in what:
    rep("            if (active && o >= 0 && o < Lb) {", "            if (active && never && o >= 0 && o < Lb) {")
if "storel2" in what:   # the same store instructions, into 64 KB per workgroup slot that stays in L2
    rep("        float* yr = y + row * (long long)T;", "        float* yr = y + (blockIdx.x & 1023) * 16384;")
    rep("if (a0 + o < T) yr[a0 + o] = re[n1];", "if (a0 + o < T) yr[(a0 + o) & 8191] = re[n1];")
    rep("if (b0 + o < T) yr[b0 + o] = im[n1];", "if (b0 + o < T) yr[(b0 + o) & 8191] = im[n1];")
if "loadl2" in what:    # the same load instructions, from the first 1024 rows only (L2 / MALL resident)
    rep("const float* xr = x + (long long)((unsigned)item / (unsigned)npairs) * T;", "const float* xr = x + (long long)(((unsigned)item / (unsigned)npairs) & 1023) * T;")
if "nopad" in what or "nostage" in what:
    rep("const int nl = ms < 0 ? (-ms < WL ? -ms : WL) : 0;", "const int nl = 0;")
    rep("const int r0 = T - ms; ", "const int r0 = WL;")
if "nostage" in what:
    rep("if (m < T) *reinterpret_cast<float4*>(P + l) = v;", "if (m < T && never) *reinterpret_cast<float4*>(P + l) = v;")
    rep("if (m + c < T && l + c >= 0 && l + c < WL) P[l + c] = e[c];", "if (never && m + c < T && l + c >= 0 && l + c < WL) P[l + c] = e[c];")
if "notw" in what:
    rep("const float2 w = s_t1[k * Q + t];", "const float2 w = make_float2(hre[k], him[k]);", 2)
    rep("const float2 w = s_t2[k * R3 + n3];", "const float2 w = make_float2(him[k], hre[k]);", 2)
if "noxch" in what:
    s, n = re.subn(r"(for \(int \w+ = 0; \w+ < (?:16|R3); \+\+\w+\)) (s_x\[[^\]]+\] = make_float2\([^;]+\);)", r"\1 if (never) \2", s)
    assert n == 4, n
    s, n = re.subn(r"const float2 v = (s_x\[[^;]+\]);", r"float2 v = make_float2(re[0], im[1]); if (never) v = \1;", s)
    assert n == 4, n
if "vol" in what:      # plain ds_read_b64 / ds_write_b64 for the exchange (no ds_read2 / b128 merging)
    rep("    float2* s_x = s_t2 + Q + sub * XCH;  // [XCH] per sub-group\n", "    float2* s_x = s_t2 + Q + sub * XCH;  // [XCH] per sub-group\n    typedef float f2v __attribute__((ext_vector_type(2)));\n    volatile __attribute__((address_space(3))) f2v* vx = (volatile __attribute__((address_space(3))) f2v*)s_x;\n")
    s, n = re.subn(r"s_x\[([^\]]+)\] = make_float2\(([^;]+)\);", r"vx[\1] = (f2v){\2};", s)
    assert n == 4, n
    s, n = re.subn(r"const float2 v = s_x\[([^;]+)\];", r"const f2v v = vx[\1];", s)
    assert n == 4, n
if "voltw" in what:    # the same for the twiddle reads
    rep("    float2* s_t2 = s_t1 + N;             // [k2][n3] w_Q^{n3 k2}\n", "    float2* s_t2 = s_t1 + N;             // [k2][n3] w_Q^{n3 k2}\n    typedef float f2w __attribute__((ext_vector_type(2)));\n    volatile __attribute__((address_space(3))) f2w* vt1 = (volatile __attribute__((address_space(3))) f2w*)s_t1;\n    volatile __attribute__((address_space(3))) f2w* vt2 = (volatile __attribute__((address_space(3))) f2w*)s_t2;\n")
    rep("const float2 w = s_t1[k * Q + t];", "const f2w w = vt1[k * Q + t];", 2)
    rep("const float2 w = s_t2[k * R3 + n3];", "const f2w w = vt2[k * R3 + n3];", 2)
if "noin" in what:
    rep("            re[n1] = P[p];\n            im[n1] = has_b ? P[Lb + p] : 0.f;", "            re[n1] = hre[n1]; im[n1] = him[n1];\n            if (never) { re[n1] = P[p]; im[n1] = has_b ? P[Lb + p] : 0.f; }")
open(p, "w").write(s)
PY
cd $W
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -c csrc/fir_fft.hip -o csrc/_obj/fir_fft.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probe/liblbm_fir_$NAME.so csrc/_obj/*.o
echo built $NAME
