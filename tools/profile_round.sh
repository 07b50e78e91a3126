#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# 1) rocprofv3 --kernel-trace --stats of the default bench.py run; 2) HBM traffic counters (separate --pmc passes)
#    of one 16384-epoch launch of each hot kernel.  Results land in gpurun_out/prof_<tag>/.
set -u
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --no-extras > $out/bench_trace.log 2>&1 || { echo "trace failed"; tail -5 $out/bench_trace.log; exit 1; }
for st in vq vqs vqb conv1 conv2 conv3 conv4 fir welch_scaled affine_post; do
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_$st -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --stage $st --reps 3 > $out/fetch_$st.log 2>&1 || { echo "fetch $st failed"; tail -5 $out/fetch_$st.log; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write_$st -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --stage $st --reps 3 > $out/write_$st.log 2>&1 || { echo "write $st failed"; tail -5 $out/write_$st.log; exit 1; }
done
find $out -name "*.csv" | head -40
