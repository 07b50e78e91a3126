#!/bin/bash
# usage: tools/pmc.sh <tag> <stage> [epochs]   -- runs the SQ counter pass and a GRBM pass for one stage
set -u
tag=$1; stage=$2; ep=${3:-16384}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sq -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --stage $stage --epochs $ep --reps 3 > $out/sq.log 2>&1 || { echo "sq pass failed"; tail -5 $out/sq.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/sq2 -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --stage $stage --epochs $ep --reps 3 > $out/sq2.log 2>&1 || { echo "sq2 pass failed"; tail -5 $out/sq2.log; exit 1; }
find $out -name "*counter_collection.csv" | head
