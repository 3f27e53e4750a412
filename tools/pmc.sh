#!/bin/bash
# Runs ON the GPU box: one rocprofv3 --kernel-trace --stats pass and separate --pmc passes of one command
# (counters never share a run with trace domains other than the kernel trace; MI355X_MICROARCH.md §rocprofv3 PMC slots).
#   usage: tools/pmc.sh <out dir under gpurun_out/> <python script> [args ...]     (env vars are inherited)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
SCRIPT=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $SCRIPT "$@" > $OUT/trace.log 2>&1; echo "trace rc=$?"
i=0
for PMC in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $PMC -d $OUT/pmc_$i --output-format csv -- python3 $SCRIPT "$@" > $OUT/pmc_$i.log 2>&1; echo "pmc $i rc=$?"
done
find $OUT -name "*.db" -delete 2>/dev/null
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.json; cat $OUT/summary.json
