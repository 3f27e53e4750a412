#!/bin/bash
# Runs ON the GPU box (via gpurun): parity tests, the bench line, and rocprofv3 summaries.
# Everything is written under gpurun_out/ (merged back by gpurun); summaries worth keeping are
# copied into profiles/ by tools/summarize_profile.py afterwards.
#   usage: QG_GIT_HEAD=<short hash> [SKIP_TESTS=1] tools/gpu_round.sh <tag> [workload ...]     (the box receives a snapshot without .git;
#          SKIP_TESTS=1: only the profile passes — one gpurun call does not fit the tests and nine workloads)
set -o pipefail
TAG=${1:-r01}; shift
WLS=${@:-c3L c4L c3T c3Td c2T c5TF c5B c5L reduce long_k c2L w16 u8 w32T reduceW}
export QG_GIT_HEAD=${QG_GIT_HEAD:-unknown}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
echo "== pytest -m gpu" | tee $OUT/progress.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/progress.log
tail -3 $OUT/pytest_gpu.log
echo "== smoke" | tee -a $OUT/progress.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/progress.log
echo "== bench" | tee -a $OUT/progress.log
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/progress.log
cat $OUT/bench.json
fi
for WL in $WLS; do
  echo "== rocprofv3 kernel-trace $WL" | tee -a $OUT/progress.log
  (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_${WL}_trace --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps $( case $WL in c3T|c3Td|c5TF|c5B|long_k|w32T) echo 10;; *) echo 200;; esac ) --warmup 5 --no-extra --no-cpu > $OUT/prof_${WL}_trace.log 2>&1); echo "trace rc=$?" | tee -a $OUT/progress.log
  for PMC in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE"; do
    N=$(echo $PMC | tr ' ' '_' | cut -c1-24)
    echo "== rocprofv3 pmc $WL $N" | tee -a $OUT/progress.log
    (cd /tmp && timeout -k 10 600 rocprofv3 --pmc $PMC -d $OUT/prof_${WL}_pmc_$N --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 3 --warmup 1 --prewarm 0 --no-extra --no-cpu > $OUT/prof_${WL}_pmc_$N.log 2>&1); echo "pmc rc=$?" | tee -a $OUT/progress.log
  done
done
# keep the merged-back payload small: drop rocprof's per-process databases, keep csv
find $OUT -name "*.db" -delete 2>/dev/null
du -sh $OUT | tee -a $OUT/progress.log
