#!/bin/bash
# Runs ON the GPU box (via gpurun): every opt-in fuzzer of tests/ once, with the seed given (new inputs each time), one JSON
# summary line per fuzzer into gpurun_out/<tag>_fuzz_all.jsonl.  A mismatch ends the run with the fuzzer's exit code.
#   usage: tools/fuzz_all.sh <tag> <seed>
set -o pipefail
TAG=${1:-fz}; SEED=${2:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_fuzz_all.jsonl
cd $GRAFT_REPO_ROOT
: > $OUT
run() { echo "== $*" >&2; timeout -k 10 420 python "$@" | tail -1 >> $OUT || exit $?; }
run tests/extended_fuzz.py 2500 $SEED
run tests/extended_fuzz_shapes.py 600 $SEED
run tests/extended_fuzz_tree_forms.py 2500 $SEED
run tests/extended_fuzz_cplx_fixed.py 2000 $SEED
run tests/extended_fuzz_cplx_fixed.py 2500 $SEED all
run tests/extended_fuzz_eltwise.py 1200 $SEED
run tests/extended_fuzz_cplx_eltwise.py 600 $SEED
run tests/extended_fuzz_pingpong.py 600 $SEED
run tests/extended_fuzz_misc.py 3 $SEED
run tests/extended_fuzz_wide.py 900 $SEED
run tests/extended_fuzz_words.py 1500 $SEED
cat $OUT
