#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + PMC passes of the default bench.py command.
# Usage: tools/profile_gpu.sh <tag> [extra bench args]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
echo "== kernel trace" 
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench_kt.log 2>&1 || { echo "kernel-trace run failed"; tail -5 $OUT/bench_kt.log; exit 1; }
tail -1 $OUT/bench_kt.log
pass() { # name counters...
  local name=$1; shift
  echo "== pmc $name: $*"
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 0 > $OUT/bench_pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; tail -3 $OUT/bench_pmc_$name.log; }
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT
pass c SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
find $OUT -name "*.csv" | head -50
