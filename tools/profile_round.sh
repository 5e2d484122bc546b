#!/bin/bash
# Runs on the GPU box: for one BASELINE config, the bench line, a rocprofv3 kernel trace and the PMC passes of the
# same command; summaries land in gpurun_out/prof_<tag>_config<N>/ (copy what is to be judged into profiles/).
# Usage: tools/profile_round.sh <tag> <config> <steps for the traced runs> [extra bench args]
set -o pipefail
TAG=$1; CFG=$2; STEPS=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_config${CFG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "== bench config $CFG"
python3 $ROOT/bench.py --config $CFG "$@" > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
cut -c1-400 $OUT/bench.json
QUIET="--config $CFG --no-cpu-baseline --no-linear-scan --no-e2e --no-traffic --steps $STEPS --warmup 0"
echo $((STEPS + 1)) > $OUT/frames.txt   # frames of a traced run: the first frame of the scene + the steps
echo "== kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $ROOT/bench.py $QUIET > $OUT/bench_kt.log 2>&1 || { echo "kernel-trace run failed"; tail -5 $OUT/bench_kt.log; exit 1; }
pass() { # name counters...
  local name=$1; shift
  echo "== pmc $name: $*"
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py $QUIET > $OUT/bench_pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; tail -3 $OUT/bench_pmc_$name.log; }
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT
pass c SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
wc -l $OUT/summary.txt
