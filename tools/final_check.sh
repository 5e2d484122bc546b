#!/bin/bash
# What the driver does at the end of a round, on one GPU box: the GPU suite, smoke() and the default bench line (outputs under gpurun_out/).
# usage (from the repo root, on the GPU box): bash tools/final_check.sh
set -x
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6_t_final.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_t_final.log
tail -4 gpurun_out/r6_t_final.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_smoke.log 2>&1; tail -3 gpurun_out/r6_smoke.log
python bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err; tail -c 400 gpurun_out/r6_bench_default.json
