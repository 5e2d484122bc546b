cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5_t14.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_t14.log
grep -v "^Exc\|^Trace\|^  File\|^TypeE" gpurun_out/r5_t14.log | tail -3
python __graft_entry__.py smoke 2>&1 | grep smoke
bash tools/profile_round.sh r05c 5 1 --steps 2 --warmup 0 2>&1 | tail -2
timeout -k 10 300 python bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/r5_bench_default.json')); print({k:d.get(k) for k in ('value','ms_per_step','kernel_ms_per_rank','first_frame_ms','value_e2e','gpu_over_cpu_like_for_like')}, d['roofline']['frac'], d['roofline']['traffic_source'], d['cpu_baseline']['value'])"
