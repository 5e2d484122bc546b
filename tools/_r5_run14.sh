cd $GRAFT_REPO_ROOT
export RTMI_DIST_BACKEND=gloo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --config 2 > gpurun_out/r5_bench_gloo2.json 2> gpurun_out/r5_bench_gloo2.err; echo "rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5_bench_gloo2.json') if l.startswith('{')][-1])
print({k:d.get(k) for k in ('value','n_gpus','ms_per_step','kernel_ms_per_rank','kernel_ms_slowest_fastest','first_frame_ms_per_rank','gather_ms_per_rank','rccl_ranks')}, d.get('parity_check'))
PY
unset RTMI_DIST_BACKEND
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5_t10.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5_t10.log
grep -v "^Exc\|^Trace\|^  File\|^TypeE" gpurun_out/r5_t10.log | tail -4
timeout -k 10 400 python bench.py --config 5 --steps 2 --warmup 0 --no-traffic --no-cpu-baseline > gpurun_out/r5_bench5b.json 2> /dev/null; python -c "
import json; d=json.load(open('gpurun_out/r5_bench5b.json')); print(5, d['value'], d['ms_per_step'], d['kernel_ms_per_rank'], d['launch']['bands'], d['roofline']['frac'])"
