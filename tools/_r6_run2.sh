set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6_t2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_t2.log
tail -15 gpurun_out/r6_t2.log
python tools/shard_perf.py 1920 512 > gpurun_out/r6_shard_perf.txt 2>&1
cat gpurun_out/r6_shard_perf.txt
(python tools/ab_tuning.py 1920 512 rtow -- wait_thresh=48 wait_thresh=56 wait_thresh=58 chunk_samples=16 chunk_samples=24) > gpurun_out/r6_ab3.txt 2>&1
cat gpurun_out/r6_ab3.txt
