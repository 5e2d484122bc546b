set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r6_t3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_t3.log
tail -15 gpurun_out/r6_t3.log
python tools/shard_perf.py 1920 512 > gpurun_out/r6_shard_perf2.txt 2>&1
cat gpurun_out/r6_shard_perf2.txt
python tools/ab_rounds.py --rounds 2 librtmi.so librtmi_ab_de4aa63.so > gpurun_out/r6_ab4.txt 2>&1
cat gpurun_out/r6_ab4.txt
