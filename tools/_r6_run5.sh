set -x
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6_t5.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_t5.log
tail -15 gpurun_out/r6_t5.log
(python tools/ab_tuning.py 1920 64 grid -- walk_start=1 cam_entry=2 walk_start=1,cam_entry=2 wait_thresh=48 wait_thresh=56; python tools/ab_tuning.py 1920 256 grid -- walk_start=1) > gpurun_out/r6_ab6.txt 2>&1
cat gpurun_out/r6_ab6.txt
