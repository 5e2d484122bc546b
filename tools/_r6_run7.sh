set -x
mkdir -p gpurun_out
(python tools/ab_tuning.py 1920 256 grid -- wait_thresh=40 wait_thresh=48 wait_thresh=36 cam_entry=1 walk_start=1,cam_entry=1,wait_thresh=52 lds_top_nodes=200) > gpurun_out/r6_ab8.txt 2>&1
cat gpurun_out/r6_ab8.txt
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "walk_starts or camera_entries or launch_info or fuzz or config4" > gpurun_out/r6_t7.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_t7.log
tail -5 gpurun_out/r6_t7.log
