set -x
mkdir -p gpurun_out
(python tools/ab_tuning.py 1920 64 grid -- wait_thresh=48 wait_thresh=44 wait_thresh=40 walk_start=1,wait_thresh=48 walk_start=1,wait_thresh=44 cam_entry=2,wait_thresh=48 cam_entry=2,wait_thresh=44 walk_start=1,cam_entry=2,wait_thresh=48; python tools/ab_tuning.py 1920 256 grid -- wait_thresh=48 walk_start=1,wait_thresh=48 cam_entry=2,wait_thresh=48) > gpurun_out/r6_ab7.txt 2>&1
cat gpurun_out/r6_ab7.txt
