set -x
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6_t9.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_t9.log
tail -6 gpurun_out/r6_t9.log
python tools/ab_libs.py 1920 256 grid librtmi.so librtmi_ab_way2batch.so > gpurun_out/r6_ab10.txt 2>&1
cat gpurun_out/r6_ab10.txt
(python tools/ab_tuning.py 1920 256 grid -- wait_thresh=40 wait_thresh=48 walk_start=1,cam_entry=1,wait_thresh=52) > gpurun_out/r6_ab11.txt 2>&1
cat gpurun_out/r6_ab11.txt
