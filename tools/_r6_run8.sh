set -x
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r6_t8.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_t8.log
tail -5 gpurun_out/r6_t8.log
python tools/ab_libs.py 1920 256 grid librtmi.so librtmi_ab_way2batch.so > gpurun_out/r6_ab9.txt 2>&1
cat gpurun_out/r6_ab9.txt
