set -x
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r6_t1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_t1.log
tail -5 gpurun_out/r6_t1.log
(python tools/ab_tuning.py 1920 512 rtow -- cam_entry=1; python tools/ab_tuning.py 1200 100 rtow -- cam_entry=1; python tools/ab_tuning.py 1920 64 grid -- cam_entry=1; python tools/ab_tuning.py 800 1024 cornell -- gen_ahead=1) > gpurun_out/r6_ab1.txt 2>&1
cat gpurun_out/r6_ab1.txt
python tools/ab_rounds.py --rounds 2 librtmi.so librtmi_ab_noga.so librtmi_ab_de4aa63.so librtmi_ab_r5final.so > gpurun_out/r6_ab2.txt 2>&1
cat gpurun_out/r6_ab2.txt
