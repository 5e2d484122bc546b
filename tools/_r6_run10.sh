set -x
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "chains or config5 or fuzz_slice or cost_ordered or scheduling or golden" > gpurun_out/r6_t10.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_t10.log
tail -6 gpurun_out/r6_t10.log
python tools/ab_tuning.py 800 1024 cornell -- chain_resolve=1 > gpurun_out/r6_ab12.txt 2>&1
cat gpurun_out/r6_ab12.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6_kt_c5 -- python3 $GRAFT_REPO_ROOT/tools/one_frame.py cornell 800 1024 > $GRAFT_REPO_ROOT/gpurun_out/r6_kt_c5.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r6_kt_c5/*/*kernel_stats.csv | head -8
