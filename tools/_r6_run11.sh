set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6_kt_c5_new -- python3 $R/tools/one_frame.py 5 1024 2 > $R/gpurun_out/r6_kt_c5_new.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6_kt_c5_old -- python3 $R/tools/one_frame.py 5 1024 2 chain_resolve=1 > $R/gpurun_out/r6_kt_c5_old.log 2>&1
for d in new old; do echo "== $d"; cat $R/gpurun_out/r6_kt_c5_$d/*/*kernel_stats.csv | head -7; tail -2 $R/gpurun_out/r6_kt_c5_$d.log; done
