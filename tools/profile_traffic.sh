#!/bin/bash
# HBM-side traffic of the default bench.py command (separate --pmc passes, as the microarch guide prescribes).
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 $ROOT/bench.py --no-cpu-baseline --no-traffic --steps 2 --warmup 0 > $OUT/bench_$c.log 2>&1 || echo "pass $c failed"
done
python3 $ROOT/tools/summarize_prof.py $OUT 2>/dev/null | grep -E "rtmi_" || true
for c in FETCH_SIZE WRITE_SIZE; do f=$(find $OUT/$c -name "*counter_collection.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    k = (row["Kernel_Name"][:40], row["Counter_Name"])
    acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in sorted(acc.items()):
    if "rtmi" in k[0]:
        print(k[0], k[1], "mean KB per dispatch", v / n, "n", n)
PY
done
