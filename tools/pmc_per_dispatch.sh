#!/bin/bash
# per-dispatch PMC values (no averaging) for tools/one_render.py
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcd_$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/one_render.py "$@" > $OUT/log$i.txt 2>&1 || echo "pass $i failed"
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "rtmi_trace" in r["Kernel_Name"] or "rtmi_wavefront" in r["Kernel_Name"]:
        rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for d, v in rows.items():
    print("dispatch", d, " ".join(f"{k}={x:.4g}" for k, x in v.items()))
PY
done
