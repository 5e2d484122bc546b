#!/bin/bash
# tools/instr_budget.py under rocprofv3 for the three views; prints executed wave-instructions per 64 segments
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in normal trapped sky; do
  rm -rf $R/gpurun_out/ib_$v
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/ib_$v -- python3 $R/tools/instr_budget.py $v > $R/gpurun_out/ib_$v.log 2>&1
  python3 - <<PY
import csv,glob,re
log=open("$R/gpurun_out/ib_$v.log").read()
m=re.search(r"VIEW (\w+) segments (\d+) samples (\d+) node_tests (\d+) sphere_tests (\d+) kernel_ms ([0-9.]+)",log)
seg,smp,nt,stt,ms=int(m.group(2)),int(m.group(3)),int(m.group(4)),int(m.group(5)),float(m.group(6))
f=glob.glob("$R/gpurun_out/ib_$v/**/*counter_collection.csv",recursive=True)[0]
acc={}
for r in csv.DictReader(open(f)):
    if re.search(r"rtmi_trace_kernel<2, false", r["Kernel_Name"]):
        acc[r["Counter_Name"]]=float(r["Counter_Value"])  # (one dispatch of the shipped variant)
rounds=seg/64.0
print(f"$v: {seg/smp:.2f} segments/sample, {nt/seg:.2f} box tests and {stt/seg:.2f} sphere tests per segment, kernel {ms:.2f} ms; per 64 segments: "
      f"VALU {acc['SQ_INSTS_VALU']/rounds:.0f}  SALU {acc['SQ_INSTS_SALU']/rounds:.0f}  LDS {acc['SQ_INSTS_LDS']/rounds:.0f} wave-instructions, "
      f"lanes active {100*acc['SQ_THREAD_CYCLES_VALU']/(64*acc['SQ_INSTS_VALU']):.0f} %, {ms*1e6/rounds*2.36:.0f} kcycles... ")
PY
done
