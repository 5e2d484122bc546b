"""Static instruction counts per phase of the round-based kernel: compiles rtmi_device.hip with -DRTMI_MARKS (comment
lines at the phase boundaries) and counts VALU / SALU / LDS / memory instructions between consecutive marks in program
order.  Loops are counted once -- multiply by the trip counts of tools/prof_stamps.py.  usage: isa_phases.py [kernel-substring]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmi_loader
pkg = rtmi_loader.load()
src = [c for c in pkg.CSRC if c.endswith("rtmi_device.hip")][0]
flags = [f for f in pkg.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
out = "/tmp/rtmi_marks.s"
subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-DRTMI_MARKS", "-I", "include", "-S", "--cuda-device-only", src, "-o", out],
               check=True, stderr=subprocess.DEVNULL)
want = sys.argv[1] if len(sys.argv) > 1 else "rtmi_trace_kernelILi2ELb0ELb0ELi0E"
txt = open(out).read()
m = re.search(r"^(_Z[\w]*%s[\w]*):" % re.escape(want), txt, re.M)
body = txt[m.end():txt.index(".Lfunc_end", m.end())].splitlines()
print(m.group(1))
cur, rows, order = "prologue", {}, []
for l in body:
    t = l.strip()
    if t.startswith("; @@"):
        cur = t[4:]
        continue
    if not t or t[0] in ".;" or t.endswith(":") or t.startswith(("L_", "//")):
        continue
    op = t.split()[0]
    kind = ("lds" if op.startswith("ds_") else "mem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
            "valu" if op.startswith("v_") else "wait" if op in ("s_waitcnt", "s_nop") else "salu")
    if cur not in rows:
        rows[cur] = dict(valu=0, salu=0, lds=0, mem=0, wait=0)
        order.append(cur)
    rows[cur][kind] += 1
tot = dict(valu=0, salu=0, lds=0, mem=0, wait=0)
for k in order:
    r = rows[k]
    print(f"  {k:16s} valu {r['valu']:5d}  salu {r['salu']:5d}  lds {r['lds']:4d}  mem {r['mem']:4d}  wait/nop {r['wait']:4d}")
    for q in tot:
        tot[q] += r[q]
print(f"  {'total':16s} valu {tot['valu']:5d}  salu {tot['salu']:5d}  lds {tot['lds']:4d}  mem {tot['mem']:4d}  wait/nop {tot['wait']:4d}")
