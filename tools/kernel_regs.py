"""Register / scratch / LDS usage of every kernel of a HIP source, from -Rpass-analysis=kernel-resource-usage.
usage: kernel_regs.py [source.hip] [extra -D flags ...]   (default: csrc/rtmi_device.hip)"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmi_loader
pkg = rtmi_loader.load()
args = sys.argv[1:]
src = args[0] if args and not args[0].startswith("-") else [c for c in pkg.CSRC if c.endswith("rtmi_device.hip")][0]
extra = [a for a in args if a.startswith("-")]
flags = [f for f in pkg.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
r = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + extra + ["-I", "include", "--cuda-device-only", "-c", src, "-o", "/tmp/_regs.o",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
if r.returncode != 0:
    print(r.stderr[-3000:])
    sys.exit(1)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark: +(\w[\w ]*\w)(?: \[[\w/]+\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(RtmiLaunch\)|void ", "", name)
    print(f"{name:60s} VGPRs {v.get('VGPRs', -1):3d}  AGPRs {v.get('AGPRs', 0):3d}  SGPRs {v.get('TotalSGPRs', v.get('SGPRs', -1)):3d}  scratch {v.get('ScratchSize', 0):4d}  "
          f"occupancy {v.get('Occupancy', -1)}  LDS {v.get('LDS Size', 0)}")
