"""Per-branch lane table of the round-based kernel (VERDICT r2 #1a): for every piece of a round -- work fetch, primary
rays, segment set-up, the draw service's passes, the material branches, the sky branch and its replay, the record
store -- how often it runs per round (with at least one lane), with how many lanes, and how many instructions it is.

  python tools/branch_census.py --build                 # on the CPU: librtmi_prof.so (-DRTMI_PROF) + the static counts
  python tools/branch_census.py <width> <spp> [rtow|grid|cornell]   # on the GPU box: prints the table

Two diagnostic builds: -DRTMI_PROF=1 (cycle stamps and the walk's lane counts; only THIS one walks with the C++ node step, so its
times differ from the shipped build) and -DRTMI_PROF=2 (the branch census, counted with wave-uniform scalars -- ballot + popcount --
around the hand-written node loops of the shipped build).  Counts are those of the shipped kernel either way: same rounds, the
vote rules are the same."""
import ctypes as C, json, os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmi_loader
pkg = rtmi_loader.load()
here = os.path.dirname(pkg.LIB_PATH)
prefix = "librtmi_prof"
for a in sys.argv[1:]:
    if a.startswith("--libs="):  # another pair of diagnostic libraries (an older build for an A/B): <prefix>1.so, <prefix>2.so
        prefix = a.split("=", 1)[1]
prof_lib = os.path.join(here, prefix + "1.so")    # -DRTMI_PROF=1: cycle stamps + walk lanes
census_lib = os.path.join(here, prefix + "2.so")  # -DRTMI_PROF=2: branch census
static_json = os.path.join(here, "librtmi_prof_static.json")


def static_counts():
    """instructions between consecutive ISA marks of the shipped variant (-DRTMI_MARKS build, program order)"""
    src = [c for c in pkg.CSRC if c.endswith("rtmi_device.hip")][0]
    flags = [f for f in pkg.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    out = "/tmp/rtmi_marks.s"
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-DRTMI_MARKS", "-I", "include", "-S", "--cuda-device-only", src, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    txt = open(out).read()
    res = {}
    for want in ("rtmi_trace_kernelILi2ELb0ELb0ELi0E", "rtmi_trace_kernelILi2ELb0ELb1ELi0E"):
        m = re.search(r"^(_Z[\w]*%s[\w]*):" % re.escape(want), txt, re.M)
        body = txt[m.end():txt.index(".Lfunc_end", m.end())].splitlines()
        cur, rows = "prologue", {}
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
            rows.setdefault(cur, dict(valu=0, salu=0, lds=0, mem=0, wait=0))[kind] += 1
        res[want] = rows
    return res


if "--build" in sys.argv:
    for lib, mode in ((prof_lib, 1), (census_lib, 2)):
        cmd = ["/opt/rocm/bin/hipcc"] + pkg.HIPCC_FLAGS + [f"-DRTMI_PROF={mode}", "-I", "include", "-o", lib] + pkg.CSRC + ["-ldl"]
        subprocess.run(cmd, check=True)
    json.dump(static_counts(), open(static_json, "w"), indent=1)
    print("built", prof_lib, static_json)
    sys.exit(0)

pos = [a for a in sys.argv[1:] if not a.startswith('--')]
w, spp = int(pos[0]), int(pos[1])
which = pos[2] if len(pos) > 2 else "rtow"
if which == "grid":
    objs, mats, kw = pkg.workloads.big_grid(316)
    kw.update(image_width=w, samples_per_pixel=spp)
elif which == "cornell":
    objs, mats, kw = pkg.workloads.cornell_like()
    kw.update(image_width=w, samples_per_pixel=spp)
else:
    objs, mats = pkg.make_world_spheres(12345)
    kw = dict(image_width=w, samples_per_pixel=spp, max_depth=50)
cam = pkg.camera_setup(pkg.camera_params(**kw))
v = [0] * 128
for lib_path in (prof_lib, census_lib):
    pkg._lib = None
    pkg.LIB_PATH = lib_path
    with pkg.Scene(cam, objs, mats, accel=pkg.ACCEL_BRUTE if '--brute' in sys.argv else pkg.ACCEL_BVH) as sc:
        sc.render_rows(0, cam.img_height, 7)
        if lib_path == prof_lib:
            ms = sc.last_kernel_ms()
        big = not sc.launch_info()["scene_in_lds"]
        out = (C.c_ulonglong * 128)()
        assert pkg.lib().rtmi_prof_read(sc._h, out) == 0
    for i, x in enumerate(out):
        v[i] += int(x)
rounds = v[64 + 2 * 21]
static = json.load(open(static_json)) if os.path.exists(static_json) else {}
st = static.get("rtmi_trace_kernelILi2ELb0ELb1ELi0E" if big else "rtmi_trace_kernelILi2ELb0ELb0ELi0E", {})


def instr(*marks):
    tot = dict(valu=0, salu=0, lds=0, mem=0)
    for m in marks:
        for k in tot:
            tot[k] += st.get(m, {}).get(k, 0)
    return tot


rounds = max(1, rounds)
tot_cycles = sum(v[8:8 + 24])
print(f"# {which} {cam.img_width}x{cam.img_height}x{spp}: census build kernel {ms:.1f} ms, {rounds} wave-rounds, "
      f"{v[65 + 2 * 8] / rounds:.1f} shading lanes per round, {tot_cycles / rounds:.0f} stamped cycles per wave-round")
lt, ll, lp, nt, nl, npk = v[32:38]
print(f"# walk: node trips/round {nt / rounds:.2f} at {nl / max(1, nt):.1f} lanes stepping, {npk / max(1, nt):.1f} parked at a leaf; "
      f"leaf trips/round {lt / rounds:.2f} at {ll / max(1, lt):.1f} lanes, {lp / max(1, lt):.1f} parked at a node")
# (label, census slot or None, time slots)
rows = [
    ("work fetch", 0, (0,)),
    ("GEN: primary ray (jitter block, pixel sample, first disk attempt)", 1, (1,)),
    ("GEN: defocus-disk retry trips", 2, (20,)),
    ("BEGIN: reciprocals, stack reset", 3, (17,)),
    ("BEGIN: pad classes", 3, (18,)),
    ("BEGIN: peeled leaves (the ground sphere)", 3, (19, 2)),
    ("WALK: node steps + votes (C++ step in this build)", None, (3,)),
    ("WALK: leaf steps", None, (21,)),
    ("draw requests", 8, (4,)),
    ("draws: owner attempt 0", 4, ()),
    ("draws: owner attempt 1 (time: both owner passes)", 5, (5,)),
    ("draws: shared passes", 6, (6,)),
    ("draws: raw word for a Dielectric hit (rides in owner pass 0)", 7, ()),
    ("draws: p / |p| of the accepted point", 8, (7,)),
    ("SHADE common: normalize(ray.direction) for Dielectric + sky", 22, (8,)),
    ("hit record (p, outward normal, front face, material fetch)", 9, (9,)),
    ("Lambertian scatter", 10, ()),
    ("Metallic scatter (time: Lambertian + Metallic block)", 11, (10,)),
    ("Dielectric scatter", 12, (11,)),
    ("  of those: Schlick draw", 13, ()),
    ("continue: attenuation push, depth, next ray", 9, (12,)),
    ("sky colour", 14, (13,)),
    ("attenuation replay (lanes whose path has a chain)", 15, (14,)),
    ("sample end: record store, next sample", 16, (15,)),
    ("  of those: ended black (depth limit / absorbed)", 17, ()),
    ("loop glue", None, (16,)),
]
print(f"{'branch':66s} {'runs/round':>10s} {'lanes/run':>9s} {'lanes/round':>11s} {'cycles/round':>12s} {'share':>6s}")
for label, slot, tslots in rows:
    cyc = sum(v[8 + t] for t in tslots)
    ctxt = f"{cyc / rounds:12.0f} {100.0 * cyc / max(1, tot_cycles):5.1f}%" if tslots else ""
    if slot is None:
        print(f"{label:66s} {'':10s} {'':9s} {'':11s} {ctxt}")
    else:
        n, l = v[64 + 2 * slot], v[65 + 2 * slot]
        print(f"{label:66s} {n / rounds:10.3f} {l / max(1, n):9.2f} {l / rounds:11.2f} {ctxt}")
if st:
    print("# static instruction counts between the ISA marks of the shipped variant (program order; rarely taken slow paths included):")
    for m, c in st.items():
        print(f"#   {m:18s} valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} mem {c['mem']:3d}")
