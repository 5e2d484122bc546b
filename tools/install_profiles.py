"""Copies what tools/profile_round.sh left under gpurun_out/prof_<tag>_config<N>/ into profiles/<name>_config<N>_* and
writes the matching round entry of profiles/hbm_traffic.json from the FETCH_SIZE / WRITE_SIZE passes.
usage: install_profiles.py <tag> <name> <round> <config> [<config> ...]     e.g. install_profiles.py r03a r03 3 3 2 4 5

The traced runs of profile_round.sh render 1 + STEPS frames (the first frame of a scene + the timed steps, no warm-up): a
frame that is rendered in bands has several trace dispatches, and the traffic of a step is their sum."""
import glob, json, os, re, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name, rnd = sys.argv[1], sys.argv[2], int(sys.argv[3])
traffic_path = os.path.join(root, "profiles", "hbm_traffic.json")
traffic = json.load(open(traffic_path))
for cfg in sys.argv[4:]:
    src = os.path.join(root, "gpurun_out", f"prof_{tag}_config{cfg}")
    dst = os.path.join(root, "profiles", f"{name}_config{cfg}_")
    shutil.copy(os.path.join(src, "summary.txt"), dst + "rocprof_summary.txt")
    shutil.copy(os.path.join(src, "bench.json"), dst + "bench.json")
    ks = glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], dst + "kernel_stats.csv")
    bench = json.loads([ln for ln in open(os.path.join(src, "bench.json")) if ln.startswith("{")][-1])
    frames = int(open(os.path.join(src, "frames.txt")).read()) if os.path.exists(os.path.join(src, "frames.txt")) else 2
    per, calls = {}, {}
    for line in open(os.path.join(src, "summary.txt")):
        m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)\s+\(n=(\d+)\)", line)
        if not m:
            continue
        k = m.group(1)
        if re.search(r"rtmi_trace_kernel<\d+, true,", k):  # the scene's one cost probe launch (the counting variant): not a timed dispatch
            continue
        key = "trace" if "rtmi_trace_kernel" in k else "resolve" if "rtmi_resolve" in k else None
        if key:
            per.setdefault(key, {})[m.group(2)] = float(m.group(3))
            calls[key] = int(m.group(4))
    t = per["trace"]
    bands = max(1, calls["trace"] // frames)
    entry = {
        "config": str(cfg), "round": rnd, "width": int(re.search(r"(\d+)x\d+,", bench["config"]["workload"]).group(1)),
        "spp": int(re.search(r"(\d+) spp", bench["config"]["workload"]).group(1)), "n_gpus": 1,
        "accel": re.search(r"accel=(\w+)", bench["config"]["workload"]).group(1),
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh {tag} {cfg}) of "
                  f"`python3 bench.py --config {cfg} --no-cpu-baseline --no-linear-scan --no-e2e`; profiles/{name}_config{cfg}_rocprof_summary.txt",
        "per_dispatch_KB": per, "trace_dispatches_per_step": bands,
        "bytes_per_launch": int((2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024 * bands),
        "resolve_bytes_per_step": int((2 * per["resolve"]["FETCH_SIZE"] + per["resolve"]["WRITE_SIZE"]) * 1024 * bands) if "resolve" in per else None,
        "note": "2*FETCH_SIZE + WRITE_SIZE (KB, gfx950 correction of MI355X_MICROARCH.md) summed over the trace dispatches of one step "
                "(a frame rendered in bands has several); WRITE_SIZE tallies 64 B per write request and the sample records leave as lone "
                "16-byte stores: it reads 3.65x their bytes (tools/ubench/write_size_calib.hip)",
    }
    traffic["entries"] = [e for e in traffic["entries"] if not (str(e.get("config")) == str(cfg) and e.get("round") == rnd)] + [entry]
    print(f"config {cfg}: {bands} trace dispatch(es) per step, trace {entry['bytes_per_launch'] / 1e9:.2f} GB per step")
json.dump(traffic, open(traffic_path, "w"), indent=1)
