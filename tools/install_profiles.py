"""Copies what tools/profile_round.sh left under gpurun_out/prof_<tag>_config<N>/ into profiles/<name>_config<N>_* and
refreshes the matching round entry of profiles/hbm_traffic.json from the FETCH_SIZE / WRITE_SIZE passes.
usage: install_profiles.py <tag> <name> <round> <config> [<config> ...]     e.g. install_profiles.py r02c r02 2 3 2 4 5"""
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
    per = {}
    for line in open(os.path.join(src, "summary.txt")):
        m = re.match(r"(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)\s+\(n=", line)
        if not m:
            continue
        k = m.group(1)
        key = "rtmi_trace_kernel<bvh>" if "rtmi_trace_kernel" in k else "rtmi_resolve_kernel" if "rtmi_resolve" in k else None
        if key:
            per.setdefault(key, {})[m.group(2)] = float(m.group(3))
    for e in traffic["entries"]:
        if str(e.get("config")) == str(cfg) and e.get("round") == rnd:
            e["per_dispatch_KB"] = per
            t = per["rtmi_trace_kernel<bvh>"]
            e["bytes_per_launch"] = int((2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024)
            e["source"] = re.sub(r"profile_round\.sh \w+ ", f"profile_round.sh {tag} ", e["source"])
            print(f"config {cfg}: trace launch {e['bytes_per_launch'] / 1e9:.2f} GB")
json.dump(traffic, open(traffic_path, "w"), indent=1)
