"""Merge gpurun_out/pmc_entry_*.json (written by tools_pmc.sh on the GPU box) into
profiles/pmc_latest.json: one entry per (bins, dtype, gpus, workload); a newer entry replaces the
older one for the same key.  bench.py replays `hbm_bytes_per_launch` as roofline.traffic."""
import glob, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "pmc_latest.json")
try:
    d = json.load(open(path))
except (OSError, ValueError):
    d = {}
entries = d.get("entries", [d] if d.get("bins") else [])
key = lambda e: (e["bins"], e["dtype"], e.get("gpus", 1), e.get("workload", "dense"))
table = {key(e): e for e in entries}
for f in sorted(glob.glob(os.path.join(root, "gpurun_out", "pmc_entry_%s*.json" % (sys.argv[1] if len(sys.argv) > 1 else "")))):
    e = json.load(open(f))
    table[key(e)] = e
    print("merged", f, "x%.3f" % e["traffic_over_algorithmic"])
json.dump({"entries": [table[k] for k in sorted(table)]}, open(path, "w"), indent=1)
