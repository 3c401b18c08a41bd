#!/usr/bin/env python3
"""Per-kernel means of the counter passes written by tools/gpu_pmc.sh -> profiles/<tag>_counters.json.
usage: python tools/pmc_any_summary.py <tag> [kernel-name-substring ...]"""
import collections, csv, glob, json, os, sys
tag, want = sys.argv[1], sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"pmc_{tag}")
out = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(src, "g*", "*", "*_counter_collection.csv"))):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        if "anonymous namespace" not in k or "at::native" in k:
            continue
        name = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        out[name][c] = round(sum(v) / len(v), 1)
        out[name]["launches"] = len(v)
dst = os.path.join(root, "profiles", f"{tag}_counters.json")
json.dump({"note": "means over all launches of the kernel in `bench.py --frames 4` (all resolution levels mixed); one rocprofv3 --pmc pass per counter group",
           "kernels": out}, open(dst, "w"), indent=1)
for k, v in out.items():
    if not want or any(w in k for w in want):
        print(k, json.dumps(v))
