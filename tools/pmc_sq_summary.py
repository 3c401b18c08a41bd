#!/usr/bin/env python3
"""Per-kernel means of the SQ counter passes written by tools/gpu_pmc_sq.sh -> profiles/<tag>_sq_counters.json."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"sq_{tag}")
out = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(src, "*", ""))):
    files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not files:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        if "anonymous namespace" not in k or "at::native" in k:
            continue
        name = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        out[name][c] = round(sum(v) / len(v), 3)
        out[name]["launches"] = len(v)
dst = os.path.join(root, "profiles", f"{tag}_sq_counters.json")
json.dump({"note": "means over the launches of `bench.py --frames 4 --batch 4` (all resolutions of the U-Net mixed); one rocprofv3 --pmc pass per counter",
           "kernels": out}, open(dst, "w"), indent=1)
for k, v in out.items():
    if "wino" in k or "mlp" in k:
        print(k, v)
