#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes written by tools/gpu_profile.sh into
profiles/<tag>_pmc_traffic.json and refresh profiles/traffic.json (read by bench.py).

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB and,
on gfx950, FETCH_SIZE reports exactly half of the bytes of a wide coalesced (16 B/lane) read stream
(/opt/skills/guides/MI355X_MICROARCH.md, section HBM); WRITE_SIZE is exact for 16-B/lane stores."""
import collections, csv, glob, json, os, sys

tag, config = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "C2")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")


def agg(sub, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))[0])):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


f, w = agg("pmc_fetch", "FETCH_SIZE"), agg("pmc_write", "WRITE_SIZE")
out = {}
for k in sorted(f):
    if "anonymous namespace" not in k or "at::native" in k:
        continue
    name = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    name = name.split("(")[0].replace(", 0>", ">") if "conv3x3" in name else name.split("(")[0]
    if name.startswith("conv3x3h_kernel<"):      # <CIN, EPI, ACC_IN, UPS, groups, taps>: bench.py names the first four; the 5x5 form by itself
        a = [x.strip() for x in name[len("conv3x3h_kernel<"):-1].split(",")]
        name = "conv5x5h_kernel<16>" if len(a) > 5 and a[5] == "5" else "conv3x3h_kernel<" + ", ".join(a[:4]) + ">"
    fm = sum(f[k]) / len(f[k])
    wm = sum(w[k]) / len(w[k]) if k in w else 0.0
    out[name] = {"launches": len(f[k]), "FETCH_SIZE_KiB_avg": round(fm, 1), "WRITE_SIZE_KiB_avg": round(wm, 1),
                 "hbm_bytes_per_launch": round((2 * fm + wm) * 1024)}
# bench.py times the fused ConvBlock as ONE class, "convblock_kernel": every instantiation (plain, with the pooling / projection /
# 1x1-output epilogues), every 3rd launch.  The same mix here: the launch-weighted mean over all of them.
blk = {k: v for k, v in out.items() if k.startswith("convblock_kernel<") or k.startswith("convblock_pipe_kernel<")}
if blk:
    n = sum(v["launches"] for v in blk.values())
    out["convblock_kernel"] = {"launches": n, "instantiations": sorted(blk),
                               "hbm_bytes_per_launch": round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in blk.values()) / n)}
dst = os.path.join(root, "profiles", f"{tag}_pmc_traffic.json")
json.dump({"config": config, "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE halves wide reads)",
           "kernels": out}, open(dst, "w"), indent=1)
tj = os.path.join(root, "profiles", "traffic.json")
allc = json.load(open(tj)) if os.path.exists(tj) else {}
# the kernel sources these bytes were measured on: bench.py quotes the traffic only while the sources still hash to this
import hashlib
hs = hashlib.sha256()
for fn in sorted(glob.glob(os.path.join(root, "rvdd-release_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "rvdd-release_amd", "csrc", "*.inc"))
                 + glob.glob(os.path.join(root, "rvdd-release_amd", "csrc", "*.h"))):
    hs.update(open(fn, "rb").read())
allc[config] = {"source": os.path.basename(dst), "csrc_sha256_16": hs.hexdigest()[:16],
                "kernels": {k: v["hbm_bytes_per_launch"] for k, v in out.items()}}
json.dump(allc, open(tj, "w"), indent=1)
print(open(dst).read()[:1500])
