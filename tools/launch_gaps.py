#!/usr/bin/env python3
"""Per-launch table of one frame-step from a rocprofv3 --kernel-trace CSV: for every launch position of the step (kernel
name in launch order) the median duration and the median idle gap between the end of the launch in front of it and its own
start, over all the frame-steps of the trace.  usage: python tools/launch_gaps.py <kernel_trace.csv> <launches per step>

The step's period is found by its first kernel (ha_green_kernel: one per frame-step)."""
import csv, statistics, sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "at::native" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
starts = [i for i, r in enumerate(rows) if name(r).startswith("ha_green_kernel")]
steps = [rows[a:b] for a, b in zip(starts, starts[1:])]
n = statistics.mode(len(s) for s in steps)
steps = [s for s in steps if len(s) == n][2:]          # the steady state of the step (first steps of a video differ)
print(f"{len(steps)} frame-steps of {n} launches")
tot_d = tot_g = 0.0
print(f"{'#':>3} {'kernel':60s} {'dur us':>8} {'gap us':>8}")
for k in range(n):
    d = statistics.median((int(s[k]["End_Timestamp"]) - int(s[k]["Start_Timestamp"])) / 1e3 for s in steps)
    g = statistics.median((int(s[k]["Start_Timestamp"]) - int(s[k - 1]["End_Timestamp"])) / 1e3 for s in steps) if k else 0.0
    tot_d += d
    tot_g += g
    print(f"{k:3d} {name(steps[0][k])[:60]:60s} {d:8.2f} {g:8.2f}")
period = statistics.median((int(b[0]["Start_Timestamp"]) - int(a[0]["Start_Timestamp"])) / 1e3 for a, b in zip(steps, steps[1:]))
print(f"sum of durations {tot_d:.1f} us, sum of gaps inside a step {tot_g:.1f} us, step period {period:.1f} us")
