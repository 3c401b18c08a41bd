#!/usr/bin/env python3
"""TV-L1 flow throughput: HIP path vs the reference's native library
(oracle/_ref/libBridge.so, built by oracle/Makefile) on the host cores, raw-resolution 640x360 pairs
(the flow size of a 1280x720 RGB frame).  One JSON line."""
import ctypes, json, os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", str(min(len(os.sched_getaffinity(0)), 16)))
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from rvdd_release_amd.util._ops import ops_runtime
from rvdd_release_amd import synth

h, w, npairs = 360, 640, int(os.environ.get("PAIRS", "6"))
seq = synth.make_sequence(npairs + 1, 2 * h, 2 * w, iso=3200, seed=77, device="cuda")
gray = seq.raw.mean(dim=1).contiguous()            # channel mean of the packed raw (library.py:165-167)
rt = ops_runtime(0)
rt.tvl1flow(gray[1], gray[0])                      # warm-up (allocates the pyramid)
torch.cuda.synchronize()
t0 = time.perf_counter(); iters = []
flows = []
for t in range(1, npairs + 1):
    u, it = rt.tvl1flow(gray[t], gray[t - 1], want_iterations=True)
    flows.append(u); iters.append(it)
torch.cuda.synchronize()
gpu_s = (time.perf_counter() - t0) / npairs
out = {"metric": "TV-L1 flows/sec, 640x360 pairs", "gpu_flows_per_s": round(1 / gpu_s, 2), "gpu_ms_per_flow": round(1e3 * gpu_s, 2),
       "mean_iterations": float(np.mean(iters)), "median_flow_px": [float(flows[0][0].median()), float(flows[0][1].median())]}
# the dataset's offline case: independent pairs, two per cooperative launch
torch.cuda.synchronize(); t0 = time.perf_counter()
fb = rt.tvl1flow_batch(gray[1:npairs + 1].contiguous(), gray[0:npairs].contiguous())
torch.cuda.synchronize(); bat_s = (time.perf_counter() - t0) / npairs
out.update({"gpu_batched_flows_per_s": round(1 / bat_s, 2), "gpu_batched_ms_per_flow": round(1e3 * bat_s, 2),
            "batched_equals_single": bool(all(torch.equal(fb[i], flows[i]) for i in range(npairs)))})
ref = os.path.join(REPO, "oracle", "_ref", "libBridge.so")
if os.path.exists(ref):
    lib = ctypes.CDLL(ref); lib.tvl1flow.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 2; lib.tvl1flow.restype = None
    g = gray.cpu().numpy(); u = np.zeros(2 * h * w, np.float32)
    n = min(3, npairs); t0 = time.perf_counter(); worst = 0.0; mean = 0.0
    for t in range(1, n + 1):
        a, b = np.ascontiguousarray(g[t]), np.ascontiguousarray(g[t - 1])
        lib.tvl1flow(a.ctypes.data, b.ctypes.data, u.ctypes.data, w, h)
        d = np.abs(u.reshape(2, h, w) - flows[t - 1].cpu().numpy()); worst = max(worst, float(d.max())); mean = max(mean, float(d.mean()))
    cpu_s = (time.perf_counter() - t0) / n
    out.update({"cpu_reference_flows_per_s": round(1 / cpu_s, 3), "cpu_threads": int(os.environ["OMP_NUM_THREADS"]), "gpu_over_cpu": round(cpu_s / gpu_s, 1),
                "gpu_vs_reference_max_abs_px": worst, "gpu_vs_reference_mean_abs_px": mean})
print(json.dumps(out))
sys.stdout.flush()
rt.close()        # before the interpreter tears the HIP runtime down (a profiler's exit handlers run after that)
