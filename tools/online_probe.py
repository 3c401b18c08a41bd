"""Why the online-flow mode runs slower inside the driver-form bench than on its own: the same quick_config
call (a) in a fresh process, (b) behind N seconds of the headline load in the same process, (c) behind the same load with that
load's runtime still alive.  usage: python tools/online_probe.py fresh|heated|alive [heat_steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
mode = sys.argv[1]
heat = int(sys.argv[2]) if len(sys.argv) > 2 else 6
keep = None
if mode in ("heated", "alive"):
    import torch
    from safetensors.torch import load_file
    from rvdd_release_amd import synth
    from rvdd_release_amd.runtime import RvddRuntime
    arch, stem, fut, iso, H, W, T, B, gflop = bench.CONFIGS["C2"]
    rt = RvddRuntime(arch, fut, B, H, W, 0)
    rt.load_state_dict(load_file(os.path.join(bench.REPO, "weights", stem + ".safetensors")))
    seqs = [synth.make_sequence(T, H, W, iso=iso, seed=2000 + b, device="cuda") for b in range(B)]
    raw = torch.stack([s.raw for s in seqs], 1).contiguous()
    fprev = torch.stack([s.flow_prev for s in seqs], 1).contiguous()
    outs = torch.empty(T - 1, B, 3, H, W, device="cuda")
    t0 = time.perf_counter()
    for _ in range(heat):
        bench.advance(rt, raw, fprev, None, outs, T, fut, False)
    torch.cuda.synchronize()
    print(mode, "heating: %d steps, %.1f s" % (heat, time.perf_counter() - t0))
    if mode == "alive":
        keep = (rt, raw, fprev, outs)
    else:
        rt.close()
        del rt, raw, fprev, outs, seqs
        torch.cuda.empty_cache()
r = bench.quick_config("C2", 2, 0, online_flow=True)
print(mode, "online", r["value"], "frames/s, conv launch", r["avg_launch_us"], "us,", r["ms_per_step"], "ms per step")
r = bench.quick_config("C2", 2, 0, online_flow=False)
print(mode, "offline", r["value"], "frames/s, conv launch", r["avg_launch_us"], "us,", r["ms_per_step"], "ms per step")
