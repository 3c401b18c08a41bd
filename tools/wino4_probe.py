"""F(4x4,3x3) Winograd kernel (option wino4) against the default F(2x2,3x3) path: max-abs difference of two frame-steps
at several sizes (forced at every size), then frames/s of C2-shaped work with and without it."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from safetensors.torch import load_file
from rvdd_release_amd import synth
from rvdd_release_amd.runtime import RvddRuntime
sd = load_file(ROOT + "/weights/recurrent-convunet+feat-iso3200.safetensors")
for B, H, W in ((1, 64, 96), (2, 72, 104), (1, 36, 52), (2, 180, 320), (2, 720, 1280)):
    seqs = [synth.make_sequence(3, H, W, iso=3200, seed=60 + b, device="cuda") for b in range(B)]
    st = lambda f: torch.stack([f(s) for s in seqs], 0)
    outs = []
    for w4 in (0, 2):
        rt = RvddRuntime("convunet+feat", 0, B, H, W, 0)
        rt.set_option("conv_kernel", 2)
        rt.set_option("wino4", w4)
        rt.load_state_dict(sd)
        o = [rt.step(st(lambda s: s.raw[0]), st(lambda s: s.raw[1]), None, st(lambda s: s.flow_prev[1]), None).clone(),
             rt.step(None, st(lambda s: s.raw[2]), None, st(lambda s: s.flow_prev[2]), None).clone()]
        outs.append(o)
        rt.close()
    print(f"B={B} {H}x{W}: max|wino4 - wino2| frame 1 {float((outs[0][0]-outs[1][0]).abs().max()):.3e}  frame 2 {float((outs[0][1]-outs[1][1]).abs().max()):.3e}", flush=True)
B, H, W, T = 8, 720, 1280, 10
seqs = [synth.make_sequence(T, H, W, iso=3200, seed=70 + b, device="cuda") for b in range(B)]
raw = torch.stack([s.raw for s in seqs], 1).contiguous(); fl = torch.stack([s.flow_prev for s in seqs], 1).contiguous()
for w4 in (0, 1, 0, 1):
    rt = RvddRuntime("convunet+feat", 0, B, H, W, 0)
    rt.set_option("wino4", w4)
    rt.load_state_dict(sd)
    out = torch.empty(B, 3, H, W, device="cuda")
    def run():
        rt.reset()
        for t in range(1, T):
            rt.step(raw[t - 1] if t == 1 else None, raw[t], None, fl[t], None, out=out)
    run(); torch.cuda.synchronize(); t0 = time.perf_counter(); run(); run(); torch.cuda.synchronize()
    print(f"wino4={w4}: {2 * (T - 1) * B / (time.perf_counter() - t0):.1f} frames/s", flush=True)
    rt.close()
