"""frames/s of C2-shaped work with the F(4x4,3x3) kernel on (timing only; used with diagnostic builds)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from safetensors.torch import load_file
from rvdd_release_amd import synth
from rvdd_release_amd.runtime import RvddRuntime
sd = load_file(ROOT + "/weights/recurrent-convunet+feat-iso3200.safetensors")
B, H, W, T = 8, 720, 1280, 8
seqs = [synth.make_sequence(T, H, W, iso=3200, seed=70 + b, device="cuda") for b in range(B)]
raw = torch.stack([s.raw for s in seqs], 1).contiguous(); fl = torch.stack([s.flow_prev for s in seqs], 1).contiguous()
for w4 in (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1").split(",")):
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
