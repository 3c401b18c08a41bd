"""Diagnostic (librvdd_hip.so built with `make STAMPS=1`): per-phase cycles of the fused ConvBlock kernel, wave 0 of
workgroup 0 of the LAST block of a C4 frame-step, read from the first pixel of the recurrent features."""
import sys, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from safetensors.torch import load_file
from rvdd_release_amd import synth
from rvdd_release_amd.runtime import RvddRuntime
B, H, W = 4, 720, 1280
sd = load_file(ROOT + "/weights/recurrent-ConvNeXtUnet+feat-future-iso3200.safetensors")
rt = RvddRuntime("next+feat", 1, B, H, W, 0)
rt.load_state_dict(sd)
s = synth.make_sequence(4, H, W, seed=1, device="cuda")
st = lambda x: torch.stack([x] * B, 0)
for t in (1, 2):
    rt.step(st(s.raw[t - 1]) if t == 1 else None, st(s.raw[t]), st(s.raw[t + 1]), st(s.flow_prev[t]), st(s.flow_next[t]))
torch.cuda.synchronize()
_, feat = rt.get_state()
if os.environ.get("RVDD_NEXT_PIPE") == "1":
    a = feat[0, :16, 0, 0].cpu().tolist()
    f, b = a[:8], a[8:]
    print("front (wave 0), cycles per tile: loop top %.0f, waiting for halo chunks %.0f, taps %.0f, LayerNorm + exchange %.0f, barriers A + B %.0f" % tuple(x / f[0] for x in f[1:6]))
    print("back  (wave 4), cycles per tile: loop top %.0f, barrier A %.0f, exchange read + barrier B %.0f, MLP of four rows %.0f" % tuple(x / b[0] for x in b[1:4] + [b[4] + b[5] + b[6]]))
    print("      of which: set-up of the two row pairs %.0f, their 12 pairs of hidden blocks %.0f, epilogues %.0f" % tuple(x / b[0] for x in b[4:7]))
    sys.exit(0)
v = feat[0, :5, 0, 0].cpu().tolist()
n = v[0]
print("tiles", n, "cycles per tile (100 MHz ticks? shader cycles): dw %.0f ln+exch %.0f mlp %.0f wait %.0f total %.0f" % (v[1]/n, v[2]/n, v[3]/n, v[4]/n, sum(v[1:])/n))
