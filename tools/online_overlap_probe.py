"""C2 with the flow recomputed online (validate.py --val_flow_from_denoised), the B sequences as G groups of B / G on G streams:
a group's chain is serial (output t-1 -> TV-L1 -> step t), but the groups are independent, so one group's latency-bound
coarse TV-L1 scales can run under another group's convolutions.  usage: python tools/online_overlap_probe.py [G ...]
Prints frames/s per G and whether the outputs equal those of G = 1 bit for bit."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from safetensors.torch import load_file
from rvdd_release_amd import synth
from rvdd_release_amd.runtime import RvddRuntime

arch, stem, fut, iso, H, W, T, B, gflop = bench.CONFIGS["C2"]
dev = torch.device("cuda", 0)
seqs = [synth.make_sequence(T, H, W, iso=iso, seed=2000 + b, device="cuda") for b in range(B)]
raw = torch.stack([s.raw for s in seqs], 1).contiguous()
fprev = torch.stack([s.flow_prev for s in seqs], 1).contiguous()
del seqs
n_out = T - 1
weights = load_file(os.path.join(bench.REPO, "weights", stem + ".safetensors"))
steps = 2
ref = None
for G in [int(a) for a in sys.argv[1:]] or [1, 2]:
    per = B // G
    rts, streams, views = [], [], []
    outs = torch.empty(n_out, B, 3, H, W, device=dev)
    for g in range(G):
        rt = RvddRuntime(arch, fut, per, H, W, 0)
        rt.load_state_dict(weights)
        rt.set_option("tvl1_async", 1)
        rts.append(rt)
        streams.append(torch.cuda.Stream(device=dev) if G > 1 else torch.cuda.current_stream(dev))
        sl = slice(g * per, (g + 1) * per)
        views.append((raw[:, sl].contiguous(), fprev[:, sl].contiguous(), torch.empty(n_out, per, 3, H, W, device=dev)))

    def run():
        for rt in rts:
            rt.reset()
        for t in range(1, T):
            for g in range(G):
                r, f, o = views[g]
                with torch.cuda.stream(streams[g]):
                    fp = f[t] if t == 1 else bench.flow_from_denoised(rts[g], o[t - 2], r[t])
                    rts[g].step(r[t - 1] if t == 1 else None, r[t], None, fp, None, out=o[t - 1])

    torch.cuda.synchronize()
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    for g in range(G):
        outs[:, g * per:(g + 1) * per] = views[g][2]
    same = None
    if ref is None:
        ref = outs.clone()
    else:
        same = bool(torch.equal(ref, outs))
    print(f"G={G}: {steps * n_out * B / el:.1f} frames/s, {1e3 * el / steps:.1f} ms per step" + ("" if same is None else f", outputs equal G=1: {same}"), flush=True)
    for rt in rts:
        rt.psnr_l1(views[0][2][0, :1], views[0][2][0, :1])      # reads the asynchronous batches' control word
        rt.close()
    del rts, views, outs
    torch.cuda.empty_cache()
