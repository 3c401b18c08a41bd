#!/usr/bin/env python3
"""Probe: does running two half-batches on two HIP streams (so that one kernel's tail and filter-bank prologue sit
under the other stream's kernel) beat one full batch?  Config C2, frames/s over all sequences.
usage: python tools/two_stream_probe.py [B_total]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safetensors.torch import load_file
from rvdd_release_amd import synth
from rvdd_release_amd.runtime import RvddRuntime

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, T = 720, 1280, 12
BT = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sd = load_file(os.path.join(REPO, "weights", "recurrent-convunet+feat-iso3200.safetensors"))
seqs = [synth.make_sequence(T, H, W, iso=3200, seed=2000 + i, device="cuda") for i in range(BT)]


def run(nstreams):
    B = BT // nstreams
    rts, streams, data, outs = [], [], [], []
    for k in range(nstreams):
        rt = RvddRuntime("convunet+feat", 0, B, H, W, 0)
        rt.load_state_dict(sd)
        rts.append(rt)
        streams.append(torch.cuda.Stream() if nstreams > 1 else torch.cuda.current_stream())
        ss = seqs[k * B:(k + 1) * B]
        data.append((torch.stack([s.raw for s in ss], 1).contiguous(), torch.stack([s.flow_prev for s in ss], 1).contiguous()))
        outs.append(torch.empty(T - 1, B, 3, H, W, device="cuda"))

    def one_pass():
        for rt in rts:
            rt.reset()
        for t in range(1, T):
            for k in range(nstreams):
                with torch.cuda.stream(streams[k]):
                    raw, fl = data[k]
                    rts[k].step(raw[t - 1] if t == 1 else None, raw[t], None, fl[t], None, out=outs[k][t - 1])
    one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        one_pass()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fps = 3 * (T - 1) * BT / dt
    for rt in rts:
        rt.close()
    return fps, torch.cat([o for o in outs], 1)


for n in (1, 2, 1, 2, 4):
    fps, out = run(n)
    print(f"B_total={BT} streams={n}: {fps:.1f} frames/s  checksum {float(out.double().sum()):.6f}")
