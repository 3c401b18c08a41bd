#!/usr/bin/env python3
"""Headline benchmark: output frames/s of the recurrent denoise+demosaic hot
path on N MI355X (one process per GPU), BASELINE.json's metric.

  python bench.py                       # N=1, config C2, finishes in minutes
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
      --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input already
resident in HBM: B independent sequences of T frames advanced in lockstep on
each GPU (T-1-future output frames per sequence).  Sequences share nothing, so
ranks never exchange data inside the timed region (weak scaling); the only
collectives are the barrier/max that bracket it and one all-gather of the
per-frame PSNR values afterwards.

The JSON line carries `roofline` for the dominant kernel (the 48->48 3x3 conv,
f32 MFMA: FLOP-bound, peak 157.3 TFLOP/s) measured with HIP events around each
of its launches inside the timed region, and `cpu_baseline`: the CPU oracle
(oracle/rvdd_oracle.py, a torch-CPU restatement of the reference's PyTorch
path) timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

CONFIGS = {
    # name: (arch, weights stem, future, iso, H, W, T, algorithmic GFLOP per output frame @ HxW)
    "C1": ("convunet", "recurrent-convunet-iso3200", 0, 3200, 256, 256, 8, 25.50),
    "C2": ("convunet+feat", "recurrent-convunet+feat-iso3200", 0, 3200, 720, 1280, 30, 435.025),
    "C3": ("convunet+feat", "recurrent-convunet+feat-future-iso12800", 1, 12800, 720, 1280, 30, 437.413),
    "C4": ("next+feat", "recurrent-ConvNeXtUnet+feat-future-iso3200", 1, 3200, 720, 1280, 30, 401.998),
}
DESCR = {
    "C1": "RVDD-basic (recurrent convunet) ISO3200 256x256 8-frame sequences",
    "C2": "recurrent convunet+feat ISO3200 1280x720 30-frame sequences",
    "C3": "recurrent convunet+feat+future ISO12800 1280x720 30-frame sequences",
    "C4": "recurrent ConvNeXtUnet+feat+future ISO3200 1280x720 30-frame sequences",
}
FP32_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, chip table (vector = matrix f32)
_CONV = "conv3x3_kernel<48, 1, false>" if os.environ.get("RVDD_CONV") == "direct" else "wino3x3_kernel<1, false>"
DOMINANT = {"convunet": _CONV, "convunet+feat": _CONV, "next": "mlp_kernel", "next+feat": "mlp_kernel"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=4, help="sequences advanced in lockstep per GPU")
    ap.add_argument("--frames", type=int, default=0, help="override frames per sequence")
    ap.add_argument("--cpu-frames", type=int, default=8, help="frames of the CPU-oracle sample (0 = skip)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--all-kernel-events", action="store_true",
                    help="diagnostic: bracket EVERY launch of every kernel (costs ~6 %% of the frame rate); "
                         "default brackets every 4th launch of the dominant kernel only")
    args = ap.parse_args()

    from safetensors.torch import load_file
    from rvdd_release_amd import shard, synth
    from rvdd_release_amd.runtime import RvddRuntime

    arch, stem, fut, iso, H, W, T, gflop_frame = CONFIGS[args.config]
    if args.frames:
        T = args.frames
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: "
                         "launch with torch.distributed.run")
    # RVDD_BENCH_ONE_GPU=1: rehearsal of the multi-process path on a one-GPU box -- every rank on GPU 0,
    # collectives over gloo on host tensors.  Never a measurement.
    rehearsal = os.environ.get("RVDD_BENCH_ONE_GPU") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    rank, local_rank, world, dist = shard.init_distributed("gloo" if rehearsal else "nccl", dev)    # nccl = RCCL on ROCm
    coll_dev = None if rehearsal else dev

    B = args.batch
    sd = load_file(os.path.join(REPO, "weights", stem + ".safetensors"))
    rt = RvddRuntime(arch, fut, B, H, W, dev_index)
    rt.load_state_dict(sd)

    # ---- synthetic inputs, resident in HBM, [T,B,...] so that a time slice is contiguous
    cfg_id = int(args.config[1])
    my_seqs = shard.shard_sequences(B * world, rank, world)   # weak scaling: B sequences per GPU
    seqs = [synth.make_sequence(T, H, W, iso=iso, seed=1000 * cfg_id + sid, device=str(dev)) for sid in my_seqs]
    raw = torch.stack([s.raw for s in seqs], 1).contiguous()
    fprev = torch.stack([s.flow_prev for s in seqs], 1).contiguous()
    fnext = torch.stack([s.flow_next for s in seqs], 1).contiguous()
    gt = torch.stack([s.gt for s in seqs], 1).contiguous()
    n_out = T - 1 - fut
    outs = torch.empty(n_out, B, 3, H, W, dtype=torch.float32, device=dev)

    def one_step():
        rt.reset()                                            # FirstOfVideo
        for t in range(1, T - fut):
            rt.step(raw[t - 1] if t == 1 else None, raw[t], raw[t + 1] if fut else None, fprev[t],
                    fnext[t] if fut else None, out=outs[t - 1])

    def barrier():
        shard.barrier(dist, dev)

    for _ in range(args.warmup):
        one_step()
    barrier()
    if not args.no_kernel_events:
        if args.all_kernel_events:
            rt.profile_select(None, 1)
        else:
            rt.profile_select(DOMINANT[arch], 4)
        rt.profile_enable(True)
    t0 = time.perf_counter()
    rt.timer_start()
    for _ in range(args.steps):
        one_step()
    ev_ms = rt.timer_stop_ms()
    barrier()
    wall = time.perf_counter() - t0
    prof = rt.profile_read() if not args.no_kernel_events else []
    rt.profile_enable(False)

    elapsed = shard.max_over_ranks(wall, dist, coll_dev)
    frames_total = args.steps * n_out * B * world
    fps = frames_total / elapsed

    # ---- task PSNR of every output frame of the last step (outside the timed region)
    psnr = torch.tensor([[rt.psnr_l1(outs[k], gt[k + 1])[1] for k in range(n_out)]], dtype=torch.float64, device=dev)
    psnr_mean = float(shard.gather_metrics(psnr.cpu() if rehearsal else psnr, dist).mean().item())    # the one collate collective

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HIP events around its launches, this rank)
    roofline = None
    kernels = {}
    for p in prof:
        if p["launches"]:
            kernels[p["name"]] = dict(launches=p["launches"], avg_us=1e3 * p["ms"] / p["launches"],
                                      total_ms=p["ms"], tflops=(p["flops"] / (p["ms"] * 1e9)) if p["ms"] else 0.0,
                                      gbps=(p["bytes"] / (p["ms"] * 1e6)) if p["ms"] else 0.0,
                                      bytes_per_launch=p["bytes"] / p["launches"])
    dom = DOMINANT[arch]
    if dom in kernels:
        k = kernels[dom]
        # HBM bytes per launch from the PMC passes (tools/gpu_profile.sh + tools/pmc_summary.py); counters
        # cannot be read from inside this process, so the last committed measurement is quoted
        traffic = None
        try:
            tj = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
            traffic = tj[args.config]["kernels"].get(dom)
        except Exception:
            pass
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(k["tflops"], 2), "peak": FP32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(k["tflops"] / FP32_PEAK_TFLOPS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": round(k["bytes_per_launch"]),
                    "launches": k["launches"], "avg_launch_us": round(k["avg_us"], 2),
                    "events": "every launch" if args.all_kernel_events else "every 4th launch"}
        if dom.startswith("wino3x3"):
            # `achieved` counts the ALGORITHMIC flops of the 3x3 conv (2*9*Cin*Cout per pixel).  The
            # Winograd F(2x2,3x3) kernel executes 16 instead of 36 multiplies per 2x2 outputs, so frac > 1 is
            # possible; the fraction of the MFMA peak the kernel actually issues is reported beside it.
            roofline["executed_mfma_tflops"] = round(k["tflops"] * 16.0 / 36.0, 2)
            roofline["executed_mfma_frac"] = round(k["tflops"] * 16.0 / 36.0 / FP32_PEAK_TFLOPS, 4)

    # ---- CPU baseline: the oracle on this host's cores, bounded sample, sequence 0
    cpu = None
    if world == 1 and args.cpu_frames > 0:
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        import rvdd_oracle as O
        cores = os.cpu_count() or 1
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            pass
        # a 1-GPU box grants a 16-CPU share of a larger host: more threads than the share only
        # oversubscribe it (256 threads measured 40x slower than 16)
        cores = int(os.environ.get("RVDD_CPU_THREADS", min(cores, 16)))
        torch.set_num_threads(cores)
        orc = O.RecurrentOracle(sd, future=fut)
        r0, p0, n0 = raw[:, 0].cpu(), fprev[:, 0].cpu(), fnext[:, 0].cpu()
        nf = min(args.cpu_frames, n_out)
        times, worst, ppsnr = [], 0.0, 1e9
        for t in range(1, 1 + nf):
            tc = time.perf_counter()
            den = orc.step(r0[t - 1][None], r0[t][None], r0[t + 1][None] if fut else None, p0[t][None],
                           n0[t][None] if fut else None, first=(t == 1))
            times.append(time.perf_counter() - tc)
            g = outs[t - 1, 0].cpu()
            d = (g - den[0]).double()
            worst = max(worst, float(d.abs().max()))
            mse = float((d * d).mean())
            ppsnr = min(ppsnr, 200.0 if mse == 0 else 10 * torch.log10(torch.tensor(4.0 / mse)).item())
        timed = times[1:] if len(times) > 1 else times        # first frame = warm-up
        cpu_fps = len(timed) / sum(timed)
        cpu = {"value": round(cpu_fps, 4), "unit": "frames/s", "cores": cores, "kind": "port",
               "sample": f"{len(timed)} frame(s) of sequence 0 of the same workload after 1 warm-up frame, "
                         f"torch {torch.__version__} CPU ops, {cores} threads",
               "gpu_vs_cpu_max_abs_diff": worst, "gpu_vs_cpu_parity_psnr_db": round(ppsnr, 2)}

    line = {
        "metric": "frames/sec (whole job), recurrent video denoise+demosaic inference", "value": round(fps, 3),
        "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, not a measurement)" if rehearsal else ""),
        "config": {"workload": f"{args.config}: {DESCR[args.config]}" + (f" (run with --frames {T})" if args.frames else ""),
                   "arch": arch, "checkpoint": stem,
                   "frame": f"{W}x{H}", "frames_per_sequence": T, "sequences_per_gpu": B,
                   "output_frames_per_step_per_gpu": n_out * B, "parallelism": f"sequences sharded over {world} GPU(s)"},
        "fps_per_gpu": round(fps / world, 3), "ms_per_frame": round(1e3 * elapsed / (frames_total / world), 3),
        "gpu_event_ms_per_step": round(ev_ms / args.steps, 3),
        "algorithmic_gflop_per_frame": gflop_frame,
        "whole_path_tflops": round(fps / world * gflop_frame * (H * W) / (CONFIGS[args.config][4] * CONFIGS[args.config][5]) / 1e3, 2),
        "whole_path_frac_of_fp32_peak": round(fps / world * gflop_frame / 1e3 / FP32_PEAK_TFLOPS, 4),
        "task_psnr_db": round(psnr_mean, 3),
        "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
    }
    if cpu:
        line["gpu_over_cpu"] = round(fps / cpu["value"], 1)
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
