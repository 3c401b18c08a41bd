#!/usr/bin/env python3
"""Headline benchmark: output frames/s of the recurrent denoise+demosaic hot
path on N MI355X (one process per GPU), BASELINE.json's metric.

  python bench.py                                  # N=1, config C2, finishes in minutes
  python bench.py --gpus N --steps K --warmup W    # launches its own N ranks (torch.distributed.run child)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W          # or is launched as one of them

A "step" is one pass of the hot path over one batch of synthetic input already
resident in HBM: B independent sequences of T frames advanced in lockstep on
each GPU (T-1-future output frames per sequence).  Sequences share nothing, so
ranks never exchange data inside the timed region; the only collectives are the
barrier/max that bracket it, one all-gather of the per-frame PSNR values
afterwards and, with --collate-outputs, one all-gather of the output frames
(timed on its own, outside the compute region).

  N = 1   config C2 (the configuration the metric is quoted on), B = 8 sequences in lockstep
  N > 1   config C5 (B = 8 sequences of 90 frames per GPU: 64 sequences on 8 GPUs), weak scaling;
          --scaling strong fixes the total at --sequences (64) instead and runs each rank's share in groups of B

The JSON line carries `roofline` for the dominant kernel (the 48->48 3x3 conv; by
default on the F16 matrix pipe with split f32 operands, dense peak 2.5 PFLOP/s),
measured with HIP events around a uniform sample of its launches inside the timed
region -- `achieved` / `frac` are ALGORITHMIC flops, the executed-MFMA
and HBM fractions sit beside them --, `cpu_baseline`: the CPU oracle
(oracle/rvdd_oracle.py, a torch-CPU restatement of the reference's PyTorch path)
timed on this host's cores on a bounded sample of the same workload, and
`other_configs`: a short run of each of BASELINE.json's other configurations (and of C2 with the flow recomputed online by TV-L1)
(C3, C4, C5's per-GPU share, C1) after the timed region, so that the driver's
record holds them too.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

CONFIGS = {
    # name: (arch, weights stem, future, iso, H, W, T, default B, algorithmic GFLOP per output frame @ HxW)
    "C1": ("convunet", "recurrent-convunet-iso3200", 0, 3200, 256, 256, 8, 8, 25.50),
    "C2": ("convunet+feat", "recurrent-convunet+feat-iso3200", 0, 3200, 720, 1280, 30, 8, 435.025),
    "C3": ("convunet+feat", "recurrent-convunet+feat-future-iso12800", 1, 12800, 720, 1280, 30, 8, 437.413),
    "C4": ("next+feat", "recurrent-ConvNeXtUnet+feat-future-iso3200", 1, 3200, 720, 1280, 30, 8, 401.998),
    "C5": ("convunet+feat", "recurrent-convunet+feat-iso3200", 0, 3200, 720, 1280, 90, 8, 435.025),
}
DESCR = {
    "C1": "RVDD-basic (recurrent convunet) ISO3200 256x256 8-frame sequences",
    "C2": "recurrent convunet+feat ISO3200 1280x720 30-frame sequences",
    "C3": "recurrent convunet+feat+future ISO12800 1280x720 30-frame sequences",
    "C4": "recurrent ConvNeXtUnet+feat+future ISO3200 1280x720 30-frame sequences",
    "C5": "recurrent convunet+feat ISO3200 1280x720 90-frame sequences, 8 per GPU (64 on 8 GPUs)",
}
FP32_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, chip table (vector = matrix f32)
F16_PEAK_TFLOPS = 2500.0          # same table: dense F16 / BF16 MFMA (~2.5 PF; the 5 PF figure is 2:1 sparsity)
# what the F16 matrix pipe SUSTAINS on this part under its package power limit: nothing but independent dense
# v_mfma_f32_16x16x32_f16 on every SIMD for 3 s (tools/mfma_f16_shapes_power.hip, profiles/r04_mfma_f16_shapes_power.txt:
# 1 926 TFLOP/s at ~1.85 GHz; the nominal 2.5 PFLOP/s needs 2.4 GHz, which the package does not hold with the pipe full)
F16_SUSTAINED_TFLOPS = 1926.0
F16_SUSTAINED_SOURCE = "profiles/r04_mfma_f16_shapes_power.txt (tools/mfma_f16_shapes_power.hip: dense 16x16x32 f16 MFMAs only, 3 s, package power limit)"
# the convunet's plain 48 -> 48 3x3 conv: the split-f16 kernel (default), or the f32-MFMA kernels (RVDD_CONV=f32 | winograd | direct)
_CONV = {"direct": "conv3x3_kernel<48, 1, false>", "winograd": "wino3x3_kernel<1, false>",
         "f32": "wino3x3_kernel<1, false>"}.get(os.environ.get("RVDD_CONV", ""), "conv3x3h_kernel<48, 1, false, false>")
_NEXT = "convblock_kernel"
DOMINANT = {"convunet": _CONV, "convunet+feat": _CONV, "next": _NEXT, "next+feat": _NEXT}
# The launches of a kernel class inside one frame-step repeat with period 11 (plain 48->48 3x3 conv; 14 when the
# upsample is not fused) or 25 (ConvNeXt ConvBlock) over the four resolution levels; bracketing every 3rd launch (3 is
# coprime with all of them) samples every position equally often.
EVENT_STRIDE = 3
# MFMA flops a kernel EXECUTES per algorithmic (direct 3x3 conv) flop: Winograd F(2x2,3x3) multiplies 16 times per
# 2x2 outputs where the direct form multiplies 36 times
EXECUTED_PER_ALGORITHMIC = {"wino3x3": 16.0 / 36.0,
                            # split-f16 kernel: three F16 MFMAs (hi.hi, hi.lo, lo.hi) per product, K = 432 padded to 448
                            "conv3x3h": 3.0 * 448.0 / 432.0,
                            # fused ConvBlock, split-f16 MLP: 114 F16 MFMAs (16x16x32) per 16 pixels = 116,736 flop per pixel
                            # against 41,568 algorithmic (depth-wise 4,704 on the vector ALU + 36,864 in the two 1x1 convs)
                            "convblock_kernel": 116736.0 / 41568.0}


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="default: C2 on one GPU, C5 on several")
    ap.add_argument("--batch", type=int, default=0, help="sequences advanced in lockstep per GPU (default: the config's)")
    ap.add_argument("--frames", type=int, default=0, help="override frames per sequence")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch sequences per GPU; strong: --sequences in total, split over the GPUs")
    ap.add_argument("--sequences", type=int, default=64, help="total sequences with --scaling strong")
    ap.add_argument("--online-flow", action="store_true",
                    help="validate.py's --val_flow_from_denoised at full size (validate.py:16-38, 81-82): the flow towards the "
                         "previous frame of every step after the first is TV-L1 (rvdd_tvl1flow_batch) from the current noisy "
                         "frame to the re-mosaicked previous OUTPUT, inside the timed region; a mode of its own, labelled in "
                         "the JSON line, never the headline metric")
    ap.add_argument("--online-groups", type=int, default=ONLINE_GROUPS,
                    help="--online-flow: the sequences of a GPU as this many independent groups on as many HIP streams (1 = round 5's single stream)")
    ap.add_argument("--collate-outputs", action="store_true",
                    help="after the timed region, all-gather the output frames of the last group (timed separately)")
    ap.add_argument("--cpu-frames", type=int, default=10, help="timed frames of the CPU-oracle sample (0 = skip)")
    ap.add_argument("--cpu-frames-8", type=int, default=4, help="timed frames of the second CPU sample at 8 threads (0 = skip)")
    ap.add_argument("--cpu-frames-wide", type=int, default=3,
                    help="timed frames of a third CPU sample at min(affinity, cgroup quota, 64) threads, when that exceeds 16 (0 = skip)")
    ap.add_argument("--no-exact-ab", action="store_true",
                    help="skip the second, untimed-region-external run on the exact-f32-product kernels (one GPU only)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short runs of the other configurations after the timed region (one GPU, default config only)")
    ap.add_argument("--other-steps", type=int, default=6,
                    help="timed steps of each of those short runs (six: 2-6 s each, enough to resolve 1 %%; round 5 ran two)")
    ap.add_argument("--no-gpu-sampler", action="store_true", help="do not sample shader clock / package power (rocm-smi) during the timed regions")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--all-kernel-events", action="store_true",
                    help="diagnostic: bracket EVERY launch of every kernel (costs ~6 %% of the frame rate); "
                         f"default brackets every {EVENT_STRIDE}rd launch of the dominant kernel only")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a CHILD process group
    (nothing in this process has touched the GPU), relay their output, return their exit code."""
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith("ROCPROF") for k in os.environ):
        # under rocprofv3 the profiler's preloaded library has ALREADY initialised the GPU in this process (with --pmc
        # it has): starting a launcher from here is a fork + exec behind an initialised GPU, which takes GPU boxes down
        raise SystemExit("bench.py --gpus N cannot start its own ranks under rocprofv3: profile one rank "
                         "(`rocprofv3 ... -- python3 bench.py --config C5`), or put the profiler inside the launcher")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


class StubRuntime:
    """RVDD_BENCH_STUB=1: a stand-in for the HIP runtime on the CPU so that tests/test_shard_gloo.py can drive this
    file's launcher, sharding, collectives and JSON line without a GPU.  Never a measurement, and says so."""

    def __init__(self, B, H, W):
        self.B, self.H, self.W = B, H, W
        self._t0 = 0.0

    def load_state_dict(self, sd):
        pass

    def reset(self):
        pass

    def step(self, raw_prev, raw_cur, raw_next, flow_prev, flow_next, out=None):
        import torch
        up = torch.nn.functional.interpolate(raw_cur[:, 1:4], scale_factor=2, mode="nearest")
        out.copy_(up)
        return out

    def psnr_l1(self, den, gt):
        import math
        d = (den.double() - gt.double())
        return float(100 * d.abs().mean()), 10 * math.log10(4.0 / max(float((d * d).mean()), 1e-30))

    def timer_start(self):
        self._t0 = time.perf_counter()

    def timer_stop_ms(self):
        return 1e3 * (time.perf_counter() - self._t0)

    def profile_select(self, *a):
        pass

    def profile_enable(self, on):
        pass

    def profile_read(self):
        return []


def host_cpu_share():
    """What this process may use of the host's CPUs: affinity mask and cgroup quota (the evidence behind `cores`)."""
    info = {"host_cpus": os.cpu_count()}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        info["affinity"] = None
    info["cgroup_cpu_max"] = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().strip()
        except OSError:
            continue
        info["cgroup_cpu_max"] = f"{path}: {txt}"
        try:
            if path.endswith("cpu.max"):
                q, per = txt.split()
                info["cgroup_cpus"] = None if q == "max" else round(int(q) / int(per), 2)
            else:
                q = int(txt)
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                info["cgroup_cpus"] = None if q < 0 else round(q / per, 2)
        except Exception:
            info["cgroup_cpus"] = None
        break
    try:
        info["loadavg_1min"] = os.getloadavg()[0]
    except OSError:
        pass
    return info


class GpuSampler:
    """Shader clock and package power of one GPU, sampled by a thread of this process while a timed region runs, so that the
    driver's record can tell a slow box from a regression (the boxes of this pool hold 1.86-1.98 GHz under the same load).
    Source: `rocm-smi -d <dev> --showpower --showclocks` every `period` s (the tool tools/power_probe.sh has used since round 2;
    a child process that reads sysfs -- it launches nothing on the GPU).  Never raises: no tool, no numbers."""

    def __init__(self, dev_index=0, period=0.4):
        import threading
        self.dev, self.period = dev_index, period
        self.samples = []                       # (perf_counter, sclk MHz or None, W or None)
        # the child must not inherit a profiler's preload (it would initialise the GPU in rocm-smi, whose `env python3` hop is then refused)
        self._env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF"))}
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self.error = None
        self._thread.start()

    def _run(self):
        import re
        while not self._stop.is_set():
            t = time.perf_counter()
            try:
                txt = subprocess.run(["rocm-smi", "-d", str(self.dev), "--showpower", "--showclocks"], capture_output=True,
                                     text=True, timeout=10, env=self._env).stdout
                mc = re.search(r"sclk[^\n]*?\((\d+)Mhz\)", txt)
                mp = re.search(r"Power \(W\):\s*([\d.]+)", txt)
                self.samples.append((0.5 * (t + time.perf_counter()), int(mc.group(1)) if mc else None, float(mp.group(1)) if mp else None))
            except Exception as e:             # rocm-smi absent or hung: record why, stop sampling
                self.error = f"{type(e).__name__}: {e}"
                return
            self._stop.wait(max(0.0, self.period - (time.perf_counter() - t)))

    def window(self, t0, t1):
        """Summary of the samples taken inside [t0, t1] (perf_counter times)."""
        rows = [r for r in self.samples if t0 <= r[0] <= t1]
        clk = [r[1] for r in rows if r[1] is not None]
        pw = [r[2] for r in rows if r[2] is not None]
        out = {"samples": len(rows), "source": "rocm-smi --showpower --showclocks, a thread of this process, inside the timed region only"}
        if clk:
            out.update(sclk_mhz_mean=round(sum(clk) / len(clk), 1), sclk_mhz_min=min(clk), sclk_mhz_max=max(clk))
        if pw:
            out.update(package_w_mean=round(sum(pw) / len(pw), 1), package_w_max=max(pw))
        if self.error:
            out["error"] = self.error
        return out

    def close(self):
        self._stop.set()
        self._thread.join(timeout=15)


def flow_from_denoised(rt, den, raw_cur):
    """validate.py:16-38 for B sequences: moving = channel mean of remosaick(previous output), target = channel mean of
    the current packed raw frame, both mapped to [0,1] (library.py:67-68, :165-167) -> TV-L1 on the device."""
    moving = ((den[:, 1, 0::2, 0::2] + den[:, 2, 0::2, 1::2]) + (den[:, 0, 1::2, 0::2] + den[:, 1, 1::2, 1::2])) * 0.125 + 0.5
    target = raw_cur.mean(dim=1) * 0.5 + 0.5
    return rt.tvl1flow_batch(target.contiguous(), moving.contiguous())


def advance(rt, raw, fprev, fnext, outs, T, fut, online_flow):
    """One group of B sequences in lockstep through all its frames: the loop of validate.py:73-88 (FirstOfVideo reset, then
    set_input / test per frame), output frame t - 1 into outs[t - 1].  THE step of every run of this file, timed region or not."""
    rt.reset()
    for t in range(1, T - fut):
        fp = fprev[t]
        if online_flow and t > 1:
            fp = flow_from_denoised(rt, outs[t - 2], raw[t])
        rt.step(raw[t - 1] if t == 1 else None, raw[t], raw[t + 1] if fut else None, fp, fnext[t] if fut else None, out=outs[t - 1])


ONLINE_GROUPS = 2      # the online-flow mode's default: the B sequences as two groups of B / 2 on two HIP streams (see OnlineGroups)


class OnlineGroups:
    """The online-flow loop with the B sequences as G independent groups of B / G, each with its own runtime handle and HIP
    stream.  A group's chain is serial (output t-1 -> TV-L1 -> step t) but the groups share nothing, so one group's
    latency-bound coarse TV-L1 scales run beside another group's convolutions.  Bit-identical outputs to G = 1 (a sequence's
    arithmetic does not depend on its batch: tools/online_overlap_probe.py, profiles/r05ad_online_flow_two_streams.txt:
    G = 2 +4.6 %, G = 4 -13 %).  Every group's work of a frame is enqueued before the next frame's, the host never waits."""

    def __init__(self, make_rt, B, G, dev):
        import torch
        if B % G:
            raise SystemExit(f"--online-groups {G} does not divide the batch {B}")
        self.G, self.per, self.dev = G, B // G, dev
        self.rts = [make_rt(self.per) for _ in range(G)]
        for rt in self.rts:
            rt.set_option("tvl1_async", 1)       # the flow batch stays on the stream: no host round trip per frame
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(G)]

    def advance(self, raw, fprev, outs, T):
        """raw [T,B,4,h,w], fprev [T,B,2,h,w], outs [T-1,B,3,H,W]: one pass over all frames (advance()'s loop, per group)."""
        import torch
        cur = torch.cuda.current_stream(self.dev)
        for st in self.streams:
            st.wait_stream(cur)
        for rt in self.rts:
            rt.reset()
        for t in range(1, T):
            for g, (rt, st) in enumerate(zip(self.rts, self.streams)):
                sl = slice(g * self.per, (g + 1) * self.per)
                with torch.cuda.stream(st):
                    fp = fprev[t, sl] if t == 1 else flow_from_denoised(rt, outs[t - 2, sl], raw[t, sl])
                    rt.step(raw[t - 1, sl] if t == 1 else None, raw[t, sl], None, fp, None, out=outs[t - 1, sl])
        for st in self.streams:
            cur.wait_stream(st)

    def close(self):
        import torch
        for rt in self.rts:
            rt.psnr_l1(torch.zeros(1, 3, 2, 2, device=self.dev), torch.zeros(1, 3, 2, 2, device=self.dev))   # reads the asynchronous batches' control word
            rt.close()


def csrc_hash():
    """sha256[:16] over the kernel sources (what tools/pmc_summary.py stamps profiles/traffic.json with)."""
    import glob
    import hashlib
    hs = hashlib.sha256()
    d = os.path.join(REPO, "rvdd-release_amd", "csrc")
    for fn in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.inc")) + glob.glob(os.path.join(d, "*.h"))):
        hs.update(open(fn, "rb").read())
    return hs.hexdigest()[:16]


def roofline_of(dom, k, config, arch, events_note):
    """The `roofline` object of a kernel class from its event-timed launches (k: launches, avg_us, tflops, gbps, bytes_per_launch)."""
    # HBM bytes per launch from the PMC passes of the same command (tools/gpu_profile.sh + tools/pmc_summary.py:
    # average over ALL launches of the kernel in a frame-step, like the event sample); counters cannot be read
    # from inside this process, so the last committed measurement is quoted together with its source file
    traffic, traffic_src = None, None
    try:
        tj = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
        ent = tj.get(config) or tj.get("C2" if config in ("C5", "C3") else config)      # C3 / C5: C2's maps and launch mix
        traffic, traffic_src = ent["kernels"].get(dom), ent.get("source")
        # a committed measurement is quoted only for the kernel sources it was taken on: after a kernel change without a
        # re-profile (tools/gpu_profile.sh + tools/pmc_summary.py) the line says so instead of quoting stale bytes
        if ent.get("csrc_sha256_16") != csrc_hash():
            traffic_src = f"{traffic_src} -- STALE: measured on csrc {ent.get('csrc_sha256_16')}, this build is {csrc_hash()}; traffic withheld"
            traffic = None
    except Exception:
        pass
    factor = next((f for pre, f in EXECUTED_PER_ALGORITHMIC.items() if dom.startswith(pre)), 1.0)
    executed = k["tflops"] * factor
    split = dom.startswith("conv3x3h") or (dom == "convblock_kernel" and os.environ.get("RVDD_NEXT_SPLIT") != "0")
    if dom == "convblock_kernel" and not split:
        factor, executed = 1.0, k["tflops"]
    peak = F16_PEAK_TFLOPS if split else FP32_PEAK_TFLOPS
    # `achieved` / `frac`: ALGORITHMIC flops of the layer (the direct conv's 2*9*Cin*Cout per pixel) over the
    # measured launch time -- the contract's convention; `frac_executed_mfma` counts the MFMA flops the kernel issues
    # (split products, K padding; Winograd executes fewer than algorithmic) and is the matrix pipe's utilisation;
    # `frac_hbm` is the same launch against the 8 TB/s HBM roof
    return {"bound": "mfma", "kernel": dom, "achieved": round(k["tflops"], 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(k["tflops"] / peak, 4),
            "frac_algorithmic": round(k["tflops"] / peak, 4),
            "executed_mfma_tflops": round(executed, 2), "frac_executed_mfma": round(executed / peak, 4),
            # the same against what the matrix pipe sustains under the package power limit (F16 forms only)
            "peak_sustained": F16_SUSTAINED_TFLOPS if split else None,
            "peak_sustained_source": F16_SUSTAINED_SOURCE if split else None,
            "frac_executed_sustained": round(executed / F16_SUSTAINED_TFLOPS, 4) if split else None,
            "frac_hbm": round(k["gbps"] / 8000.0, 4), "hbm_peak_gbps": 8000.0,
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": round(k["bytes_per_launch"]),
            "mfma_dtype": "f16 (f32 operands split hi + lo, f32 accumulation)" if split else "f32",
            "hbm_gbps_algorithmic": round(k["gbps"], 1),
            "what_is_executed": ("F16 MFMA flops the kernel executes in its two 1x1 convs (114 MFMAs per 16 pixels; the depth-wise 7x7, "
                                 "LayerNorm and GELU run on the vector ALU and bound the kernel: DESIGN.md 4.4)")
                                if split and dom == "convblock_kernel" else
                                "F16 MFMA flops the kernel executes: 3 MFMAs per product (hi.hi, hi.lo, lo.hi), K 432 padded to 448"
                                if split else
                                "MFMA flops the kernel executes (Winograd F(2x2,3x3): 16/36 of the direct conv's)"
                                if factor != 1.0 else "the kernel's algorithmic flops (all executed on MFMA)",
            "algorithmic_equiv_tflops": round(k["tflops"], 2),
            "launches": k["launches"], "avg_launch_us": round(k["avg_us"], 2),
            "events": events_note}


def kernel_table(prof):
    kernels = {}
    for p in prof:
        if p["launches"]:
            kernels[p["name"]] = dict(launches=p["launches"], avg_us=1e3 * p["ms"] / p["launches"],
                                      total_ms=p["ms"], tflops=(p["flops"] / (p["ms"] * 1e9)) if p["ms"] else 0.0,
                                      gbps=(p["bytes"] / (p["ms"] * 1e6)) if p["ms"] else 0.0,
                                      bytes_per_launch=p["bytes"] / p["launches"])
    return kernels


def quick_config(name, steps, dev_index, online_flow=False, batch=None, sampler=None, online_groups=ONLINE_GROUPS):
    """A short run of another configuration (default: at its default batch): 1 warm-up step, `steps` timed steps (wall clock
    between device synchronisations) of the SAME loop the headline run times (advance), HIP events around every 3rd launch of
    its dominant kernel and that kernel's `roofline`.  Inputs synthetic, resident in HBM."""
    import torch
    from safetensors.torch import load_file
    from rvdd_release_amd import synth
    from rvdd_release_amd.runtime import RvddRuntime
    arch, stem, fut, iso, H, W, T, B, gflop = CONFIGS[name]
    B = batch or B
    dev = torch.device("cuda", dev_index)
    sd = load_file(os.path.join(REPO, "weights", stem + ".safetensors"))

    def make_rt(b):
        r = RvddRuntime(arch, fut, b, H, W, dev_index)
        r.load_state_dict(sd)
        return r
    og = None
    if online_flow and online_groups > 1 and not fut and B % online_groups == 0:
        og = OnlineGroups(make_rt, B, online_groups, dev)
        rt = og.rts[0]           # (metrics, and the event sample: group 0's launches, B / G sequences each)
    else:
        rt = make_rt(B)
        if online_flow:
            rt.set_option("tvl1_async", 1)       # the flow batch stays on the stream: no host round trip per frame
    seqs = [synth.make_sequence(T, H, W, iso=iso, seed=1000 * int(name[1]) + b, device=str(dev)) for b in range(B)]
    raw = torch.stack([s.raw for s in seqs], 1).contiguous()
    fprev = torch.stack([s.flow_prev for s in seqs], 1).contiguous()
    fnext = torch.stack([s.flow_next for s in seqs], 1).contiguous() if fut else None
    gt_last = torch.stack([s.gt[T - 1 - fut] for s in seqs], 0).contiguous()
    del seqs
    n_out = T - 1 - fut
    outs = torch.empty(n_out, B, 3, H, W, dtype=torch.float32, device=dev)

    def one_pass():
        if og is not None:
            og.advance(raw, fprev, outs, T)
        else:
            advance(rt, raw, fprev, fnext, outs, T, fut, online_flow)
    one_pass()
    torch.cuda.synchronize()
    rt.profile_select(DOMINANT[arch], EVENT_STRIDE)
    rt.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    el = t1 - t0
    kernels = kernel_table(rt.profile_read())
    rt.profile_enable(False)
    psnr_last = rt.psnr_l1(outs[n_out - 1], gt_last)[1]      # (also the call that reads an asynchronous flow batch's control word)
    finite = bool(torch.isfinite(outs).all())
    if og is not None:
        og.close()
    else:
        rt.close()
    res = {"workload": f"{name}: {DESCR[name]}" + (" -- with the flow towards the previous frame recomputed by TV-L1 from every previous "
                                                  "output inside the timed loop (validate.py --val_flow_from_denoised)" if online_flow else ""),
           "value": round(steps * n_out * B / el, 2), "unit": "frames/s",
           "ms_per_step": round(1e3 * el / steps, 3), "steps": steps, "warmup": 1, "sequences_in_lockstep": B,
           "output_frames_per_step": n_out * B, "finite": finite, "task_psnr_db_last_frame": round(psnr_last, 3)}
    if og is not None:
        res["online_groups"] = {"groups": og.G, "sequences_per_group": og.per,
                                "what": "independent groups of sequences on their own HIP streams; outputs bit-identical to one group; "
                                        "the dominant kernel's events are group 0's launches (B / G sequences each)"}
    if sampler is not None:
        res["gpu_clock_power"] = sampler.window(t0, t1)
    dom = DOMINANT[arch]
    if dom in kernels:
        res["dominant_kernel"] = dom
        res["avg_launch_us"] = round(kernels[dom]["avg_us"], 2)
        res["launches_sampled"] = kernels[dom]["launches"]
        res["roofline"] = roofline_of(dom, kernels[dom], name, arch, f"every {EVENT_STRIDE}rd launch (uniform over the launches of a frame-step)")
    del raw, fprev, fnext, outs
    torch.cuda.empty_cache()
    return res


def main():
    args = parse_args()
    # the host driver only supports dmabuf IPC: RCCL (and any sharing of device memory between processes) needs this
    # before the first HIP call of the process, whoever started it (torch.distributed.run directly, or self_launch)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: "
                         "run `python bench.py --gpus N` (it launches its own ranks) or use torch.distributed.run")

    import torch
    from rvdd_release_amd import shard, synth

    config = args.config or ("C2" if args.gpus == 1 else "C5")
    arch, stem, fut, iso, H, W, T, B_default, gflop_frame = CONFIGS[config]
    if args.frames:
        T = args.frames
    B = args.batch or B_default
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    stub = os.environ.get("RVDD_BENCH_STUB") == "1"
    # RVDD_BENCH_ONE_GPU=1: rehearsal of the multi-process path on a one-GPU box -- every rank on GPU 0,
    # collectives over gloo on host tensors.  Never a measurement.
    rehearsal = os.environ.get("RVDD_BENCH_ONE_GPU") == "1"
    if stub:
        dev = torch.device("cpu")
        rank, local_rank, world, dist = shard.init_distributed("gloo")
        coll_dev = None
        H, W = 32, 48
        sd = {}
        rt = StubRuntime(B, H, W)
    else:
        from safetensors.torch import load_file
        from rvdd_release_amd.runtime import RvddRuntime
        dev_index = 0 if rehearsal else local_rank
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        rank, local_rank, world, dist = shard.init_distributed("gloo" if rehearsal else "nccl", dev)  # nccl = RCCL on ROCm
        coll_dev = None if rehearsal else dev
        sd = load_file(os.path.join(REPO, "weights", stem + ".safetensors"))
        rt = RvddRuntime(arch, fut, B, H, W, dev_index)
        rt.load_state_dict(sd)
    on_host = stub or rehearsal

    # ---- which sequences this rank advances (static block partition, shard.py; no data-path collective)
    if args.scaling == "strong":
        total_seqs = args.sequences
        if total_seqs % (world * B):
            raise SystemExit(f"--sequences {total_seqs} must be a multiple of gpus x batch = {world * B}")
    else:
        total_seqs = B * world
    my_seqs = list(shard.shard_sequences(total_seqs, rank, world))
    groups = [my_seqs[i:i + B] for i in range(0, len(my_seqs), B)]     # one group = B sequences in lockstep

    # ---- synthetic inputs, resident in HBM, [T,B,...] so that a time slice is contiguous
    cfg_id = int(config[1])
    n_out = T - 1 - fut
    inputs = []
    gt = None
    for gi, grp in enumerate(groups):
        seqs = [synth.make_sequence(T, H, W, iso=iso, seed=1000 * cfg_id + sid, device=str(dev)) for sid in grp]
        raw = torch.stack([s.raw for s in seqs], 1).contiguous()
        fprev = torch.stack([s.flow_prev for s in seqs], 1).contiguous()
        fnext = torch.stack([s.flow_next for s in seqs], 1).contiguous() if fut else None
        inputs.append((raw, fprev, fnext))
        if gi == len(groups) - 1:
            gt = torch.stack([s.gt for s in seqs], 1).contiguous()     # ground truth of the last group (task PSNR)
        del seqs
    outs = torch.empty(n_out, B, 3, H, W, dtype=torch.float32, device=dev)   # outputs of the group being advanced

    og = None
    if args.online_flow and not stub:
        rt.set_option("tvl1_async", 1)       # the flow batch stays on the stream: no host round trip per frame
        if args.online_groups > 1 and not fut and B % args.online_groups == 0:
            def make_rt(b):
                r = RvddRuntime(arch, fut, b, H, W, dev_index)
                r.load_state_dict(sd)
                return r
            og = OnlineGroups(make_rt, B, args.online_groups, dev)

    def one_step(rt=rt, outs=outs):
        for raw, fprev, fnext in inputs:
            if og is not None and rt is og_main:
                og.advance(raw, fprev, outs, T)
            else:
                advance(rt, raw, fprev, fnext, outs, T, fut, args.online_flow)
    og_main = rt

    def barrier():
        shard.barrier(dist, None if on_host else dev)

    for _ in range(args.warmup):
        one_step()
    # shader clock and package power through every timed region of this process (rank 0 of a single-node run samples its own GPU)
    under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith("ROCPROF") for k in os.environ)
    sampler = GpuSampler(dev_index) if (rank == 0 and not stub and not args.no_gpu_sampler and not under_profiler) else None
    barrier()
    if not args.no_kernel_events:
        if args.all_kernel_events:
            rt.profile_select(None, 1)
        else:
            rt.profile_select(DOMINANT[arch], EVENT_STRIDE)
        rt.profile_enable(True)
    t0 = time.perf_counter()
    rt.timer_start()
    for _ in range(args.steps):
        one_step()
    ev_ms = rt.timer_stop_ms()
    barrier()
    t1 = time.perf_counter()
    wall = t1 - t0
    clock_power = sampler.window(t0, t1) if sampler is not None else None
    prof = rt.profile_read() if not args.no_kernel_events else []
    rt.profile_enable(False)
    if og is not None:
        prof = []                # the launches ran on the groups' handles, not on `rt`

    elapsed = shard.max_over_ranks(wall, dist, coll_dev)
    n_ranks_seen = shard.count_ranks(dist, coll_dev)             # from the collective itself, not from the environment
    # which library ran those collectives, on tensors living where (a plain `python bench.py` has no group at all)
    distributed = {"process_group": dist is not None,
                   "backend": dist.get_backend() if dist is not None else None,
                   "collective_tensors_on": str(coll_dev) if coll_dev is not None else "cpu",
                   "n_ranks_seen_from": "all_reduce(SUM) of ones" if dist is not None else "no collective (single process)",
                   "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    frames_total = args.steps * n_out * total_seqs
    fps = frames_total / elapsed

    # ---- task PSNR of every output frame of the last group of the last step (outside the timed region)
    psnr = torch.tensor([[rt.psnr_l1(outs[k], gt[k + 1])[1] for k in range(n_out)]], dtype=torch.float64, device=dev)
    psnr_mean = float(shard.gather_metrics(psnr.cpu() if on_host else psnr, dist).mean().item())    # the collate collective

    # ---- optional: collate the output frames themselves (north star: "one all-gather only to collate outputs")
    collate = None
    if args.collate_outputs:
        src = outs.cpu() if rehearsal else outs
        barrier()
        tc = time.perf_counter()
        gathered = shard.gather_outputs(src, dist)
        barrier()
        dt = time.perf_counter() - tc
        nbytes = src.numel() * 4
        collate = {"ms": round(1e3 * dt, 3), "bytes_per_rank": nbytes, "gathered_shape": list(gathered.shape),
                   "recv_GBps_per_rank": round(nbytes * (world - 1) / dt / 1e9, 2) if world > 1 else None,
                   "what": "all-gather of the output frames of the last group, outside the timed region"}
        del gathered

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    if sampler is not None and (world > 1 or args.no_other_configs or args.online_flow or config != "C2" or args.batch or args.frames):
        sampler.close()          # nothing else is timed in this process
        sampler = None

    # ---- roofline of the dominant kernel (HIP events around a uniform sample of its launches, this rank)
    roofline = None
    kernels = kernel_table(prof)
    dom = DOMINANT[arch]
    if dom in kernels:
        roofline = roofline_of(dom, kernels[dom], config, arch, "every launch" if args.all_kernel_events else
                               f"every {EVENT_STRIDE}rd launch (uniform over the launches of a frame-step)")

    # ---- the same workload on the kernels that multiply f32 operands directly (f32 MFMA): what the split-f16 matrix path
    # buys, and how far apart the two paths' frames are.  One GPU only, outside the timed region.
    exact = None
    if world == 1 and not stub and not args.no_exact_ab and not args.online_flow:
        rt2 = RvddRuntime(arch, fut, B, H, W, dev_index)
        rt2.set_option("next_split" if arch.startswith("next") else "conv_kernel", 0 if arch.startswith("next") else 4)
        rt2.load_state_dict(sd)
        outs2 = torch.empty_like(outs)
        one_step(rt2, outs2)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        one_step(rt2, outs2)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t2
        d = (outs2 - outs).abs()
        exact = {"value": round(n_out * len(my_seqs) / el2, 3), "unit": "frames/s", "steps": 1,
                 "kernels": "RVDD_NEXT_SPLIT=0 (f32-MFMA 1x1 convs)" if arch.startswith("next") else "RVDD_CONV=f32 (f32-MFMA Winograd / direct 3x3 convs)",
                 "max_abs_diff_vs_default": float(d.max()), "mean_abs_diff_vs_default": float(d.mean()),
                 "what": "every output frame of the last group of sequences, default (split-f16 matrix path) against exact-f32 products"}
        rt2.close()
        del outs2, d

    # ---- the other configurations of BASELINE.json, a short run each (outside the timed region): C3, C4, C5's per-GPU
    # share and C1 at their default batch, so that the driver's record holds more than the headline config
    other = None
    if world == 1 and not stub and not rehearsal and not args.no_other_configs and not args.online_flow and config == "C2" \
            and not args.batch and not args.frames:
        other = {}
        for name in ("C3", "C4", "C5", "C1"):
            try:
                other[name] = quick_config(name, args.other_steps, dev_index, sampler=sampler)
            except Exception as e:          # a failure here must not take the headline line with it
                other[name] = {"error": f"{type(e).__name__}: {e}"}
        # C1 as BASELINE.json states it: ONE sequence of 8 frames (the default batch above runs eight in lockstep)
        for key, kw in (("C1_one_sequence", dict(name="C1", batch=1)), ("C2_online_flow", dict(name="C2", online_flow=True))):
            try:
                other[key] = quick_config(steps=args.other_steps * (8 if key.startswith("C1") else 1), dev_index=dev_index, sampler=sampler, **kw)
            except Exception as e:
                other[key] = {"error": f"{type(e).__name__}: {e}"}
    if sampler is not None:
        sampler.close()

    # ---- CPU baseline: the oracle on this host's cores, bounded sample, sequence 0
    cpu = None
    if world == 1 and args.cpu_frames > 0 and not stub:
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        import rvdd_oracle as O
        share = host_cpu_share()
        cores = share["affinity"] or os.cpu_count() or 1
        if share.get("cgroup_cpus"):
            cores = min(cores, max(1, int(share["cgroup_cpus"])))
        # a 1-GPU box grants a 16-CPU share of a larger host: more threads than the share only
        # oversubscribe it (256 threads measured 40x slower than 16); `host_cpu_share` is the evidence
        cores = int(os.environ.get("RVDD_CPU_THREADS", min(cores, 16)))
        raw0, fprev0, fnext0 = inputs[-1]
        r0, p0 = raw0[:, 0].cpu(), fprev0[:, 0].cpu()
        n0 = fnext0[:, 0].cpu() if fut else None

        def cpu_sample(threads, warm, timed, check):
            torch.set_num_threads(threads)
            orc = O.RecurrentOracle(sd, future=fut)
            nf = min(warm + timed, n_out)
            times, worst, ppsnr = [], 0.0, 1e9
            task_cpu, task_gpu = [], []
            for t in range(1, 1 + nf):
                tc = time.perf_counter()
                den = orc.step(r0[t - 1][None], r0[t][None], r0[t + 1][None] if fut else None, p0[t][None],
                               n0[t][None] if fut else None, first=(t == 1))
                times.append(time.perf_counter() - tc)
                if check:
                    d = (outs[t - 1, 0].cpu() - den[0]).double()
                    worst = max(worst, float(d.abs().max()))
                    mse = float((d * d).mean())
                    ppsnr = min(ppsnr, 200.0 if mse == 0 else 10 * torch.log10(torch.tensor(4.0 / mse)).item())
                    # task PSNR (models/recurrent_model.py:512-525) of the SAME frames from both sides: the north
                    # star's 0.01 dB bar is on this difference
                    g0 = gt[t, 0].cpu()[None]
                    task_cpu.append(O.psnr(den, g0))
                    task_gpu.append(O.psnr(outs[t - 1, 0].cpu()[None], g0))
            tt = times[warm:] if len(times) > warm else times
            return len(tt) / sum(tt), len(tt), worst, ppsnr, task_cpu, task_gpu

        v, n_t, worst, ppsnr, task_cpu, task_gpu = cpu_sample(cores, 2, args.cpu_frames, True)
        cpu = {"value": round(v, 4), "unit": "frames/s", "cores": cores, "kind": "port",
               "sample": f"{n_t} frame(s) of sequence 0 of the same workload after 2 warm-up frames, B = 1, "
                         f"torch {torch.__version__} CPU ops, {cores} threads of {os.cpu_count()} host CPUs",
               "host_cpu_share": share,
               "gpu_vs_cpu_max_abs_diff": worst, "gpu_vs_cpu_parity_psnr_db": round(ppsnr, 2),
               "task_psnr_db_cpu": round(sum(task_cpu) / len(task_cpu), 4),
               "task_psnr_db_gpu_same_frames": round(sum(task_gpu) / len(task_gpu), 4),
               "task_psnr_max_abs_diff_db": round(max(abs(a - b) for a, b in zip(task_cpu, task_gpu)), 6)}
        if args.cpu_frames_8 > 0 and cores > 8:
            v8, n8 = cpu_sample(8, 1, args.cpu_frames_8, False)[:2]
            cpu["value_8_threads"] = round(v8, 4)
            cpu["sample_8_threads"] = f"{n8} frame(s) after 1 warm-up frame, 8 threads (BASELINE.md section 2 used 8)"
        wide = min(share["affinity"] or 0, 64)
        if share.get("cgroup_cpus"):
            wide = min(wide, int(share["cgroup_cpus"]))
        if args.cpu_frames_wide > 0 and wide > cores:
            vw, nw = cpu_sample(wide, 1, args.cpu_frames_wide, False)[:2]
            cpu["value_wide"] = round(vw, 4)
            cpu["sample_wide"] = f"{nw} frame(s) after 1 warm-up frame, {wide} threads = min(affinity, cgroup quota, 64)"

    par = (f"{total_seqs} sequences sharded over {world} GPU(s), {len(groups)} group(s) of {B} in lockstep per GPU"
           if args.scaling == "strong" else f"sequences sharded over {world} GPU(s), {B} per GPU in lockstep")
    data = "synthetic"
    if args.online_flow:
        data += " (ONLINE FLOW: TV-L1 from the previous output inside the timed region, validate.py --val_flow_from_denoised; not the headline metric"
        data += (f"; the {B} sequences as {og.G} independent groups on {og.G} HIP streams)" if og is not None else ")")
    if rehearsal:
        data += " (REHEARSAL: all ranks on one GPU, not a measurement)"
    if stub:
        data += " (STUB: CPU stand-in for the HIP runtime, launcher/collective test only, not a measurement)"
    split_path = (_CONV.startswith("conv3x3h") if not arch.startswith("next") else
                  os.environ.get("RVDD_NEXT_SPLIT") != "0")
    line = {
        "metric": "frames/sec (whole job), recurrent video denoise+demosaic inference", "value": round(fps, 3),
        "unit": "frames/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "distributed": distributed, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32 (products as 3 split-f16 MFMAs, f32 accumulation)" if split_path else "f32", "data": data,
        "arithmetic": ("f32 in, f32 out, f32 accumulation; the convunet's 48-channel 3x3 convs multiply on the F16 matrix pipe with "
                       "every f32 operand split into two f16 halves (3 MFMAs per product): as close to the reference as the "
                       "f32-MFMA kernels (tests/split_precision_study.py, DESIGN.md 4); RVDD_CONV=f32 runs those instead")
                      if (not arch.startswith("next") and _CONV.startswith("conv3x3h")) else
                      ("f32 in, f32 out, f32 accumulation; the ConvBlock's two 1x1 convs multiply on the F16 matrix pipe with every f32 "
                       "operand split into two f16 halves (DESIGN.md 4.4); RVDD_NEXT_SPLIT=0 runs the f32-MFMA form")
                      if (arch.startswith("next") and os.environ.get("RVDD_NEXT_SPLIT") != "0")
                      else "f32 throughout (f32 MFMA, f32 VALU)",
        "config": {"workload": f"{config}: {DESCR[config]}" + (f" (run with --frames {T})" if args.frames else ""),
                   "arch": arch, "checkpoint": stem,
                   "frame": f"{W}x{H}", "frames_per_sequence": T, "sequences_per_gpu": len(my_seqs),
                   "sequences_in_lockstep": B, "sequences_total": total_seqs,
                   "output_frames_per_step_per_gpu": n_out * len(my_seqs), "parallelism": par},
        "fps_per_gpu": round(fps / world, 3), "ms_per_frame": round(1e3 * elapsed / (frames_total / world), 3),
        "gpu_event_ms_per_step": round(ev_ms / args.steps, 3),
        "algorithmic_gflop_per_frame": gflop_frame,
        "whole_path_algorithmic_tflops": round(fps / world * gflop_frame / 1e3, 2),
        "task_psnr_db": round(psnr_mean, 3),
        "gpu_clock_power": clock_power,
        "roofline": roofline, "cpu_baseline": cpu, "exact_f32_kernels": exact, "other_configs": other, "collate": collate,
        "kernels": kernels,
    }
    if cpu:
        line["gpu_over_cpu"] = round(fps / cpu["value"], 1)
    print(json.dumps(line), flush=True)
    if og is not None:
        og.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
