"""The N>1 path of bench.py on CPU: 2 processes, gloo, the same shard.py helpers
bench.py uses on RCCL.  Each rank advances its own sequences (here with the CPU
oracle standing in for the HIP step) and the one all-gather collates the
per-frame metrics; the result must equal the single-process run."""
import os
import subprocess
import sys
import textwrap

import torch

from conftest import REPO

WORKER = textwrap.dedent("""
    import os, sys, json, torch
    sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "oracle"))
    import rvdd_oracle as O
    from safetensors.torch import load_file
    from rvdd_release_amd import shard, synth
    torch.set_num_threads(2)
    rank, local_rank, world, dist = shard.init_distributed("gloo")
    B, T, H, W = 2, 3, 32, 48
    sd = load_file(os.path.join({repo!r}, "weights", "recurrent-convunet-iso3200.safetensors"))
    ids = shard.shard_sequences(B * world, rank, world)
    rows = []
    for sid in ids:
        s = synth.make_sequence(T, H, W, seed=900 + sid)
        outs = O.RecurrentOracle(sd, future=0).run_sequence(s.raw, s.flow_prev)
        rows.append([O.psnr(outs[k][None], s.gt[k + 1][None]) for k in range(T - 1)])
    shard.barrier(dist)
    allm = shard.gather_metrics(torch.tensor(rows, dtype=torch.float64), dist)
    tmax = shard.max_over_ranks(1.0 + rank, dist)
    if rank == 0:
        print("RESULT " + json.dumps(dict(ids=list(ids), metrics=allm.tolist(), tmax=tmax, world=world)))
    if dist is not None:
        dist.destroy_process_group()
""")


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run(nproc, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(repo=REPO))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    if nproc == 1:
        cmd = [sys.executable, str(script)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_shard_partition_is_exact():
    from rvdd_release_amd import shard
    for n, w in ((64, 8), (5, 2), (7, 8), (1, 4)):
        got = [s for r in range(w) for s in shard.shard_sequences(n, r, w)]
        assert got == list(range(n))
    assert list(shard.shard_sequences(64, 3, 8)) == list(range(24, 32))     # seq s -> rank s // 8


def test_two_ranks_equal_one_process(tmp_path):
    two = _run(2, tmp_path)
    assert two["world"] == 2 and two["ids"] == [0, 1] and two["tmax"] == 2.0
    assert len(two["metrics"]) == 4
    # the same four sequences in one process (weak scaling: 2 per rank x 2 ranks)
    import rvdd_oracle as O
    from conftest import load_weights
    from rvdd_release_amd import synth
    sd = load_weights("recurrent-convunet-iso3200")
    for sid in range(4):
        s = synth.make_sequence(3, 32, 48, seed=900 + sid)
        outs = O.RecurrentOracle(sd, future=0).run_sequence(s.raw, s.flow_prev)
        want = [O.psnr(outs[k][None], s.gt[k + 1][None]) for k in range(2)]
        assert max(abs(a - b) for a, b in zip(want, two["metrics"][sid])) < 1e-4


def _bench(args, tmp_path):
    """bench.py's OWN launcher (`python bench.py --gpus N`, no torch.distributed.run around it) with the CPU stub
    standing in for the HIP runtime: ranks over gloo, the same sharding, collectives and JSON line as on the GPU node."""
    import json
    env = dict(os.environ, RVDD_BENCH_STUB="1", OMP_NUM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True,
                         env=env, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_self_launch_two_ranks_weak(tmp_path):
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "4", "--batch", "2", "--collate-outputs"], tmp_path)
    assert r["n_gpus"] == 2 and r["n_ranks_seen"] == 2 and r["scaling"] == "weak"
    assert r["config"]["workload"].startswith("C5")           # the multi-GPU default configuration
    assert r["config"]["sequences_total"] == 4 and r["config"]["sequences_per_gpu"] == 2
    assert r["value"] > 0 and r["steps"] == 2 and r["warmup"] == 1
    assert abs(r["value"] - 2 * 3 * 4 / (r["ms_per_step"] * 2 / 1e3)) / r["value"] < 1e-3    # frames of ALL ranks / max time
    assert r["collate"]["gathered_shape"] == [2, 3, 2, 3, 32, 48]
    assert "STUB" in r["data"] and r["cpu_baseline"] is None and r["vs_baseline"] is None


def test_bench_self_launch_strong_scaling(tmp_path):
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--frames", "3", "--batch", "2", "--scaling", "strong",
                "--sequences", "8"], tmp_path)
    assert r["scaling"] == "strong" and r["n_ranks_seen"] == 2
    assert r["config"]["sequences_total"] == 8 and r["config"]["sequences_per_gpu"] == 4
    assert r["config"]["output_frames_per_step_per_gpu"] == 2 * 4
    one = _bench(["--gpus", "1", "--steps", "1", "--warmup", "0", "--frames", "3", "--batch", "2", "--config", "C5",
                  "--scaling", "strong", "--sequences", "8"], tmp_path)
    assert one["n_gpus"] == 1 and one["config"]["sequences_per_gpu"] == 8
    assert abs(one["task_psnr_db"] - r["task_psnr_db"]) < 20       # same generator; both finite numbers


def test_bench_refuses_mismatched_world(tmp_path):
    env = dict(os.environ, RVDD_BENCH_STUB="1", WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         env=env, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=3" in out.stderr


def test_world_size_one_under_a_launcher_has_a_live_group(tmp_path):
    """`torch.distributed.run --nproc-per-node 1` (the driver's N = 1 launch form): the process group IS initialised
    and the collectives run (the GPU suite checks the same on RCCL); a plain `python` run has no group.  Through the
    stub step of bench.py, which prints which is which."""
    import json
    for launcher in (True, False):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        env.update(RVDD_BENCH_STUB="1", OMP_NUM_THREADS="2")
        tail = [os.path.join(REPO, "bench.py"), "--gpus", "1", "--config", "C5", "--frames", "4", "--steps", "1", "--warmup", "0",
                "--collate-outputs", "--cpu-frames", "0"]
        cmd = ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())] if launcher else [sys.executable]) + tail
        out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        d = line["distributed"]
        assert d["process_group"] is launcher and d["backend"] == ("gloo" if launcher else None), d
        assert line["n_ranks_seen"] == 1 and line["collate"]["gathered_shape"][0] == 1
