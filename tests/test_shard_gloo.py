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
