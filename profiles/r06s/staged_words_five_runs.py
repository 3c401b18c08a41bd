import ctypes, json, os, sys, collections
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch, bench
from safetensors.torch import load_file
from rvdd_release_amd import synth
from rvdd_release_amd.runtime import RvddRuntime
lib = ctypes.CDLL(os.path.join(bench.REPO, "rvdd-release_amd", "librvdd_hip.so"))
lib.rvdd_debug_read_a.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
arch, stem, fut, iso, H, W, _, B, _ = bench.CONFIGS["C2"]
sd = load_file(os.path.join(bench.REPO, "weights", stem + ".safetensors"))
seqs = [synth.make_sequence(3, H, W, iso=iso, seed=4100 + b, device="cuda") for b in range(B)]
st = lambda f: torch.stack([f(s) for s in seqs], 0)
N = 256 * 512 * 40
dumps = []; outs = []
for run in range(5):
    rt = RvddRuntime(arch, fut, B, H, W, 0)
    rt.load_state_dict(sd)
    o = rt.step(st(lambda s: s.raw[0]), st(lambda s: s.raw[1]), None, st(lambda s: s.flow_prev[1]), None).clone()
    torch.cuda.synchronize()
    buf = np.zeros(N, dtype=np.uint32)
    lib.rvdd_debug_read_a(buf.ctypes.data, N)
    dumps.append(buf.reshape(256, 512, 40).copy()); outs.append(o)
    rt.close()
print("output frames equal to run 0:", [bool(torch.equal(o, outs[0])) for o in outs])
D = np.stack(dumps)                      # [run][wg][tid][40]
res = D[:, :, :, 16:32]
# majority value per word
ref = np.where((res[0] == res[1]) | (res[0] == res[2]), res[0], res[1])
for run in range(5):
    bad = (res[run] != ref)              # [wg][tid][16]
    n = int(bad.any(2).sum())
    print(f"run {run}: threads whose staged words differ from the majority: {n}")
    if n:
        wg, tid = np.nonzero(bad.any(2))
        print("   by (wave, quarter):", sorted(collections.Counter((int(t) // 64, int(t) % 64 // 16) for t in tid).items()))
        print("   by word (4*(2e+f) + {hi0,hi1,lo0,lo1}):", sorted(collections.Counter(int(k) for k in np.nonzero(bad)[2]).items()))
        print("   tiles (y0, x0, soff, b):", sorted(collections.Counter((int(D[run, w, 0, 32]), int(D[run, w, 0, 33]), int(np.int32(D[run, w, 0, 34])), int(D[run, w, 0, 35])) for w in wg).items())[:12])
        for w, t in list(zip(wg, tid))[:6]:
            print("   wg %d tid %d got %s want %s" % (w, t, [hex(int(x)) for x in res[run, w, t]], [hex(int(x)) for x in ref[w, t]]))
