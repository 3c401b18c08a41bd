"""The C-ABI library: it loads, exports exactly what include/rvdd.h declares, and
fails loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest
import torch

from conftest import REPO

HEADER = os.path.join(REPO, "include", "rvdd.h")


def declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rvdd_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from rvdd_release_amd import _lib
    assert declared_functions() == _lib.exported_symbols()


def test_library_exports_every_declared_symbol():
    from rvdd_release_amd import _lib
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\sT\s+(rvdd_[a-z0-9_]+)", out))
    assert exported == set(declared_functions())
    assert b"gfx950" in lib.rvdd_version()


def test_documented_options_are_the_options_the_library_knows():
    """include/rvdd.h documents rvdd_set_option's names one by one; the library lists the names it accepts in the message it
    raises for an unknown one.  The two lists are the same set (an option removed from one and not the other is a stale ABI)."""
    from rvdd_release_amd import _lib
    txt = open(HEADER).read()
    doc = txt[txt.index("Known names:"):txt.index("int rvdd_set_option(")]
    documented = set(re.findall(r'^ \*   "([a-z0-9_]+)"', doc, flags=re.M))
    blob = open(_lib.LIB_PATH, "rb").read()
    m = re.search(rb"unknown option '%s' \(known: ([a-z0-9_, ]+)\)", blob)
    assert m, "the library's unknown-option message was not found"
    known = set(m.group(1).decode().split(", "))
    assert documented == known, (sorted(documented - known), sorted(known - documented))


def test_library_carries_gfx950_code_object():
    from rvdd_release_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    assert b"conv3x3_kernel" in blob and b"warp48_kernel" in blob and b"convblock_pipe_kernel" in blob


def test_create_validates_arguments():
    from rvdd_release_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    for bad in (_lib.RvddCfg(7, 0, 1, 64, 64, 0), _lib.RvddCfg(0, 2, 1, 64, 64, 0), _lib.RvddCfg(0, 0, 0, 64, 64, 0),
                _lib.RvddCfg(0, 0, 1, 63, 64, 0), _lib.RvddCfg(0, 0, 1, 8, 64, 0)):
        assert lib.rvdd_create(C.byref(bad), C.byref(h)) == -1          # RVDD_ERR_ARG
        assert lib.rvdd_last_error(None)
        assert not h.value
    assert lib.rvdd_create(None, C.byref(h)) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly_no_cpu_fallback():
    from rvdd_release_amd.runtime import RvddRuntime
    with pytest.raises(RuntimeError, match="no HIP device available.*no CPU fallback"):
        RvddRuntime("convunet", 0, 1, 64, 64, 0)
    from rvdd_release_amd.util.flow_utils import warp
    with pytest.raises(RuntimeError, match="GPU tensors only"):
        warp(torch.zeros(1, 3, 16, 16), torch.zeros(1, 2, 16, 16), "bicubic")


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under rvdd-release_amd/ may reference it."""
    root = os.path.join(REPO, "rvdd-release_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                src = open(os.path.join(dp, f)).read()
                assert "rvdd_oracle" not in src and "oracle/" not in src, os.path.join(dp, f)


def test_no_kernel_spills_to_scratch():
    """Register spills of the BUILT library's kernels, read from its code objects (tools/kernel_resources.py).  No kernel
    may spill vector registers or use scratch memory (a scratch reload is an s_waitcnt vmcnt(0) in the middle of a
    kernel's memory pipeline: the fused-upsample Winograd kernel lost 7 % to nine of them), and the kernels of the
    hot path (3x3 convs, ConvNeXt block, pre-stages) must not spill scalar registers either."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import kernel_resources
    from rvdd_release_amd import _lib
    rows = kernel_resources.kernel_table(_lib.LIB_PATH)
    names = {r["name"] for r in rows}
    assert len(rows) >= 50 and any(n.startswith("wino3x3_ups_kernel") for n in names), sorted(names)
    bad = [(r["name"], r.get("vgpr_spill_count", 0), r.get("private_segment_fixed_size", 0)) for r in rows
           if r.get("vgpr_spill_count", 0) or r.get("private_segment_fixed_size", 0)]
    assert not bad, bad
    # scalar spills go to lanes of a vector register (v_writelane / v_readlane, no memory): none in the kernels that
    # carry a frame-step; a handful in the epilogue variants that run once (OUT3: the 1x1 conv 48 -> 3) or three times
    # (the fused ConvBlock's pooling epilogue, split-f16 form) per frame-step, or whose two role branches (the pipelined
    # ConvBlock: front and back waves) each keep their own set of loop invariants
    hot = ("wino3x3", "conv3x3", "convblock_kernel", "convblock_pipe_kernel", "proj1x1", "warp48", "netin", "ha_")
    once = ("convblock_kernel<true, false, ", "convblock_kernel<false, true, true>", "convblock_pipe_kernel", "wino3x3_kernel<4, false>")
    # conv3x3h_kernel holds two forms of its tile loop since round 4 (with and without the block floating point's scaling,
    # chosen per workgroup): the instantiations with the most loop invariants (second pass of the two-source layers, fused
    # upsample, bottleneck sum, fused 1x1 output, the 16-channel first layer) keep a few of them in lanes of a vector register -- written once at the top
    # of the kernel, read back with one v_readlane per tile at most; the plain layers (60 % of the launches) keep none
    few = ("conv3x3h_kernel<48, 0, true, ", "conv3x3h_kernel<48, 1, true, ", "conv3x3h_kernel<48, 1, false, true, ",
           "conv3x3h_kernel<48, 3, ", "conv3x3h_kernel<48, 4, ", "conv3x3h_kernel<16, ")
    once = once + few
    # the plain 48 -> 48 instantiations keep ONE loop invariant in a vector lane since the tile's own offsets moved to chunk 6
    # of the MFMA loop (+0.4 % on C2, LABBOOK.md 4.1d): pinned at that one
    plain = ("conv3x3h_kernel<48, 0, false, false, 1, 3, 3>", "conv3x3h_kernel<48, 1, false, false, 1, 3, 3>")
    # the output-channel-split instantiations (conv3x3h.hip MT = 1: launches with at most a third of a tile per CU, where a launch
    # is one tile per workgroup) are not the hot path: pinned on their own at what they have (the fused-upsample one 31: round 6's
    # interior-tile form of its halo fetch added five scalars to the 26 -- and took 85 vector instructions per item out of the loop)
    small = [r for r in rows if r["name"].startswith("conv3x3h_kernel<48, ") and r["name"].endswith(", 1>")]
    assert small and all(r.get("sgpr_spill_count", 0) <= 31 for r in small), [(r["name"], r["sgpr_spill_count"]) for r in small]
    rows = [r for r in rows if r not in small]
    bad = [(r["name"], r["sgpr_spill_count"]) for r in rows if r["name"].startswith(hot) and not r["name"].startswith(once)
           and r.get("sgpr_spill_count", 0) > (1 if r["name"].startswith(plain) else 0)]
    assert not bad, bad
    # the fused-upsample instantiations (three launches per frame-step) carry the interpolation's row / column constants on
    # top of the tile loop's, and since round 6 two forms of the fetch (interior tiles: no address / weight arithmetic; border tiles:
    # the general form): 27 scalars in vector lanes, pinned there -- at most nine v_readlane in the chunk loop's code, none per chunk
    ups = ("conv3x3h_kernel<48, 1, false, true, ",)
    bad = [(r["name"], r["sgpr_spill_count"]) for r in rows if r["name"].startswith(once) and not r["name"].startswith(ups)
           and r.get("sgpr_spill_count", 0) > 16]
    assert not bad, bad
    bad = [(r["name"], r["sgpr_spill_count"]) for r in rows if r["name"].startswith(few) and not r["name"].startswith(ups)
           and r.get("sgpr_spill_count", 0) > 14]
    assert not bad, bad
    bad = [(r["name"], r["sgpr_spill_count"]) for r in rows if r["name"].startswith(ups) and r.get("sgpr_spill_count", 0) > 27]
    assert not bad, bad
    # the pipelined ConvBlock is C4's hot kernel (about 25 launches per frame-step), not a once-per-step variant: its plain
    # instantiation is pinned at what it has today, the pooling / 1x1-output ones at a handful
    pipe = {r["name"]: r.get("sgpr_spill_count", 0) for r in rows if r["name"].startswith("convblock_pipe_kernel")}
    assert pipe, sorted(names)
    bad = [(n, c) for n, c in pipe.items() if c > (2 if n.startswith("convblock_pipe_kernel<false, false") else 8)]
    assert not bad, bad
