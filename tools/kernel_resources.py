#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel in the BUILT library, read from the code objects inside
rvdd-release_amd/librvdd_hip.so (no recompilation: the numbers are those of the binary that runs).

    python tools/kernel_resources.py [library.so]    # table (default: the in-tree library)
    python tools/kernel_resources.py --spills        # only the kernels that spill or use scratch; exit 1 if any

The .hip_fatbin section holds one clang offload bundle per translation unit; each bundle's gfx950 entry is an ELF code
object whose NT_AMDGPU_METADATA note lists, per kernel, .vgpr_count / .agpr_count / .vgpr_spill_count /
.sgpr_spill_count / .private_segment_fixed_size (scratch bytes per lane) / .group_segment_fixed_size (static LDS).
tests/test_abi.py runs `kernel_table()` and pins the hot kernels at zero spills.
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "rvdd-release_amd", "librvdd_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
          ".private_segment_fixed_size", ".group_segment_fixed_size", ".max_flat_workgroup_size")


def _demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return [re.sub(r"\(.*$", "", n.replace("void ", "", 1).replace("(anonymous namespace)::", "")) for n in out.splitlines()]


def kernel_table(lib=LIB):
    """-> list of dicts {name, vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count,
    private_segment_fixed_size, group_segment_fixed_size, max_flat_workgroup_size}, one per kernel of the library."""
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, st in enumerate(starts):
            piece = os.path.join(tmp, f"bundle{i}.bin")
            open(piece, "wb").write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, f"co{i}.o")
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={piece}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            cur = None
            for ln in notes.splitlines():
                ln = ln.strip()
                m = re.match(r"^-?\s*(\.[a-z_]+):\s*(.*)$", ln)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip().strip("'")
                if key == ".agpr_count":             # first key of a kernel entry (keys are sorted)
                    cur = {"agpr_count": int(val)}
                    rows.append(cur)
                elif cur is not None and key in FIELDS:
                    cur[key[1:]] = int(val)
                elif cur is not None and key == ".name":
                    cur["mangled"] = val
    names = _demangle([r.get("mangled", "?") for r in rows])
    for r, n in zip(rows, names):
        r["name"] = n
    return sorted(rows, key=lambda r: r["name"])


def main():
    only = "--spills" in sys.argv
    libs = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = kernel_table(libs[0]) if libs else kernel_table()
    bad = [r for r in rows if r.get("vgpr_spill_count", 0) or r.get("sgpr_spill_count", 0) or r.get("private_segment_fixed_size", 0)]
    print(f"{'kernel':58s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>7s}")
    for r in (bad if only else rows):
        print(f"{r['name'][:58]:58s} {r.get('vgpr_count', 0):5d} {r.get('agpr_count', 0):5d} {r.get('sgpr_count', 0):5d} "
              f"{r.get('vgpr_spill_count', 0):6d} {r.get('sgpr_spill_count', 0):6d} {r.get('private_segment_fixed_size', 0):7d} "
              f"{r.get('group_segment_fixed_size', 0):7d}")
    print(f"{len(rows)} kernels, {len(bad)} with spills or scratch")
    return 1 if (only and bad) else 0


if __name__ == "__main__":
    sys.exit(main())
