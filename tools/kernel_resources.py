#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of a built library (reads the code objects' metadata notes; CPU only).

    python tools/kernel_resources.py [path/to/lib.so] [-v]

Prints the kernels that use scratch memory (spills) -- the product library must have none -- and with -v every kernel.
Also importable: kernel_resources(lib) -> list of dicts (tests/test_host_cpu.py::test_no_kernel_uses_scratch)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_resources(lib=None):
    lib = lib or os.path.join(REPO, "dynamicscaler_amd", "libdynscaler_hip.so")
    out = []
    with tempfile.TemporaryDirectory() as td:
        work = os.path.join(td, "lib.so")
        shutil.copy(lib, work)                       # llvm-objdump drops the extracted bundles next to its input
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", work], check=True, capture_output=True, cwd=td)
        for f in sorted(os.listdir(td)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(td, f)],
                                   capture_output=True, text=True, check=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                def field(name, cast=int, default=0):
                    m = re.search(r"\." + name + r":\s+(\S+)", blk)
                    return cast(m.group(1)) if m else default
                out.append({"name": field("name", str, "?"), "vgpr": field("vgpr_count"), "agpr": int(blk.split()[0]),
                            "sgpr": field("sgpr_count"), "scratch": field("private_segment_fixed_size"),
                            "lds": field("group_segment_fixed_size"), "spill_vgpr": field("vgpr_spill_count"),
                            "spill_sgpr": field("sgpr_spill_count")})
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    rows = kernel_resources(args[0] if args else None)
    bad = [r for r in rows if r["scratch"] or r["spill_vgpr"]]
    print(f"{len(rows)} kernels, {len(bad)} with scratch / spills")
    for r in (rows if "-v" in sys.argv else bad):
        print(f'{r["vgpr"]:4d} v {r["agpr"]:4d} a {r["sgpr"]:4d} s  scratch {r["scratch"]:5d}  lds {r["lds"]:6d}  {r["name"][:150]}')
    sys.exit(1 if bad else 0)
