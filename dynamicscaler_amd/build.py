"""Build libdynscaler_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m dynamicscaler_amd.build [--force]

Objects are rebuilt only when a source (or a header) is newer.  The .so stays next to this file so it
travels to the GPU box with the repo snapshot (git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libdynscaler_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["error.cpp", "tile_ops.hip", "gemm.hip", "attention.hip", "norm.hip", "misc.hip", "encoders.hip", "wide.hip", "unet_program.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-ffp-contract=on"]


# per-source flags.  attention.hip: -O3's SLP vectoriser pairs adjacent fp32 adds / multiplies into v_pk_*_f32, which cost several
# times their two halves in the gaps between MFMAs (csrc/attention.hip, DS_ATTN_PK)
SOURCE_FLAGS = {"attention.hip": ["-fno-slp-vectorize"]}


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


# diagnostic variants of the library (same ABI, loaded through DS_HIP_LIBRARY by tests/hazard_probe.py only):
# "barebarrier" = round 1's K-step barrier without the lgkmcnt(0) in front of it (profiles/r2_notes.md)
VARIANTS = {"barebarrier": ["-DDS_EXP_BARE_BARRIER"],
            # the tuning / diagnostic environment switches (DS_GEMM_TILE, DS_GEMM_GROUP_M, DS_GEMM_BIG_MIN, DS_CONV_TAPS_INNER, DS_ATTN_QB,
            # DS_TATTN_VALU, DS_GN_SPARSE_WGS, DS_GN_CHUNK_RULE): compiled out of the product, read by this variant (csrc/common.h)
            "tune": ["-DDS_TUNING_ENV=1"], "attnplain": ["-DDS_ATTN_NO_XCD_REMAP"], "nt0": ["-DDS_EXP_NT=0", "-DDS_EXP_STREAM_NT=0"],
            "attnnarrow": ["-DDS_ATTN_NARROW_STORES"],
            "attn4": ["-DDS_ATTN_WGS=4"], "attn2": ["-DDS_ATTN_WGS=2"],
            "attnvt": ["-DDS_ATTN_TRV=0"],     # V transposed on its way into LDS (rounds 1-3) instead of transposing LDS reads
            "noslp": ["-fno-slp-vectorize"],      # every file without -O3's pairing of fp32 ops (attention.hip always is)
            # round 2's GroupNorm kernel choice (by instance COUNT): breaks batch invariance at full size (profiles/r3_notes.md section 7)
            "gncount": ["-DDS_EXP_GN_COUNT_THRESHOLD"],
            # round 3's K loop on v_mfma_f32_32x32x16_f16 (the product uses 16x16x32 since round 4: profiles/r4_notes.md)
            "mfma32": ["-DDS_MFMA16=0"],
            # tuning variants of the 16x16x32 K loop (A/B: tools/gpu_ab.sh): where the LDS-DMA pieces go, how far ahead fragments are read
            # one tile per workgroup, strips over stage 0 (round 3's structure) instead of the persistent big tiles
            "nopersist": ["-DDS_PERSIST=0"],
            "setprio": ["-DDS_SETPRIO_HI=1"], "nt32": ["-DDS_EXP_NT=11"],
            # fp32 rows: 8 consecutive channels / columns per thread (round 3) instead of two groups of 4 contiguous across the wave
            "now4": ["-DDS_EXP_NO_W4=1"],
            # with ds_gemm_f16_stats (GroupNorm statistics from the producer's epilogue: opt-in feature, profiles/r4_notes.md section 3)
            "gemmstats": ["-DDS_GEMM_STATS=1", "-DDS_PERSIST=0"],   # (with the persistent tile loop one of its kernels spills)
            # round 6: GEMM kernels held to fewer registers than two waves per SIMD allow (room for co-resident waves of other kernels)
            "regcap208": ["-DDS_GEMM_VGPR_CAP=208", "-DDS_TUNING_ENV=1"], "regcap192": ["-DDS_GEMM_VGPR_CAP=192", "-DDS_TUNING_ENV=1"],
            # round 6 A/B: 16-row epilogue strips of the wave's whole width on the 256 x 320 tile (320-byte store pieces) instead of 32 rows x two tiles
            "epihalf": ["-DDS_EPI_HALF=1", "-DDS_TUNING_ENV=1"],
            "m16p1": ["-DDS_M16_PSPAN4=1"], "m16p3": ["-DDS_M16_PSPAN4=3"], "m16p4": ["-DDS_M16_PSPAN4=4"], "m16a3": ["-DDS_M16_AHEAD=3"], "m16a1": ["-DDS_M16_AHEAD=1"]}


def build(force=False, verbose=True, variant=None):
    extra, obj_dir, lib = [], OBJ, LIB          # the product; a variant builds into its own directory / library
    if variant is not None:
        extra = VARIANTS[variant]
        obj_dir = os.path.join(HERE, "_build_" + variant)
        lib = os.path.join(HERE, f"libdynscaler_hip_{variant}.so")
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "dynscaler_hip.h")]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _newer([src] + headers, obj):
            cmd = [HIPCC] + FLAGS + SOURCE_FLAGS.get(s, []) + extra + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for cmd, r in ex.map(run, jobs):
            if verbose and (r.stdout or r.stderr):
                sys.stderr.write(r.stdout + r.stderr)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or jobs or _newer(objs, lib):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    return lib


def build_diag(verbose=True):
    """libdynscaler_diag.so (csrc/diag.hip): diagnostics outside the product ABI (the LDS / register poison launch of the test suite)."""
    src = os.path.join(CSRC, "diag.hip")
    lib = os.path.join(HERE, "libdynscaler_diag.so")
    if _newer([src], lib):
        cmd = [HIPCC, "-O2", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", "-x", "hip", src, "-o", lib]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if verbose and (r.stdout or r.stderr):
            sys.stderr.write(r.stdout + r.stderr)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    return lib


def build_examples(verbose=True):
    """examples/unet_host: a C++ host of the UNet that uses only include/dynscaler_hip.h and the HIP runtime (the non-Python
    boundary of ds_unet_*); tests/test_gpu_unet_c.py runs it.  Rebuilt when its source, the header or the library is newer."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "examples", "unet_host.cpp")
    exe = os.path.join(root, "examples", "unet_host")
    if not os.path.exists(src):
        return None
    if _newer([src, os.path.join(root, "include", "dynscaler_hip.h"), LIB], exe):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(root, "include"), src, "-L", HERE,
               "-ldynscaler_hip", "-Wl,-rpath,$ORIGIN/../dynamicscaler_amd", "-o", exe]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if verbose and (r.stdout or r.stderr):
            sys.stderr.write(r.stdout + r.stderr)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    return exe


if __name__ == "__main__":
    v = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    print(build(force="--force" in sys.argv, variant=v))
    if v is None:
        print(build_diag())
        print(build_examples())
        for name in VARIANTS:          # variant libraries already in the tree are kept in step with the sources (a stale one is refused at load)
            if os.path.exists(os.path.join(HERE, f"libdynscaler_hip_{name}.so")):
                print(build(force="--force" in sys.argv, variant=name))
