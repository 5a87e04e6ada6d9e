#!/usr/bin/env python3
"""Round 6: the one-launch register-resident GroupNorm (csrc/norm.hip gn_resident_kernel) against the two-launch form, per shape.
Uses the "tune" build (DS_GN_RESIDENT is read there only): one child process per setting.   python tools/bench_gn_resident.py [E ...]"""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" not in sys.argv:
    Es = [a for a in sys.argv[1:]] or ["16", "2"]
    res = {}
    for mode in ("1", "0"):
        env = dict(os.environ, DS_HIP_LIBRARY=os.path.join(REPO, "dynamicscaler_amd", "libdynscaler_hip_tune.so"), DS_GN_RESIDENT=mode)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + Es, env=env, capture_output=True, text=True)
        rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        if not rows:
            print("child failed", r.stderr[-1500:])
        for row in rows:
            res.setdefault(row["key"], {})[mode] = row
    print(f"{'shape':44s} | resident us | two-launch us | ratio | resident GB/s (algorithmic 1 read + 1 write) | max |diff| vs two-launch")
    for k, v in res.items():
        a, b = v.get("1"), v.get("0")
        if a and b:
            print(f"{k:44s} | {a['us']:9.1f} | {b['us']:9.1f} | {a['us'] / b['us']:.2f} | {a['bytes'] / a['us'] / 1e3:7.0f} | {a['digest']} {b['digest']}")
    sys.exit(0)
import hashlib, torch
sys.path.insert(0, REPO)
from dynamicscaler_amd import ops
d = torch.device("cuda:0")
T = 16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for E in [int(a) for a in sys.argv[1:] if a != "--child"]:
    for C, H, W in ((320, 40, 64), (640, 40, 64), (960, 40, 64), (640, 20, 32), (1280, 20, 32), (1920, 20, 32), (1280, 10, 16), (2560, 10, 16), (1280, 5, 8), (2560, 5, 8)):
        for dt in (torch.float32, torch.float16):
            for joint in (False, True):
                ninst, rows = (E, T * H * W) if joint else (E * T, H * W)
                if rows > 5120 or rows <= 256:
                    continue
                torch.manual_seed(0)
                x = (torch.randn(ninst * rows, C, device=d) * 0.5 + 0.1).to(dt)
                g, be = torch.rand(C, device=d) + 0.5, torch.randn(C, device=d) * 0.1
                y = ops.groupnorm(x, g, be, ninst, rows, C, 1e-5, True)
                us = timeit(lambda: ops.groupnorm(x, g, be, ninst, rows, C, 1e-5, True))
                el = 4 if dt == torch.float32 else 2
                print(json.dumps(dict(key=f"E={E} C={C} {H}x{W} {'jointT' if joint else 'frame '} {'f32' if el == 4 else 'f16'} rows={rows}", us=us,
                                      bytes=ninst * rows * C * (el + 2), digest=hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:8])), flush=True)
