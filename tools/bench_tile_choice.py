"""Forced-tile A/B on the shapes of an 8-GPU rank's share / one sphere view ([cond | uncond] pair of one 512 x 320 x 16f window): the
default choose_tile pick against each tile id that applies.  Uses the "tune" build (DS_GEMM_TILE is read there only, once per
process: one child process per tile id):   python tools/bench_tile_choice.py [evals=2]"""
import hashlib, json, os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILES = {-1: "default", -2: "default(2)", 0: "128x64", 1: "128x128", 2: "256x256", 3: "256x320", 4: "128x128deep", 5: "128x64deep", 6: "128x320sw", 7: "128x256sw"}
if "--child" not in sys.argv:
    E = sys.argv[1] if len(sys.argv) > 1 else "2"
    table = {}
    for t in TILES:
        env = dict(os.environ, DS_HIP_LIBRARY=os.path.join(REPO, "dynamicscaler_amd", "libdynscaler_hip_tune.so"))
        env.pop("DS_GEMM_TILE", None)
        if t >= 0:
            env["DS_GEMM_TILE"] = str(t)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), E, "--child"], env=env, capture_output=True, text=True)
        rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
        if not rows:
            print(f"tile {TILES[t]}: child failed\n{r.stderr[-800:]}")
        for row in rows:
            table.setdefault((row["name"], row["M"], row["N"], row["K"]), {})[t] = row
    print(f"{'shape':16s} {'M':>7s} {'N':>5s} {'K':>6s} | " + " ".join(f"{TILES[t]:>11s}" for t in TILES) + " | best vs default")
    for key, cols in table.items():
        ref = cols.get(-1, {}).get("sha")
        cells = []
        for t in TILES:
            c = cols.get(t)
            cells.append(f"{'--':>11s}" if c is None or c["us"] is None else f"{c['us']:9.1f}{' ' if c['sha'] == ref else '!'}u")
        ok = {t: c["us"] for t, c in cols.items() if c["us"] is not None}
        best = min(ok, key=ok.get)
        print(f"{key[0]:16s} {key[1]:7d} {key[2]:5d} {key[3]:6d} | " + " ".join(cells) + f" | {TILES[best]} {ok[best] / ok[-1]:.2f}")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicscaler_amd import ops, _lib

d = torch.device("cuda:0")
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = 16


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rnd(*shape):
    return (torch.randn(*shape, device=d) * 0.5).half()


torch.manual_seed(0)
cases = []
for C, H, W in [(320, 40, 64), (640, 20, 32), (1280, 10, 16), (1280, 5, 8)]:
    M = E * T * H * W
    cases += [(f"L{C} out/proj", M, C, C, {}, True), (f"L{C} qkv", M, 3 * C, C, {}, False), (f"L{C} ff1", M, 8 * C, C, {"epilogue": _lib.DS_EPI_GEGLU}, False),
              (f"L{C} ff2", M, C, 4 * C, {}, True),
              (f"L{C} conv3", M, C, 9 * C, dict(a_mode=_lib.DS_A_CONV3, cin=C, lda=C, conv=(E * T, H, W, H, W, 1, 0)), False),
              (f"L{C} conv3 2C", M, C, 18 * C, dict(a_mode=_lib.DS_A_CONV3, cin=2 * C, lda=2 * C, conv=(E * T, H, W, H, W, 1, 0)), False),
              (f"L{C} tconv", M, C, 3 * C, dict(a_mode=_lib.DS_A_TCONV, cin=C, lda=C, tconv=(T, H * W)), True)]
for name, m, n, k, kw, res in cases:
    cin = kw.get("cin", k)
    A, Wt = rnd(m, cin), rnd(n, k)
    b = torch.randn(n, device=d)
    R = rnd(m, n) if res else None
    try:
        out = ops.gemm(A, Wt, b, R, M=m, N=n, K=k, **kw)
        us = timeit(lambda: ops.gemm(A, Wt, b, R, M=m, N=n, K=k, **kw)) * 1e6
        sha = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:12]
    except Exception as e:      # noqa: BLE001
        us, sha = None, None
    print(json.dumps(dict(name=name, M=m, N=n, K=k, us=us, sha=sha)), flush=True)
