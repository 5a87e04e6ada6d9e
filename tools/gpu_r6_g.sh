#!/bin/bash
# round 6, call G: half-block full-width epilogue strips on the 256 x 320 tile -- GEMM tests, per-shape and in-step A/B against the "nohalf" build
O=gpurun_out/r6_g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm or conv or linear or geglu or tail_split" 2>&1 | tail -3 | tee $O/gemm_tests.txt
cat > $O/shapes.py <<'PY'
import hashlib, json, os, sys, torch
sys.path.insert(0, os.getcwd())
from dynamicscaler_amd import ops, _lib
d = torch.device("cuda:0")
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
torch.manual_seed(0)
def rnd(*s): return (torch.randn(*s, device=d) * 0.5).half()
E, T = 16, 16
cases = []
M = E * T * 40 * 64
cases += [("out/proj +res16", M, 320, 320, {}, "h"), ("out/proj +res32 out32", M, 320, 320, dict(epilogue=_lib.DS_EPI_RES_F32 | _lib.DS_EPI_OUT_F32), "f"), ("q / proj_in nobias", M, 320, 320, {}, None),
          ("qkv", M, 960, 320, {}, None), ("ff2 +res16", M, 320, 1280, {}, "h"), ("tconv +res16", M, 320, 960, dict(a_mode=_lib.DS_A_TCONV, cin=320, lda=320, tconv=(T, 40 * 64)), "h"),
          ("conv3", M, 320, 2880, dict(a_mode=_lib.DS_A_CONV3, cin=320, lda=320, conv=(E * T, 40, 64, 40, 64, 1, 0)), None)]
M3 = E * T * 10 * 16
cases += [("L3 conv3 (256x320 by the rounds rule)", M3, 1280, 11520, dict(a_mode=_lib.DS_A_CONV3, cin=1280, lda=1280, conv=(E * T, 10, 16, 10, 16, 1, 0)), None),
          ("L3 qkv 3840", M3, 3840, 1280, {}, None), ("L3 out +res16", M3, 1280, 1280, {}, "h")]
for name, m, n, k, kw, res in cases:
    cin = kw.get("cin", k)
    A, W = rnd(m, cin), rnd(n, k)
    b = torch.randn(n, device=d) if "nobias" not in name and "qkv" not in name else None
    R = None if res is None else (rnd(m, n) if res == "h" else torch.randn(m, n, device=d))
    out = ops.gemm(A, W, b, R, M=m, N=n, K=k, **kw)
    us = timeit(lambda: ops.gemm(A, W, b, R, M=m, N=n, K=k, **kw))
    print(json.dumps(dict(name=name, M=m, N=n, K=k, us=round(us, 1), sha=hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:10])), flush=True)
PY
for lib in product nohalf; do
  if [ $lib = nohalf ]; then export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_nohalf.so; else unset DS_HIP_LIBRARY; fi
  timeout 600 python $O/shapes.py 2>/dev/null | sed "s/^/$lib /" | tee -a $O/shapes.txt
done
B="--steps 4 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-configs 0 --wide-step 0"
pick() { python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('$1', d['ms_per_step'], d['result_sha256'])"; }
export DS_OPERAND_POLICY=f16
for rep in 1 2 3; do
  unset DS_HIP_LIBRARY; timeout 600 python bench.py $B 2>/dev/null | pick "cfg3 half" | tee -a $O/step_ab.txt
  export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_nohalf.so; timeout 600 python bench.py $B 2>/dev/null | pick "cfg3 nohalf" | tee -a $O/step_ab.txt
done
