#!/bin/bash
# round 6, call O: L2 touch of the dense A rows two K-steps ahead (build variant "l2touch") -- GEMM tests on it, per-shape and in-step A/B
O=gpurun_out/r6_o; mkdir -p $O
L=$PWD/dynamicscaler_amd/libdynscaler_hip_l2touch.so   # (git apply tools/exp/gemm_l2_touch.patch && python -m dynamicscaler_amd.build --variant l2touch)
DS_HIP_LIBRARY=$L timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm or linear or geglu or tail_split or layernorm_fold or ln" 2>&1 | tail -3 | tee $O/gemm_tests.txt
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
M1, M2, M3 = E * T * 40 * 64, E * T * 20 * 32, E * T * 10 * 16
G = _lib.DS_EPI_GEGLU
cases = [("L1 out/proj +res16", M1, 320, 320, {}, "h"), ("L1 out/proj +res32 out32", M1, 320, 320, dict(epilogue=_lib.DS_EPI_RES_F32 | _lib.DS_EPI_OUT_F32), "f"),
         ("L1 q nobias", M1, 320, 320, {}, None), ("L1 qkv", M1, 960, 320, {}, None), ("L1 geglu", M1, 2560, 320, dict(epilogue=G), None),
         ("L1 ff2 +res16", M1, 320, 1280, {}, "h"),
         ("L2 out +res16", M2, 640, 640, {}, "h"), ("L2 qkv", M2, 1920, 640, {}, None), ("L2 geglu", M2, 5120, 640, dict(epilogue=G), None), ("L2 ff2 +res16", M2, 640, 2560, {}, "h"),
         ("L3 out +res16", M3, 1280, 1280, {}, "h"), ("L3 qkv", M3, 3840, 1280, {}, None), ("L3 geglu", M3, 10240, 1280, dict(epilogue=G), None), ("L3 ff2 +res16", M3, 1280, 5120, {}, "h")]
for name, m, n, k, kw, res in cases:
    A, W = rnd(m, k), rnd(n, k)
    b = torch.randn(n, device=d) if ("nobias" not in name and "qkv" not in name) else None
    R = None if res is None else (rnd(m, n) if res == "h" else torch.randn(m, n, device=d))
    out = ops.gemm(A, W, b, R, M=m, N=n, K=k, **kw)
    us = timeit(lambda: ops.gemm(A, W, b, R, M=m, N=n, K=k, **kw))
    print(json.dumps(dict(name=name, M=m, N=n, K=k, us=round(us, 1), sha=hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:10])), flush=True)
PY
for rep in 1 2; do
for lib in product l2touch; do
  if [ $lib = l2touch ]; then export DS_HIP_LIBRARY=$L; else unset DS_HIP_LIBRARY; fi
  timeout 600 python $O/shapes.py 2>/dev/null | sed "s/^/$lib /" | tee -a $O/shapes_$rep.txt
done; done
B="--steps 4 --warmup 1 --no-cpu-baseline --no-roofline --full-panorama 0 --other-configs 0 --wide-step 0"
pick() { python -c "import sys,json; l=[x for x in sys.stdin if x.startswith('{')]; d=json.loads(l[-1]); print('$1', d['ms_per_step'], d['result_sha256'])"; }
export DS_OPERAND_POLICY=f16
for rep in 1 2 3; do
  unset DS_HIP_LIBRARY; timeout 600 python bench.py $B 2>/dev/null | pick "cfg3 product" | tee -a $O/step_ab.txt
  export DS_HIP_LIBRARY=$L; timeout 600 python bench.py $B 2>/dev/null | pick "cfg3 l2touch" | tee -a $O/step_ab.txt
done
