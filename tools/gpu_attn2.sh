#!/bin/bash
O=gpurun_out/attn2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | tail -3 | tee -a $O/summary.txt
for v in "" attnnarrow "" attnnarrow; do
  if [ -n "$v" ]; then export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_$v.so; else unset DS_HIP_LIBRARY; fi
  python tools/bench_attention.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
done
unset DS_HIP_LIBRARY
for rep in 1 2 3; do
  for V in base attnnarrow; do
    if [ $V = base ]; then unset DS_HIP_LIBRARY; else export DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_$V.so; fi
    timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --full-panorama 0 > $O/${V}_$rep.json 2> $O/${V}_$rep.err
    python - $O/${V}_$rep.json $V <<'PY' | tee -a $O/summary.txt
import json,sys
print(sys.argv[2], round(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["ms_per_step"],1))
PY
  done
done
