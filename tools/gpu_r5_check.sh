#!/bin/bash
# round-5 mid-round check: kernel / C-program / wide tests, the policy tests of the full-size suite, the default bench line, and the
# rank-share step (col2, one tile per batch, cond / uncond on two streams = two rank-steps of an 8-GPU cfg3 run) with and without the
# tail split of the persistent GEMM ("tune" variant: DS_GEMM_TAIL_SPLIT=0)
O=gpurun_out/${1:-r5check}; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unet_c.py tests/test_gpu_wide.py -q -x 2>&1 | tail -5 | tee $O/quick.txt
timeout 2400 python -m pytest tests/test_gpu_fullsize.py -q -x -s -k "cfg1_full_size or ring_pipeline_with_the_real_unet or poison or batch_equals" 2>&1 | grep -v Warning | tail -60 | tee $O/fullsize.txt
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/summary.txt
cut -c1-600 $O/bench_default.json | tee -a $O/summary.txt
share() { DS_SPLIT_CFG=2 timeout 600 python bench.py --config col2 --steps 8 --warmup 3 --streams 2 --tile-batch 1 --no-cpu-baseline --no-roofline --full-panorama 0 --wide-step 0 2> $O/share_$1.err | tail -1 > $O/share_$1.json; echo "share $1: $(grep -o '"ms_per_step": [0-9.]*' $O/share_$1.json) $(grep -o '"latent_after_timed_steps": "[0-9a-f]*"' $O/share_$1.json)" | tee -a $O/summary.txt; }
python -m dynamicscaler_amd.build --variant tune > /dev/null 2>&1
share split_a
DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_tune.so DS_GEMM_TAIL_SPLIT=0 share nosplit_a
share split_b
DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_tune.so DS_GEMM_TAIL_SPLIT=0 share nosplit_b
