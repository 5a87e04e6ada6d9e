#!/bin/bash
# round-5 mid-round check: tail-split kernel test, the policy / mid-schedule tests of the full-size suite, the multi-rank rehearsals
# (bench.py per_rank fields), and the rank-share step (col2, one tile per batch = two rank-steps of an 8-GPU cfg3 run): cond / uncond on
# two streams with and without the launch-share hint, and the one-stream pair batch
O=gpurun_out/${1:-r5check}; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "tail_split or gemm" 2>&1 | tail -3 | tee $O/quick.txt
timeout 3000 python -m pytest tests/test_gpu_fullsize.py -q -x -s -k "cfg1_full_size or ring_pipeline_with_the_real_unet or mid_schedule or batch_equals or execution_mode" 2>&1 | grep -E "^\{|passed|failed|Error|error" | cut -c1-400 | tee $O/fullsize.txt
timeout 2400 python -m pytest tests/test_gpu_multirank.py -q -x 2>&1 | tail -5 | tee $O/multirank.txt
share() { tag=$1; shift; env "$@" timeout 600 python bench.py --config col2 --steps 8 --warmup 3 --tile-batch 1 --no-cpu-baseline --no-roofline --full-panorama 0 --wide-step 0 $SHARE_ARGS 2> $O/share_$tag.err | tail -1 > $O/share_$tag.json; echo "share $tag: $(grep -o '"ms_per_step": [0-9.]*' $O/share_$tag.json) $(grep -o '"latent_after_timed_steps": "[0-9a-f]*"' $O/share_$tag.json)" | tee -a $O/summary.txt; }
for rep in a b; do
  SHARE_ARGS="--streams 2" share hint_$rep DS_SPLIT_CFG=2 DS_SHARE_LAUNCHES=1
  SHARE_ARGS="--streams 2" share nohint_$rep DS_SPLIT_CFG=2 DS_SHARE_LAUNCHES=0
  SHARE_ARGS="--streams 1" share pair1s_$rep DS_SPLIT_CFG=0
  SHARE_ARGS="--streams 1" share pair1s_nosplit_$rep DS_SPLIT_CFG=0 DS_HIP_LIBRARY=$PWD/dynamicscaler_amd/libdynscaler_hip_tune.so DS_GEMM_TAIL_SPLIT=0
done
( time timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; echo "bench rc=$? $(grep real $O/bench_time.txt)" | tee -a $O/summary.txt
cut -c1-300 $O/bench_default.json | tee -a $O/summary.txt
