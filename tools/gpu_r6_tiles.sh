#!/bin/bash
# round 6: forced-tile A/B incl. the split-ring two-per-CU forms (tile ids 6, 7), big batch (16 evaluations) and pair batch (2)
O=gpurun_out/r6_tiles; mkdir -p $O
timeout 1500 python tools/bench_tile_choice.py 16 > $O/tiles_e16.txt 2>&1
timeout 900 python tools/bench_tile_choice.py 2 > $O/tiles_e2.txt 2>&1
cat $O/tiles_e16.txt $O/tiles_e2.txt
