#!/bin/bash
# cProfile of the sphere loop at gen_pano_360's stage-1 geometry (tools/bench_sphere.py): where the host time goes
O=gpurun_out/${1:-sphere}; mkdir -p $O
M=${2:-t2v}; S=${3:-13}
timeout 1200 python -c "
import cProfile, pstats, sys, io
sys.argv = ['bench_sphere.py', '--model', '$M', '--steps', '$S']
pr = cProfile.Profile()
pr.enable()
exec(compile(open('tools/bench_sphere.py').read(), 'tools/bench_sphere.py', 'exec'), {'__file__': 'tools/bench_sphere.py', '__name__': '__main__'})
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
open('$O/cprofile_$M.txt', 'w').write(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(30)
open('$O/cprofile_${M}_tottime.txt', 'w').write(s.getvalue())
" 2>&1 | tail -2 | tee $O/line_$M.txt
