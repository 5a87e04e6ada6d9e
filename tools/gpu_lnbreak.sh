#!/bin/bash
O=gpurun_out/lnbreak; mkdir -p $O
DS_BENCH_BREAKDOWN=$O/shape_base.csv timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --full-panorama 0 > $O/rb.json 2> $O/rb.err
DS_FOLD_LN=1 DS_BENCH_BREAKDOWN=$O/shape_ln.csv timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --full-panorama 0 > $O/rv.json 2> $O/rv.err
python - $O <<'PY' | tee $O/summary.txt
import csv,sys
O=sys.argv[1]
b={(r['kernel'],r['shape']):r for r in csv.DictReader(open(f"{O}/shape_base.csv"))}
v={(r['kernel'],r['shape']):r for r in csv.DictReader(open(f"{O}/shape_ln.csv"))}
tb=sum(float(r['ms_per_step']) for r in b.values()); tv=sum(float(r['ms_per_step']) for r in v.values())
print("sum base",round(tb,1),"ln",round(tv,1))
keys=sorted(set(b)|set(v), key=lambda k:-abs(float(v.get(k,{'ms_per_step':0})['ms_per_step'])-float(b.get(k,{'ms_per_step':0})['ms_per_step'])))
for k in keys[:20]:
    print(k, "base", b.get(k,{}).get('ms_per_step'), b.get(k,{}).get('launches'), "ln", v.get(k,{}).get('ms_per_step'), v.get(k,{}).get('launches'))
PY
