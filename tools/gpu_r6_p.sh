#!/bin/bash
# round 6, call P: texture-address / L1 counters of the step's top GEMM shapes (tools/pmc_shapes.py): is the short-K K loop bound by the
# CU's vector-memory path?  Three small --pmc passes (counter blocks hold few counters at a time), the program directly after "--".
R=$PWD; O=$R/gpurun_out/r6_p; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 $R/tools/pmc_shapes.py > $O/p1.log 2>&1; echo "p1 rc=$?" | tee -a $O/summary.txt
timeout 600 rocprofv3 --pmc TA_DATA_STALLED_BY_TC_CYCLES TA_BUFFER_READ_LDS_WAVEFRONTS GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python3 $R/tools/pmc_shapes.py > $O/p2.log 2>&1; echo "p2 rc=$?" | tee -a $O/summary.txt
timeout 600 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_GATE_EN GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -- python3 $R/tools/pmc_shapes.py > $O/p3.log 2>&1; echo "p3 rc=$?" | tee -a $O/summary.txt
cd $R
f1=$(find $O/p1 -name "*counter_collection.csv" | head -1); f2=$(find $O/p2 -name "*counter_collection.csv" | head -1); f3=$(find $O/p3 -name "*counter_collection.csv" | head -1)
python3 tools/pmc_sq_summary.py $O/r6_pmc_ta_tcp.json $f1 $f2 $f3 > $O/sum.log 2>&1; echo "summary rc=$?" | tee -a $O/summary.txt
find $O -name "*counter_collection.csv" -size +20M -delete; find $O -name "*.db" -delete
python3 - <<PY
import json
j=json.load(open("$O/r6_pmc_ta_tcp.json"))
for k,v in j.items():
    cyc=v.get("GRBM_GUI_ACTIVE",0)/8.0
    if not cyc: continue
    print(f"{k[:60]:60s} us {v['avg_us_profiled']:8.1f} TA busy {v.get('TA_TA_BUSY',0)/256/cyc:.2f} addr-stalled-by-TC {v.get('TA_ADDR_STALLED_BY_TC_CYCLES',0)/256/cyc:.2f} data-stalled-by-TC {v.get('TA_DATA_STALLED_BY_TC_CYCLES',0)/256/cyc:.2f} TCP pending-stall {v.get('TCP_PENDING_STALL_CYCLES',0)/256/cyc:.2f} tagconflict {v.get('TCP_READ_TAGCONFLICT_STALL_CYCLES',0)/256/cyc:.2f}")
PY
