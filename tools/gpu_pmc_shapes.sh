#!/bin/bash
# SQ / LDS / HBM counter passes on the top shapes of a step (tools/pmc_shapes.py), summarised by tools/pmc_sq_summary.py
S=${1:-pmcs}; R=$PWD; O=$R/gpurun_out/$S; mkdir -p $O
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_available.txt 2>&1
pick() { python3 - "$@" <<'PY'
import re,sys
avail=set(re.findall(r"\b((?:SQ|GRBM|TCC|TCP|TA)_[A-Z0-9_]+|FETCH_SIZE|WRITE_SIZE)\b", open(sys.argv[1]).read()))
print(" ".join(c for c in sys.argv[2:] if c in avail))
PY
}
A=$(pick $O/counters_available.txt SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE)
B=$(pick $O/counters_available.txt SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM)
echo "pass A: $A" | tee -a $O/summary.txt; echo "pass B: $B" | tee -a $O/summary.txt
timeout 900 rocprofv3 --pmc $A --output-format csv -d $O/pmcA -- python3 $R/tools/pmc_shapes.py > $O/pmcA.log 2>&1; echo "A rc=$?" | tee -a $O/summary.txt
timeout 900 rocprofv3 --pmc $B --output-format csv -d $O/pmcB -- python3 $R/tools/pmc_shapes.py > $O/pmcB.log 2>&1; echo "B rc=$?" | tee -a $O/summary.txt
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcF -- python3 $R/tools/pmc_shapes.py > $O/pmcF.log 2>&1; echo "F rc=$?" | tee -a $O/summary.txt
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcW -- python3 $R/tools/pmc_shapes.py > $O/pmcW.log 2>&1; echo "W rc=$?" | tee -a $O/summary.txt
cd $R
fa=$(find $O/pmcA -name "*counter_collection.csv" | head -1); fb=$(find $O/pmcB -name "*counter_collection.csv" | head -1)
ff=$(find $O/pmcF -name "*counter_collection.csv" | head -1); fw=$(find $O/pmcW -name "*counter_collection.csv" | head -1)
python3 tools/pmc_sq_summary.py $O/pmc_sq_summary.json $fa $fb $ff $fw > $O/pmc_sq_summary.log 2>&1; echo "summary rc=$?" | tee -a $O/summary.txt
find $O -name "*counter_collection.csv" -size +20M -delete; find $O -name "*.db" -delete
du -sh $O | tee -a $O/summary.txt
