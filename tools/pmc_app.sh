#!/bin/bash
# usage: tools/pmc_app.sh <tag>  -- SQ / LDS / cache counter passes over tools/app_bench.py (one --pmc pass per group)
tag=$1
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i + 1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $repo/gpurun_out/${tag}_g$i -- python3 $repo/tools/app_bench.py 65536 3 > $repo/gpurun_out/${tag}_g$i.log 2>&1 || { echo "pass $i failed"; tail -5 $repo/gpurun_out/${tag}_g$i.log; exit 1; }
done
for g in 1 2 3 4; do python3 $repo/tools/pmc_lds.py $repo/gpurun_out/${tag}_g$g | grep -i "mixed\|4800"; done
