#!/bin/bash
# usage: tools/pmc_groups.sh <tag> <python script and args...>
# the four SQ / LDS / cache counter groups over `python3 <script> <args>`, one rocprofv3 --pmc pass per group (counters only:
# never combined with tracing on this pool); then tools/pmc_lds.py over every pass
tag=$1; shift
repo=$(cd "$(dirname "$0")/.." && pwd)
script=$1; shift
case $script in /*) ;; *) script=$PWD/$script;; esac     # (pmc_cmd.sh runs the command from /tmp)
set -- "$script" "$@"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i + 1))
  $repo/tools/pmc_cmd.sh ${tag}_g$i "$grp" "$@" || { echo "pass $i failed"; tail -5 $repo/gpurun_out/${tag}_g$i.log; exit 1; }
done
for g in 1 2 3 4; do python3 $repo/tools/pmc_lds.py $repo/gpurun_out/${tag}_g$g; done
