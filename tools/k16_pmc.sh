#!/bin/bash
# usage (GPU box, repo root): tools/k16_pmc.sh [hops]  -- counter passes (separate rocprofv3 --pmc runs, counters only) over tools/k16_ab.py:
# both 16384-point kernels in one process, HBM bytes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE) and the SQ / TA / TCP picture of each
hops=${1:-20000}
set -o pipefail
tools/pmc_cmd.sh k16pmc_fetch "FETCH_SIZE" $PWD/tools/k16_ab.py $hops 1 8 || exit 1
tools/pmc_cmd.sh k16pmc_write "WRITE_SIZE" $PWD/tools/k16_ab.py $hops 1 8 || exit 1
tools/pmc_cmd.sh k16pmc_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" $PWD/tools/k16_ab.py $hops 1 8 || exit 1
tools/pmc_cmd.sh k16pmc_sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" $PWD/tools/k16_ab.py $hops 1 8 || exit 1
cat gpurun_out/k16pmc_summary.txt
