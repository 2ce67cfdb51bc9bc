#!/bin/bash
# usage (GPU box, repo root): tools/pmc_flops.sh <round>  -- the f32 operations the BASELINE kernels execute per unit, by the SQ's
# instruction counters (one rocprofv3 --pmc pass, counters only) -> gpurun_out/<round>_fp32_flops.json (copy it to profiles/)
round=${1:-r06}
tools/pmc_cmd.sh fp32_flops "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU SQ_WAVES" $PWD/tools/fp32_legs.py || exit 1
python3 tools/pmc_flops_summary.py gpurun_out/fp32_flops gpurun_out/fp32_legs_plan.json > gpurun_out/${round}_fp32_flops.json || exit 1
cat gpurun_out/${round}_fp32_flops.json
