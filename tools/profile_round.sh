#!/bin/bash
# usage: tools/profile_round.sh  -- everything profiles/ holds for a round, from ONE device:
# bench line, rocprofv3 kernel trace + stats of the same command, the two HBM PMC passes, the SQ counter passes
repo=$(cd "$(dirname "$0")/.." && pwd)
cd $repo
rm -f gpurun_out/pr_csrc_sha16.txt
# (the counter summaries bench.py quotes are stamped with the kernel sources' hash: on the first pass of a round they do not exist yet and the line says traffic: null;
#  tools/profiles_from_round.py writes them, a second bench run -- tools/gpu_check.sh -- then quotes them)
timeout -k 10 400 python bench.py > gpurun_out/pr_bench.json 2> gpurun_out/pr_bench.err || exit 1
cp bench_legs.json gpurun_out/pr_bench_legs.json
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $repo/gpurun_out/pr_trace -- python3 $repo/bench.py --cpu-frames 0 > $repo/gpurun_out/pr_trace.log 2>&1) || exit 1
python3 tools/summarize_prof.py gpurun_out/pr_trace > gpurun_out/pr_kernel_trace.txt
tools/pmc_bench.sh pr_fetch FETCH_SIZE && tools/pmc_bench.sh pr_write WRITE_SIZE || exit 1
tools/pmc_bench.sh pr_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE || exit 1
tools/pmc_bench.sh pr_sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS || exit 1
python3 tools/pmc_summary.py gpurun_out/pr_sq1_bench gpurun_out/pr_sq2_bench > gpurun_out/pr_sq_counters.txt
# the two summaries bench.py quotes (stamped with the kernel sources' hash): profiles/<round>_hbm_traffic.json, <round>_pixel_pipes.json
python3 tools/pmc_round.py ${ROUND:-r06} gpurun_out/pr_fetch_bench gpurun_out/pr_fetch_micro gpurun_out/pr_write_bench gpurun_out/pr_write_micro \
    gpurun_out/pr_sq1_bench gpurun_out/pr_sq2_bench > gpurun_out/pr_round.txt || exit 1
(cd gpurun_out && find pr_* -name '*.csv' | sort) > gpurun_out/pr_files.txt   # this run's files: gpurun MERGES into the caller's gpurun_out/, older runs' stay there
tools/pmc_flops.sh ${ROUND:-r06} > gpurun_out/pr_flops.log 2>&1 || exit 1      # executed f32 operations per unit -> gpurun_out/<round>_fp32_flops.json
python3 -c "from bench import csrc_sha16; print(csrc_sha16())" > gpurun_out/pr_csrc_sha16.txt   # what tools/profiles_from_round.py checks the tree against
echo profile round done
