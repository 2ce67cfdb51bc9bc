#!/bin/bash
# usage: tools/pmc_clock.sh <lib.so> <tag>   -- GRBM_GUI_ACTIVE (busy cycles) of the STFT kernel under one library build
cd /tmp && export TMPDIR=/tmp
SGX_LIB=$1 timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /root/repo/gpurun_out/clk_$2 -- python3 /root/repo/tools/quick_bench.py --frames 1000000 > /root/repo/gpurun_out/clk_$2.log 2>&1
echo "clk $2 rc=$?"
grep "stft ch" /root/repo/gpurun_out/clk_$2.log
