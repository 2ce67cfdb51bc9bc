#!/bin/bash
# usage: tools/pmc_bench.sh <outdir> <counter>  -- one rocprofv3 --pmc pass over bench.py (short) and the microbench
out=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d /root/repo/gpurun_out/${out}_bench -- python3 /root/repo/bench.py --steps 3 --warmup 1 --cpu-frames 0 --pixel-frames 65536 > /root/repo/gpurun_out/${out}_bench.log 2>&1
echo "pmc bench $out rc=$?"
timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d /root/repo/gpurun_out/${out}_micro -- /root/repo/tools/bin/microbench > /root/repo/gpurun_out/${out}_micro.log 2>&1
echo "pmc micro $out rc=$?"
