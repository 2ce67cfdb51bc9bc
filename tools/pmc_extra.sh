#!/bin/bash
# usage: tools/pmc_extra.sh <outdir> <counters...>   -- one rocprofv3 --pmc pass over tools/quick_bench.py --extra
out=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d /root/repo/gpurun_out/$out -- python3 /root/repo/tools/quick_bench.py --extra > /root/repo/gpurun_out/$out.log 2>&1
echo "pmc $out rc=$?"
