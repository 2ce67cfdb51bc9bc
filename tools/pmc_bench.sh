#!/bin/bash
# usage: tools/pmc_bench.sh <outdir> <counters...>  -- one rocprofv3 --pmc pass over bench.py (short: 3 timed steps, no sustain
# windows, one placement, no CPU leg; configs 2, 3 and 4 -- at 20 000 hop positions -- the (l, r) and the paired legs all run) and one
# over the microbench (known byte counts: the calibration)
out=$1; shift
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $repo/gpurun_out/${out}_bench -- python3 $repo/bench.py --steps 3 --warmup 1 --sustain-s 0 --leg-sustain-s 0 --placements 1 --config4-hops 20000 --complex-frames 0 --app-frames 0 --cpu-frames 0 > $repo/gpurun_out/${out}_bench.log 2>&1
rc=$?
echo "pmc bench $out rc=$rc"
if [ -x $repo/tools/bin/microbench ]; then
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $repo/gpurun_out/${out}_micro -- $repo/tools/bin/microbench > $repo/gpurun_out/${out}_micro.log 2>&1
  rc2=$?
  echo "pmc micro $out rc=$rc2"
  [ $rc -eq 0 ] && rc=$rc2
fi
exit $rc   # a failed or timed-out profiler pass is the script's own status
