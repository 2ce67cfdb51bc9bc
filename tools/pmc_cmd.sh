#!/bin/bash
# usage: tools/pmc_cmd.sh <outdir> "<counters...>" <python script and args...>
# one rocprofv3 --pmc pass (counters only: never combined with tracing on this pool) over `python3 <script> <args>`
out=$1; ctrs=$2; shift 2
repo=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $repo/gpurun_out/$out -- python3 "$@" > $repo/gpurun_out/$out.log 2>&1
rc=$?
echo "pmc $out rc=$rc"
exit $rc
