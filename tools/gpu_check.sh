#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_check.sh [pytest args...]  -- the GPU suite, then the default bench line.
# A step killed at its limit ends the call (no further GPU step after a timeout); a test FAILURE does not.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 ${TEST_LIMIT:-900} python -m pytest tests -m gpu -q "$@" > gpurun_out/gc_tests.log 2>&1
rc=$?
tail -15 gpurun_out/gc_tests.log
if [ $rc -ge 124 ]; then echo "pytest killed (rc $rc): stopping"; exit $rc; fi
timeout -k 10 ${BENCH_LIMIT:-500} python bench.py --steps 20 --warmup 5 > gpurun_out/gc_bench.json 2> gpurun_out/gc_bench.err
brc=$?
echo "bench rc $brc, line bytes: $(wc -c < gpurun_out/gc_bench.json)"
cat gpurun_out/gc_bench.json
exit $(( rc != 0 ? rc : brc ))
