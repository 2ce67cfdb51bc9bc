#!/bin/bash
# usage (GPU box): tools/k16_var_ab.sh [reps] -- the config-4 parity tests on libsgx.so, then every library in spectrogram_rs_amd/ab/ on
# config 4 (tools/k16_ab.py), interleaved on the same device
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -x -k "config4_16384" > gpurun_out/k16_var_tests.log 2>&1
rc=$?
tail -5 gpurun_out/k16_var_tests.log
[ $rc -ne 0 ] && exit $rc
for rep in $(seq 1 ${1:-3}); do
  for lib in spectrogram_rs_amd/ab/*.so; do
    echo "== $lib (rep $rep)"
    SGX_LIB=$PWD/$lib timeout -k 10 200 python tools/k16_ab.py 100000 5 8 || exit 1
  done
done > gpurun_out/k16_var_ab.log 2>&1
grep -E "==|median|worst" gpurun_out/k16_var_ab.log
