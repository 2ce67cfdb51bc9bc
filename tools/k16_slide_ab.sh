#!/bin/bash
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -x -k "config4_16384" > gpurun_out/slide_tests.log 2>&1
rc=$?
tail -5 gpurun_out/slide_tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  for lib in spectrogram_rs_amd/libsgx.so spectrogram_rs_amd/ab/k16_noslide.so; do
    echo "== $lib (rep $rep)"
    SGX_LIB=$PWD/$lib timeout -k 10 200 python tools/k16_ab.py 100000 5 8 || exit 1
  done
done > gpurun_out/slide_ab.log 2>&1
grep -E "==|median" gpurun_out/slide_ab.log
for ch in 2 1; do for lib in spectrogram_rs_amd/libsgx.so spectrogram_rs_amd/ab/k16_noslide.so; do echo "== $lib ch $ch"; SGX_LIB=$PWD/$lib timeout -k 10 200 python tools/k16_ab.py 100000 5 $ch || exit 1; done; done > gpurun_out/slide_ab2.log 2>&1
grep -E "==|median|worst" gpurun_out/slide_ab2.log
