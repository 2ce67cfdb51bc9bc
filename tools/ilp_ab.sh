#!/bin/bash
# usage (GPU box): tools/ilp_ab.sh -- every library in spectrogram_rs_amd/ab/ plus libsgx.so on config 4 (k16_ab.py), the (l, r) stream and the mono
# rows (stereo_bench.py) and config 3 (pixel_bench.py), interleaved on one device, two rounds
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in spectrogram_rs_amd/libsgx.so spectrogram_rs_amd/ab/*.so; do
    echo "== $lib (rep $rep)"
    SGX_LIB=$PWD/$lib timeout -k 10 200 python tools/k16_ab.py 100000 5 8 2>&1 | grep -E "median|worst" || exit 1
    SGX_LIB=$PWD/$lib timeout -k 10 200 python tools/stereo_bench.py 2>&1 | grep -E "real-input|l, r\) stream" || exit 1
    SGX_LIB=$PWD/$lib timeout -k 10 200 python tools/pixel_bench.py 2>&1 | grep -E "mono" || exit 1
  done
done > gpurun_out/ilp_ab.log 2>&1
cat gpurun_out/ilp_ab.log | cut -c1-230
