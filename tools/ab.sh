#!/bin/bash
# usage: [REPS=n] [BENCH=tools/quick_bench.py] tools/ab.sh <bench args...>
# runs $BENCH (quick_bench.py: configs 2 / 4 and other windows; pixel_bench.py: config 3; stereo_bench.py; app_bench.py: W 2400 / 2205) under
# every library in spectrogram_rs_amd/ab/ (A/B builds, tools/build_variant.sh), REPS times (default 2), interleaved, on the SAME
# device: device-to-device variance on the pool is larger than most kernel deltas
for rep in $(seq 1 ${REPS:-2}); do
  for lib in spectrogram_rs_amd/ab/*.so; do
    echo "== $lib (rep $rep)"
    SGX_LIB=$PWD/$lib timeout -k 10 200 python ${BENCH:-tools/quick_bench.py} "$@" || exit 1
  done
done
