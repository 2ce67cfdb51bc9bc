#!/bin/bash
# usage: [REPS=n] tools/ab.sh <quick_bench args...> -- runs every library under spectrogram_rs_amd/ab/ (A/B builds) REPS times
# (default 2), interleaved, on the SAME device (device-to-device variance on the pool is larger than most kernel deltas)
for rep in $(seq 1 ${REPS:-2}); do
  for lib in spectrogram_rs_amd/ab/*.so; do
    echo "== $lib (rep $rep)"
    SGX_LIB=$PWD/$lib timeout -k 10 200 python tools/quick_bench.py "$@" || exit 1
  done
done
