#!/bin/bash
# usage: [REPS=n] tools/ab_app.sh [frames]  -- tools/app_bench.py under every library in spectrogram_rs_amd/ab/, interleaved, on ONE device
for rep in $(seq 1 ${REPS:-2}); do
  for lib in spectrogram_rs_amd/ab/*.so; do
    echo "== $lib (rep $rep)"
    SGX_LIB=$PWD/$lib timeout -k 10 200 python tools/app_bench.py "$@" || exit 1
  done
done
