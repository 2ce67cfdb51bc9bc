#!/bin/bash
# usage: [REPS=n] tools/ab_k1.sh -- tools/k1_ab.py under every library in spectrogram_rs_amd/ab/, interleaved, on ONE device
for rep in $(seq 1 ${REPS:-2}); do
  for lib in spectrogram_rs_amd/ab/*.so; do
    case $lib in *pitch*) p=1;; *) p=0;; esac
    SGX_AB_PITCH16K=$p SGX_LIB=$PWD/$lib timeout -k 10 200 python tools/k1_ab.py 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
