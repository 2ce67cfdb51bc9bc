#!/bin/bash
# usage: [SRC=file.hip] tools/build_variant.sh <name> <extra hipcc flags...> -- builds spectrogram_rs_amd/ab/<name>.so with
# one kernel file (default stft4096_wg.hip) recompiled under the extra flags (ablation / A-B builds)
set -e
name=$1; shift
cd "$(dirname "$0")/../spectrogram_rs_amd/csrc"
make -s
mkdir -p ../ab build/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result \
    --offload-arch=gfx950 -munsafe-fp-atomics "$@" -c ${SRC:-stft4096_wg.hip} -o build/ab/$name.o
objs=$(ls build/*.o | grep -v ${SRC:-stft4096_wg.hip}.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -o ../ab/$name.so $objs build/ab/$name.o
echo built ../ab/$name.so
