#!/bin/bash
# usage: [SRC=file.hip] tools/build_variant.sh <name> <extra hipcc flags...> -- builds spectrogram_rs_amd/ab/<name>.so with
# one kernel file (default stft4096_wg.hip) recompiled under the extra flags (ablation / A-B builds).  A variant that issues add-TID LDS stores
# goes through the same ISA check as the product.
set -e
name=$1; shift
cd "$(dirname "$0")/../spectrogram_rs_amd/csrc"
make -s
mkdir -p ../ab build/ab
FLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result --offload-arch=gfx950 -munsafe-fp-atomics"
/opt/rocm/bin/hipcc $FLAGS "$@" -c ${SRC:-stft4096_wg.hip} -o build/ab/$name.o
/opt/rocm/bin/hipcc $FLAGS "$@" -S --cuda-device-only ${SRC:-stft4096_wg.hip} -o build/ab/$name.s 2>/dev/null
if grep -q addtid build/ab/$name.s; then python3 ../../tools/isa_check_addtid.py build/ab/$name.s; fi
objs=$(ls build/*.o | grep -v ${SRC:-stft4096_wg.hip}.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -o ../ab/$name.so $objs build/ab/$name.o
echo built ../ab/$name.so
