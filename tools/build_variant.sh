#!/bin/bash
# usage: [SRC=file.hip] tools/build_variant.sh <name> <extra hipcc flags...> -- builds spectrogram_rs_amd/ab/<name>.so with
# one kernel file (default stft4096_wg.hip) recompiled under the extra flags (ablation / A-B builds).  The variant's device assembly
# goes through the same 16-byte-store hazard check as the product (an A/B build that trips it computes garbage now and then).
set -e
name=$1; shift
cd "$(dirname "$0")/../spectrogram_rs_amd/csrc"
make -s
mkdir -p ../ab build/ab
FLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result --offload-arch=gfx950 -munsafe-fp-atomics"
/opt/rocm/bin/hipcc $FLAGS "$@" -c ${SRC:-stft4096_wg.hip} -o build/ab/$name.o
/opt/rocm/bin/hipcc $FLAGS "$@" -S --cuda-device-only ${SRC:-stft4096_wg.hip} -o build/ab/$name.s 2>/dev/null
if grep -q buffer_store_dwordx4 build/ab/$name.s; then python3 ../../tools/isa_check_store16.py build/ab/$name.s; fi
if grep -q addtid build/ab/$name.s; then python3 ../../tools/isa_check_addtid.py build/ab/$name.s; fi
objs=$(ls build/*.o | grep -v ${SRC:-stft4096_wg.hip}.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -o ../ab/$name.so $objs build/ab/$name.o
echo built ../ab/$name.so
