#!/bin/bash
# Builds an experimental copy of the C-ABI library with extra flags on ONE translation unit:
#   scripts/build_variant.sh NAME FILE.hip "FLAGS"   ->  libtsd_amd/lib/variants/libtsdgpu_NAME.so
# Run it with TSDGPU_LIB=libtsd_amd/lib/variants/libtsdgpu_NAME.so (developer switch of capi.py).
set -e
cd "$(dirname "$0")/../libtsd_amd/csrc"
name=$1; unit=$2; flags=$3
make -s
mkdir -p build/variants ../lib/variants
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None"
case $unit in ols.hip|fft.hip|ols_long.hip) base="$base -fno-slp-vectorize -ffp-contract=fast";; esac
/opt/rocm/bin/hipcc $base $flags -c $unit -o build/variants/${name}.o
objs=""
for f in common fir ols ols_long fft sos resample polyphase ola sharded detect vecops; do
  if [ "$f.hip" == "$unit" ]; then objs="$objs build/variants/${name}.o"; else objs="$objs build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libtsdgpu_${name}.so $objs
echo "built libtsd_amd/lib/variants/libtsdgpu_${name}.so"
