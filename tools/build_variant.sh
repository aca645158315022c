#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>  -> variants/lib_<name>.so
# Rebuilds ONE translation unit (VARIANT_SRC, default stft_fast.hip) with the extra flags and links
# it with the objects of the regular build (run make first); A/B the results with tools/ab_libs.sh.
set -e
cd /root/repo/pydrobert-speech_amd/csrc
name=$1; shift
src=${VARIANT_SRC:-stft_fast.hip}
mkdir -p /root/repo/variants
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=fast -fno-slp-vectorize"
[ "$src" = stft_fast.hip ] && flags="$flags -fno-signed-zeros"
/opt/rocm/bin/hipcc $flags "$@" -c $src -o /tmp/variant_$name.o
objs=""
for o in capi stft_generic stft_fast post pre si si_fft comm; do
  if [ "$o.hip" = "$src" ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o /root/repo/variants/lib_$name.so
echo built variants/lib_$name.so
