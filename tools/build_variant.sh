#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags for stft_fast.hip...>  -> gpurun_out/variants/lib_<name>.so
set -e
cd /root/repo/pydrobert-speech_amd/csrc
name=$1; shift
mkdir -p /root/repo/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=fast -fno-signed-zeros "$@" -c stft_fast.hip -o /tmp/stft_fast_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC capi.o stft_generic.o /tmp/stft_fast_$name.o post.o pre.o si.o si_fft.o -o /root/repo/variants/lib_$name.so
echo built variants/lib_$name.so
