#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>  -> variants/lib_<name>.so
# Rebuilds the fused STFT kernel's translation units with the extra flags and links them with the other objects of
# the regular build (run make first); A/B the results with tools/ab_libs.sh.
#   VARIANT_GEOMS="32_16_25_4" (default): the dispatcher (-DPDS_DEV_ONLY512) and that geometry alone -- seconds;
#   such a library serves the 512-point, 25-row plans only (the headline and configs[2..3] workloads)
#   VARIANT_GEOMS=all: every geometry of stft_geoms.def (as many parallel jobs as cores)
#   VARIANT_SRC=post.hip (or another non-STFT unit): that unit alone, the STFT units from the regular build
set -e
cd /root/repo/pydrobert-speech_amd/csrc
name=$1; shift
mkdir -p /root/repo/variants /tmp/variant_$name
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=fast -fno-slp-vectorize"
others="capi stft_generic post pre si si_fft comm feed"
if [ -n "$VARIANT_SRC" ]; then
  u=${VARIANT_SRC%.hip}
  /opt/rocm/bin/hipcc $flags "$@" -c $VARIANT_SRC -o /tmp/variant_$name/$u.o
  objs="/tmp/variant_$name/$u.o stft_fast.o $(ls geom_*.o)"
  for o in $others; do [ $o = $u ] || objs="$objs $o.o"; done
else
  geoms=${VARIANT_GEOMS:-32_16_25_4}
  dev="-DPDS_DEV_ONLY512"
  if [ "$geoms" = all ]; then
    dev=""
    geoms=$(/opt/rocm/bin/hipcc -E -P -x c++ "$@" stft_geoms.def 2>/dev/null | sed -n 's/^PDS_GEOM(\([0-9]*\), *\([0-9]*\), *\([0-9]*\), *\([0-9]*\)).*/\1_\2_\3_\4/p')
  fi
  /opt/rocm/bin/hipcc $flags -fno-signed-zeros $dev "$@" -c stft_fast.hip -o /tmp/variant_$name/stft_fast.o &
  for g in $geoms; do
    IFS=_ read n1 n2 rows minw <<< "$g"
    /opt/rocm/bin/hipcc $flags -fno-signed-zeros $dev "$@" -DPDS_G_N1=$n1 -DPDS_G_N2=$n2 -DPDS_G_ROWS=$rows -DPDS_G_MINW=$minw \
      -c stft_geom.hip -o /tmp/variant_$name/geom_$g.o &
    while [ $(jobs -r | wc -l) -ge $(nproc) ]; do sleep 0.2; done
  done
  wait
  objs="$(ls /tmp/variant_$name/*.o)"
  for o in $others; do objs="$objs $o.o"; done
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -o /root/repo/variants/lib_$name.so
echo built variants/lib_$name.so
