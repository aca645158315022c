#!/bin/bash
# tools/isa_dev.sh [extra hipcc flags]: compiles ONE geometry of the fused STFT kernel (ISA_GEOM="n1 n2 rows minw",
# default the headline's "32 16 25 4") with --save-temps into /tmp/isa and prints each kernel's resource usage.
set -e
mkdir -p /tmp/isa && cd /tmp/isa
read n1 n2 rows minw <<< "${ISA_GEOM:-32 16 25 4}"
src=/root/repo/pydrobert-speech_amd/csrc/stft_geom.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=fast \
  -fno-signed-zeros -fno-slp-vectorize -DPDS_G_N1=$n1 -DPDS_G_N2=$n2 -DPDS_G_ROWS=$rows -DPDS_G_MINW=$minw "$@" --save-temps -c $src -o /tmp/isa/dev.o 2>&1 | grep -v warning | head -20
s=/tmp/isa/stft_geom-hip-amdgcn-amd-amdhsa-gfx950.s
python3 - $s <<'PY'
import re, sys
name = None
for line in open(sys.argv[1]):
    m = re.match(r"^(_ZN3pds16stft_wave_kernel\w+):", line)
    if m:
        name = m.group(1).replace("_ZN3pds16stft_wave_kernelI", "").replace("EEvNS_10FastParamsE", "")
        name = name.replace("Li", "").replace("ELb", ",b").replace("E", ",")
        stats = {}
    for key in ("codeLenInByte", "NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "TotalNumSgprs"):
        m = re.match(r"^; %s[:=]? *=? *(\d+)" % key, line)
        if m and name:
            stats[key] = int(m.group(1))
            if key == "Occupancy":
                print(name, stats)
PY
