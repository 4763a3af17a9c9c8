#!/bin/bash
# build_variant.sh NAME [hipcc flags...]: an experimental libgmr_amd variant under gmr_amd/lib/variants/libNAME.so
# (picked up with GMR_AMD_LIB=...; see tools/gpu_quick.sh).  Prints register / scratch use of ik_kernel<36,true>.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
mkdir -p $R/gmr_amd/lib/variants /tmp/gmr_variant_$N
cd /tmp/gmr_variant_$N
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I$R/include -Wno-unused-value --save-temps -DGMR_IK_VARIANTS "$@" -o $R/gmr_amd/lib/variants/lib$N.so $R/gmr_amd/csrc/api.hip 2>/dev/null
awk '/^_ZN3gmr9ik_kernelILi36ELb1E.*:/{f=1} f&&/; (NumVgprs|ScratchSize|Occupancy|codeLenInByte)/{printf "%s ", $0} f&&/Occupancy/{print ""; exit}' api-hip-amdgcn-amd-amdhsa-gfx950.s | sed "s/^/$N: /"
