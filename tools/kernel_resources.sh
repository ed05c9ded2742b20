#!/bin/bash
# VGPRs / spills / scratch / LDS / occupancy of every kernel of one HIP source (compile-time remarks; no GPU needed)
# usage: tools/kernel_resources.sh search.hip [name-filter]
R=$(cd "$(dirname "$0")/.." && pwd); S=${1:-search.hip}; F=${2:-.}
cd $R/pylidar-slam_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -ffp-contract=off -fno-slp-vectorize \
  -Rpass-analysis=kernel-resource-usage -c $S -o /tmp/kr_$$.o 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//' | \
  awk '/Function Name:/{n=$NF} / VGPRs:/{v=$NF} /ScratchSize/{s=$NF} /Occupancy/{o=$NF} /VGPRs Spill/{sp=$NF} /LDS Size/{print "vgpr="v, "spill="sp, "scratch="s, "occ="o, "lds="$NF, n}' | \
  c++filt | cut -c1-110 | grep -E "$F"
rm -f /tmp/kr_$$.o
