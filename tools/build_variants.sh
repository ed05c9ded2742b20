#!/bin/bash
# dev: builds of libicp_mi355x.so with different build-time knobs of the kNN selection path, for A/B runs on the GPU box
# usage: tools/build_variants.sh name:"-DKNN_W1=4 -DKNN_PREFETCH=0" ...   -> tools/variants/libicp_<name>.so
set -eu
ROOT=$(cd $(dirname $0)/.. && pwd); C=$ROOT/pylidar-slam_amd/csrc; O=$ROOT/tools/variants; mkdir -p $O
make -C $C -j8 > /dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$C -Wall -Wno-unused-function -Wno-unused-result -ffp-contract=off"
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c $C/search.hip -o $O/search_$name.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $C/build/api.o $C/build/hash_grid.o $O/search_$name.o $C/build/gauss_newton.o $C/build/projection.o $C/build/grid_sample.o $C/build/projective.o -o $O/libicp_$name.so && rm $O/search_$name.o && echo built $name ) &
done
wait
