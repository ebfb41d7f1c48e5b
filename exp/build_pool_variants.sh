#!/bin/bash
# Diagnostics: variants of the pooled triangulation kernel with extra compiler flags as separate libraries for exp/ab_pool.sh.
#   bash exp/build_pool_variants.sh NAME "flags" [NAME2 "flags2"] ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/pose2sim_amd/csrc
OBJS=$(ls $CSRC/_build/*.o | grep -v p2s_tri_pool.o)
mkdir -p $ROOT/exp/bin
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread $flags -I $ROOT/include -I $CSRC -c $CSRC/p2s_tri_pool.hip -o /tmp/pool_$name.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $ROOT/exp/bin/libp2s_$name.so $OBJS /tmp/pool_$name.o
  echo built $name "($flags)"
done
