#!/bin/bash
# Diagnostics: build variants of the fused triangulation kernel (compile-time switches) as separate libraries for
# exp/ab_bench.py.   bash exp/build_variants.sh NAME "-DFLAG1 -DFLAG2" [NAME2 "..."] ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/pose2sim_amd/csrc
OBJS=$(ls $CSRC/_build/*.o | grep -v p2s_tri_fused.o)
mkdir -p $ROOT/exp/bin
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread $flags -I $ROOT/include -I $CSRC -c $CSRC/p2s_tri_fused.hip -o /tmp/fused_$name.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $ROOT/exp/bin/libp2s_$name.so $OBJS /tmp/fused_$name.o
  echo built $name "($flags)"
done
