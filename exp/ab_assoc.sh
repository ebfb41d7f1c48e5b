#!/bin/bash
# A/B of association-kernel variants on one box: cfg3 bench (twice, interleaved) and the oracle sweep for each library.
# usage: exp/ab_assoc.sh out_dir lib1.so lib2.so ...   ("default" = the shipped library)
out=$1; shift
mkdir -p $out
for rep in 1 2; do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    if [ "$lib" = default ]; then unset P2S_LIB; else export P2S_LIB=$lib; fi
    python bench.py --config cfg3 --no-cpu-baseline > $out/cfg3_${tag}_$rep.json 2> $out/err_${tag}_$rep.log || exit 1
    python -c "import json,sys; d=json.loads(open('$out/cfg3_${tag}_$rep.json').read().strip().splitlines()[-1]); print('$tag', $rep, round(d['ms_per_step'],2), 'ms')"
  done
done
for lib in "$@"; do
  tag=$(basename $lib .so)
  if [ "$lib" = default ]; then unset P2S_LIB; else export P2S_LIB=$lib; fi
  python tests/sweeps/sweep_assoc.py > $out/sweep_${tag}.log 2>&1
  echo "$tag: $(grep -c 'frames with different proposals' $out/sweep_${tag}.log) workloads; worst: $(grep -o 'max |d affinity| [0-9.e+-]*' $out/sweep_${tag}.log | sort -g -k4 | tail -1); differing: $(grep -o 'different proposals [0-9]*' $out/sweep_${tag}.log | tr '\n' ' ')"
done
