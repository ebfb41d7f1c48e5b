#!/bin/bash
# VALU / LDS / wait counters of the triangulation kernel for bench.py variants, summarised per kernel.
# usage (GPU box): bash profiles/pmc_tri.sh out_dir "<bench args 1>" "<bench args 2>" ...
out=$1; shift
mkdir -p $out
ROOT=$(pwd)
export TMPDIR=/tmp
i=0
for args in "$@"; do
  i=$((i+1))
  cd /tmp
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $ROOT/$out/pmc_$i -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline $args > $ROOT/$out/pmc_$i.log 2>&1 || exit 1
  cd $ROOT
  python - $out/pmc_$i "$args" <<'PY'
import csv, glob, os, sys
d, args = sys.argv[1], sys.argv[2]
acc = {}
for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(path)):
        if 'p2s_tri' not in row['Kernel_Name']:
            continue
        k = (row['Kernel_Name'].split('(')[0][-40:], row['Counter_Name'], row['Dispatch_Id'])
        acc[k] = acc.get(k, 0.0) + float(row['Counter_Value'])
per = {}
for (kn, cn, _), v in acc.items():
    per.setdefault((kn, cn), []).append(v)
print('##', args)
kernels = sorted({k for k, _ in per})
for kn in kernels:
    g = {cn: sum(v) / len(v) for (k, cn), v in per.items() if k == kn}
    w = max(g.get('SQ_WAVES', 1), 1)
    print(kn, 'waves %.0f' % w, 'VALU/wave %.1f' % (g['SQ_INSTS_VALU'] / w), 'SALU/wave %.1f' % (g['SQ_INSTS_SALU'] / w),
          'LDS/wave %.1f' % (g['SQ_INSTS_LDS'] / w), 'valu-active cycles/wave %.0f' % (4 * g['SQ_ACTIVE_INST_VALU'] / w),
          'wave cycles %.0f' % (4 * g['SQ_WAVE_CYCLES'] / w))
PY
  rm -rf $out/pmc_$i
done
