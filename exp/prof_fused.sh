#!/bin/bash
# Diagnostics on the GPU box: kernel trace + VALU counters of the cfg2 / cfg2_clean bench (fused triangulation kernel).
#   gpurun -- 'bash exp/prof_fused.sh s2 [extra bench.py arguments, e.g. --tri-path twotiles]'
set -o pipefail
R=${1:-s2}
shift
EXTRA="$@"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for cfg in cfg2 cfg2_clean; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$cfg" -- python "$ROOT/bench.py" --config $cfg --steps 50 --warmup 5 --no-cpu-baseline $EXTRA > "$OUT/trace_$cfg.log" 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_$cfg" -- python "$ROOT/bench.py" --config $cfg --steps 5 --warmup 1 --no-cpu-baseline $EXTRA > "$OUT/pmc_$cfg.log" 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc2_$cfg" -- python "$ROOT/bench.py" --config $cfg --steps 5 --warmup 1 --no-cpu-baseline $EXTRA > "$OUT/pmc2_$cfg.log" 2>&1
done
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for cfg in ('cfg2', 'cfg2_clean'):
    for f in glob.glob(f'{out}/trace_{cfg}/**/*kernel_stats.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if 'p2s' in row['Name']:
                print(cfg, 'kernel', row['Name'][:60], 'calls', row['Calls'], 'avg_ns', row['AverageNs'])
    for tag in ('pmc', 'pmc2'):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob(f'{out}/{tag}_{cfg}/**/*counter_collection.csv', recursive=True):
            for row in csv.DictReader(open(f)):
                if 'p2s' in row['Kernel_Name']:
                    a = acc[(row['Kernel_Name'][:40], row['Counter_Name'])]
                    a[0] += float(row['Counter_Value']); a[1] += 1
        for (k, c), (v, n) in sorted(acc.items()):
            print(cfg, tag, k, c, f'{v / n:.4g}', f'(n={n})')
PY
