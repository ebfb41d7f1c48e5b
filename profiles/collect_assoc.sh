#!/bin/bash
# Evidence for the association kernel on the GPU box:  gpurun --timeout 900 -- 'bash profiles/collect_assoc.sh r02'
# bench cfg3, rocprofv3 kernel trace and SQ counters of the same command, the phase timeline (exp/assoc_trace.py) and
# the oracle sweeps of both kernel forms; small summaries land in gpurun_out/<round>/assoc/ (copy to profiles/<round>/).
set -o pipefail
R=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R/assoc
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py --config cfg3 --no-cpu-baseline > "$OUT/bench_cfg3.json" 2> "$OUT/bench_cfg3.err" || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python "$ROOT/bench.py" --config cfg3 --no-cpu-baseline > "$OUT/trace.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_valu" -- python "$ROOT/bench.py" --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_valu.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/pmc_lds" -- python "$ROOT/bench.py" --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/pmc_lds.log" 2>&1 || exit 1
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
stats = sorted(glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime)
if stats:
    rows = list(csv.reader(open(stats[-1])))
    with open(os.path.join(out, 'kernel_stats_cfg3.csv'), 'w') as fh:
        csv.writer(fh).writerows(rows)
    for r in rows[:4]:
        print(r)
acc = {}
for sub in ('pmc_valu', 'pmc_lds'):
    paths = sorted(glob.glob(os.path.join(out, sub, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)
    for path in paths[-1:]:
        for row in csv.DictReader(open(path)):
            if 'p2s_assoc' not in row['Kernel_Name']:
                continue
            key = (row['Counter_Name'], row['Dispatch_Id'])
            acc[key] = acc.get(key, 0.0) + float(row['Counter_Value'])
per = {}
for (name, _), v in acc.items():
    per.setdefault(name, []).append(v)
with open(os.path.join(out, 'pmc_assoc_cfg3.csv'), 'w') as fh:
    fh.write('counter,average_per_launch,launches\n')
    for name in sorted(per):
        fh.write('%s,%.0f,%d\n' % (name, sum(per[name]) / len(per[name]), len(per[name])))
        print(name, '%.4g' % (sum(per[name]) / len(per[name])))
PY
rm -rf "$OUT/trace" "$OUT/pmc_valu" "$OUT/pmc_lds"
python exp/assoc_trace.py > "$OUT/assoc_trace.log" 2>&1
tail -3 "$OUT/assoc_trace.log"
python tests/sweeps/sweep_assoc.py > "$OUT/sweep_assoc.log" 2>&1
P2S_SWEEP_FORM=general python tests/sweeps/sweep_assoc.py > "$OUT/sweep_assoc_general.log" 2>&1
grep frames "$OUT/sweep_assoc.log" "$OUT/sweep_assoc_general.log"
cat "$OUT/bench_cfg3.json"
