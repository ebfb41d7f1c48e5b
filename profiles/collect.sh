#!/bin/bash
# Collect the round's evidence on the GPU box:  gpurun -- 'bash profiles/collect.sh r01'
# Writes raw files under gpurun_out/<round>/ ; profiles/summarize.py turns them into profiles/<round>/.
set -eo pipefail
R=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench_cfg2_default.json" 2> "$OUT/bench_cfg2_default.err"
for cfg in cfg2_clean cfg3 cfg4 cfg5 single; do
  timeout -k 10 300 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python "$ROOT/bench.py" --no-cpu-baseline > "$OUT/trace.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_valu" -- python "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/pmc_valu.log" 2>&1
cd "$ROOT"
python profiles/summarize.py "$R" > "$OUT/summary.log" 2>&1
tail -5 "$OUT/summary.log"
cat "$OUT/bench_cfg2_default.json"
