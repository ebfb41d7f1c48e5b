#!/bin/bash
# Collect the round's evidence on the GPU box:  gpurun --timeout 1200 -- 'bash profiles/collect.sh r02'
# Writes raw files under gpurun_out/<round>/, summarises them there (profiles/summarize.py) into gpurun_out/<round>/final/
# -- copy that to profiles/<round>/ and its traffic.json to profiles/traffic.json afterwards.
set -o pipefail
R=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench_cfg2_default.json" 2> "$OUT/bench_cfg2_default.err"
for cfg in cfg2_clean cfg3 cfg4 cfg5_tenth single; do
  timeout -k 10 400 python bench.py --config $cfg --no-cpu-baseline > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python "$ROOT/bench.py" --no-cpu-baseline > "$OUT/trace.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_valu" -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/pmc_valu.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_wait" -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/pmc_wait.log" 2>&1
cd "$ROOT"
# summarise on the box and drop the raw rocprofv3 output (gpurun brings back at most 64 MiB)
python profiles/summarize.py "$R" > "$OUT/summary.log" 2>&1
mkdir -p "$OUT/final" && cp -r profiles/$R/. "$OUT/final/" && cp profiles/traffic.json "$OUT/final/traffic.json"
rm -rf "$OUT/trace" "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_valu "$OUT"/pmc_wait
tail -5 "$OUT/summary.log"
cat "$OUT/bench_cfg2_default.json"
