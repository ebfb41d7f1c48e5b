#!/bin/bash
# Collect the round's evidence on the GPU box:  gpurun --timeout 1190 -- 'bash profiles/collect.sh r03'
# Writes raw files under gpurun_out/<round>/, summarises them there (profiles/summarize.py) into gpurun_out/<round>/final/
# -- copy that to profiles/<round>/ and its traffic.json to profiles/traffic.json afterwards.
# Per configuration with a triangulation kernel worth a roofline line (cfg2 = BASELINE configs[1], cfg4 = the configs[3]
# shard): kernel trace + stats, FETCH_SIZE and WRITE_SIZE in passes of their own, the SQ instruction counters; kernel
# traces only for cfg3 (association) and cfg5_tenth (32 cameras).
set -o pipefail
R=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench_cfg2_default.json" 2> "$OUT/bench_cfg2_default.err"
for cfg in cfg2_clean cfg3 cfg4 cfg5_tenth single; do
  timeout -k 10 400 python bench.py --config $cfg --no-cpu-baseline > "$OUT/bench_$cfg.json" 2> "$OUT/bench_$cfg.err"
done
cd /tmp
for cfg in cfg2 cfg4; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$cfg" -- python "$ROOT/bench.py" --config $cfg --no-cpu-baseline > "$OUT/trace_$cfg.log" 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$cfg" -- python "$ROOT/bench.py" --config $cfg --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/pmc_fetch_$cfg.log" 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$cfg" -- python "$ROOT/bench.py" --config $cfg --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/pmc_write_$cfg.log" 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_valu_$cfg" -- python "$ROOT/bench.py" --config $cfg --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/pmc_valu_$cfg.log" 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc_wait_$cfg" -- python "$ROOT/bench.py" --config $cfg --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/pmc_wait_$cfg.log" 2>&1
done
for cfg in cfg3 cfg5_tenth; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$cfg" -- python "$ROOT/bench.py" --config $cfg --no-cpu-baseline > "$OUT/trace_$cfg.log" 2>&1
done
cd "$ROOT"
# summarise on the box and drop the raw rocprofv3 output (gpurun brings back at most 64 MiB)
python profiles/summarize.py "$R" > "$OUT/summary.log" 2>&1
mkdir -p "$OUT/final" && cp -r profiles/$R/. "$OUT/final/" && cp profiles/traffic.json "$OUT/final/traffic.json"
rm -rf "$OUT"/trace_* "$OUT"/pmc_fetch_* "$OUT"/pmc_write_* "$OUT"/pmc_valu_* "$OUT"/pmc_wait_*
tail -12 "$OUT/summary.log"
cat "$OUT/bench_cfg2_default.json"
