# HBM traffic and kernel time of one bench configuration for one library: exp/traffic_cfg.sh <out_dir> <cfg> [LIB=path]
OUT=$(pwd)/gpurun_out/$1; cfg=$2; lib=${3#LIB=}
ROOT=$(pwd); mkdir -p "$OUT"; export TMPDIR=/tmp; export P2S_LIB=$lib
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/f" -- python "$ROOT/bench.py" --config $cfg --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/f.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/w" -- python "$ROOT/bench.py" --config $cfg --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/w.log" 2>&1
cd "$ROOT"
python bench.py --config $cfg --no-cpu-baseline > "$OUT/bench.json" 2>/dev/null
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
tot = {}
for tag, name in (('f', 'FETCH_SIZE'), ('w', 'WRITE_SIZE')):
    per = collections.defaultdict(float)
    for f in glob.glob(f'{out}/{tag}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if 'p2s_tri' in row['Kernel_Name'] and row['Counter_Name'] == name:
                per[row['Dispatch_Id']] += float(row['Counter_Value'])
    tot[name] = sum(per.values()) / max(len(per), 1)
d = json.loads(open(f'{out}/bench.json').read().strip().split('\n')[-1])
print('kernel_ms %.4f frac %.4f; per launch: read %.1f MB (2x FETCH), write %.1f MB' % (d['roofline']['kernel_ms'], d['roofline']['frac'], 2 * tot['FETCH_SIZE'] * 1024 / 1e6, tot['WRITE_SIZE'] * 1024 / 1e6))
PY
rm -rf "$OUT/f" "$OUT/w"
