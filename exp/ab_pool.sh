# A/B of the triangulation kernel forms on one box: exp/ab_pool.sh <out.log> <config> "<[LIB=path] bench args>" ...
out=$1; cfg=$2; shift 2
mkdir -p $(dirname $out); : > $out
for v in "$@"; do
  echo "== $v" >> $out
  lib=""; args="$v"
  case "$v" in LIB=*) lib="${v%% *}"; lib="${lib#LIB=}"; args="${v#* }";; esac
  P2S_LIB=$lib timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --steps 200 $args >> $out 2>&1 || exit 1
done
python - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l); r=d['roofline']; s=d['config']['search']
        print('  kernel_ms %.4f frac %.4f  evals %.0f passes %.0f screened %.0f spasses %.0f units %.0f' % (r['kernel_ms'], r['frac'], s['subsets_evaluated_per_step'], s['evaluation_passes_per_step'], s['screened_subsets_per_step'], s['screen_passes_per_step'], s['units_entering_search_per_step']))
PY
