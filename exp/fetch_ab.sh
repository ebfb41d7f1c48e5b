# HBM read traffic and kernel time of cfg2 with a variant library: exp/fetch_ab.sh <lib or ""> <tag>
lib=$1; tag=$2
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/f_$tag
P2S_LIB=$lib timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/f_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --preroll-ms 0 --no-cpu-baseline > /tmp/f_$tag.log 2>&1
python - $tag <<'PY'
import csv,glob,sys
f=sorted(glob.glob('/tmp/f_%s/**/*counter_collection.csv'%sys.argv[1],recursive=True))[-1]
v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'pool' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
import collections
print(sys.argv[1],'FETCH_SIZE sum/dispatch KB (x2 = bytes):', sum(v)/max(1,len(v))*1, 'rows', len(v))
PY
