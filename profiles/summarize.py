"""Turn the raw rocprofv3 output of profiles/collect.sh (gpurun_out/<round>/) into the small files kept
under profiles/<round>/ and refresh profiles/traffic.json (what bench.py reports as roofline.traffic)."""
import csv
import glob
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else 'r02'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', R)
DST = os.path.join(ROOT, 'profiles', R)
os.makedirs(DST, exist_ok=True)


def short(name):
    name = name.replace('void ', '').replace('(anonymous namespace)::', '')
    name = name.split('(')[0]
    name = name.split('<')[0]
    return name.strip()


def counter_avg(subdir, counter):
    """kernel -> average counter value per dispatch (values of one dispatch are summed over its rows)."""
    per = {}
    paths = sorted(glob.glob(os.path.join(SRC, subdir, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)
    for path in paths[-1:]:                                  # the latest run only
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row['Counter_Name'] != counter:
                    continue
                key = (short(row['Kernel_Name']), path + ':' + row['Dispatch_Id'])
                per[key] = per.get(key, 0.0) + float(row['Counter_Value'])
    out = {}
    for (k, _), v in per.items():
        out.setdefault(k, []).append(v)
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


for f in glob.glob(os.path.join(SRC, 'bench_*.json')):
    shutil.copy(f, DST)
stats = sorted(glob.glob(os.path.join(SRC, 'trace', '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime)
if stats:
    rows = list(csv.reader(open(stats[-1])))
    for r in rows[1:]:                                      # torch's generator kernels have names of several KB
        if len(r[0]) > 160:
            r[0] = r[0][:150] + '...[name cut]'
    with open(os.path.join(DST, 'kernel_stats_cfg2.csv'), 'w') as fh:
        csv.writer(fh, quoting=csv.QUOTE_ALL).writerows(rows)

# instruction mix / issue utilisation per kernel
names = ['SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_WAIT_INST_ANY']
valu = {n: counter_avg('pmc_valu', n)[0] for n in names}
names2 = ['SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SMEM', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE']
wait = {n: counter_avg('pmc_wait', n)[0] for n in names2}
kernels = sorted(k for k in valu['SQ_WAVES'] if k.startswith('p2s_'))
if kernels:
    with open(os.path.join(DST, 'pmc_valu_cfg2.csv'), 'w') as fh:
        fh.write('kernel,' + ','.join(names + names2) + ',valu_insts_per_wave,valu_issue_us_at_2.4GHz(insts*4.15cyc/1024 SIMDs)\n')
        for k in kernels:
            vals = [valu[n].get(k, 0.0) for n in names] + [wait[n].get(k, 0.0) for n in names2]
            per_wave = vals[1] / max(vals[0], 1.0)
            issue_us = vals[1] * 4.15 / 1024.0 / 2400.0
            fh.write(k + ',' + ','.join('%.0f' % v for v in vals) + ',%.1f,%.1f\n' % (per_wave, issue_us))
            print(k, 'VALU insts/wave %.1f' % per_wave, 'issue time %.1f us' % issue_us)

fetch, nf = counter_avg('pmc_fetch', 'FETCH_SIZE')
write, _ = counter_avg('pmc_write', 'WRITE_SIZE')
rows, total = [], 0.0
for k in sorted(fetch):
    if not k.startswith('p2s_'):
        continue
    rd = 2.0 * fetch[k] * 1024.0              # FETCH_SIZE is in KB and counts 128-B requests as 64 B on gfx950
    wr = write.get(k, 0.0) * 1024.0
    rows.append((k, nf[k], fetch[k], write.get(k, 0.0), rd, wr, rd + wr))
with open(os.path.join(DST, 'pmc_hbm_traffic_cfg2.csv'), 'w') as fh:
    fh.write('kernel,dispatches,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_read_bytes(2x FETCH, gfx950 correction),hbm_write_bytes,total_bytes_per_launch\n')
    for r in rows:
        fh.write('%s,%d,%.1f,%.1f,%.0f,%.0f,%.0f\n' % r)
        print(r)
tri = [r for r in rows if r[0].startswith('p2s_tri_')]
if tri:
    # cfg2 is one chunk: one launch of each kernel per step (the one-launch kernel: a single row)
    total = sum(r[6] for r in tri)
    with open(os.path.join(ROOT, 'profiles', 'traffic.json'), 'w') as fh:
        json.dump({'cfg2': total,
                   '_note': 'HBM bytes per step of bench.py --config cfg2 (every triangulation kernel of a step, inputs rotated over 5 buffers), rocprofv3 '
                            '--pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM '
                            f'section); source: profiles/{R}/pmc_hbm_traffic_cfg2.csv (profiles/collect.sh + summarize.py)'}, fh, indent=1)
    print('traffic per step', total)
