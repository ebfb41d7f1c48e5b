"""Turn the raw rocprofv3 output of profiles/collect.sh (gpurun_out/<round>/) into the small files kept under
profiles/<round>/ and refresh profiles/traffic.json: what bench.py reports as roofline.traffic -- per configuration the
HBM bytes of one step together with a fingerprint of the kernel sources they were measured on (bench.py drops the figure
when the sources have changed since)."""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else 'r03'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', R)
DST = os.path.join(ROOT, 'profiles', R)
os.makedirs(DST, exist_ok=True)
TRI_SOURCES = ['p2s_tri_pool.hip', 'p2s_tri_fused.hip', 'p2s_tri.hip', 'p2s_tri_deep.hip', 'p2s_tri_dev.h', 'p2s_internal.h', 'p2s_api.hip']


def sources_fingerprint():
    h = hashlib.sha1()
    for name in TRI_SOURCES:
        with open(os.path.join(ROOT, 'pose2sim_amd', 'csrc', name), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


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

traffic = {}
for cfg in ('cfg2', 'cfg3', 'cfg4', 'cfg5_tenth'):
    stats = sorted(glob.glob(os.path.join(SRC, f'trace_{cfg}', '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime)
    if stats:
        rows = list(csv.reader(open(stats[-1])))
        for r in rows[1:]:                                      # torch's generator kernels have names of several KB
            if len(r[0]) > 160:
                r[0] = r[0][:150] + '...[name cut]'
        with open(os.path.join(DST, f'kernel_stats_{cfg}.csv'), 'w') as fh:
            csv.writer(fh, quoting=csv.QUOTE_ALL).writerows(rows)
        for r in rows[1:]:
            if 'p2s_' in r[0]:
                print(cfg, 'kernel', short(r[0]), 'calls', r[1], 'avg ns', r[3])

    # instruction mix / issue utilisation per kernel
    names = ['SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_WAIT_INST_ANY']
    valu = {n: counter_avg(f'pmc_valu_{cfg}', n)[0] for n in names}
    names2 = ['SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SMEM', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE']
    wait = {n: counter_avg(f'pmc_wait_{cfg}', n)[0] for n in names2}
    kernels = sorted(k for k in valu['SQ_WAVES'] if k.startswith('p2s_'))
    if kernels:
        with open(os.path.join(DST, f'pmc_valu_{cfg}.csv'), 'w') as fh:
            fh.write('kernel,' + ','.join(names + names2) + ',valu_insts_per_wave\n')
            for k in kernels:
                vals = [valu[n].get(k, 0.0) for n in names] + [wait[n].get(k, 0.0) for n in names2]
                per_wave = vals[1] / max(vals[0], 1.0)
                fh.write(k + ',' + ','.join('%.0f' % v for v in vals) + ',%.1f\n' % per_wave)
                print(cfg, k, 'VALU insts per launch %.4g, per wave %.1f' % (vals[1], per_wave))

    fetch, nf = counter_avg(f'pmc_fetch_{cfg}', 'FETCH_SIZE')
    write, _ = counter_avg(f'pmc_write_{cfg}', 'WRITE_SIZE')
    rows = []
    for k in sorted(fetch):
        if not k.startswith('p2s_'):
            continue
        rd = 2.0 * fetch[k] * 1024.0              # FETCH_SIZE is in KB and counts 128-B requests as 64 B on gfx950
        wr = write.get(k, 0.0) * 1024.0
        rows.append((k, nf[k], fetch[k], write.get(k, 0.0), rd, wr, rd + wr))
    if rows:
        with open(os.path.join(DST, f'pmc_hbm_traffic_{cfg}.csv'), 'w') as fh:
            fh.write('kernel,dispatches,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_read_bytes(2x FETCH, gfx950 correction),hbm_write_bytes,total_bytes_per_launch\n')
            for r in rows:
                fh.write('%s,%d,%.1f,%.1f,%.0f,%.0f,%.0f\n' % r)
                print(cfg, r)
        tri = [r for r in rows if r[0].startswith('p2s_tri_')]
        if tri:
            # a step is `launches_per_step` launches of the triangulation kernel (bench.py's line says how many: shards of
            # more than 2^31 bytes of observations go in chunks); the counters are averages per launch
            per_step = 1
            try:
                line = [l for l in open(os.path.join(SRC, f'bench_{cfg}.json' if cfg != 'cfg2' else 'bench_cfg2_default.json')) if l.startswith('{')][-1]
                per_step = json.loads(line)['roofline'].get('launches_per_step') or 1
            except Exception as exc:
                print(cfg, 'launches per step unknown:', exc)
            traffic[cfg] = {'bytes_per_step': per_step * sum(r[6] for r in tri), 'launches_per_step': per_step, 'kernels': [r[0] for r in tri],
                            'source': f'profiles/{R}/pmc_hbm_traffic_{cfg}.csv', 'sources_sha1': sources_fingerprint()}
            print(cfg, 'traffic per step', traffic[cfg]['bytes_per_step'])
if traffic:
    traffic['_note'] = ('HBM bytes per step of bench.py --config <cfg> (every triangulation kernel of a step, inputs rotated over >= 1 GiB of '
                        'buffers), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM '
                        'section); sources_sha1 = fingerprint of the kernel sources measured (profiles/summarize.py), bench.py reports the figure '
                        'only while it matches')
    with open(os.path.join(ROOT, 'profiles', 'traffic.json'), 'w') as fh:
        json.dump(traffic, fh, indent=1)
