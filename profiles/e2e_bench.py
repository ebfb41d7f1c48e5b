"""End-to-end wall time of Pose2Sim.triangulation-equivalent stages on a synthetic single-person trial written to
disk as OpenPose JSON folders (C cameras x F frames): where the time goes once the kernels take < 1 ms.
python profiles/e2e_bench.py [F] [C] -> one JSON line."""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import e2e_common as ec  # noqa: E402
from pose2sim_amd import poseio, skeletons, synth, triangulation, trc  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ids, names, swap = skeletons.keypoints('HALPE_26')
wl = synth.make_config(F, C, len(ids), 1, seed=1)
root = tempfile.mkdtemp(prefix='p2s_e2e_')
try:
    t0 = time.time()
    trial = ec.write_trial(root, 'trial', wl['cams'], ec.people_from_xyl(wl['xyl'], ids, 26), json_subdir='pose')
    t_write = time.time() - t0
    cfg = ec.base_config(trial, False)
    os.chdir(root)
    stages = {}
    orig_load, orig_trc = poseio.load_observations, trc.write_trc

    def timed(name, fn):
        def wrapper(*a, **k):
            t = time.time()
            out = fn(*a, **k)
            stages[name] = stages.get(name, 0.0) + time.time() - t
            return out
        return wrapper
    poseio.load_observations = timed('json_ingest_s', orig_load)
    trc.write_trc = timed('trc_write_s', orig_trc)
    eng_holder = {}
    orig_engine = triangulation._make_engine

    def make_engine():
        e = orig_engine()
        e.triangulate = timed('engine_host_call_s', e.triangulate)
        eng_holder['e'] = e
        return e
    triangulation._make_engine = make_engine
    triangulation.triangulate_all(cfg)          # first run: library load, page cache
    stages.clear()
    t0 = time.time()
    triangulation.triangulate_all(cfg)
    total = time.time() - t0
    out = {'frames': F, 'cams': C, 'json_files': F * C, 'total_s': round(total, 3), 'writing_the_synthetic_trial_s': round(t_write, 1)}
    out.update({k: round(v, 3) for k, v in stages.items()})
    out['other_host_s (listing, tracking, interpolation, trimming, logging)'] = round(total - sum(stages.values()), 3)
    print(json.dumps(out))
finally:
    os.chdir('/')
    shutil.rmtree(root, ignore_errors=True)
