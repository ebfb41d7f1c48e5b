"""End-to-end wall time of the multi-person association stage on a synthetic trial written to disk
(C cameras x P persons x F frames of OpenPose JSON): ingest, kernel, proposal extraction, JSON rewrite.
python profiles/e2e_assoc_bench.py [F] [C] [P] -> one JSON line."""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import e2e_common as ec  # noqa: E402
from pose2sim_amd import personAssociation as pa, poseio, skeletons, synth  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 8
Pn = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ids, names, swap = skeletons.keypoints('HALPE_26')
wl = synth.make_config(F, C, len(ids), Pn, seed=3, p_missing_cam=0.0)
rng = np.random.default_rng(0)
people = ec.people_from_xyl(wl['xyl'], ids, 26)
for per_cam in people:                                   # persons in random order per camera
    for c in range(C):
        if per_cam[c]:
            per_cam[c] = [per_cam[c][i] for i in rng.permutation(len(per_cam[c]))]
root = tempfile.mkdtemp(prefix='p2s_e2e_assoc_')
try:
    t0 = time.time()
    trial = ec.write_trial(root, 'trial', wl['cams'], people, json_subdir='pose')
    t_write = time.time() - t0
    cfg = ec.base_config(trial, True)
    os.chdir(root)
    stages = {}

    def timed(name, fn):
        def wrapper(*a, **k):
            t = time.time()
            out = fn(*a, **k)
            stages[name] = stages.get(name, 0.0) + time.time() - t
            return out
        return wrapper
    poseio.read_people_batch = timed('json_ingest_s', poseio.read_people_batch)
    pa.person_index_per_cam = timed('proposal_extraction_s', pa.person_index_per_cam)
    pa.rewrite_json_files_batch = timed('json_rewrite_s', pa.rewrite_json_files_batch)
    orig_engine = pa._make_engine

    def make_engine():
        e = orig_engine()
        e.associate = timed('engine_host_call_s', e.associate)
        return e
    pa._make_engine = make_engine
    pa.associate_all(cfg)
    stages.clear()
    t0 = time.time()
    pa.associate_all(cfg)
    total = time.time() - t0
    out = {'frames': F, 'cams': C, 'persons': Pn, 'json_files': F * C, 'total_s': round(total, 3), 'writing_the_synthetic_trial_s': round(t_write, 1)}
    out.update({k: round(v, 3) for k, v in stages.items()})
    out['other_host_s'] = round(total - sum(stages.values()), 3)
    print(json.dumps(out))
finally:
    os.chdir('/')
    shutil.rmtree(root, ignore_errors=True)
