"""Host-side ingest rate: native parser (csrc/p2s_ingest.cpp) vs a json.load loop on the same files.
python profiles/ingest_bench.py [n_files] -> one JSON line."""
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pose2sim_amd.ingest import JsonBatch  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
d = tempfile.mkdtemp(prefix='p2s_ingest_')
try:
    rng = np.random.default_rng(0)
    paths = []
    for i in range(N):
        kp = rng.uniform(0, 2000, 78).astype(np.float32)
        p = os.path.join(d, f'cam_{i:06d}.json')
        with open(p, 'w') as fh:
            json.dump({'version': 1.3, 'people': [{'person_id': [-1], 'pose_keypoints_2d': [float(x) for x in kp],
                                                    'face_keypoints_2d': [], 'hand_left_keypoints_2d': [], 'hand_right_keypoints_2d': [],
                                                    'pose_keypoints_3d': [], 'face_keypoints_3d': [], 'hand_left_keypoints_3d': [],
                                                    'hand_right_keypoints_3d': []}]}, fh)
        paths.append(p)
    res = {'files': N, 'bytes_per_file': os.path.getsize(paths[0]), 'cpus': os.cpu_count(), 'native_files_per_s': {}}
    for nt in (1, 2, 4, 8, 16):
        t0 = time.perf_counter()
        with JsonBatch(paths, nt) as b:
            out = np.empty((N, 26, 3), np.float32)
            b.gather_keypoints(list(range(26)), 1, np.arange(N) * 78, 78, out)
        res['native_files_per_s'][str(nt)] = round(N / (time.perf_counter() - t0))
    t0 = time.perf_counter()
    m = min(N, 5000)
    for p in paths[:m]:
        with open(p) as fh:
            json.load(fh)
    res['json_load_files_per_s'] = round(m / (time.perf_counter() - t0))
    print(json.dumps(res))
finally:
    shutil.rmtree(d, ignore_errors=True)
