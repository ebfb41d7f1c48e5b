"""Native associated-pose JSON writer (csrc/p2s_rewrite.cpp) against the Python restatement of the reference's
rewrite_json_files (personAssociation.py:552-580, json.load + json.dumps): same bytes, same files removed.
Host-only code: runs without a GPU."""
import json
import os
import random

import numpy as np
import pytest

import __graft_entry__ as entry
from pose2sim_amd import personAssociation as pa
from test_ingest import DOCS


@pytest.fixture(scope='module', autouse=True)
def built():
    entry.build_hip()


EXTRA = [
    '{"version": 1.3, "people": [{"person_id": [-1], "pose_keypoints_2d": [1.5, 2.25, 0.9, 100.123456789, 3e2, -0.0], "face_keypoints_2d": []},'
    ' {"person_id": [-1], "pose_keypoints_2d": [7, 8, 9.000001, 1E-7, 1e22, 123456789012345678901234567890]}, {"x": null}]}',
    '{"people": [{"s": "a\\n\\t\\"q\\"\\\\ \\/ \\b\\f\\r \\u00e9 \\ud83d\\ude00 \\ud800 café 中 \U0001F600 \x7f"}, [1, [2, {"k": [true, false, null]}]], "str", 7, -0, -0.0, 1e400, -1e400, NaN, Infinity, -Infinity], "k": "v"}',
    '{"a": 1, "people": [{"p": 1}], "a": 2, "people": [{"p": 2}, {"p": 3}], "b": {"z": 1, "z": 2, "y": [ ]}}',
    '{"people": [], "nested": {"people": [1, 2, 3]}}',
    '{"nopeople": 1}', '{"people": {"0": 1}}', '{"people": "abc"}', '[{"people": []}]', '{}',
    ' \n\t{ "people" :\t[ { } ,\n{ "a" : [ ] } ] , "z" : 0.1 }\r\n',
    '{"people": [{"v": [0.1, 0.2, 0.30000000000000004, 1e-05, 0.0001, 1e16, 1e15, 5e-324, 1.7976931348623157e308, 2.5, 100.0]}]}',
    '{"people": [{"i": ' + '9' * 4300 + '}]}', '{"people": [{"i": ' + '9' * 4301 + '}]}',
]


def _python_rewrite(tmp, src_files, proposals, tag):
    """What the reference's rewrite_json_files leaves on disk (personAssociation.py:552-580), stated with json: the
    source document with its people replaced by one entry per proposal ({} where the camera does not see the person);
    nothing at all when anything goes wrong (unreadable source, no 'people' list, a person index out of range)."""
    out = []
    for cam, src in enumerate(src_files):
        try:
            with open(src, 'r') as fh:
                doc = json.load(fh)
            picked = [{} if np.isnan(row[cam]) else doc['people'][int(row[cam])] for row in proposals]   # looked up only when needed
            out.append(json.dumps({**doc, 'people': picked}))
        except Exception:
            out.append(None)
    return out


def test_native_rewrite_equals_json_dumps(tmp_path):
    rng = random.Random(5)
    docs = [d for d in DOCS if isinstance(d, str)] + EXTRA + [d for d in DOCS if isinstance(d, bytes)]
    srcs = []
    for i, d in enumerate(docs):
        p = tmp_path / f'src_{i:03d}.json'
        p.write_bytes(d if isinstance(d, bytes) else d.encode())
        srcs.append(str(p))
    srcs.append(str(tmp_path / 'missing.json'))
    n_cams = len(srcs)
    for trial in range(6):
        k = [0, 1, 2, 3, 1, 4][trial]
        proposals = np.array([[rng.choice([np.nan, 0, 0, 1, 2, 5]) for _ in range(n_cams)] for _ in range(k)], dtype=float).reshape(k, n_cams)
        want = _python_rewrite(str(tmp_path), srcs, proposals, trial)
        dst = [str(tmp_path / f'nat_{trial}_{c}.json') for c in range(n_cams)]
        pa.rewrite_json_files_batch([dst], [srcs], [proposals], n_cams)
        got = [open(d).read() if os.path.exists(d) else None for d in dst]
        for c in range(n_cams):
            assert got[c] == want[c], (trial, c, docs[c][:80] if c < len(docs) else 'missing', proposals[:, c])


def test_native_rewrite_random_documents(tmp_path):
    rng = random.Random(9)

    def rnd_value(depth=0):
        k = rng.random()
        if depth > 3 or k < 0.35:
            return rng.choice([rng.uniform(-2000, 2000), float(np.float32(rng.uniform(0, 1))), rng.randint(-5, 5), True, False, None,
                               'x' * rng.randint(0, 3) + rng.choice(['', 'é', '"', '\\', '\n', ' ', '\U0001F600']), 1e-7 * rng.random(), 1e20 * rng.random()])
        if k < 0.7:
            return [rnd_value(depth + 1) for _ in range(rng.randint(0, 4))]
        return {rng.choice(['a', 'b', 'pose_keypoints_2d', 'é', 'k' * rng.randint(1, 3)]): rnd_value(depth + 1) for _ in range(rng.randint(0, 4))}
    srcs = []
    for i in range(150):
        doc = {'version': 1.3, 'people': [rnd_value(1) for _ in range(rng.randint(0, 4))], 'extra': rnd_value(1)}
        if rng.random() < 0.2:
            doc.pop('people')
        text = json.dumps(doc, ensure_ascii=rng.random() < 0.5, indent=rng.choice([None, None, 2]), separators=rng.choice([None, (',', ':')]))
        p = tmp_path / f'r_{i:03d}.json'
        p.write_text(text, encoding='utf-8')
        srcs.append(str(p))
    n = len(srcs)
    proposals = np.array([[rng.choice([np.nan, 0, 1, 2, 3]) for _ in range(n)] for _ in range(3)], dtype=float)
    want = _python_rewrite(str(tmp_path), srcs, proposals, 'r')
    dst = [str(tmp_path / f'nat_r_{c}.json') for c in range(n)]
    pa.rewrite_json_files_batch([dst], [srcs], [proposals], n)
    got = [open(d).read() if os.path.exists(d) else None for d in dst]
    assert sum(w is not None for w in want) > 20
    for c in range(n):
        assert got[c] == want[c], (c, open(srcs[c]).read()[:200])
