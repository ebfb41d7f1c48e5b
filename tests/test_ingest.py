"""Native OpenPose-JSON ingest (csrc/p2s_ingest.cpp) against Python's json module, which is what the
reference parses with (triangulation.py:607-653, :77-90; personAssociation.py:260-274, :84-91).
Host-only code: these tests run without a GPU (the shared library loads on any machine)."""
import json
import os
import random

import numpy as np
import pytest

import __graft_entry__ as entry
from pose2sim_amd import poseio


@pytest.fixture(scope='module', autouse=True)
def built():
    entry.build_hip()


def _write(tmp_path, docs):
    paths = []
    for i, t in enumerate(docs):
        p = str(tmp_path / f'doc_{i:04d}.json')
        with open(p, 'wb') as fh:
            fh.write(t if isinstance(t, bytes) else t.encode())
        paths.append(p)
    return paths


def _python_view(path):
    """What the reference's readers can get out of a file: None if json.load raises, else
    (len(people) or None when there is no people list, [per person: list of numbers | 'nolist'])."""
    try:
        with open(path, 'r') as fh:
            js = json.load(fh)
    except Exception:
        return None
    if not isinstance(js, dict) or not isinstance(js.get('people'), list):
        return ('nopeople', [])
    persons = []
    for p in js['people']:
        if isinstance(p, dict) and isinstance(p.get('pose_keypoints_2d'), list):
            persons.append(p['pose_keypoints_2d'])
        else:
            persons.append('nolist')
    return (len(js['people']), persons)


DOCS = [
    '{"version":1.3,"people":[{"person_id":[-1],"pose_keypoints_2d":[1.5,2.25,0.9,100.123456789,3e2,-0.0,7,8,9]}]}',
    '{"people":[]}', '{"people":[{}]}', '{"people":[{"pose_keypoints_2d":[]}]}', '{"people":[1,"a",null,[1,2,3]]}',
    '{"people":[{"pose_keypoints_2d":{"a":1}}]}', '{"people":{"a":1}}', '{"people":3}', '{"nopeople":[]}', '[1,2,3]', '3', '"s"',
    '', ' ', '{', '[1,2', '{"people":[}', '{"people":[{"pose_keypoints_2d":[1,2,]}]}', '{"people":[],}', "{'people':[]}",
    '{"people":[]} x', '{"people":[]}\n\n  ', '\n\t {"people" : [ { "pose_keypoints_2d" : [ 1 , 2 , 3 ] } ] } \r\n',
    '{"people":[{"pose_keypoints_2d":[NaN,Infinity,-Infinity,1,2,3]}]}', '{"people":[{"pose_keypoints_2d":[-NaN]}]}',
    '{"people":[{"pose_keypoints_2d":[01]}]}', '{"people":[{"pose_keypoints_2d":[1.]}]}', '{"people":[{"pose_keypoints_2d":[.5]}]}',
    '{"people":[{"pose_keypoints_2d":[1e]}]}', '{"people":[{"pose_keypoints_2d":[+1]}]}', '{"people":[{"pose_keypoints_2d":[-]}]}',
    '{"people":[{"pose_keypoints_2d":[1E+2,1e-2,-1.5E3,0e0,-0,0.0]}]}',
    '{"people":[{"pose_keypoints_2d":[1e400,-1e400,1e-400,4.9e-324,2.2250738585072014e-308,1.7976931348623157e308]}]}',
    '{"people":[{"pose_keypoints_2d":[123456789012345678901234567890,0.1234567890123456789012345678901234567890,9007199254740993]}]}',
    '{"people":[{"a":1}],"people":[{"pose_keypoints_2d":[4,5,6]},{"pose_keypoints_2d":[7,8,9]}]}',
    '{"people":[{"pose_keypoints_2d":[1,2,3],"pose_keypoints_2d":[4,5,6,7,8,9]}]}',
    '{"peopl\\u0065":[{"pose_keypoints_2d":[1,2,3]}]}', '{"people":[{"pose_keypoints\\u005f2d":[1,2,3]}]}',
    '{"s":"a\\nb\\t\\"q\\"\\\\ \\/ \\b\\f\\r \\u00e9 \\ud83d\\ude00 \\ud800","people":[]}', '{"s":"bad \\x escape","people":[]}',
    '{"s":"bad \\u12g4","people":[]}', '{"s":"raw\ttab","people":[]}', '{"s":"raw\nnewline","people":[]}', '{"s":"del \x7f ok","people":[]}',
    '{"s":"café 中文 \U0001F600","people":[]}', b'{"s":"\xff\xfe","people":[]}', b'{"s":"\xc0\xaf","people":[]}',
    b'{"s":"\xed\xa0\x80","people":[]}', b'\xef\xbb\xbf{"people":[]}', b'{"s":"\xe4\xb8","people":[]}',
    '{"people":[{"pose_keypoints_2d":[1,2,3]}],"deep":' + '[' * 100 + ']' * 100 + '}',
    '{"people":[{"pose_keypoints_2d":[true,false,null]}]}', '{"people":[{"pose_keypoints_2d":[1,"2",3]}]}',
    '{"people":[{"pose_keypoints_2d":[1,[2],3]}]}', '{"a":tru,"people":[]}', '{"a":nul,"people":[]}', '{"a":Infinit,"people":[]}',
    '{"people":[{"pose_keypoints_2d":[1,2,3]},{"pose_keypoints_2d":[4,5]},{"pose_keypoints_2d":[6,7,8,9,10,11]}]}',
    '{"a":{"b":[{"c":{}},[],{}]},"people":[{"x":[{"pose_keypoints_2d":[9,9,9]}],"pose_keypoints_2d":[1,2,3]}]}',
    '{"people":[{"pose_keypoints_2d":[1,2,3]}]}{"people":[]}', '{"people":[{"pose_keypoints_2d":[1 2]}]}', '{"people" [ ]}', '{"people":[]:1}',
    '{1:2,"people":[]}', '{"a":1 "people":[]}',
]


def test_parser_accepts_and_rejects_what_json_load_does(tmp_path):
    from pose2sim_amd.ingest import (P2S_JSON_NO_PEOPLE_LIST, P2S_JSON_PERSON_NO_LIST, P2S_JSON_PERSON_NOT_NUMERIC,
                                     P2S_JSON_UNREADABLE, JsonBatch)
    paths = _write(tmp_path, DOCS) + [str(tmp_path / 'missing.json'), '', str(tmp_path)]
    with JsonBatch(paths, 3) as b:
        lengths = b.person_lengths
        for i, p in enumerate(paths):
            view = _python_view(p) if p else None
            doc = DOCS[i] if i < len(DOCS) else p
            if view is None:
                assert b.counts[i] == P2S_JSON_UNREADABLE, (i, doc)
                continue
            if view[0] == 'nopeople':
                assert b.counts[i] == P2S_JSON_NO_PEOPLE_LIST, (i, doc)
                continue
            assert b.counts[i] == view[0], (i, doc)
            for n, person in enumerate(view[1]):
                got_len = lengths[b.person_base[i] + n]
                if person == 'nolist':
                    assert got_len == P2S_JSON_PERSON_NO_LIST, (i, doc)
                elif any(not isinstance(v, (int, float, bool, type(None))) for v in person):
                    assert got_len == P2S_JSON_PERSON_NOT_NUMERIC, (i, doc)
                else:
                    assert got_len == len(person), (i, doc)
                    vals, _ = b.gather_people([i], [n], max(1, len(person)), np.float64)
                    want = np.array([np.nan if v is None else float(v) for v in person] + [np.nan] * (1 - min(1, len(person))))
                    assert np.array_equal(vals[0], want, equal_nan=True), (i, doc)


def test_numbers_are_correctly_rounded_like_python_float(tmp_path):
    rng = random.Random(7)
    tokens = []
    for _ in range(20000):
        kind = rng.random()
        if kind < 0.3:      # what a pose estimator writes: float32 values printed by Python
            tokens.append(repr(float(np.float32(rng.uniform(-10, 4000)))))
        elif kind < 0.5:
            tokens.append(repr(rng.uniform(-1, 1) * 10 ** rng.randint(-320, 308)))
        elif kind < 0.7:    # long digit strings, halfway cases
            digits = ''.join(rng.choice('0123456789') for _ in range(rng.randint(1, 40)))
            tokens.append(('-' if rng.random() < 0.5 else '') + (digits.lstrip('0') or '0') + '.' + ''.join(rng.choice('0123456789') for _ in range(rng.randint(1, 30))))
        elif kind < 0.85:
            tokens.append('%d' % rng.randint(-10 ** rng.randint(1, 25), 10 ** rng.randint(1, 25)))
        else:
            m = '%d.%de%s%d' % (rng.randint(0, 9), rng.randint(0, 10 ** 17), rng.choice(['', '+', '-']), rng.randint(0, 330))
            tokens.append(m)
    tokens += ['9007199254740993', '9007199254740992.5', '0.1', '1e23', '8.41e21', '2.2250738585072011e-308', '5e-324', '2.4703282292062327e-324', '2.4703282292062328e-324']
    doc = '{"people":[{"pose_keypoints_2d":[' + ','.join(tokens) + ']}]}'
    path = _write(tmp_path, [doc])[0]
    want = np.array([float(v) for v in json.load(open(path))['people'][0]['pose_keypoints_2d']])
    from pose2sim_amd.ingest import JsonBatch
    with JsonBatch([path]) as b:
        vals, inexact = b.gather_people([0], [0], len(tokens), np.float64)
        assert inexact == 0
        assert np.array_equal(vals[0], want)
        v32, inexact32 = b.gather_people([0], [0], len(tokens), np.float32)
        with np.errstate(over='ignore'):
            w32 = want.astype(np.float32)
        assert np.array_equal(v32[0], w32)
        assert inexact32 == int((w32.astype(np.float64) != want).sum())


def _reference_extract(paths_f, keypoints_ids, nb_persons):
    """extract_files_frame_f (triangulation.py:607-653) restated with json.load, one frame."""
    out = np.full((nb_persons, len(paths_f), len(keypoints_ids), 3), np.nan)
    for n in range(nb_persons):
        for c, p in enumerate(paths_f):
            try:
                with open(p, 'r') as fh:
                    js = json.load(fh)
                for k, kid in enumerate(keypoints_ids):
                    try:
                        kp = js['people'][n]['pose_keypoints_2d']
                        out[n, c, k] = [kp[kid * 3], kp[kid * 3 + 1], kp[kid * 3 + 2]]
                    except Exception:
                        pass
            except Exception:
                pass
    return out


def _make_trial(tmp_path, F, C, Kj, seed, f32=True, broken=True):
    rng = np.random.default_rng(seed)
    root = tmp_path / f'pose_{seed}'
    dirs = [f'cam_{c + 1:02d}_json' for c in range(C)]
    for c, d in enumerate(dirs):
        os.makedirs(root / d)
        for f in range(F):
            r = rng.random()
            if r < 0.05:
                continue                                      # missing file
            path = root / d / f'cam_{c + 1:02d}_{f:06d}.json'
            if r < 0.08 and broken:
                path.write_text('{"people": [')             # truncated
                continue
            people = []
            for _ in range(rng.integers(0, 4)):
                n = Kj * 3 if rng.random() > 0.1 else int(rng.integers(0, Kj * 3))     # short lists
                v = rng.uniform(0, 2000, n)
                v = v.astype(np.float32).astype(np.float64) if f32 else v
                if rng.random() < 0.1 and n >= 3:
                    v[0::3] = np.nan                          # a person whose x are all NaN
                people.append({'person_id': [-1], 'pose_keypoints_2d': [float(x) for x in v]})
            if rng.random() < 0.03:
                people.append({'person_id': [-1]})            # no keypoint list
            path.write_text(json.dumps({'version': 1.3, 'people': people}))
    names = poseio.list_json_files(str(root), dirs)
    return str(root), dirs, names


@pytest.mark.parametrize('f32', [True, False])
def test_load_observations_matches_reference_extraction(tmp_path, f32):
    F, C, Kj = 40, 3, 6
    root, dirs, names = _make_trial(tmp_path, F, C, Kj, 11 + f32, f32)
    maps = poseio.frame_file_map(names)
    ids = [4, 0, 5, 2, 7]                                     # 7: beyond the list -> NaN
    nb = 4                                                     # one more than any file holds
    got = poseio.load_observations(root, dirs, maps, (2, F), ids, nb)
    assert got.dtype == (np.float32 if f32 else np.float64)
    for fi, f in enumerate(range(2, F)):
        paths_f = [os.path.join(root, dirs[c], maps[c].get(f, 'none')) for c in range(C)]
        want = _reference_extract(paths_f, ids, nb)
        assert np.array_equal(got[fi].astype(np.float64), want, equal_nan=True), f
    one = poseio.load_observations(root, dirs, maps, (0, F), ids, 1)
    assert np.array_equal(one[2:, 0], got[:, 0], equal_nan=True)

    # multi-person mode: the person count is the maximum over ALL files (triangulation.py:784)
    root, dirs, names = _make_trial(tmp_path, F, C, Kj, 31 + f32, f32, broken=False)
    maps = poseio.frame_file_map(names)
    got, nb = poseio.load_observations(root, dirs, maps, (5, 20), ids, 0, json_files_names=names, count_all_persons=True)
    assert nb == max(poseio.count_persons(os.path.join(root, dirs[c], nm)) for c in range(C) for nm in names[c])
    for fi, f in enumerate(range(5, 20)):
        paths_f = [os.path.join(root, dirs[c], maps[c].get(f, 'none')) for c in range(C)]
        assert np.array_equal(got[fi].astype(np.float64), _reference_extract(paths_f, ids, nb), equal_nan=True), f


def test_multi_person_count_raises_like_the_reference_on_a_broken_file(tmp_path):
    root, dirs, names = _make_trial(tmp_path, 30, 2, 4, 5)
    maps = poseio.frame_file_map(names)
    # count_persons_in_json has no try/except (triangulation.py:87-89): the truncated files raise
    with pytest.raises(json.JSONDecodeError):
        max(poseio.count_persons(os.path.join(root, dirs[c], nm)) for c in range(2) for nm in names[c])
    from pose2sim_amd.ingest import JsonBatch
    paths, _ = poseio._trial_paths(root, dirs, names)
    with JsonBatch(paths) as b, pytest.raises(json.JSONDecodeError):
        poseio.max_persons_in_trial(b, paths)


def test_people_batches_match_read_json_and_persons_combinations(tmp_path):
    from pose2sim_amd import personAssociation as pa
    F, C, Kj = 60, 3, 5
    root, dirs, names = _make_trial(tmp_path, F, C, Kj, 21)
    maps = poseio.frame_file_map(names)
    paths = [os.path.join(root, dirs[c], maps[c].get(f, 'none')) for f in range(F) for c in range(C)]
    # read_json: every kept person has the full length here, so drop the short ones from the comparison set
    full = []
    for p in paths:
        people = poseio.read_people(p)
        full.append(all(len(x) == Kj * 3 for x in people))
    sel = [p for p, ok in zip(paths, full) if ok]
    n_people, rows, Kj3 = poseio.read_people_batch(sel)
    assert Kj3 == Kj * 3
    want = [poseio.read_people(p) for p in sel]
    assert list(n_people) == [len(w) for w in want]
    assert np.array_equal(rows, np.array([x for w in want for x in w], dtype=np.float64).reshape(-1, Kj3), equal_nan=True)
    with pytest.raises(ValueError):
        poseio.read_people_batch(paths)                        # ragged persons are refused, as before

    # persons_combinations' count + read_json's i-th person (personAssociation.py:84-91, :200-202)
    k3 = 2 * 3
    counts, tracked = pa.single_person_candidates(paths, k3)
    row = 0
    for i, p in enumerate(paths):
        try:
            people = json.load(open(p))['people']
            n = len([q for q in people if not all(np.isnan(q['pose_keypoints_2d'][::3]))])
        except Exception:
            n = 0
        assert counts[i] == n, p
        listed = poseio.read_people(p)
        for j in range(n):
            v = listed[j][k3:k3 + 3] if j < len(listed) else []
            w = np.array(v if len(v) == 3 else [np.nan] * 3, dtype=float)
            assert np.array_equal(tracked[row], w, equal_nan=True), (p, j)
            row += 1
    assert row == len(tracked)


def test_ingest_throughput_report(tmp_path, capsys):
    """Not a pass/fail number: prints files/s of the native parser next to json.load on the same files."""
    import time
    rng = np.random.default_rng(0)
    paths = []
    for i in range(4000):
        kp = rng.uniform(0, 2000, 78).astype(np.float32)
        p = tmp_path / f'f_{i:06d}.json'
        p.write_text(json.dumps({'version': 1.3, 'people': [{'person_id': [-1], 'pose_keypoints_2d': [float(x) for x in kp],
                                                            'face_keypoints_2d': [], 'hand_left_keypoints_2d': [], 'hand_right_keypoints_2d': [],
                                                            'pose_keypoints_3d': [], 'face_keypoints_3d': [], 'hand_left_keypoints_3d': [],
                                                            'hand_right_keypoints_3d': []}]}))
        paths.append(str(p))
    from pose2sim_amd.ingest import JsonBatch
    t0 = time.perf_counter()
    with JsonBatch(paths) as b:
        out = np.empty((len(paths), 26, 3), np.float32)
        b.gather_keypoints(list(range(26)), 1, np.arange(len(paths)) * 78, 78, out)
    t1 = time.perf_counter()
    ref = np.array([json.load(open(p))['people'][0]['pose_keypoints_2d'] for p in paths]).reshape(len(paths), 26, 3)
    t2 = time.perf_counter()
    assert np.array_equal(out.astype(np.float64), ref)
    with capsys.disabled():
        print(f'\n[ingest] native {len(paths) / (t1 - t0):.0f} files/s, json.load {len(paths) / (t2 - t1):.0f} files/s')
