"""The drop-in boundary: Pose2Sim.triangulation(config) / Pose2Sim.personAssociation(config) with the
reference's three config forms.  Config discovery (level detection, session + trial deep merge, project_dir
injection, exclude_from_batch) is compared with what the reference's read_config_files returned for the same
tree (tests/golden/make_golden_config.py); the stage dispatch runs on a synthetic two-trial session."""
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import e2e_common as ec  # noqa: E402
from test_e2e_trc import OracleEngine  # noqa: E402

from pose2sim_amd import Pose2Sim, skeletons, synth, triangulation  # noqa: E402


def _norm(dicts, root):
    out = []
    for d in dicts:
        d = json.loads(json.dumps(d))
        pd = d['project'].get('project_dir')
        if pd is not None:
            d['project']['project_dir'] = os.path.relpath(os.path.realpath(pd), os.path.realpath(root))
        out.append(d)
    return out


def test_config_discovery_matches_reference(golden_dir, tmp_path, monkeypatch):
    g = json.load(open(os.path.join(golden_dir, 'config_cases.json')))
    root = str(tmp_path / 'session')
    os.makedirs(os.path.join(root, 'calibration'))
    open(os.path.join(root, 'Config.toml'), 'w').write(g['session'])
    for name, text in g['trials'].items():
        os.makedirs(os.path.join(root, name))
        open(os.path.join(root, name, 'Config.toml'), 'w').write(text)
    forms = {'session_path': (root, None), 'trial_path': (os.path.join(root, 'Trial_1'), None), 'cwd_session': (None, root),
             'cwd_trial': (None, os.path.join(root, 'Trial_2')),
             'dict': ({'project': {'project_dir': os.path.join(root, 'Trial_1')}, 'pose': {}}, None)}
    # The reference builds project_dir as join(config_dir, relpath(trial)) (Pose2Sim.py:158): with an absolute
    # session path that only resolves when the working directory sits as deep as the session directory, as it did
    # when the golden was recorded; a sibling directory reproduces that here.
    sibling = tmp_path / 'elsewhere'
    sibling.mkdir()
    for name, (config, cwd) in forms.items():
        monkeypatch.chdir(cwd or str(sibling))
        level, dicts = Pose2Sim.read_config_files(config)
        want = g['cases'][name]
        assert level == want['level'], name
        assert _norm(dicts, root) == want['dicts'], name
    monkeypatch.chdir(str(sibling))
    with pytest.raises(FileNotFoundError):
        Pose2Sim.read_config_files(str(tmp_path / 'nowhere'))


def test_batch_session_runs_every_trial_but_the_excluded_one(tmp_path, monkeypatch):
    ids, names, swap = skeletons.keypoints('HALPE_26')
    root = str(tmp_path / 'session')
    wl = synth.make_config(30, 4, len(ids), 1, seed=31)
    for t in ('Trial_1', 'Trial_2', 'Trial_3'):
        ec.write_trial(root, t, wl['cams'], ec.people_from_xyl(wl['xyl'], ids, 26), json_subdir='pose')
    session = ("[project]\nmulti_person = false\nframe_rate = 60\nframe_range = []\nexclude_from_batch = ['Trial_3']\n"
               "[pose]\npose_model = 'HALPE_26'\nvid_img_extension = 'mp4'\n"
               "[personAssociation]\nlikelihood_threshold_association = 0.3\n"
               "[personAssociation.single_person]\nreproj_error_threshold_association = 20\ntracked_keypoint = 'Neck'\n"
               "[personAssociation.multi_person]\nreconstruction_error_threshold = 0.1\nmin_affinity = 0.2\n"
               "[triangulation]\nreproj_error_threshold_triangulation = 15\nlikelihood_threshold_triangulation = 0.3\n"
               "min_cameras_for_triangulation = 2\nmax_distance_m = 1.0\ninterp_if_gap_smaller_than = 20\ninterpolation = 'linear'\n"
               "remove_incomplete_frames = false\nsections_to_keep = 'all'\nmin_chunk_size = 10\nfill_large_gaps_with = 'last_value'\n"
               "show_interp_indices = true\nmake_c3d = false\nundistort_points = false\nhandle_LR_swap = false\n"
               "[logging]\nuse_custom_logging = true\n")
    open(os.path.join(root, 'Config.toml'), 'w').write(session)
    for t, text in (('Trial_1', ''), ('Trial_2', '[project]\nframe_range = [5, 25]\n'), ('Trial_3', '')):
        open(os.path.join(root, t, 'Config.toml'), 'w').write(text)
    monkeypatch.setattr(triangulation, '_make_engine', lambda: OracleEngine())
    monkeypatch.chdir(root)
    Pose2Sim.triangulation(root)                                  # session path: batch over the trials
    out1 = sorted(os.listdir(os.path.join(root, 'Trial_1', 'pose-3d')))
    out2 = sorted(os.listdir(os.path.join(root, 'Trial_2', 'pose-3d')))
    assert len(out1) == 1 and out1[0].startswith('Trial_1_0-2') and out2 == ['Trial_2_5-24.trc']   # auto range = shortest camera
    assert not os.path.exists(os.path.join(root, 'Trial_3', 'pose-3d'))
    # the same trial through its own path and through None (= cwd) rewrites an identical file
    first = open(os.path.join(root, 'Trial_2', 'pose-3d', out2[0])).read()
    Pose2Sim.triangulation(os.path.join(root, 'Trial_2'))
    assert open(os.path.join(root, 'Trial_2', 'pose-3d', out2[0])).read() == first
    monkeypatch.chdir(os.path.join(root, 'Trial_2'))
    Pose2Sim.triangulation()
    assert open(os.path.join(root, 'Trial_2', 'pose-3d', out2[0])).read() == first
    with pytest.raises(NotImplementedError):
        Pose2Sim.kinematics(root)
