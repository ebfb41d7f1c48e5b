"""Boundary golden: what the reference's Pose2Sim.read_config_files / determine_level return for a
session tree with session-level and trial-level Config.toml files (deep merge, project_dir injection,
exclude_from_batch), config given as a session path, a trial path and None (= cwd).  -> config_cases.json"""
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

SESSION = '''
[project]
multi_person = false
frame_rate = 60
frame_range = []
exclude_from_batch = ['Trial_3']
[pose]
pose_model = 'HALPE_26'
vid_img_extension = 'mp4'
[triangulation]
reproj_error_threshold_triangulation = 15
min_cameras_for_triangulation = 2
interpolation = 'cubic'
[logging]
use_custom_logging = true
'''
TRIALS = {
    'Trial_1': "[project]\nframe_range = [10, 50]\n[triangulation]\nmin_cameras_for_triangulation = 3\n",
    'Trial_2': "[project]\nmulti_person = true\n[personAssociation.multi_person]\nmin_affinity = 0.3\n",
    'Trial_3': "[project]\nframe_rate = 30\n",
}


def build_tree(root):
    os.makedirs(os.path.join(root, 'calibration'))
    with open(os.path.join(root, 'Config.toml'), 'w') as fh:
        fh.write(SESSION)
    for name, text in TRIALS.items():
        os.makedirs(os.path.join(root, name))
        with open(os.path.join(root, name, 'Config.toml'), 'w') as fh:
            fh.write(text)


def norm(dicts, root):
    out = []
    for d in dicts:
        d = json.loads(json.dumps(d))
        pd = d['project'].get('project_dir')
        if pd is not None:
            d['project']['project_dir'] = os.path.relpath(os.path.realpath(pd), os.path.realpath(root))
        out.append(d)
    return out


def gen():
    ref_shim.load()
    import importlib
    ref = importlib.import_module('Pose2Sim.Pose2Sim')
    root = tempfile.mkdtemp(prefix='p2s_cfg_')
    cases = {}
    try:
        build_tree(root)
        cwd = os.getcwd()
        try:
            level, dicts = ref.read_config_files(root)
            cases['session_path'] = {'level': level, 'dicts': norm(dicts, root)}
            level, dicts = ref.read_config_files(os.path.join(root, 'Trial_1'))
            cases['trial_path'] = {'level': level, 'dicts': norm(dicts, root)}
            os.chdir(root)
            level, dicts = ref.read_config_files(None)
            cases['cwd_session'] = {'level': level, 'dicts': norm(dicts, root)}
            os.chdir(os.path.join(root, 'Trial_2'))
            level, dicts = ref.read_config_files(None)
            cases['cwd_trial'] = {'level': level, 'dicts': norm(dicts, root)}
            given = {'project': {'project_dir': os.path.join(root, 'Trial_1')}, 'pose': {}}
            level, dicts = ref.read_config_files(given)
            cases['dict'] = {'level': level, 'dicts': norm(dicts, root)}
        finally:
            os.chdir(cwd)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    with open(os.path.join(HERE, 'config_cases.json'), 'w') as fh:
        json.dump({'session': SESSION, 'trials': TRIALS, 'cases': cases}, fh, indent=1, sort_keys=True)
    print({k: (v['level'], [d['project']['project_dir'] for d in v['dicts']]) for k, v in cases.items()})


if __name__ == '__main__':
    gen()
