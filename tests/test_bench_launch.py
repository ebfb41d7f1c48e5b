"""`python bench.py --gpus N` must not be mis-runnable (round 2 verdict): without a launcher it starts the N ranks
itself, as fresh children, before anything touches the GPU; with a launcher whose world size disagrees it exits
non-zero.  Rehearsed here without a GPU in the bench's dry mode (gloo, no engine)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, env=env, capture_output=True, text=True,
                          timeout=timeout)


def _json_lines(text):
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith('{'):
            out.append(json.loads(line))
    return out


def test_gpus_2_without_launcher_starts_two_ranks():
    r = _run(['--gpus', '2', '--steps', '3'], {'P2S_BENCH_DRY': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                       # rank 0 alone prints
    assert lines[0]['n_gpus'] == 2 and lines[0]['ranks_seen'] == 2 and lines[0]['steps'] == 3


def test_world_size_mismatch_is_refused():
    r = _run(['--gpus', '4'], {'P2S_BENCH_DRY': '1', 'WORLD_SIZE': '1', 'RANK': '0'})
    assert r.returncode != 0
    assert 'WORLD_SIZE=1' in r.stderr and '--gpus 4' in r.stderr
    r = _run(['--gpus', '1'], {'P2S_BENCH_DRY': '1', 'WORLD_SIZE': '2', 'RANK': '0'})
    assert r.returncode != 0


def test_single_gpu_needs_no_launcher():
    r = _run(['--gpus', '1'], {'P2S_BENCH_DRY': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_lines(r.stdout)[0]['n_gpus'] == 1


def test_launcher_parent_never_imports_torch():
    """The parent of the spawned ranks must stay clear of the GPU runtime: the spawn happens before `import torch`."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    main_body = src[src.index('def main():'):]
    assert main_body.index('spawn_ranks(') < main_body.index('import torch')
    head = src[:src.index('def make_workload')]
    assert 'import torch' not in head
