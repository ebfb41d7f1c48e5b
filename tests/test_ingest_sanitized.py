"""The JSON parser and rewriter read files a user hands them, so the host code (ingest, associated-pose writer,
.trc rows) is also run under AddressSanitizer + UBSan (host build, g++): the acceptance corpus of test_ingest.py plus a few thousand random mutations of valid documents
(truncations, byte flips, duplicated slices).  Any out-of-bounds access, leak or undefined behaviour fails."""
import os
import random
import subprocess

import pytest

from test_ingest import DOCS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('asan') / 'ingest_driver')
    cmd = ['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-omit-frame-pointer', '-pthread',
           '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'native', 'ingest_driver.cpp'),
           os.path.join(ROOT, 'pose2sim_amd', 'csrc', 'p2s_ingest.cpp'), os.path.join(ROOT, 'pose2sim_amd', 'csrc', 'p2s_rewrite.cpp'),
           os.path.join(ROOT, 'pose2sim_amd', 'csrc', 'p2s_trc.cpp'), '-o', exe]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def _mutations(rng, n):
    base = [d.encode() if isinstance(d, str) else d for d in DOCS if len(d) > 20]
    out = []
    for _ in range(n):
        b = bytearray(rng.choice(base))
        for _ in range(rng.randint(1, 4)):
            kind = rng.random()
            if kind < 0.3 and len(b) > 2:
                del b[rng.randrange(len(b)):]                              # truncate
            elif kind < 0.6 and b:
                b[rng.randrange(len(b))] = rng.randrange(256)              # flip a byte
            elif kind < 0.8 and len(b) > 4:
                i, j = sorted(rng.sample(range(len(b)), 2)); b[i:i] = b[i:j]   # duplicate a slice
            else:
                b.insert(rng.randrange(len(b) + 1), rng.choice(b'{}[],:"\\-+.eE0123456789 \n\tNItfn'))
        out.append(bytes(b))
    return out


def test_parser_under_asan_and_ubsan(driver, tmp_path):
    rng = random.Random(3)
    docs = [d.encode() if isinstance(d, str) else d for d in DOCS] + _mutations(rng, 3000)
    docs.append(b'[' * 100000)                        # nesting far beyond the depth limit
    docs.append(b'{"people":[{"pose_keypoints_2d":[' + b','.join(b'1.5' for _ in range(200000)) + b']}]}')
    paths = []
    for i, d in enumerate(docs):
        p = tmp_path / f'm_{i:05d}.json'
        p.write_bytes(d)
        paths.append(str(p))
    paths += [str(tmp_path / 'missing.json'), '', str(tmp_path)]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    for threads in ('1', '8'):
        r = subprocess.run([driver, threads, str(tmp_path / f'rows_{threads}.trc')], input='\n'.join(paths) + '\n', capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
        assert 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, r.stderr[-4000:]
        assert r.stdout.startswith(f'files {len(paths)} ')
