"""The RCCL leg of the multi-GPU path on the one GPU a test box has: a process group with backend 'nccl' (= RCCL
on ROCm) and world_size 1, through which parallel.sharded_triangulate and bench.py's asynchronous packed
all-gather run exactly the calls the 8-GPU job makes (device tensors, uint8 payload, async handles).  The
multi-rank logic itself (shard bounds, padding, reassembly) is covered on CPU with gloo and world_size 2 in
test_distributed_cpu.py."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%%d' %% int(sys.argv[1]), rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    from pose2sim_amd import parallel, synth
    from pose2sim_amd.engine import Engine
    wl = synth.make_config(300, 5, 26, 2, seed=3)
    eng = Engine(0); eng.set_calibration(wl['P'])
    prm = eng.tri_params(15.0, 0.3, 2)
    direct = eng.triangulate(wl['xyl'], prm)
    os.environ['P2S_FORCE_COLLECTIVE'] = '1'
    gathered = parallel.sharded_triangulate(lambda x: eng.triangulate(x, prm), wl['xyl'])
    for a, b in zip(direct, gathered):
        assert np.array_equal(a, b, equal_nan=True)
    # the product path's form: results stay on the GPU in one packed buffer (Engine.triangulate_packed), the all-gather
    # takes that buffer as it is and the host copy happens once
    F, Pn, K = wl['xyl'].shape[0], wl['xyl'].shape[1], wl['xyl'].shape[3]
    packed = eng.triangulate_packed(wl['xyl'], prm, pad_blocks=(F + 5) * Pn)       # padded like a largest shard
    assert packed.buf.is_cuda and packed.n_blocks == F * Pn
    Qd, ed, nd, md = parallel.unpack_results(packed.buf.cpu().numpy(), (F + 5) * Pn, K)
    assert np.array_equal(Qd[:F * Pn].reshape(direct[0].shape), direct[0], equal_nan=True)
    assert np.array_equal(ed[:F * Pn].reshape(direct[1].shape), direct[1], equal_nan=True)
    assert np.array_equal(nd[:F * Pn].reshape(direct[2].shape), direct[2]) and np.array_equal(md[:F * Pn].reshape(direct[3].shape), direct[3])
    exact = eng.triangulate_packed(wl['xyl'], prm, pad_blocks=F * Pn)
    whole = parallel.gather_results(exact, F, Pn, K, host_copy_on=0)
    for a, b in zip(direct, whole):
        assert np.array_equal(np.asarray(a).reshape(np.asarray(b).shape), b, equal_nan=True)
    # the product's exchange: the points' section of the device buffer is gathered as it is, the per-unit tables come to
    # this rank's host only, the report's column sums are reduced
    from pose2sim_amd import postproc
    Qf, means, tables = parallel.gather_trajectory(exact, F, Pn, K, skipna=True, host_copy_on=0)
    assert np.array_equal(Qf, direct[0], equal_nan=True)
    assert np.array_equal(tables.err, direct[1], equal_nan=True) and np.array_equal(tables.n_excl, direct[2]) and np.array_equal(tables.mask, direct[3])
    assert np.array_equal(means[:, :, 0], postproc.frame_means(direct[1].reshape(-1, K).astype(np.float64)).reshape(F, Pn), equal_nan=True)
    ids = np.tile(np.arange(Pn), (F, 1))
    sums = parallel.reduce_report_sums(tables, ids, np.array([[0, F]] * Pn), 5)
    np.testing.assert_allclose(sums['err_sum'][1], np.nansum(direct[1][:, 1].astype(np.float64), axis=0), rtol=1e-13)
    assert sums['frames'][0] == F
    # bench.py's form: packed device buffer, asynchronous all-gather into a second buffer
    n = 1 << 20
    src = torch.arange(n, dtype=torch.int64, device='cuda').to(torch.uint8)
    dst = torch.empty(n, dtype=torch.uint8, device='cuda')
    work = dist.all_gather_into_tensor(dst, src, async_op=True)
    work.wait(); torch.cuda.synchronize()
    assert torch.equal(dst, src)
    t = torch.tensor([1.5], dtype=torch.float64, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
    assert float(t.item()) == 1.5
    dist.destroy_process_group()
    print('RCCL-OK')
''') % ROOT


@pytest.mark.gpu
def test_rccl_path_with_one_rank(tmp_path):
    import __graft_entry__ as entry
    entry.build_hip()
    script = tmp_path / 'rccl_one_rank.py'
    script.write_text(SCRIPT)
    port = 29500 + os.getpid() % 2000
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, str(script), str(port)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and 'RCCL-OK' in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.gpu
def test_bench_multi_rank_path_with_one_rank():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, RANK / WORLD_SIZE from the
    environment, backend nccl), with one rank: process group, double-buffered asynchronous all-gather of the
    packed results inside the timed region, max over ranks."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', P2S_BENCH_FORCE_COLLECTIVE='1')
    port = 29600 + os.getpid() % 2000
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '10', '--warmup', '2',
           '--no-cpu-baseline']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['steps'] == 10 and line['value'] > 1e8
    assert 'all-gather' in line['config']['parallelism']
