"""N > 1 path on CPU: 2 processes with the environment a distributed launcher sets (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_PORT -- nothing of torch is needed), host control plane of pyimsegm_amd.distributed (the role gloo played in
round 1) -- sharding, per-round gather, max-over-ranks, the group model across ranks, a failing rank."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import sys, numpy as np
    sys.path.insert(0, %r)
    from pyimsegm_amd.distributed import Group, segment_batch_sharded
    g = Group(backend='host')
    assert g.world == 2
    images = [np.full((6, 8, 3), i, dtype=np.uint8) for i in range(5)]      # ragged: 5 images, 2 ranks
    assert g.shard(5) == ([0, 2, 4] if g.rank == 0 else [1, 3])
    seen = []
    def fake_segment(img):
        seen.append(int(img[0, 0, 0]))
        return np.full(img.shape[:2], 10 * int(img[0, 0, 0]) + g.rank, dtype=np.int32)
    out = segment_batch_sharded(images, fake_segment, g)
    t = g.max_over_ranks(1.0 + g.rank)
    assert t == 2.0
    feats = g.gather_objects(np.ones((2 + g.rank, 3)) * g.rank)
    model = g.broadcast_object({'fitted_on': None if feats is None else sum(len(f) for f in feats)})
    assert model['fitted_on'] == 5
    if g.rank == 0:
        assert [int(o[0, 0]) for o in out] == [0, 11, 20, 31, 40], [int(o[0, 0]) for o in out]
        assert all(o.shape == (6, 8) and o.dtype == np.int32 for o in out)
        print('RANK0_OK')
    else:
        assert out is None
    assert seen == g.shard(5)
    # several images of a rank in flight (worker threads): same sharding, same result order
    out2 = segment_batch_sharded(images, lambda img: np.full(img.shape[:2], 10 * int(img[0, 0, 0]) + g.rank, dtype=np.int32),
                                 g, nb_workers=3)
    if g.rank == 0:
        assert [int(o[0, 0]) for o in out2] == [0, 11, 20, 31, 40]
    else:
        assert out2 is None
    # group model across ranks (pipelines.py:142-155): features of image i from rank i mod world, fitted on rank 0, broadcast
    from pyimsegm_amd.distributed import estim_model_classes_group_sharded
    model, blocks = estim_model_classes_group_sharded(
        images, lambda img: np.full((2, 3), float(img[0, 0, 0])), lambda fts: {'mean': float(fts.mean()), 'rows': len(fts)}, g)
    assert model == {'mean': 2.0, 'rows': 10}, model
    if g.rank == 0:
        assert [float(b[0, 0]) for b in blocks] == [0., 1., 2., 3., 4.]
    # a rank that fails keeps the collective pattern alive and the error surfaces on every rank
    def failing(img):
        if g.rank == 1 and int(img[0, 0, 0]) == 3:
            raise ValueError('boom')
        return np.zeros(img.shape[:2], dtype=np.int32)
    try:
        segment_batch_sharded(images, failing, g)
        raise SystemExit('no error raised')
    except (ValueError, RuntimeError) as ex:
        assert ('boom' in str(ex)) == (g.rank == 1)
    assert segment_batch_sharded([], failing, g) == ([] if g.rank == 0 else None)
    g.close()
''') % ROOT


def test_two_rank_host_plane(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, OMP_NUM_THREADS='1', RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT='29617', TORCHELASTIC_RUN_ID='pytest%d' % os.getpid())
        procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, out[-2000:] + err[-2000:]
    assert 'RANK0_OK' in outs[0][0]


WORKER_EIGHT = textwrap.dedent('''
    import sys, numpy as np
    sys.path.insert(0, %r)
    from pyimsegm_amd.distributed import Group, segment_batch_sharded, estim_model_classes_group_sharded, worker_threads_per_rank
    g = Group(backend='host')
    assert g.world == 8
    # BASELINE configs[3]: 64 images over 8 ranks, image i on rank i mod 8 (SURVEY 8e) -- and a ragged count
    assert g.shard(64) == list(range(g.rank, 64, 8)) and g.shard(19) == list(range(g.rank, 19, 8))
    assert g.shard(3) == ([g.rank] if g.rank < 3 else [])
    images = [np.full((5, 7, 3), i, dtype=np.uint8) for i in range(19)]
    out = segment_batch_sharded(images, lambda img: np.full(img.shape[:2], 100 * int(img[0, 0, 0]) + g.rank, dtype=np.int32), g,
                                nb_workers=3)
    if g.rank == 0:
        assert [int(o[0, 0]) for o in out] == [100 * i + i %% 8 for i in range(19)]
        print('RANK0_OK')
    else:
        assert out is None
    assert g.max_over_ranks(float(g.rank)) == 7.0 and g.min_over_ranks(float(g.rank)) == 0.0
    assert g.any_over_ranks(g.rank == 5) is True and g.any_over_ranks(False) is False
    blocks = g.gather_objects({'rank': g.rank, 'payload': np.arange(1000 * (g.rank + 1))})
    if g.rank == 0:
        assert [b['rank'] for b in blocks] == list(range(8)) and [len(b['payload']) for b in blocks] == [1000 * (r + 1) for r in range(8)]
    assert g.broadcast_object({'k': 11} if g.rank == 0 else None) == {'k': 11}
    model, _ = estim_model_classes_group_sharded(images, lambda img: np.full((2, 3), float(img[0, 0, 0])),
                                                 lambda fts: {'mean': float(fts.mean()), 'rows': len(fts)}, g)
    assert model == {'mean': 9.0, 'rows': 38}, model
    # worker threads per rank: never more than the CPUs a rank has (eight ranks on a small host: at least one each)
    assert worker_threads_per_rank(8, 4, cpus=3) == 3 and worker_threads_per_rank(8, 4, cpus=1) == 1
    # a rank that dies inside a step: every rank sees an error, nobody hangs in the gather
    def failing(img):
        if g.rank == 6:
            raise ValueError('boom')
        return np.zeros(img.shape[:2], dtype=np.int32)
    try:
        segment_batch_sharded(images, failing, g)
        raise SystemExit('no error raised')
    except (ValueError, RuntimeError) as ex:
        assert ('boom' in str(ex)) == (g.rank == 6)
    g.barrier()
    g.close()
''') % ROOT


def test_eight_rank_host_plane(tmp_path):
    """the world size of the driver's scaling run, which cannot be rehearsed on the one-GPU test box: eight processes on the host
    control plane (hub accept loop, gathers of ragged payloads, broadcast, the group model, sharding of 64 / 19 / 3 images, a rank
    that fails inside a step) -- reference: the pool map of imsegm/utilities/experiments.py:386-411"""
    script = tmp_path / 'worker8.py'
    script.write_text(WORKER_EIGHT)
    procs = []
    for rank in range(8):
        env = dict(os.environ, OMP_NUM_THREADS='1', RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='8', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT='29619', TORCHELASTIC_RUN_ID='pytest8_%d' % os.getpid())
        procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, out[-2000:] + err[-2000:]
    assert 'RANK0_OK' in outs[0][0]


def test_control_plane_rejects_stray_connections(tmp_path):
    """a connection that announces a rank outside 1..world-1 (or one that is taken) is dropped, the job still forms"""
    import socket
    import struct
    import threading
    import time
    sys.path.insert(0, ROOT)
    from pyimsegm_amd.distributed import _Star
    path = str(tmp_path / 'hub.sock')
    os.environ['IMSEGM_COMM_SOCKET'] = path
    try:
        hub = {}
        t = threading.Thread(target=lambda: hub.setdefault('star', _Star(0, 2, timeout=30.)))
        t.start()
        deadline = time.time() + 10
        while not os.path.exists(path) and time.time() < deadline:
            time.sleep(0.02)
        stray = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        while True:                                      # (the file appears with bind(), connections are taken after listen())
            try:
                stray.connect(path)
                break
            except OSError:
                assert time.time() < deadline, 'the hub never listened'
                time.sleep(0.02)
        stray.sendall(struct.pack('<i', 7))              # not a rank of this job
        mute = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        mute.connect(path)
        mute.close()                                     # connects and never says who it is: dropped, the hub stays up
        peer = _Star(1, 2, timeout=30.)
        t.join(30)
        assert sorted(hub['star'].peers) == [1]
        peer.close()
        hub['star'].close()
        stray.close()
        assert not os.path.exists(path)
    finally:
        del os.environ['IMSEGM_COMM_SOCKET']


def test_control_plane_directory_must_be_a_private_directory_not_a_link(tmp_path, monkeypatch):
    """the hub's socket lives in <runtime dir>/imsegm-<uid>: a symbolic link planted there (to a directory the user owns, mode
    0700 -- os.stat would be satisfied) is refused"""
    sys.path.insert(0, ROOT)
    from pyimsegm_amd.distributed import _Star
    target = tmp_path / 'elsewhere'
    target.mkdir(mode=0o700)
    runtime = tmp_path / 'run'
    runtime.mkdir()
    os.symlink(str(target), str(runtime / ('imsegm-%d' % os.getuid())))
    monkeypatch.setenv('XDG_RUNTIME_DIR', str(runtime))
    monkeypatch.delenv('IMSEGM_COMM_SOCKET', raising=False)
    with pytest.raises(RuntimeError, match='private'):
        _Star(0, 1, timeout=5.)


def _fake_sysfs(root, devices):
    """a sysfs tree with PCI devices {bus id: numa node} and nodes {node: cpulist}"""
    nodes = {}
    for bus_id, (node, cpulist) in devices.items():
        d = root / 'bus' / 'pci' / 'devices' / bus_id
        d.mkdir(parents=True)
        (d / 'numa_node').write_text('%d\n' % node)
        if node >= 0:
            nodes[node] = cpulist
    for node, cpulist in nodes.items():
        d = root / 'devices' / 'system' / 'node' / ('node%d' % node)
        d.mkdir(parents=True)
        (d / 'cpulist').write_text(cpulist + '\n')
    return str(root)


def test_numa_placement_of_a_rank(tmp_path):
    """bind_to_device_numa_node: CPUs of the GPU's NUMA node from sysfs; an even share of the allowed CPUs when the kernel
    does not say; nothing for a single rank without NUMA information"""
    sys.path.insert(0, ROOT)
    from pyimsegm_amd import distributed as dist
    assert dist.parse_cpu_list('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    allowed = sorted(os.sched_getaffinity(0))
    half = max(1, len(allowed) // 2)
    lists = ['%s' % ','.join(str(c) for c in allowed[:half]), '%s' % ','.join(str(c) for c in allowed[half:] or allowed[:1])]
    sysfs = _fake_sysfs(tmp_path / 'sys', {'0000:05:00.0': (0, lists[0]), '0000:85:00.0': (1, lists[1]), '0000:c1:00.0': (-1, '')})
    assert dist.numa_node_cpus('0000:85:00.0', sysfs) == (1, dist.parse_cpu_list(lists[1]))
    assert dist.numa_node_cpus('0000:c1:00.0', sysfs) == (None, None)           # single-node box: -1
    assert dist.numa_node_cpus('0000:ff:00.0', sysfs) == (None, None)           # no such device
    got = dist.bind_to_device_numa_node('0000:05:00.0', world=2, local_rank=0, sysfs=sysfs, apply=False)
    assert got['numa_node'] == 0 and got['cpus'] == half and 'numa node' in got['how']
    got = dist.bind_to_device_numa_node('0000:c1:00.0', world=2, local_rank=1, sysfs=sysfs, apply=False)
    assert got['numa_node'] is None and got['cpus'] == half and 'even share' in got['how']
    got = dist.bind_to_device_numa_node('0000:c1:00.0', world=1, sysfs=sysfs, apply=False)
    assert got['cpus'] == len(allowed) and got['how'].startswith('unchanged')
    assert dist.worker_threads_per_rank(8, 12, cpus=4) == 4 and dist.worker_threads_per_rank(1, 3, cpus=64) == 3


def test_two_ranks_bind_to_the_nodes_of_their_gpus(tmp_path):
    """two processes, each applies the affinity of 'its' device for real and reports what the kernel then allows it"""
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip('needs two CPUs')
    half = len(allowed) // 2
    lists = [','.join(str(c) for c in allowed[:half]), ','.join(str(c) for c in allowed[half:])]
    sysfs = _fake_sysfs(tmp_path / 'sys', {'0000:05:00.0': (0, lists[0]), '0000:85:00.0': (1, lists[1])})
    code = textwrap.dedent('''
        import os, sys, threading
        sys.path.insert(0, %r)
        from pyimsegm_amd import distributed as dist
        rank = int(sys.argv[1])
        got = dist.bind_to_device_numa_node(['0000:05:00.0', '0000:85:00.0'][rank], world=2, local_rank=rank, sysfs=%r)
        seen = []
        t = threading.Thread(target=lambda: seen.append(sorted(os.sched_getaffinity(0))))     # threads started later inherit it
        t.start(); t.join()
        print('AFFINITY', got['numa_node'], sorted(os.sched_getaffinity(0)) == seen[0], ','.join(str(c) for c in seen[0]))
    ''') % (ROOT, sysfs)
    procs = [subprocess.Popen([sys.executable, '-c', code, str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
             for r in range(2)]
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err[-2000:]
        line = [ln for ln in out.splitlines() if ln.startswith('AFFINITY')][-1].split()
        assert line[1] == str(r) and line[2] == 'True' and line[3] == lists[r]
