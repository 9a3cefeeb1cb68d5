"""N > 1 path on CPU: 2 processes started by torch.distributed.run (the launcher of the benchmark contract), host control
plane of pyimsegm_amd.distributed (the role gloo played in round 1) -- sharding, per-round gather, max-over-ranks,
the group model across ranks, a failing rank."""
import os
import subprocess
import sys
import textwrap

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


def test_two_rank_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29617', str(script)]
    env = dict(os.environ, OMP_NUM_THREADS='1')
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert 'RANK0_OK' in res.stdout
