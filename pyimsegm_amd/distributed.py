"""Multi-GPU plumbing: one process per GPU, images of a batch sharded across ranks, results
gathered on rank 0 (RCCL over xGMI when the backend is ``nccl``; ``gloo`` on CPU for tests).

The reference's only parallelism is a process pool over images
(``imsegm/utilities/experiments.py:392-403``, used at ``pipelines.py:142-150``); this is its multi-GPU
counterpart.  The hot path itself has no data-path collective: every image is segmented entirely on
one GPU.  ``torch`` is used for the process group only.
"""
import os

import numpy as np


class Group(object):
    """thin wrapper around ``torch.distributed`` that degrades to a single process"""

    def __init__(self, backend=None):
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.dist = None
        self.torch = None
        self.device = 'cpu'
        if self.world > 1 or 'RANK' in os.environ:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            if backend is None:
                backend = 'nccl' if torch.cuda.is_available() else 'gloo'
            if backend == 'nccl':
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device('cuda', self.local_rank)
                dist.init_process_group('nccl', device_id=self.device)
            else:
                dist.init_process_group(backend)
            self.backend = backend
        # the HIP library of this process binds to the rank's GPU
        os.environ.setdefault('IMSEGM_HIP_DEVICE', str(self.local_rank))

    # -- work partition --------------------------------------------------------------------------
    def shard(self, n_items):
        """indices of the items this rank owns: item i -> rank i mod world"""
        return list(range(self.rank, n_items, self.world))

    # -- collectives -------------------------------------------------------------------------------
    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
            if self.backend == 'nccl':
                self.torch.cuda.synchronize()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_arrays(self, array, dst=0, keep_on_device=False):
        """gather one equally-shaped array per rank on ``dst`` (numpy, or a device array exposing
        ``__cuda_array_interface__``); returns the list of numpy arrays there, else None"""
        if self.dist is None:
            return [array]
        if hasattr(array, '__cuda_array_interface__'):
            t = self.torch.as_tensor(array, device=self.device)      # zero copy: already in this GPU's HBM
        else:
            t = self.torch.from_numpy(np.ascontiguousarray(array)).to(self.device)
        out = [self.torch.empty_like(t) for _ in range(self.world)] if self.rank == dst else None
        self.dist.gather(t, out, dst=dst)
        if self.rank != dst:
            return None
        return out if keep_on_device else [o.cpu().numpy() for o in out]

    def gather_objects(self, obj, dst=0):
        """gather small picklable objects (feature matrices for the group model) on ``dst``"""
        if self.dist is None:
            return [obj]
        out = [None] * self.world if self.rank == dst else None
        self.dist.gather_object(obj, out, dst=dst)
        return out

    def broadcast_object(self, obj, src=0):
        if self.dist is None:
            return obj
        box = [obj]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


def segment_batch_sharded(list_images, segment_fn, group, nb_workers=1):
    """segment a batch of equally-sized images sharded over the ranks of ``group``.

    ``segment_fn(image) -> label map`` runs on this rank's GPU.  Every rank processes the images
    ``i = rank, rank + world, ...``; label maps travel to rank 0 with one gather per round of
    images.  Returns the full list of label maps on rank 0, ``None`` elsewhere.

    ``nb_workers`` > 1 keeps that many images of this rank in flight: worker threads, each with its
    own HIP stream (``_hip.default_context`` is per thread), so that the host stages of one image
    (class probabilities, edge weights) overlap the kernels of another.
    """
    n = len(list_images)
    mine = group.shard(n)
    rounds = (n + group.world - 1) // group.world
    results = [None] * n if group.rank == 0 else None
    shape = np.asarray(list_images[0]).shape[:2]
    pool = futures = None
    if nb_workers > 1 and len(mine) > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=int(nb_workers))
        futures = [pool.submit(segment_fn, list_images[i]) for i in mine]
    for rnd in range(rounds):
        if rnd < len(mine):
            segm = futures[rnd].result() if futures else segment_fn(list_images[mine[rnd]])
            segm = np.ascontiguousarray(segm, dtype=np.int32)
        else:  # ragged tail: this rank has no image in the last round, contribute a dummy
            segm = np.full(shape, -1, dtype=np.int32)
        parts = group.gather_arrays(segm, dst=0)
        if group.rank == 0:
            for r, part in enumerate(parts):
                idx = rnd * group.world + r
                if idx < n:
                    results[idx] = part
    if pool is not None:
        pool.shutdown()
    return results
