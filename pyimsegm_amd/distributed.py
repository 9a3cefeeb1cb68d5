"""Multi-GPU plumbing: one process per GPU, images of a batch sharded across ranks, label maps gathered on rank 0.

The reference's only parallelism is a process pool over images (``imsegm/utilities/experiments.py:392-403``, used at
``pipelines.py:142-150``); this is its multi-GPU counterpart.  The hot path itself has no data-path
collective: every image is segmented entirely on one GPU.  Two planes, neither of them PyTorch:

* **data plane** -- RCCL, bound directly with ctypes (``librccl.so``: ``ncclGetUniqueId``, ``ncclCommInitRank``, grouped
  ``ncclSend`` / ``ncclRecv``): label maps travel from every rank's HBM to rank 0's HBM over xGMI, a whole round of
  images per call, on the stream of the calling context (:class:`DeviceGather`);
* **control plane** -- a few small host-side exchanges (barrier, maximum of a float, the RCCL unique id, feature
  matrices / model parameters of the group model): length-prefixed pickles over a Unix-domain socket with rank 0 as the
  hub (one node, as the benchmark contract says; :class:`_Star`).  The same plane carries numpy arrays when no GPU is
  present -- that is what the world-size-2 CPU tests run on.

Ranks are taken from the environment a launcher such as ``python -m torch.distributed.run`` sets (``RANK``,
``LOCAL_RANK``, ``WORLD_SIZE``, ``MASTER_PORT``); nothing of ``torch`` is imported.
"""
import ctypes as C
import os
import pickle
import socket
import stat
import struct
import time

import numpy as np


# ---------------------------------------------------------------------------------------------------------
# control plane
# ---------------------------------------------------------------------------------------------------------
def _send_msg(sock, data):
    sock.sendall(struct.pack('<Q', len(data)))
    sock.sendall(data)


def _recv_exact(sock, size):
    chunks = []
    while size:
        chunk = sock.recv(min(size, 1 << 20))
        if not chunk:
            raise ConnectionError('peer closed the control connection')
        chunks.append(chunk)
        size -= len(chunk)
    return b''.join(chunks)


def _recv_msg(sock):
    size, = struct.unpack('<Q', _recv_exact(sock, 8))
    return _recv_exact(sock, size)


class _Star(object):
    """host-side exchanges of one node: rank 0 listens on a Unix-domain socket, every other rank keeps one connection"""

    def __init__(self, rank, world, timeout=300.):
        self.rank, self.world = rank, world
        tag = '%s-%s' % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'run'))
        tag = ''.join(ch if ch.isalnum() or ch in '-_' else '_' for ch in tag)[:60]
        # the socket lives in a directory only this user can enter (0700, ownership checked): nobody else can connect
        # to the hub or put a socket of their own at the agreed path
        base = os.path.join(os.environ.get('XDG_RUNTIME_DIR') or '/tmp', 'imsegm-%d' % os.getuid())
        self.path = os.environ.get('IMSEGM_COMM_SOCKET')
        if not self.path:
            os.makedirs(base, mode=0o700, exist_ok=True)
            st = os.lstat(base)             # (lstat: a symbolic link planted at the predictable /tmp path is refused, not followed)
            if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
                raise RuntimeError('control-plane directory %s is not a directory private to this user' % base)
            self.path = os.path.join(base, '%s.sock' % tag)
        self.peers = {}
        self.sock = None
        self.server = None
        deadline = time.time() + timeout
        if rank == 0:
            try:
                os.unlink(self.path)
            except OSError:
                pass
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(self.path)
            srv.listen(world)
            srv.settimeout(timeout)
            self.server = srv
            try:
                while len(self.peers) < world - 1:
                    srv.settimeout(max(deadline - time.time(), 0.05))
                    conn, _ = srv.accept()
                    # a rank sends its number right after connecting: a connection that stays silent (or closes) is a stray one
                    # and is dropped like one with an invalid number -- it does not take the hub down
                    conn.settimeout(min(10., timeout))
                    try:
                        peer, = struct.unpack('<i', _recv_exact(conn, 4))
                    except (socket.timeout, ConnectionError, OSError):
                        conn.close()
                        continue
                    if not (1 <= peer < world) or peer in self.peers:       # a stray or duplicate connection: drop it
                        conn.close()
                        continue
                    conn.settimeout(None)
                    self.peers[peer] = conn
            except BaseException:
                self.close()
                raise
        else:
            while True:
                sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    sock.connect(self.path)
                    sock.sendall(struct.pack('<i', rank))
                    self.sock = sock
                    break
                except OSError:
                    sock.close()
                    if time.time() > deadline:
                        raise RuntimeError('rank %d: no control connection to rank 0 at %s' % (rank, self.path))
                    time.sleep(0.05)

    def gather(self, data):
        """bytes of every rank on rank 0 (list indexed by rank), None elsewhere"""
        if self.rank == 0:
            out = [data] + [None] * (self.world - 1)
            for peer, conn in self.peers.items():
                out[peer] = _recv_msg(conn)
            return out
        _send_msg(self.sock, data)
        return None

    def bcast(self, data):
        """bytes of rank 0 on every rank"""
        if self.rank == 0:
            for conn in self.peers.values():
                _send_msg(conn, data)
            return data
        return _recv_msg(self.sock)

    def close(self):
        if self.rank == 0:
            for conn in self.peers.values():
                conn.close()
            self.peers = {}
            if self.server is not None:
                self.server.close()
                self.server = None
            try:
                os.unlink(self.path)
            except OSError:
                pass
        elif self.sock is not None:
            self.sock.close()


# ---------------------------------------------------------------------------------------------------------
# data plane: RCCL through ctypes
# ---------------------------------------------------------------------------------------------------------
class _NcclUniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]


NCCL_INT8, NCCL_UINT8, NCCL_INT32, NCCL_FLOAT64 = 0, 1, 2, 8


class Rccl(object):
    """one RCCL communicator of this process (``ncclCommInitRank`` over all ranks of the group)"""

    def __init__(self, group):
        path = os.environ.get('IMSEGM_RCCL_LIBRARY', '')
        names = [path] if path else ['librccl.so', 'librccl.so.1', '/opt/rocm/lib/librccl.so']
        lib = None
        for name in names:
            try:
                lib = C.CDLL(name)
                break
            except OSError:
                continue
        if lib is None:
            raise RuntimeError('librccl.so not found')
        vp = C.c_void_p
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [C.c_int]
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, _NcclUniqueId, C.c_int]
        lib.ncclCommDestroy.argtypes = [vp]
        lib.ncclSend.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
        lib.ncclRecv.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
        lib.ncclGroupStart.argtypes = []
        lib.ncclGroupEnd.argtypes = []
        for fn in ('ncclGetUniqueId', 'ncclCommInitRank', 'ncclCommDestroy', 'ncclSend', 'ncclRecv', 'ncclGroupStart', 'ncclGroupEnd'):
            getattr(lib, fn).restype = C.c_int
        self.lib, self.group = lib, group
        self.comm = None

    def connect(self):
        """collective over all ranks: unique id from rank 0 over the control plane, ``ncclCommInitRank``"""
        from pyimsegm_amd import _hip
        lib, group = self.lib, self.group
        _hip._check(_hip.load_library().imsegm_set_device(group.device_index))
        uid = _NcclUniqueId()
        status = lib.ncclGetUniqueId(C.byref(uid)) if group.rank == 0 else 0
        raw = group._bcast_bytes(C.string_at(C.byref(uid), 128) if (group.rank == 0 and status == 0) else b'')
        self._ok(status, 'ncclGetUniqueId')
        if len(raw) != 128:
            raise RuntimeError('rank 0 could not create the RCCL unique id')
        C.memmove(C.byref(uid), raw, 128)
        comm = C.c_void_p()
        self._ok(lib.ncclCommInitRank(C.byref(comm), group.world, uid, group.rank), 'ncclCommInitRank')
        self.comm = comm

    def _ok(self, status, what):
        if status != 0:
            raise RuntimeError('%s failed: %s' % (what, self.lib.ncclGetErrorString(status).decode('utf-8', 'replace')))

    def gather_to_root(self, send_ptr, recv_ptr, nbytes, stream, root=0):
        """every rank sends ``nbytes`` from ``send_ptr``; the root receives rank r's block at ``recv_ptr + r * nbytes``
        (device pointers; one grouped call = one launch per peer pair, all links in parallel)"""
        lib, g = self.lib, self.group
        vp = C.c_void_p
        self._ok(lib.ncclGroupStart(), 'ncclGroupStart')
        if g.rank == root:
            for peer in range(g.world):
                self._ok(lib.ncclRecv(vp(recv_ptr + peer * nbytes), nbytes, NCCL_UINT8, peer, self.comm, vp(stream)), 'ncclRecv')
        self._ok(lib.ncclSend(vp(send_ptr), nbytes, NCCL_UINT8, root, self.comm, vp(stream)), 'ncclSend')
        self._ok(lib.ncclGroupEnd(), 'ncclGroupEnd')

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


# ---------------------------------------------------------------------------------------------------------
# placement of a rank on the host: the NUMA node of its GPU
# ---------------------------------------------------------------------------------------------------------
def parse_cpu_list(text):
    """``'0-3,8,10-11'`` (the format of /sys/devices/system/node/node*/cpulist) -> sorted list of CPU numbers"""
    cpus = set()
    for part in text.replace('\n', '').split(','):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return sorted(cpus)


def numa_node_cpus(pci_bus_id, sysfs='/sys'):
    """(NUMA node, its CPUs) of the PCI device ``pci_bus_id`` from sysfs; (None, None) when the kernel does not say (a
    single-node box reports -1, containers may hide the files)"""
    try:
        with open(os.path.join(sysfs, 'bus', 'pci', 'devices', pci_bus_id, 'numa_node')) as fp:
            node = int(fp.read().strip())
        if node < 0:
            return None, None
        with open(os.path.join(sysfs, 'devices', 'system', 'node', 'node%d' % node, 'cpulist')) as fp:
            cpus = parse_cpu_list(fp.read())
        return (node, cpus) if cpus else (None, None)
    except (OSError, ValueError):
        return None, None


def bind_to_device_numa_node(pci_bus_id, world=1, local_rank=0, sysfs='/sys', apply=True):
    """Pin this process (and the threads it starts from now on) to the CPUs of the NUMA node its GPU hangs off, so that the
    pages it touches first -- input staging, result arrays -- and the threads that feed the GPU sit next to it.  Ranks that share
    a node keep the whole node (the kernel balances them).  Without the sysfs entries (or on a single-node box) the rank takes
    an even share ``cpus // world`` of what it may run on, or nothing is changed when ``world`` is 1.
    Returns a dict for the benchmark record: {'numa_node', 'cpus': count, 'how'}."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else list(range(os.cpu_count() or 1))
    node, cpus = numa_node_cpus(pci_bus_id, sysfs) if pci_bus_id else (None, None)
    how = 'numa node of the GPU'
    if cpus:
        cpus = [c for c in cpus if c in set(allowed)]
    if not cpus:
        node, how = None, 'even share of the allowed CPUs (no NUMA information)'
        if world <= 1:
            return {'numa_node': None, 'cpus': len(allowed), 'how': 'unchanged (one rank, no NUMA information)'}
        share = max(1, len(allowed) // world)
        cpus = allowed[(local_rank % world) * share:(local_rank % world + 1) * share] or allowed
    if apply and hasattr(os, 'sched_setaffinity'):
        try:
            os.sched_setaffinity(0, cpus)
        except OSError as ex:
            return {'numa_node': node, 'cpus': len(allowed), 'how': 'unchanged (%s)' % ex}
    return {'numa_node': node, 'cpus': len(cpus), 'how': how}


def worker_threads_per_rank(world, wanted, cpus=None):
    """images (worker threads) in flight a rank may run: what it wants, capped at its share of the host's CPUs"""
    if cpus is None:
        cpus = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    return max(1, min(int(wanted), max(1, int(cpus))))


class Group(object):
    """the ranks of one job on one node; degrades to a single process"""

    def __init__(self, backend=None, single=False):
        """``backend``: None = RCCL when a GPU is visible and the job has more than one rank (or was launched by a
        distributed launcher), host sockets otherwise; ``'host'`` / ``'gloo'`` force the host plane (CPU tests);
        ``single``: a group of this process alone whatever the environment says"""
        self.world = 1 if single else int(os.environ.get('WORLD_SIZE', '1'))
        self.rank = 0 if single else int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.distributed = not single and (self.world > 1 or 'RANK' in os.environ)
        self.star = _Star(self.rank, self.world) if self.world > 1 else None
        # the HIP library of this process binds to the rank's GPU
        os.environ.setdefault('IMSEGM_HIP_DEVICE', str(self.local_rank))
        self.device_index = 0
        self.rccl = None
        self.backend = 'host'
        self.rccl_error = None
        self.placement = None
        if backend in (None, 'rccl', 'nccl') and self.distributed:
            # stage 1 (local): a GPU and the library; stage 2 (collective, only if every rank passed stage 1): the communicator
            rccl = None
            try:
                from pyimsegm_amd import _hip
                n = _hip.device_count()
                if n > 0:
                    self.device_index = self.local_rank % n
                    rccl = Rccl(self)
            except Exception as ex:      # no GPU / no library: every rank falls back to the host plane
                self.rccl_error = repr(ex)
            # the host side of the rank next to its GPU (sched_setaffinity to the CPUs of the GPU's NUMA node)
            self.placement = None
            if self.world > 1 and rccl is not None and os.environ.get('IMSEGM_NO_NUMA_BIND') is None:
                try:
                    from pyimsegm_amd import _hip
                    self.placement = bind_to_device_numa_node(_hip.device_pci_bus_id(self.device_index), self.world, self.local_rank)
                except Exception as ex:
                    self.placement = {'numa_node': None, 'cpus': None, 'how': 'unchanged (%r)' % (ex, )}
            if self.min_over_ranks(1 if rccl is not None else 0) > 0:
                ok = 0
                try:
                    rccl.connect()
                    ok = 1
                except Exception as ex:
                    self.rccl_error = repr(ex)
                if self.min_over_ranks(ok) > 0:
                    self.rccl, self.backend = rccl, 'rccl'
                else:
                    rccl.close()

    # -- work partition --------------------------------------------------------------------------
    def shard(self, n_items):
        """indices of the items this rank owns: item i -> rank i mod world"""
        return list(range(self.rank, n_items, self.world))

    # -- control plane ---------------------------------------------------------------------------
    def _gather_bytes(self, data):
        return [data] if self.star is None else self.star.gather(data)

    def _bcast_bytes(self, data):
        return data if self.star is None else self.star.bcast(data)

    def barrier(self):
        if self.star is not None:
            self.star.gather(b'')
            self.star.bcast(b'')

    def gather_objects(self, obj, dst=0):
        """gather small picklable objects (feature matrices for the group model) on rank 0"""
        if dst != 0:
            raise ValueError('rank 0 is the hub of the control plane')
        parts = self._gather_bytes(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))
        return None if parts is None else [pickle.loads(p) for p in parts]

    def broadcast_object(self, obj, src=0):
        if src != 0:
            raise ValueError('rank 0 is the hub of the control plane')
        return pickle.loads(self._bcast_bytes(pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL) if self.rank == 0 else b''))

    def max_over_ranks(self, value):
        parts = self.gather_objects(float(value))
        return float(self.broadcast_object(max(parts) if parts is not None else None))

    def min_over_ranks(self, value):
        parts = self.gather_objects(float(value))
        return float(self.broadcast_object(min(parts) if parts is not None else None))

    def any_over_ranks(self, flag):
        return self.max_over_ranks(1.0 if flag else 0.0) > 0

    def gather_arrays(self, array, dst=0):
        """gather one numpy array per rank on rank 0 over the host plane; list of arrays there, None elsewhere"""
        array = np.ascontiguousarray(array)
        parts = self.gather_objects((array.shape, array.dtype.str, array.tobytes()), dst=dst)
        if parts is None:
            return None
        return [np.frombuffer(raw, dtype=np.dtype(dt)).reshape(shape).copy() for shape, dt, raw in parts]

    def close(self):
        if self.rccl is not None:
            self.rccl.close()
            self.rccl = None
        if self.star is not None:
            try:
                self.barrier()
            except Exception:
                pass
            self.star.close()
            self.star = None


class DeviceGather(object):
    """rounds of equally sized device buffers (label maps) from every rank's HBM to rank 0's HBM over RCCL.

    A worker stages a finished label map into a slot of the send ring (device-to-device copy on its own stream, so
    its session can start the next image at once); when ``items_per_round`` slots are filled on every rank, ONE grouped
    ``ncclSend`` / ``ncclRecv`` moves the whole round.  Rank 0 keeps the latest round of all ranks in ``recv``
    (world x items_per_round x item_bytes, in HBM).  Without RCCL (CPU tests, one process) the round travels over the
    host plane instead."""

    def __init__(self, group, item_bytes, items_per_round, ctx=None, depth=2):
        from pyimsegm_amd import _hip
        self.group, self.item_bytes, self.per_round = group, int(item_bytes), int(items_per_round)
        self.depth = max(2, int(depth))          # rounds the send ring holds: a slow rank stalls the others only beyond that
        self.ctx = ctx or _hip.default_context()
        self.hip = _hip
        self.round_bytes = self.item_bytes * self.per_round
        self.rounds_done = 0
        self.send = self.recv = None
        if group.rccl is not None:
            lib = _hip.load_library()
            p = C.c_void_p()
            _hip._check(lib.imsegm_device_alloc(group.device_index, self.depth * self.round_bytes, C.byref(p)))
            self.send = p.value
            if group.rank == 0:
                q = C.c_void_p()
                _hip._check(lib.imsegm_device_alloc(group.device_index, group.world * self.round_bytes, C.byref(q)))
                self.recv = q.value
        self.host_round = [None] * self.per_round          # host-plane fallback

    def slot_ptr(self, round_index, item):
        return self.send + (round_index % self.depth) * self.round_bytes + item * self.item_bytes

    def stage(self, round_index, item, src, worker_ctx=None, offset=0, nbytes=None):
        """put (a part of) item ``item`` of round ``round_index`` into the ring.  ``src``: a device array (anything with
        ``__cuda_array_interface__``; copied device to device on the worker's stream) or, on the host plane, a numpy array"""
        nbytes = self.item_bytes if nbytes is None else int(nbytes)
        if self.group.rccl is not None:
            ptr = src.__cuda_array_interface__['data'][0] if hasattr(src, '__cuda_array_interface__') else int(src)
            (worker_ctx or self.ctx).copy(self.slot_ptr(round_index, item) + offset, ptr, nbytes, synchronize=True)
        else:
            slot = self.host_round[item]
            if slot is None:
                slot = self.host_round[item] = {}
            if hasattr(src, '__cuda_array_interface__'):      # ranks that share one GPU (no RCCL communicator): through the host
                host = np.empty(nbytes, dtype=np.uint8)
                (worker_ctx or self.ctx).copy(host.ctypes.data, src.__cuda_array_interface__['data'][0], nbytes, synchronize=True)
                slot[offset] = host
            else:
                slot[offset] = np.array(src, copy=True)

    def flush(self, round_index):
        """collective: move round ``round_index`` (all its items staged on every rank) to rank 0"""
        g = self.group
        if g.rccl is not None:
            g.rccl.gather_to_root(self.slot_ptr(round_index, 0), self.recv or 0, self.round_bytes, self.ctx.stream)
            self.ctx.synchronize()
            out = None
        else:
            items = [[slot[k] for k in sorted(slot)] for slot in self.host_round if slot is not None]
            out = g.gather_objects(items)
            self.host_round = [None] * self.per_round
        self.rounds_done += 1
        return out

    def close(self):
        lib = self.hip.load_library()
        if self.send:
            lib.imsegm_device_free(C.c_void_p(self.send))
        if self.recv:
            lib.imsegm_device_free(C.c_void_p(self.recv))
        self.send = self.recv = None


def estim_model_classes_group_sharded(list_images, features_fn, fit_fn, group):
    """the group model across ranks (reference ``pipelines.py:142-155``): every rank extracts the
    K_i x F features of its images ``i = rank, rank + world, ...``, the blocks are gathered on rank 0 in image order,
    the model is fitted there once and broadcast (its parameters are a few hundred doubles)

    :param features_fn: image -> K x F features (runs on this rank's GPU)
    :param fit_fn: concatenated features -> fitted model (runs on rank 0)
    :return tuple(model, list(ndarray)): the model on every rank; all feature blocks on rank 0, this rank's elsewhere
    """
    n = len(list_images)
    mine = group.shard(n)
    local = [features_fn(list_images[i]) for i in mine]
    parts = group.gather_objects(list(zip(mine, local)))
    model, blocks = None, local
    if group.rank == 0:
        by_index = dict(pair for part in parts for pair in part)
        blocks = [by_index[i] for i in range(n)]
        model = fit_fn(np.nan_to_num(np.concatenate(tuple(blocks), axis=0)))
    model = group.broadcast_object(model)
    return model, blocks


def segment_batch_sharded(list_images, segment_fn, group, nb_workers=1, segment_device_fn=None):
    """segment a batch of equally-sized images sharded over the ranks of ``group``.

    ``segment_fn(image) -> label map`` runs on this rank's GPU.  Every rank processes the images
    ``i = rank, rank + world, ...``; label maps travel to rank 0 one round of images at a time.  Returns the full
    list of label maps on rank 0, ``None`` elsewhere.

    With RCCL (``group.rccl``) and ``segment_device_fn(image) -> object with .device_ptr / .ctx / .close()`` the label maps
    never visit the host of their own rank: each worker stages its finished map into the send ring in HBM, one grouped
    ``ncclSend`` / ``ncclRecv`` per round of ``nb_workers`` images per rank moves them to rank 0's HBM over xGMI, and only
    rank 0 copies them to its host.  Otherwise (CPU tests, one process) the maps travel over the host plane.

    ``nb_workers`` > 1 keeps that many images of this rank in flight: worker threads, each with its
    own HIP stream (``_hip.default_context`` is per thread).  A rank whose ``segment_fn`` fails still takes part in
    every collective (with a dummy map), the error is raised on all ranks after the last round.
    """
    n = len(list_images)
    if n == 0:
        return [] if group.rank == 0 else None
    mine = group.shard(n)
    shape = np.asarray(list_images[0]).shape[:2]
    nb_workers = max(1, int(nb_workers or 1))
    if group.rccl is not None and segment_device_fn is not None:
        return _segment_batch_rccl(list_images, segment_device_fn, group, nb_workers, mine, shape)
    rounds = (n + group.world - 1) // group.world
    results = [None] * n if group.rank == 0 else None
    pool = futures = None
    error = None
    try:
        if nb_workers > 1 and len(mine) > 1:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=nb_workers)
            futures = [pool.submit(segment_fn, list_images[i]) for i in mine]
        for rnd in range(rounds):
            segm = None
            if rnd < len(mine) and error is None:
                try:
                    segm = futures[rnd].result() if futures else segment_fn(list_images[mine[rnd]])
                    segm = np.ascontiguousarray(segm)
                except Exception as ex:            # keep the collective pattern alive, report after the loop
                    error = ex
            if segm is None:    # ragged tail (no image of this rank in the last round) or a failed image: a dummy
                segm = np.full(shape, -1, dtype=np.int32)
            parts = group.gather_arrays(segm, dst=0)
            if group.rank == 0:
                for r, part in enumerate(parts):
                    idx = rnd * group.world + r
                    if idx < n:
                        results[idx] = part
    finally:
        if pool is not None:
            pool.shutdown()
    failed = group.any_over_ranks(error is not None)
    if error is not None:
        raise error
    if failed:
        raise RuntimeError('segmentation failed on another rank')
    return results


def _segment_batch_rccl(list_images, segment_device_fn, group, nb_workers, mine, shape):
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from pyimsegm_amd import _hip
    n = len(list_images)
    per_round = nb_workers
    max_mine = (n + group.world - 1) // group.world                  # images of rank 0 = the most any rank has
    rounds = (max_mine + per_round - 1) // per_round
    item_bytes = int(shape[0]) * int(shape[1]) * 4
    ctx = _hip.default_context()
    gather = DeviceGather(group, item_bytes, per_round, ctx, depth=4)
    cond = threading.Condition()
    flushed = [0]
    aborted = [False]
    errors = []

    def work(pos):
        rnd, item = divmod(pos, per_round)
        res = None
        try:
            res = segment_device_fn(list_images[mine[pos]])
            with cond:                                               # the ring holds `gather.depth` rounds
                cond.wait_for(lambda: aborted[0] or flushed[0] >= rnd - (gather.depth - 1))
            if aborted[0]:
                return
            gather.stage(rnd, item, res.device_ptr, res.ctx)
        except Exception as ex:
            errors.append(ex)
        finally:
            if res is not None:
                res.close()

    results = [None] * n if group.rank == 0 else None
    host = _hip.pinned_empty((group.world, per_round) + tuple(shape), np.int32) if group.rank == 0 else None
    pool = ThreadPoolExecutor(max_workers=nb_workers)
    try:
        futures = [pool.submit(work, pos) for pos in range(len(mine))]
        for rnd in range(rounds):
            for fut in futures[rnd * per_round:(rnd + 1) * per_round]:
                fut.result()
            gather.flush(rnd)                                        # collective: every rank, every round
            with cond:
                flushed[0] = rnd + 1
                cond.notify_all()
            if group.rank == 0:
                ctx.copy(host.ctypes.data, gather.recv, host.nbytes, synchronize=True)
                for r in range(group.world):
                    for item in range(per_round):
                        idx = (rnd * per_round + item) * group.world + r
                        if idx < n:
                            results[idx] = host[r, item].copy()
    finally:
        with cond:                    # a failed flush / copy / interrupt: wake the workers that wait for a free ring slot,
            aborted[0] = True         # else shutdown() would wait for them forever and the error never surface
            cond.notify_all()
        pool.shutdown()
        gather.close()
    failed = group.any_over_ranks(bool(errors))
    if errors:
        raise errors[0]
    if failed:
        raise RuntimeError('segmentation failed on another rank')
    return results
