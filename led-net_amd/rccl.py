"""Direct RCCL communicators (ctypes over the librccl that torch itself loaded).

Why not ``torch.distributed.all_reduce``: the data-parallel step issues ~170 SyncBN all-reduces of
``[2, C]`` f32 buffers (per BatchNorm -- or group of BatchNorms whose statistics are ready together --
and direction, as ``torch.nn.SyncBatchNorm`` does in the reference's DDP run) plus the gradient
all-reduce.  ``ncclAllReduce`` on the caller's HIP stream is a plain stream operation: it is captured
into the step's hipGraph with the kernels around it and replayed by every rank in the same order.

DEFAULT for N > 1 (train.Trainer): ONE launch stream, so one ordered sequence of collectives per rank
(communicator 0 for the SyncBN statistics, 'grad' for the gradient exchange issued after the backward behind a stream
wait -- never concurrent).  Opt-in (LEDN_MULTI_COMM=1), validated on one rank only:
one communicator PER LAUNCH STREAM (``CommSet``): the step runs on several HIP streams (main, the
context-branch stream, the gradient-exchange stream).  Operations of one communicator must be issued
in the same order on every rank and are serialised by RCCL; a communicator per stream keeps the
streams independent -- each stream's collectives are issued in (deterministic) program order, and the
all-reduce kernels of different communicators are small enough to be co-resident, so no rank can
block another.  xGMI is point-to-point: the SyncBN messages are 0.25-2 KB (pure latency), the
gradient buckets 0.7 / 5.4 MB (one-shot over the 7 links), nothing here is bandwidth-bound.

The unique ids are created on rank 0 and broadcast through the already initialised
``torch.distributed`` group; ``world == 1`` needs no process group (self-test on one GPU).
"""
import contextlib
import ctypes as C
import glob
import os

import torch

NCCL_FLOAT32, NCCL_SUM = 7, 0


class _UniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]


class RcclError(RuntimeError):
    pass


_LIB = None


def _load():
    global _LIB
    if _LIB is not None:
        return _LIB
    libdir = os.path.join(os.path.dirname(torch.__file__), 'lib')
    cands = glob.glob(os.path.join(libdir, 'librccl.so*')) or ['librccl.so']
    lib = C.CDLL(cands[0])
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.ncclGetErrorString.argtypes = [C.c_int]
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGroupStart.argtypes = []
    lib.ncclGroupEnd.argtypes = []
    for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclCommDestroy, lib.ncclCommCount,
              lib.ncclGroupStart, lib.ncclGroupEnd):
        f.restype = C.c_int
    _LIB = lib
    return lib


class Comm:
    """One communicator over all ranks of the job; f32 sum all-reduce on the current stream."""

    def __init__(self, rank, world, device):
        self.lib = _load()
        self.rank, self.world, self.device = rank, world, device
        uid = _UniqueId()
        if rank == 0:
            self._ok(self.lib.ncclGetUniqueId(C.byref(uid)), 'ncclGetUniqueId')
        if world > 1:
            import torch.distributed as dist
            t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(device)
            dist.broadcast(t, 0)
            C.memmove(C.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
        self.comm = C.c_void_p()
        with torch.cuda.device(device):
            self._ok(self.lib.ncclCommInitRank(C.byref(self.comm), world, uid, rank), 'ncclCommInitRank')
        n = C.c_int(0)
        self._ok(self.lib.ncclCommCount(self.comm, C.byref(n)), 'ncclCommCount')
        self.nranks = n.value
        probe = torch.ones(4, dtype=torch.float32, device=device)    # first collective: buffers, proxies
        self.all_reduce(probe, probe)
        if float(probe.sum().item()) != 4.0 * world or self.nranks != world:
            raise RcclError(f'RCCL self-check failed: {probe.tolist()} (world {world}, nranks {self.nranks})')

    def _ok(self, rc, what):
        if rc != 0:
            raise RcclError(f'{what}: {self.lib.ncclGetErrorString(rc).decode()}')

    def all_reduce(self, src, dst):
        """dst = sum over ranks of src (dst may be src), on the current stream of the tensors' device."""
        for t in (src, dst):
            if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                raise RcclError('rccl.Comm.all_reduce: contiguous float32 device tensors expected')
        if src.numel() != dst.numel():
            raise RcclError('rccl.Comm.all_reduce: size mismatch')
        stream = torch.cuda.current_stream(src.device).cuda_stream
        self._ok(self.lib.ncclAllReduce(src.data_ptr(), dst.data_ptr(), src.numel(), NCCL_FLOAT32, NCCL_SUM,
                                        self.comm, C.c_void_p(stream)), 'ncclAllReduce')
        return dst

    def all_reduce_(self, t):
        return self.all_reduce(t, t)

    @contextlib.contextmanager
    def group(self):
        """ncclGroupStart/End: the all-reduces issued inside are aggregated into ONE launch."""
        self._ok(self.lib.ncclGroupStart(), 'ncclGroupStart')
        try:
            yield self
        finally:
            self._ok(self.lib.ncclGroupEnd(), 'ncclGroupEnd')

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()


class CommSet:
    """Communicators keyed by launch-stream slot (ops._slot: 0 = main, k = auxiliary stream k) plus one
    for the gradient exchange ('grad').  All are created up front, in the same order on every rank."""

    def __init__(self, rank, world, device, slots=(0, 1, 'grad')):
        self.rank, self.world, self.device = rank, world, device
        self.comms = {s: Comm(rank, world, device) for s in slots}
        self.nranks = self.comms[slots[0]].nranks
        # what RCCL itself reports (ncclCommCount) for EVERY communicator of the step, not only slot 0
        self.nranks_all = {str(s): c.nranks for s, c in self.comms.items()}

    def get(self, slot):
        c = self.comms.get(slot)
        if c is None:
            raise RcclError(f'no communicator for launch stream slot {slot!r}: collectives on this stream were '
                            f'not planned (slots: {list(self.comms)})')
        return c

    def close(self):
        for c in self.comms.values():
            c.close()
