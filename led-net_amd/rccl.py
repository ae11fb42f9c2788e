"""Direct RCCL communicator (ctypes over the librccl that torch itself loaded).

Why not ``torch.distributed.all_reduce``: the data-parallel step issues ~200 SyncBN all-reduces of a
``[2, C]`` f32 buffer (one per BatchNorm and direction, as ``torch.nn.SyncBatchNorm`` does in the
reference's DDP run) plus the gradient all-reduce.  Issued through ProcessGroupNCCL they cannot be
part of a hipGraph capture, so an N > 1 step falls back to eager launches (~11 us of dispatch gap per
kernel, 32 ms instead of 20 ms per step).  ``ncclAllReduce`` on the caller's HIP stream is a plain
stream operation: it is captured with the kernels around it and replayed by every rank in the same
order.  Collectives stay on ONE stream (the Trainer switches the two-branch streams off for N > 1).

The unique id is created on rank 0 and broadcast through the already initialised
``torch.distributed`` group; ``world == 1`` needs no process group (self-test on one GPU).
"""
import ctypes as C
import glob
import os

import torch

NCCL_FLOAT32, NCCL_SUM = 7, 0


class _UniqueId(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]


class RcclError(RuntimeError):
    pass


def _load():
    libdir = os.path.join(os.path.dirname(torch.__file__), 'lib')
    cands = glob.glob(os.path.join(libdir, 'librccl.so*')) or ['librccl.so']
    lib = C.CDLL(cands[0])
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclGetErrorString.argtypes = [C.c_int]
    lib.ncclGetErrorString.restype = C.c_char_p
    for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclCommDestroy):
        f.restype = C.c_int
    return lib


class Comm:
    """One communicator over all ranks of the job; in-place f32 sum all-reduce on the current stream."""

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
        probe = torch.ones(4, dtype=torch.float32, device=device)    # first collective: buffers, proxies
        self.all_reduce_(probe)
        if float(probe.sum().item()) != 4.0 * world:
            raise RcclError(f'RCCL self-check failed: {probe.tolist()} (world {world})')

    def _ok(self, rc, what):
        if rc != 0:
            raise RcclError(f'{what}: {self.lib.ncclGetErrorString(rc).decode()}')

    def all_reduce_(self, t):
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise RcclError('rccl.Comm.all_reduce_: contiguous float32 device tensor expected')
        stream = torch.cuda.current_stream(t.device).cuda_stream
        self._ok(self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), NCCL_FLOAT32, NCCL_SUM,
                                        self.comm, C.c_void_p(stream)), 'ncclAllReduce')
        return t

    def close(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
