"""Data-parallel plumbing: one process per GPU, gradients summed by RCCL over xGMI.

The reference's only multi-GPU mechanism is single-process ``nn.DataParallel``
(models/model_util.py:36-37, 283-284).  Here every rank owns one MI355X and a full replica; after a
backward pass the optimizer all-reduces its *flat* gradient buffer with one collective
(``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests) and the 1/world
scale is folded into the SGD kernel.  BatchNorm uses per-rank batch statistics, as DataParallel's
replicas do.
"""
import os

import torch
import torch.distributed as dist


# MCDSEG_DIST_FORCE=1: join a process group even as the only rank and run every collective through it -- the RCCL path of a
# one-GPU box (bench.py under torch.distributed.run with one rank: tests/test_trainers_gpu.py)
FORCE = os.environ.get("MCDSEG_DIST_FORCE", "0") == "1"


def is_distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE)


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract).
    Returns (rank, world, local_rank); a no-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1 and not (FORCE and "MASTER_PORT" in os.environ):
        return 0, 1, local
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = os.environ.get("MCDSEG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if os.environ.get("MCDSEG_SINGLE_DEVICE") == "1":
        local = 0  # test rigs with fewer GPUs than ranks (gloo only: RCCL refuses two ranks on one device)
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size(), local


# bench.py --gpus N: a list to which every collective on a GPU tensor appends (bytes, start event, end event) -- HIP events on the
# caller's stream around the blocking all-reduce, or around the wait for an asynchronous one (the time the compute stream
# stands still for the exchange; what overlapped with the backward pass does not show).  None: nothing is recorded.
COLLECTIVE_EVENTS = None


class _timed_collective:
    def __init__(self, flat):
        self.on = COLLECTIVE_EVENTS is not None and flat.is_cuda
        self.nbytes = flat.numel() * flat.element_size()

    def __enter__(self):
        if self.on:
            self.t0, self.t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.t0.record()

    def __exit__(self, *exc):
        if self.on:
            self.t1.record()
            COLLECTIVE_EVENTS.append((self.nbytes, self.t0, self.t1))


# MCDSEG_NATIVE_RCCL=1: the gradient exchange through the library's own communicator (include/mcdseg.h: mcdseg_comm_init /
# mcdseg_allreduce, csrc/comm.hip) instead of torch.distributed's -- RCCL called from the C ABI on the stream the caller names, no
# ProcessGroup stream in between.  The process group stays for what it is good at: the rendezvous (the communicator's id travels over
# it), barriers, the scalar exchanges.  Off by default: a one-GPU box can prove the path with one rank only
# (tests/test_trainers_gpu.py::test_native_rccl_allreduce_one_rank), RCCL refusing two ranks on one device.
NATIVE_RCCL = os.environ.get("MCDSEG_NATIVE_RCCL", "0") == "1"
_NATIVE = {"comm": None, "stream": None}


def native_comm():
    """the library's communicator over all ranks of the process group (created on first use: collective), or None"""
    if not (NATIVE_RCCL and is_distributed() and torch.cuda.is_available()):
        return None
    if _NATIVE["comm"] is None:
        import ctypes
        from ._lib import check, lib
        L = lib()
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank() == 0:
            buf = (ctypes.c_ubyte * 128)()
            check(L.mcdseg_comm_unique_id(buf), "comm_unique_id")
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        if dist.get_world_size() > 1:  # the id travels over the process group the ranks already share
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            t = ident.to(dev)
            dist.broadcast(t, src=0)
            ident = t.cpu()
        raw = (ctypes.c_ubyte * 128)(*ident.tolist())
        comm = ctypes.c_void_p()
        check(L.mcdseg_comm_init(ctypes.byref(comm), dist.get_world_size(), raw, dist.get_rank()), "comm_init")
        _NATIVE["comm"] = comm
    return _NATIVE["comm"]


def _native_all_reduce(flat, stream):
    from ._lib import check, lib
    assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
    check(lib().mcdseg_allreduce(flat.data_ptr(), flat.numel(), native_comm(), stream.cuda_stream), "allreduce")


def all_reduce_sum_(flat):
    """In-place sum of a flat buffer over all ranks (no-op when not distributed)."""
    if is_distributed():
        with _timed_collective(flat):
            if flat.is_cuda and flat.dtype == torch.float32 and native_comm() is not None:
                _native_all_reduce(flat, torch.cuda.current_stream(flat.device))  # stream-ordered on the caller's stream
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


class _AsyncSum:
    """work handle of an asynchronous all-reduce; ``wait()`` orders the caller's stream behind it"""

    def __init__(self, flat):
        self.flat = flat
        self.waited = False
        self.done = None
        if flat.is_cuda and flat.dtype == torch.float32 and native_comm() is not None:
            # the library's communicator: the exchange runs on a stream of its own behind the caller's work so far; ``wait`` orders the
            # caller's stream behind it (an event, no host synchronisation)
            cur = torch.cuda.current_stream(flat.device)
            if _NATIVE["stream"] is None:
                _NATIVE["stream"] = torch.cuda.Stream(flat.device)
            ready = torch.cuda.Event()
            ready.record(cur)
            _NATIVE["stream"].wait_event(ready)
            _native_all_reduce(flat, _NATIVE["stream"])
            self.done = torch.cuda.Event()
            self.done.record(_NATIVE["stream"])
            self.work = None
        else:
            self.work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def _order(self):
        if self.work is not None:
            self.work.wait()
        else:
            torch.cuda.current_stream(self.flat.device).wait_event(self.done)

    def wait(self):
        if self.waited:  # (FlatSGD waits again when it resets its buckets: that orders the stream it is called on, but it is the same
            self._order()  # exchange -- counted once in bench.py's ``collectives``)
            return
        with _timed_collective(self.flat):
            self._order()
        self.waited = True


def all_reduce_sum_async(flat):
    """Start the in-place sum and return a handle (None when not distributed); ``handle.wait()`` orders the caller's stream behind it"""
    if is_distributed():
        return _AsyncSum(flat)
    return None


def broadcast_(tensors, src=0):
    if is_distributed():
        for t in tensors:
            dist.broadcast(t, src=src)


def barrier():
    if is_distributed():
        dist.barrier()
