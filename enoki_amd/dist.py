"""Index-range sharding of 1-D arrays across the GPUs of one node.

Partition: rank r of P owns elements [r*N/P, (r+1)*N/P) of EVERY size-N array, so all vertical ops, compares,
selects and casts are local.  Size-1 arrays and small gather tables (size K) are replicated.  The only exchange
steps of the hot path are
    * horizontal reductions: local block reduction, then an all-reduce of ONE element;
    * gradients of replicated tables: local scatter_add into a K-buffer, then an all-reduce of K elements.
Both run as RCCL all-reduces over xGMI through torch.distributed (backend "nccl" IS RCCL on ROCm); `gloo` is
used by the CPU tests.  To keep the number of collectives per backward() at ONE, callers pack every pending
reduction into a flat buffer with `Packer`.

torch is plumbing only: tensors are zero-copy views of the device arrays (``__cuda_array_interface__``), and the
library stream is switched to torch's current stream so that kernels and collectives are ordered by the stream.
"""
import os

import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """[begin, end) of the index range owned by `rank`"""
    return (rank * n) // world, ((rank + 1) * n) // world


def env_rank_world():
    """(rank, local device index, world size).  The device index is LOCAL_RANK folded into the devices this process can
    see, so a launcher that narrows the visibility per rank (one visible GPU each) works like one that does not."""
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available() and torch.cuda.device_count() > 0:
        local %= torch.cuda.device_count()
    return int(os.environ.get("RANK", "0")), local, int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise the process group (no-op for a single process).  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_rank_world()
    # under a launcher (RANK set) the group is created even for one rank, so that a 1-GPU torchrun run
    # exercises exactly the code path of the multi-GPU runs
    if (world > 1 or "RANK" in os.environ) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # ENOKI_DIST_BACKEND=gloo: host-staged collectives, e.g. to run several ranks on ONE GPU (RCCL refuses that)
            backend = os.environ.get("ENOKI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


_own_stream = None


def adopt_torch_stream(ek_module):
    """Run the library's kernels on torch's current stream so that collectives and kernels are stream ordered.  When
    that is the (legacy) default stream, a dedicated stream is created and made current first: the default stream
    cannot be captured into a step graph (ek_hip_graph_*), and its implicit synchronisation with every other stream is
    not wanted on the hot path either."""
    global _own_stream
    if torch.cuda.current_stream() == torch.cuda.default_stream():
        torch.cuda.synchronize()
        _own_stream = torch.cuda.Stream()
        torch.cuda.set_stream(_own_stream)
    ek_module.hip_set_stream(torch.cuda.current_stream().cuda_stream)


def as_tensor(array):
    """zero-copy torch view of a device array (HIPArray / DiffArray python object)"""
    return torch.as_tensor(array, device="cuda")


def all_reduce_(tensor, op="sum"):
    """in-place all-reduce; identity for a single process"""
    if dist.is_initialized():
        rop = {"sum": dist.ReduceOp.SUM, "prod": dist.ReduceOp.PRODUCT, "max": dist.ReduceOp.MAX,
               "min": dist.ReduceOp.MIN}[op]
        dist.all_reduce(tensor, op=rop)
    return tensor


class Packer:
    """Flat staging buffers: several pending sum-reductions (scalars and K-vectors) -> ONE all-reduce.

    xGMI is point-to-point, so a ring all-reduce pays per-link latency per collective: batching the
    1-element hsum results with the K-element gradient buffers keeps backward() at a single collective.
    The collective is issued asynchronously (RCCL's own stream, ordered after the compute stream at issue
    time) and `depth` staging buffers rotate, so the all-reduce of step i overlaps the kernels of step i+1;
    a buffer is only waited for when it comes up for reuse (or in `wait_all`)."""

    def __init__(self, sizes, device, dtype=torch.float32, depth=2):
        self.sizes = list(sizes)
        self.offsets = [0]
        for s in self.sizes:
            self.offsets.append(self.offsets[-1] + s)
        self.bufs = [torch.empty(self.offsets[-1], device=device, dtype=dtype) for _ in range(depth)]
        self.work = [None] * depth
        self.cur = 0
        self.flat = self.bufs[0]

    def slot(self, i, buf=None):
        flat = self.flat if buf is None else buf
        return flat[self.offsets[i]:self.offsets[i + 1]]

    def pack(self, tensors):
        self.cur = (self.cur + 1) % len(self.bufs)
        if self.work[self.cur] is not None:        # the buffer's previous collective must have finished
            self.work[self.cur].wait()
            self.work[self.cur] = None
        self.flat = self.bufs[self.cur]
        from enoki_amd import hip as _ek
        if self.flat.is_cuda and self.flat.dtype == torch.float32 and len(tensors) <= 8 and \
                _ek.hip_stream() == torch.cuda.current_stream().cuda_stream and \
                all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in tensors):
            # one launch for all parts (ek_hip_concat) instead of one copy per part -- only when the library runs on
            # torch's current stream (adopt_torch_stream): otherwise the launch would race with the producers of
            # `tensors` and with the collective, and the per-part copies below (torch's stream) are used
            assert [t.numel() for t in tensors] == self.sizes
            _ek.hip_concat_f32(self.flat.data_ptr(), [(t.data_ptr(), t.numel()) for t in tensors])
            return
        for i, t in enumerate(tensors):
            self.slot(i).copy_(t.reshape(-1), non_blocking=True)

    def all_reduce(self, async_op=True):
        """start the all-reduce of the buffer filled by the last pack(); returns its slots (valid after wait)"""
        if dist.is_initialized():
            w = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
            self.work[self.cur] = w if async_op else None
        return [self.slot(i) for i in range(len(self.sizes))]

    def wait_all(self):
        for i, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[i] = None


def active():
    """True when collectives are live (a process group exists)"""
    return dist.is_initialized()


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value):
    """max of a python float over all ranks (CPU side channel: works with nccl and gloo)"""
    if not dist.is_initialized():
        return value
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
