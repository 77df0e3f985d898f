"""Index-range sharding of 1-D arrays across the GPUs of one node.

Partition: rank r of P owns elements [r*N/P, (r+1)*N/P) of EVERY size-N array, so all vertical ops, compares,
selects and casts are local.  Size-1 arrays and small gather tables (size K) are replicated.  The only exchange
steps of the hot path are
    * horizontal reductions: local block reduction, then an all-reduce of ONE element;
    * gradients of replicated tables: local scatter_add into a K-buffer, then an all-reduce of K elements -- or, when the
      consumer of the gradient works on the slice it owns (`gradient(table, scattered=True)`), a REDUCE-SCATTER: rank r
      receives bins [r K / P, (r + 1) K / P), half the bytes on the wire and no K-sized work replicated afterwards.
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


_TORCH_OPS = {"sum": dist.ReduceOp.SUM, "prod": dist.ReduceOp.PRODUCT, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}


def _to_tensor(value, device, dtype=None):
    """contribution -> flat torch tensor on `device` (zero-copy for device arrays and torch tensors)"""
    if isinstance(value, torch.Tensor):
        t = value
    elif hasattr(value, "__cuda_array_interface__") and torch.device(device).type == "cuda":
        t = torch.as_tensor(value, device=device)
    elif isinstance(value, (bool, int, float)):
        t = torch.tensor([value], dtype=dtype or (torch.int64 if isinstance(value, (bool, int)) else torch.float32))
    else:
        import numpy as np
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(value)))
    t = t.reshape(-1)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.device != torch.device(device):
        t = t.to(device)
    return t


class Handle:
    """A contribution to an Exchange: after flush() `tensor()` is the REDUCED value (a view of the staging buffer, valid
    until the buffer rotates back: `depth` flushes later)"""

    rider = False            # a 1-element sum that travels in an extra column of a reduce-scatter (Plan)
    scattered = False        # reduce-scattered: tensor() is the slice of the reduction that this rank owns
    owned = None             # (begin, end) of that slice in the full table

    def __init__(self):
        self._slot, self._work, self._host = None, None, None

    def tensor(self):
        if self._host is not None and self._slot is None:
            self._slot = _to_tensor(self._host[0], self._host[1], self._host[2])
        if self._slot is None:
            raise RuntimeError("Exchange.flush() has not been called for this contribution")
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._slot

    def item(self):
        if self._host is not None:
            return self._host[0]
        return self.tensor()[0].item()


class Exchange:
    """The exchange step of one backward() / one frame: every horizontal result of a sharded computation -- hsum / hprod /
    hmin / hmax partials, mask counts, gradients of replicated tables -- is registered with `add()`, and `flush()` issues
    ONE all-reduce per (dtype, reduction) group: for the usual "float32 loss + float32 table gradients" that is a single
    collective per step (xGMI is point-to-point: per-collective latency, not bytes, dominates at these sizes).  The
    collectives are asynchronous and the staging buffers rotate (`depth`), so the exchange of step i overlaps the kernels
    of step i + 1.  flush() returns a Plan; `plan.run()` repeats the same exchange on the same source buffers, which is
    what a replayed step graph (ek_hip_graph_*) needs: the sources keep their addresses, only their contents change."""

    def __init__(self, device="cpu", depth=2):
        self.device, self.depth = torch.device(device), depth
        self._pending = []                 # (handle, tensor, op)
        self._bufs = {}                    # (dtype, op, total) -> [buffers], cursor
        self._inflight = []
        self.collectives = 0

    def add(self, value, op="sum", dtype=None):
        if op not in _TORCH_OPS:
            raise ValueError(f"unknown reduction '{op}'")
        h = Handle()
        if isinstance(value, (bool, int, float)) and not (dist.is_initialized() and dist.get_world_size() > 1):
            # a host scalar with nobody to exchange it with: it is its own reduction (no device round trip)
            h._host = (value, self.device, dtype)
            return h
        self._pending.append((h, _to_tensor(value, self.device, dtype), op))
        return h

    def add_scattered(self, value, op="sum", dtype=None):
        """A K-element contribution whose REDUCED value is only needed where it is owned: after flush() the handle's
        tensor() is the slice [r * c, (r + 1) * c) of the reduction, c = ceil(K / P), on rank r (`Handle.owned` = its range
        in the table).  All scattered contributions of one (dtype, reduction) share ONE reduce-scatter, which moves half
        the bytes of the all-reduce they would otherwise ride on -- and nothing K-sized is replicated afterwards: what
        each rank does with the gradient (an optimiser step on its slice, a norm) scales with 1 / P."""
        if op not in _TORCH_OPS:
            raise ValueError(f"unknown reduction '{op}'")
        h = Handle()
        h.scattered = True
        self._pending.append((h, _to_tensor(value, self.device, dtype), op))
        return h

    def flush(self, async_op=True):
        plan = Plan(self, self._pending, async_op)
        self._pending = []
        plan.run()
        return plan

    def _buffer(self, key, total, dtype):
        entry = self._bufs.get((key, total))
        if entry is None:
            entry = self._bufs[(key, total)] = {"bufs": [torch.empty(total, device=self.device, dtype=dtype) for _ in range(self.depth)],
                                                "work": [None] * self.depth, "cur": -1}
        entry["cur"] = (entry["cur"] + 1) % self.depth
        i = entry["cur"]
        w = entry["work"][i]
        if w is not None:                         # the buffer's previous collective must have finished
            # usually it has (depth - 1 steps ago): ask first, and only make the compute stream wait when it has not --
            # a stream-wait in front of every step's graph launch keeps consecutive launches from overlapping
            done = False
            try:
                done = w.is_completed()
            except Exception:
                pass
            if not done:
                w.wait()
            entry["work"][i] = None
        return entry, i

    def wait_all(self):
        for entry in self._bufs.values():
            for i, w in enumerate(entry["work"]):
                if w is not None:
                    w.wait()
                    entry["work"][i] = None


def _world():
    return dist.get_world_size() if dist.is_initialized() else 1


def _rank():
    return dist.get_rank() if dist.is_initialized() else 0


class Plan:
    def __init__(self, exchange, items, async_op):
        self.ex, self.async_op = exchange, async_op
        self.groups, self.scattered = {}, {}
        for h, t, op in items:
            (self.scattered if h.scattered else self.groups).setdefault((t.dtype, op), []).append((h, t))
        # 1-element sums (the loss) RIDE on the reduce-scatter of their (dtype, "sum") group instead of paying a collective of
        # their own: the value goes into one extra column of EVERY row, so every rank receives sum_r(value_r) there
        for key, parts in self.scattered.items():
            if key[1] != "sum" or key not in self.groups:
                continue
            riders = [(h, t) for h, t in self.groups[key] if t.numel() == 1]
            if riders and len(parts) + len(riders) <= 8:
                for h, _ in riders:
                    h.rider = True
                parts.extend(riders)
                self.groups[key] = [(h, t) for h, t in self.groups[key] if t.numel() != 1]
                if not self.groups[key]:
                    del self.groups[key]

    def _run_scattered(self):
        """ONE reduce-scatter per (dtype, reduction): the staging buffer is laid out rank-major, row r = the chunks that
        rank r owns of every part ([P, sum of chunks]); padded where K is not a multiple of P"""
        ex, P, r = self.ex, _world(), _rank()
        for (dtype, op), parts in self.scattered.items():
            chunks = [1 if h.rider else -(-t.numel() // P) for h, t in parts]
            row = sum(chunks)
            entry, i = ex._buffer((dtype, op, "scatter"), P * row + row, dtype)
            flat = entry["bufs"][i]
            send, recv = flat[:P * row].view(P, row), flat[P * row:]
            fast = False
            if flat.is_cuda and dtype == torch.float32 and len(parts) <= 8 and \
                    all(t.is_cuda and t.is_contiguous() and (h.rider or t.numel() == P * c) for (h, t), c in zip(parts, chunks)):
                from enoki_amd import hip as _ek
                if _ek.hip_stream() == torch.cuda.current_stream().cuda_stream:
                    _ek.hip_concat_rows_f32(flat.data_ptr(), P, [(t.data_ptr(), t.numel()) for _, t in parts])   # one launch
                    fast = True
            offset = 0
            for (h, t), c in zip(parts, chunks):
                k = t.numel()
                if h.rider:
                    if not fast:
                        send[:, offset] = t[0]
                    h._slot = recv[offset:offset + 1]
                    offset += 1
                    continue
                if fast:
                    pass
                elif k == P * c:
                    send[:, offset:offset + c].copy_(t.view(P, c), non_blocking=True)
                else:                                    # ragged last chunk: zero padding (the identity of a sum)
                    if op != "sum":
                        raise ValueError("scattered reductions other than 'sum' need a table size that is a multiple of the world size")
                    send[:, offset:offset + c].zero_()
                    full = k // c
                    if full:
                        send[:full, offset:offset + c].copy_(t[:full * c].view(full, c), non_blocking=True)
                    if k > full * c:
                        send[full, offset:offset + k - full * c].copy_(t[full * c:], non_blocking=True)
                h._slot = recv[offset:offset + min(c, max(0, k - r * c))]
                h.owned = (min(r * c, k), min((r + 1) * c, k))
                offset += c
            work = None
            if dist.is_initialized() and P > 1 and dist.get_backend() != "gloo":
                work = dist.reduce_scatter_tensor(recv, flat[:P * row], op=_TORCH_OPS[op], async_op=self.async_op)
            elif dist.is_initialized() and P > 1:
                # gloo (the CPU tests) has no reduce-scatter: all-reduce the staging buffer, keep the owned row
                work = dist.all_reduce(flat[:P * row], op=_TORCH_OPS[op], async_op=False)
                recv.copy_(send[r])
                work = None
            else:
                recv.copy_(send[0], non_blocking=True)
            if dist.is_initialized():
                ex.collectives += 1
            if not self.async_op:
                if work is not None:
                    work.wait()
                work = None
            entry["work"][i] = work
            for h, _ in parts:
                h._work = work

    def run(self):
        ex = self.ex
        self._run_scattered()
        for (dtype, op), parts in self.groups.items():
            total = sum(t.numel() for _, t in parts)
            entry, i = ex._buffer((dtype, op), total, dtype)
            flat = entry["bufs"][i]
            fast = False
            if flat.is_cuda and dtype == torch.float32 and len(parts) <= 8 and all(t.is_cuda and t.is_contiguous() for _, t in parts):
                from enoki_amd import hip as _ek
                if _ek.hip_stream() == torch.cuda.current_stream().cuda_stream:
                    _ek.hip_concat_f32(flat.data_ptr(), [(t.data_ptr(), t.numel()) for _, t in parts])   # one launch
                    fast = True
            offset = 0
            for h, t in parts:
                n = t.numel()
                if not fast:
                    flat[offset:offset + n].copy_(t, non_blocking=True)
                h._slot = flat[offset:offset + n]
                offset += n
            work = None
            if dist.is_initialized():
                work = dist.all_reduce(flat, op=_TORCH_OPS[op], async_op=self.async_op)
                ex.collectives += 1
                if not self.async_op:
                    work = None
            entry["work"][i] = work
            for h, _ in parts:
                h._work = work


class Sharded:
    """Index-range sharding as a LIBRARY feature: rank r owns [begin, end) of every size-N array; tables and
    scalars are replicated.  Horizontal operations on shards go through this object, which computes the local part with
    the array module `ek` and registers it with the step's Exchange; `flush()` then finishes ALL of them with one
    all-reduce per (dtype, reduction).  Vertical operations need nothing: they are local by construction.

        sh = Sharded(ek, N)                        # rank / world from the process group (or 0 / 1)
        x = ek.Float32(...[sh.begin:sh.end])
        y = sh.hsum(f(x)); g = sh.gradient(table); hits = sh.count(mask)
        sh.flush()                                 # ONE collective for y and g, one for hits (int64)
        y.tensor(), g.tensor(), hits.item()
    """

    def __init__(self, ek, n_total, device=None, depth=2):
        self.ek = ek
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.n_total = n_total
        self.begin, self.end = shard_range(n_total, self.rank, self.world)
        self.n = self.end - self.begin
        if device is None:
            device = "cuda" if (torch.cuda.is_available() and (not dist.is_initialized() or dist.get_backend() == "nccl")) else "cpu"
        self.exchange = Exchange(device, depth)

    def _reduce(self, name, op, x):
        local = getattr(self.ek, name)(x)
        try:
            local = self.ek.detach(local)
        except Exception:
            pass
        return self.exchange.add(local, op)

    def hsum(self, x):
        return self._reduce("hsum", "sum", x)

    def hprod(self, x):
        return self._reduce("hprod", "prod", x)

    def hmax(self, x):
        return self._reduce("hmax", "max", x)

    def hmin(self, x):
        return self._reduce("hmin", "min", x)

    def count(self, mask):
        return self.exchange.add(int(self.ek.count(mask)), "sum", torch.int64)

    def any(self, mask):
        return _Predicate(self.exchange.add(int(self.ek.count(mask)), "sum", torch.int64), lambda c: c > 0)

    def all(self, mask, size=None):
        """`size`: global number of entries (default: the sharded length)"""
        return _Predicate(self.exchange.add(int(self.ek.count(mask)), "sum", torch.int64),
                          lambda c, n=(self.n_total if size is None else size): c == n)

    def gradient(self, table, scattered=False):
        """gradient of a REPLICATED table: the local scatter_add result, summed over the ranks.  scattered=True: every rank
        receives only the bins it owns ([r K / P, (r + 1) K / P), Handle.owned) through a reduce-scatter -- half the bytes
        of the all-reduce, and no K-sized work is replicated after the exchange; gather_scattered() rebuilds the full array
        where somebody needs it."""
        g = self.ek.gradient(table)
        return self.exchange.add_scattered(g, "sum") if scattered else self.exchange.add(g, "sum")

    def reduce(self, array, op="sum", scattered=False):
        """an already computed local partial (e.g. kept from a captured step graph)"""
        return self.exchange.add_scattered(array, op) if scattered else self.exchange.add(array, op)

    def gather_scattered(self, handle, size):
        """the full reduced table from its scattered slices (an all-gather; only for callers that need all of it)"""
        own = handle.tensor()
        if self.world == 1:
            return own[:size].clone()
        c = -(-size // self.world)
        padded = torch.zeros(c, device=own.device, dtype=own.dtype)
        padded[:own.numel()] = own
        out = torch.empty(self.world * c, device=own.device, dtype=own.dtype)
        dist.all_gather_into_tensor(out, padded) if dist.get_backend() != "gloo" else \
            dist.all_gather(list(out.view(self.world, c).unbind(0)), padded)
        return out[:size]

    def flush(self, async_op=True):
        return self.exchange.flush(async_op)

    def wait_all(self):
        self.exchange.wait_all()


class _Predicate:
    def __init__(self, handle, fn):
        self.handle, self.fn = handle, fn

    def item(self):
        return bool(self.fn(self.handle.item()))


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
