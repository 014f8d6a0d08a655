"""Multi-GPU side of the path: one process per GPU, one column partition per rank, no collective in the data path.

The reference shards the scan the same way: rank r of `mpirun -n P gt_mpi_gather` reads column_partitions[r] of the loader
JSON, scans it on its own and writes its own output (src/main/cpp/src/config/json_config.cc:340-417,
tools/gt_mpi_gather.cc:322-366).  The only cross-rank steps are bookkeeping: the reduction of the benchmark counters
(max time, summed counts) and, for callers that want ONE stream, an ordered concatenation of the per-rank bodies.
Works over any torch.distributed backend ("nccl" = RCCL on the GPUs, "gloo" in the CPU tests)."""
import ctypes

from . import _lib


def column_partition(loader_json_text, rank):
    """(begin, end) of rank's column partition, by the loader-JSON rules of the reference"""
    b, e = ctypes.c_int64(), ctypes.c_int64()
    if isinstance(loader_json_text, str):
        loader_json_text = loader_json_text.encode()
    if _lib.lib().gdbamd_column_partition(loader_json_text, rank, ctypes.byref(b), ctypes.byref(e)) != 0:
        raise RuntimeError(_lib.lib().gdb_mi355_last_error().decode())
    return b.value, e.value


def synthetic_partition(rank, begin, length):
    """bench.py's weak-scaling layout: every rank owns its own `length` columns, back to back"""
    b = begin + rank * length
    return b, b + length - 1


def aggregate(seconds, counters, device=None):
    """(max over ranks of seconds, per-counter sum over ranks); identity when torch.distributed is not initialised"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds), [float(c) for c in counters]
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([float(x) for x in counters], dtype=torch.float64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in c.tolist()]


def ordered_concat(body, dst=0):
    """bodies of all ranks in rank (= column) order on rank `dst`, None elsewhere: the single-stream view of P partition
    outputs.  Host-side utility (bytes objects); the per-rank pages stay where they are."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return body
    parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(body, parts, dst=dst)
    return b"".join(parts) if parts is not None else None


def ordered_concat_tensors(local, dst=0):
    """Device-side flavour of ordered_concat: `local` is this rank's body as a 1-D uint8 tensor (in HBM under the "nccl"
    backend = RCCL over xGMI, on the host under "gloo").  On rank `dst` returns ONE tensor holding the bodies of all ranks in
    rank (= column partition) order, None on the other ranks.  Sizes travel by all_gather, the bytes point to point straight
    into their place in the result - no staging through host memory, no pickling."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(sizes, mine)
    sizes = [int(x.item()) for x in sizes]
    if rank != dst:
        if sizes[rank]:
            dist.send(local.contiguous(), dst=dst)
        return None
    out = torch.empty(sum(sizes), dtype=torch.uint8, device=local.device)
    off = 0
    pending = []
    for r in range(world):             # every body has its place (prefix sum of the sizes): all receives are posted at once and
        view = out[off:off + sizes[r]]  # run side by side - over xGMI each sender has its own link to the root
        if r == dst:
            view.copy_(local)
        elif sizes[r]:
            pending.append(dist.irecv(view, src=r))
        off += sizes[r]
    for w in pending:
        w.wait()
    return out


def gather_interval(engine, begin, end, arena_bytes=None, dst=0):
    """produce-combined-VCF over all ranks: every rank scans + combines its own column interval on its GPU, rank `dst` ends up
    with the VCF bodies of all partitions in column order as one uint8 tensor in its HBM.  A rank's body is assembled as ONE page
    (the arena grows to the size of the body), so what is sent is the page itself where it lies: no clone, no concatenation;
    the root copies only its own page into the result."""
    import torch
    empty = lambda: torch.empty(0, dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))
    if arena_bytes is None:             # one page = the whole body, sent where it lies
        pages = list(engine.page_tensors(begin, end, 1 << 62))
        local = pages[0] if pages else empty()
    else:                               # a caller-limited arena is reused page after page: the pages have to be copied out and joined
        pages = [t.clone() for t in engine.page_tensors(begin, end, arena_bytes)]
        local = torch.cat(pages) if pages else empty()
    return ordered_concat_tensors(local, dst=dst)
