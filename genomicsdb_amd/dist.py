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


def paged_concat(pages, sink, page_bytes, dst=0, ring_slots=3, device=None):
    """Ordered concatenation of the ranks' page streams with BOUNDED memory on every rank: the single-stream view of the P
    per-partition outputs of `mpirun -n P gt_mpi_gather` (gt_mpi_gather.cc:322-366 writes P files; a combined stream is those
    files back to back in rank = column order).

    pages: this rank's pages, an iterable of 1-D uint8 tensors of at most `page_bytes` bytes each (HBM pages under "nccl" =
    RCCL over xGMI, host tensors under "gloo"); a page need only stay valid until the next one is asked for.
    sink(t): called on rank `dst` once per page, in rank order and in each rank's page order; `t` is valid during the call only.

    Rank `dst` never holds more than ring_slots pages of other ranks (a ring of receive buffers that is reused), a sending rank
    never more than ring_slots copies of its own pages: a rank keeps scanning while its earlier pages wait for the root, and
    stops when its ring is full.  Every page travels as an 8-byte header (its size; 0 closes the rank's stream) followed by
    the bytes, point to point; there is no collective and no tensor of the size of a whole body anywhere (the round-2
    gather_interval needed the sum of all bodies on the root).  Returns the number of bytes handed to `sink` (root) / sent."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        total = 0
        for t in pages:
            sink(t)
            total += int(t.numel())
        return total
    world, rank = dist.get_world_size(), dist.get_rank()
    ring_slots = max(2, int(ring_slots))
    if rank != dst:
        ring, hdrs, pending, total, k = [], [], [], 0, 0
        for t in pages:
            n = int(t.numel())
            if n == 0:
                continue
            if n > page_bytes:
                raise ValueError("page of %d bytes exceeds page_bytes %d" % (n, page_bytes))
            slot = k % ring_slots
            if len(ring) <= slot:
                ring.append(torch.empty(page_bytes, dtype=torch.uint8, device=t.device))
                hdrs.append(torch.zeros(1, dtype=torch.int64, device=t.device))
            elif len(pending) >= ring_slots:          # the slot's previous page must have left
                for w in pending.pop(0):
                    w.wait()
            ring[slot][:n].copy_(t)                   # the engine's arena is reused by the next page:
            if t.is_cuda:                             # the copy (torch's stream) must be over before the engine (its own
                torch.cuda.current_stream(t.device).synchronize()   # stream) is asked for the next one
            hdrs[slot].fill_(n)
            pending.append((dist.isend(hdrs[slot], dst=dst), dist.isend(ring[slot][:n], dst=dst)))
            total += n
            k += 1
        for ws in pending:
            for w in ws:
                w.wait()
        end = torch.zeros(1, dtype=torch.int64, device=device if device is not None else (ring[0].device if ring else None))
        dist.send(end, dst=dst)
        return total
    total = 0
    for t in pages:                                   # the root's own pages go straight to the sink where they lie
        if int(t.numel()):
            sink(t)
            total += int(t.numel())
    dev = device
    ring, inflight, k = [], [], 0                     # inflight: (work, view) in arrival order
    hdr = None
    for r in range(world):
        if r == dst:
            continue
        while True:
            if hdr is None:
                hdr = torch.zeros(1, dtype=torch.int64, device=dev)
            dist.recv(hdr, src=r)
            n = int(hdr.item())
            if n == 0:
                break
            if n > page_bytes:
                raise ValueError("rank %d announced a page of %d bytes, page_bytes is %d" % (r, n, page_bytes))
            if len(inflight) >= ring_slots - 1 and len(ring) >= ring_slots:   # free the oldest slot: its page goes to the sink now
                w, view = inflight.pop(0)
                w.wait()
                sink(view)
                total += int(view.numel())
            slot = k % ring_slots
            if len(ring) <= slot:
                ring.append(torch.empty(page_bytes, dtype=torch.uint8, device=dev))
            view = ring[slot][:n]
            inflight.append((dist.irecv(view, src=r), view))
            k += 1
    for w, view in inflight:
        w.wait()
        sink(view)
        total += int(view.numel())
    return total


def gather_interval_paged(engine, begin, end, sink, page_bytes=1 << 30, dst=0, ring_slots=3):
    """produce-combined-VCF over all ranks with bounded memory: every rank scans + combines its own column interval on its GPU
    page by page (pages of at most page_bytes, left in HBM), rank `dst`'s sink sees the pages of all partitions in column order
    (see paged_concat).  At the width of BASELINE configs[3] (100 000 samples: ~9 MB of text per record) a body does not fit any
    single tensor; this is the form of the concat that still works there."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    return paged_concat(engine.page_tensors(begin, end, page_bytes), sink, page_bytes, dst=dst, ring_slots=ring_slots, device=dev)
