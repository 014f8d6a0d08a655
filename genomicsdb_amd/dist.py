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


def equi_partition(counts, hist_begin, bin_size, num_parts):
    """ColumnHistogramOperator::equi_partition_and_print_bins (variant_operations.cc:769-796): [(first_column, last_column, cells)] of about
    equal cell count, and the text the reference prints for them.  counts: numpy uint64 (CombineEngine.column_histogram)"""
    import numpy as np
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    L = _lib.lib()
    L.gdbamd_equi_partition_text.restype = ctypes.c_int64
    L.gdbamd_equi_partition_text.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint64]
    n = L.gdbamd_equi_partition_text(counts.ctypes.data, counts.size, hist_begin, bin_size, num_parts, None, 0)
    if n < 0:
        raise ValueError("more partitions (%d) than histogram bins (%d)" % (num_parts, counts.size))
    buf = ctypes.create_string_buffer(n)
    L.gdbamd_equi_partition_text(counts.ctypes.data, counts.size, hist_begin, bin_size, num_parts, buf, n)
    text = buf.raw[:n].decode()
    parts = [tuple(int(x) for x in ln.split(",")) for ln in text.split("\n")[1:] if ln]
    return parts, text


def balanced_partition(counts, hist_begin, bin_size, rank, world):
    """(begin, end) of rank's column partition when the loader JSON lists none: the reference's own recipe - count the cells
    per column bin (gt_mpi_gather --produce-histogram) and cut where the running count passes total / world.  The greedy cut of
    equi_partition_and_print_bins may come out with fewer lines than ranks when the load is lumpy; the last ranks then share the last
    line's columns evenly."""
    parts, _ = equi_partition(counts, hist_begin, bin_size, world)
    if len(parts) >= world:
        parts = parts[:world - 1] + [(parts[world - 1][0], parts[-1][1], sum(p[2] for p in parts[world - 1:]))]
        b, e, _ = parts[rank]
        return b, e
    if rank < len(parts) - 1:
        return parts[rank][0], parts[rank][1]
    b, e, _ = parts[-1]
    k, m = rank - (len(parts) - 1), world - (len(parts) - 1)
    w = (e - b + 1) // m
    return b + k * w, (e if k == m - 1 else b + (k + 1) * w - 1)


_header_group = None


def _headers_group():
    """A CPU-side ("gloo") process group of all ranks for the 8-byte page headers of paged_concat: a page's size is known on the
    host, and a header that travels over the GPU backend costs the root a device-to-host synchronisation per page.  Created once
    (collectively: every rank calls paged_concat) and kept."""
    global _header_group
    import torch.distributed as dist
    if _header_group is None:
        _header_group = dist.new_group(backend="gloo")
    return _header_group


class _ThreadedWork:
    """is_completed() for a backend whose receives only make progress inside wait() (gloo): wait() runs in a helper thread.  Lets
    the polled root of paged_concat - the one "nccl" uses - run under gloo in the CPU tests."""

    def __init__(self, work):
        import threading
        self.error = None
        self._t = threading.Thread(target=self._run, args=(work,), daemon=True)
        self._t.start()

    def _run(self, work):
        try:
            work.wait()
        except BaseException as e:      # noqa: BLE001 - re-raised by is_completed()
            self.error = e

    def is_completed(self):
        if self._t.is_alive():
            return False
        if self.error is not None:
            raise self.error
        return True


def root_ring_slots(world, page_bytes, ring_slots, root_ring_bytes):
    """receive buffers per sender on the root: as many as the byte budget allows, at least 1, at most ring_slots"""
    senders = max(1, world - 1)
    return int(max(1, min(ring_slots, root_ring_bytes // (senders * max(1, page_bytes)))))


def paged_concat(pages, sink, page_bytes, dst=0, ring_slots=3, device=None, stats=None, root_ring_bytes=8 << 30, polled=None):
    """Ordered concatenation of the ranks' page streams with BOUNDED memory on every rank: the single-stream view of the P
    per-partition outputs of `mpirun -n P gt_mpi_gather` (gt_mpi_gather.cc:322-366 writes P files; a combined stream is those
    files back to back in rank = column order).

    pages: this rank's pages, an iterable of 1-D uint8 tensors of at most `page_bytes` bytes each (HBM pages under "nccl" =
    RCCL over xGMI, host tensors under "gloo"); a page need only stay valid until the next one is asked for.  BGZF pages (an
    engine created with output format "z" / "b") concatenate like any others - blocks are self-contained - and are ~6 x smaller.
    sink(t): called on rank `dst` once per page, in rank order and in each rank's page order; `t` is valid during the call only.

    A sending rank keeps ring_slots copies of its own pages in flight and stops scanning when they are full.  Rank `dst` receives
    from ALL ranks at once: every sender has its own ring of receive buffers there, so while rank r is being drained the ranks
    behind it can park pages at the root on top of their own ring instead of stalling after theirs.  The root's rings are sized
    from a byte budget (`root_ring_bytes`, default 8 GiB over all senders: 1 slot each for 8 ranks and 1 GiB pages, 3 each with
    256 MiB pages), allocated when first used.  A page travels as an 8-byte header (its size; 0 closes the rank's stream) over a
    CPU-side gloo group - the size is known on the host, no device synchronisation is paid for it - and the bytes point to point
    over the data backend; there is no collective and no tensor of the size of a whole body anywhere.

    Root side, two flavours.  polled (the default under "nccl"): ONE thread posts the receives of all senders as their headers
    arrive and slots are free, polls work.is_completed() and feeds the sink in rank order - ProcessGroupNCCL is never entered from
    two threads; only the header receives (gloo, host tensors) sit in helper threads.  threaded (the default under "gloo", whose
    receives only progress inside wait()): a receiver thread per sender.  A sink that raises does not strand the senders: the
    root keeps receiving and discarding until every rank has closed its stream, then re-raises.
    Returns the number of bytes handed to `sink` (root) / sent.  stats (a dict, optional) receives "bytes", "seconds", "blocked_s"
    (sender: waiting for a free slot; root: waiting for a page of the rank being drained)."""
    import os
    import time
    import torch
    import torch.distributed as dist
    t_begin = time.time()
    blocked = 0.0
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        total = 0
        for t in pages:
            sink(t)
            total += int(t.numel())
        if stats is not None:
            stats.update(bytes=total, seconds=time.time() - t_begin, blocked_s=0.0)
        return total
    world, rank = dist.get_world_size(), dist.get_rank()
    ring_slots = max(2, int(ring_slots))
    hgroup = _headers_group()
    if polled is None:
        env = os.environ.get("GDBAMD_CONCAT_POLLED")
        polled = (env == "1") if env is not None else dist.get_backend() != "gloo"
    if rank != dst:
        ring, pending, total, k = [], [], 0, 0
        for t in pages:
            n = int(t.numel())
            if n == 0:
                continue
            if n > page_bytes:
                raise ValueError("page of %d bytes exceeds page_bytes %d" % (n, page_bytes))
            slot = k % ring_slots
            if len(ring) <= slot:
                ring.append(torch.empty(page_bytes, dtype=torch.uint8, device=t.device))
            elif len(pending) >= ring_slots:          # the slot's previous page must have left
                t0 = time.time()
                for w in pending.pop(0)[:2]:
                    w.wait()
                blocked += time.time() - t0
            ring[slot][:n].copy_(t)                   # the engine's arena is reused by the next page:
            if t.is_cuda:                             # the copy (torch's stream) must be over before the engine (its own
                torch.cuda.current_stream(t.device).synchronize()   # stream) is asked for the next one
            h = torch.tensor([n], dtype=torch.int64)  # host tensor, gloo group: the root learns the size without touching the device
            pending.append((dist.isend(h, dst=dst, group=hgroup), dist.isend(ring[slot][:n], dst=dst), h))   # (h lives as long as its send: dropped with the slot)
            total += n
            k += 1
        for ws in pending:
            for w in ws[:2]:
                w.wait()
        dist.send(torch.zeros(1, dtype=torch.int64), dst=dst, group=hgroup)
        if stats is not None:
            stats.update(bytes=total, seconds=time.time() - t_begin, blocked_s=blocked)
        return total
    # ---- root ---------------------------------------------------------------------------------------------------------------
    import queue
    import threading
    dev = device
    total = 0
    slots = root_ring_slots(world, page_bytes, ring_slots, int(root_ring_bytes))
    failure = []                                      # first exception of the sink: from then on pages are received and dropped

    def deliver(t):
        nonlocal total
        if failure:
            return
        try:
            sink(t)
            total += int(t.numel())
        except BaseException as e:                    # noqa: BLE001 - re-raised when every sender has closed
            failure.append(e)

    def header_reader(src, out_q):                    # host tensors over gloo: the only receives that run in helper threads
        try:
            hdr = torch.zeros(1, dtype=torch.int64)
            while True:
                dist.recv(hdr, src=src, group=hgroup)
                n = int(hdr[0])
                out_q.put(n)
                if n == 0:
                    return
        except BaseException as e:                    # noqa: BLE001
            out_q.put(e)

    senders = [r for r in range(world) if r != dst]
    sizes = {r: queue.Queue() for r in senders}
    for r in senders:
        threading.Thread(target=header_reader, args=(r, sizes[r]), daemon=True).start()

    def next_size(r, block):
        try:
            n = sizes[r].get(block)
        except queue.Empty:
            return None
        if isinstance(n, BaseException):
            raise n
        if n > page_bytes and not failure:
            # a sender that breaks the contract must not strand the others: the page is received into a buffer of its own and dropped,
            # every stream is drained to its closing header, then the error is raised (like a failing sink)
            failure.append(ValueError("rank %d announced a page of %d bytes, page_bytes is %d" % (r, n, page_bytes)))
        return n

    if polled:
        # one thread: receives posted as headers arrive and slots are free, completion polled, the sink fed in rank order
        wrap = (lambda w: w) if dist.get_backend() != "gloo" else _ThreadedWork
        ring = {r: [] for r in senders}               # buffers, allocated when first used
        free = {r: list(range(slots)) for r in senders}
        inflight = {r: [] for r in senders}           # (work, view, slot) in posting order
        closed = {r: False for r in senders}          # the closing header has been read

        def post():
            posted = False
            for r in senders:
                while free[r] and not closed[r]:
                    n = next_size(r, False)
                    if n is None:
                        break
                    if n == 0:
                        closed[r] = True
                        break
                    slot = free[r].pop(0)
                    while len(ring[r]) <= slot:
                        ring[r].append(torch.empty(page_bytes, dtype=torch.uint8, device=dev))
                    view = ring[r][slot][:n] if n <= page_bytes else torch.empty(n, dtype=torch.uint8, device=dev)   # (oversize: dropped, see next_size)
                    inflight[r].append((wrap(dist.irecv(view, src=r)), view, slot))
                    posted = True
            return posted

        for r in range(world):                        # rank (= column) order, the root's own pages in their place
            if r == dst:
                for t in pages:
                    post()
                    if int(t.numel()):
                        deliver(t)
                continue
            while True:
                progressed = post()
                if inflight[r] and inflight[r][0][0].is_completed():
                    _, view, slot = inflight[r].pop(0)
                    deliver(view)
                    free[r].append(slot)
                    continue
                if closed[r] and not inflight[r]:
                    break
                if not progressed:
                    t0 = time.time()
                    time.sleep(50e-6)
                    blocked += time.time() - t0
    else:
        # a receiver thread per sender (gloo: a receive only progresses inside wait(), and a wait with a timeout closes the pair)
        def receiver(src, out_q, free_sem):
            try:
                ring, k = [], 0
                while True:
                    free_sem.acquire()                  # a buffer the sink has given back
                    n = next_size(src, True)
                    if n == 0:
                        out_q.put(None)
                        return
                    slot = k % slots
                    k += 1
                    if len(ring) <= slot:
                        ring.append(torch.empty(page_bytes, dtype=torch.uint8, device=dev))
                    view = ring[slot][:n] if n <= page_bytes else torch.empty(n, dtype=torch.uint8, device=dev)       # (oversize: dropped, see next_size)
                    dist.recv(view, src=src)
                    out_q.put(view)
            except BaseException as e:                  # noqa: BLE001 - handed to the thread that drains
                out_q.put(e)

        rx = {}
        for r in senders:
            qr, free_sem = queue.Queue(), threading.Semaphore(slots)
            th = threading.Thread(target=receiver, args=(r, qr, free_sem), daemon=True)
            th.start()
            rx[r] = (qr, free_sem, th)
        for r in range(world):
            if r == dst:
                for t in pages:
                    if int(t.numel()):
                        deliver(t)
                continue
            qr, free_sem, th = rx[r]
            while True:
                t0 = time.time()
                item = qr.get()
                blocked += time.time() - t0
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                deliver(item)
                free_sem.release()
            th.join()
    if stats is not None:
        stats.update(bytes=total, seconds=time.time() - t_begin, blocked_s=blocked, root_slots_per_sender=slots, polled=bool(polled))
    if failure:
        raise failure[0]
    return total


def gather_interval_paged(engine, begin, end, sink, page_bytes=1 << 30, dst=0, ring_slots=3, stats=None):
    """produce-combined-VCF over all ranks with bounded memory: every rank scans + combines its own column interval on its GPU
    page by page (pages of at most page_bytes, left in HBM), rank `dst`'s sink sees the pages of all partitions in column order
    (see paged_concat).  At the width of BASELINE configs[3] (100 000 samples: ~9 MB of text per record) a body does not fit any
    single tensor; this is the form of the concat that still works there."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    return paged_concat(engine.page_tensors(begin, end, page_bytes), sink, page_bytes, dst=dst, ring_slots=ring_slots, device=dev, stats=stats)
