"""N > 1 on CPU (gloo, world_size 2): the column-partition arithmetic, the reduction bench.py uses, and the ordered concat.
The device path itself needs a GPU; what is shared between ranks is only this bookkeeping - the reference's ranks do not
talk to each other during the scan either (gt_mpi_gather.cc:322-366)."""
import json
import os
import socket
import sys

import pytest

import helpers
from golden_cases import CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LOADER = {"column_partitions": [{"begin": 0, "workspace": "/tmp/ws", "array": "a0"}, {"begin": 12202, "workspace": "/tmp/ws", "array": "a1"}],
          "vid_mapping_file": "vid.json", "callset_mapping_file": "callsets.json"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from genomicsdb_amd import dist as gdist
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        begin, end = gdist.column_partition(json.dumps(LOADER), rank)
        sb, se = gdist.synthetic_partition(rank, 10_000_000, 1_000_000)
        # each rank "scans" its partition of the reference's overlapping-intervals fixture with the CPU oracle (checker only)
        case = [c for c in CASES if c[0] == "t0_overlapping_at_12202"][0]
        _, callsets, vid, ov, golden, mode = case
        cells = helpers.cells_for(callsets, vid)
        ov = dict(ov)
        ov["query_column_ranges"] = [[[begin, min(end, 1_000_000_000)]]]
        qj, _ = helpers.query_json(callsets, vid, ov, mode)
        body, nrec, _ = helpers.oracle_run(qj, cells, partition_begin=begin, with_header=False)
        dt, (recs,) = gdist.aggregate(1.0 + rank, [nrec])
        whole = gdist.ordered_concat(body)
        # the tensor flavour (the one that runs over RCCL with the pages in HBM): same bytes, point to point into place
        import torch
        t = gdist.ordered_concat_tensors(torch.frombuffer(bytearray(body), dtype=torch.uint8))
        assert (t is None) == (rank != 0)
        if t is not None:
            assert bytes(t.numpy().tobytes()) == whole
        e = gdist.ordered_concat_tensors(torch.empty(0, dtype=torch.uint8) if rank == 1 else torch.frombuffer(bytearray(b"x"), dtype=torch.uint8))
        assert e is None if rank else bytes(e.numpy().tobytes()) == b"x"
        q.put((rank, begin, end, sb, se, nrec, dt, recs, whole))
    finally:
        dist.destroy_process_group()


def test_two_ranks_partition_reduce_concat():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, b0, e0, sb0, se0, n0, dt0, recs0, whole0), (r1, b1, e1, sb1, se1, n1, dt1, recs1, whole1) = res
    assert (b0, e0) == (0, 12201) and b1 == 12202 and e1 >= 10**12          # ends derive from the next sorted begin
    assert (sb0, se0) == (10_000_000, 10_999_999) and (sb1, se1) == (11_000_000, 11_999_999)
    assert dt0 == dt1 == 2.0 and recs0 == recs1 == n0 + n1                  # max over ranks, sum over ranks
    assert whole1 is None and whole0 is not None
    # the second partition starts inside an interval: it is clipped to the partition and equals the reference's golden for
    # the query at 12202; the concat is the two bodies in column order
    golden = helpers.golden_text([c for c in CASES if c[0] == "t0_overlapping_at_12202"][0][4])
    body_golden = b"".join(l for l in golden.splitlines(True) if not l.startswith(b"#"))
    assert whole0.endswith(body_golden) and n1 == body_golden.count(b"\n")
    assert whole0.count(b"\n") == n0 + n1


def _import_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import genomicsdb_amd
    from genomicsdb_amd import dist as gdist
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        # every rank IMPORTS its own column partition from the gVCFs (like one rank of `mpirun -n 2 vcf2tiledb`): the cells that
        # begin in it plus, replayed first, the intervals that begin before it and reach into it
        begin, end = gdist.column_partition(json.dumps(LOADER), rank)
        v = os.path.join(helpers.GOLDEN, "inputs", "vid.json")
        c = os.path.join(helpers.GOLDEN, "inputs", "callsets", "t0_1_2.json")
        cells, ncells = genomicsdb_amd.import_cells(v, c, file_root=helpers.GOLDEN, column_begin=begin, column_end=min(end, 2**62))
        qj, _ = helpers.query_json("t0_1_2.json", "vid.json", {"query_column_ranges": [[[begin, min(end, 1_000_000_000)]]]}, "query")
        body, nrec, _ = helpers.oracle_run(qj, cells, partition_begin=begin, with_header=False)     # (the oracle is the checker of this CPU test)
        whole = gdist.ordered_concat(body)
        q.put((rank, begin, ncells, nrec, whole))
    finally:
        dist.destroy_process_group()


def test_two_ranks_import_their_partitions_and_concat():
    """partition-boundary replay end to end: rank 1's partition begins at 12202, inside reference blocks of all three samples;
    the concatenation of the two partition scans must be what ONE scan of the whole array gives for the two query intervals
    [0, 12201] and [12202, ...] - the interval that crosses the boundary is split there in both"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_import_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, b0, nc0, n0, whole0), (_, b1, nc1, n1, whole1) = res
    cells = helpers.cells_for("t0_1_2.json", "vid.json")
    qj, _ = helpers.query_json("t0_1_2.json", "vid.json", {"query_column_ranges": [[[0, 12201], [12202, 1_000_000_000]]]}, "query")
    want, nrec, _ = helpers.oracle_run(qj, cells, with_header=False)
    assert whole1 is None and whole0 == want and n0 + n1 == nrec
    import struct
    off, ncells_full = 0, 0
    while off < len(cells):
        off += struct.unpack_from("<Q", cells, off + 16)[0]
        ncells_full += 1
    assert nc0 + nc1 > ncells_full          # the replayed intervals exist in both partitions


def _paged_worker(rank, world, port, q, dst=0, polled=None, root_ring_bytes=8 << 30, fail_at=None, root_page_bytes=1000):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from genomicsdb_amd import dist as gdist
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        import random
        rnd = random.Random(77 + rank)
        # rank r "scans" its partition into pages of at most 1000 bytes: 0 pages on rank 1, a different number elsewhere;
        # one reused buffer stands for the engine's arena (a page is only valid until the next one is asked for)
        npages = 0 if rank == 1 else 5 + 4 * rank
        arena = torch.empty(1000, dtype=torch.uint8)
        mine = []

        def pages():
            for k in range(npages):
                n = rnd.randint(1, 1000)
                arena[:n] = torch.tensor([(rank * 50 + k + i) % 251 for i in range(n)], dtype=torch.uint8)
                mine.append(bytes(arena[:n].numpy().tobytes()))
                yield arena[:n]
        got = []
        live = {"max": 0}
        stats = {}
        def sink(t):
            if fail_at is not None and len(got) == fail_at:
                raise RuntimeError("sink failed at page %d" % fail_at)
            got.append(bytes(t.numpy().tobytes()))
        try:
            total = gdist.paged_concat(pages(), sink, page_bytes=root_page_bytes if rank == dst else 1000, dst=dst, ring_slots=3, stats=stats, polled=polled,
                                       root_ring_bytes=root_ring_bytes)
        except ValueError as e:
            assert rank == dst and root_page_bytes < 1000 and "announced a page" in str(e)
            q.put((rank, mine, got, -2))
            return
        except RuntimeError as e:
            assert rank == dst and fail_at is not None and "sink failed" in str(e)
            q.put((rank, mine, got, -1))
            return
        assert total == sum(len(m) for m in mine) if rank != dst else True
        assert stats["bytes"] == total and stats["seconds"] >= stats["blocked_s"] >= 0.0
        if rank == dst:
            assert stats["polled"] == bool(polled)
            assert stats["root_slots_per_sender"] == gdist.root_ring_slots(world, 1000, 3, root_ring_bytes)
        q.put((rank, mine, got, total))
    finally:
        dist.destroy_process_group()


def test_paged_concat_is_ordered_and_bounded():
    """three ranks, pages of different counts (one rank has none): the root's sink sees every page of every rank, in rank and
    page order, through a ring of three receive buffers (no tensor of the size of a body exists anywhere)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_paged_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = [pg for _, mine, _, _ in res for pg in mine]
    assert res[0][2] == want and len(want) == 5 + 13
    assert res[0][3] == sum(len(p) for p in want)
    assert res[1][2] == [] and res[2][2] == []


def _run_paged(world, **kw):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    dst = kw.pop("dst", 0)
    procs = [ctx.Process(target=_paged_worker, args=(r, world, port, q, dst), kwargs=kw) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("dst", [0, 1])
def test_paged_concat_polled_root_is_single_threaded_on_the_data_backend(dst):
    """The root flavour "nccl" uses: ONE thread posts the receives of all senders as their headers (a gloo side group) arrive,
    polls work.is_completed() and feeds the sink in rank order.  Under gloo a receive only completes inside wait(), so the work
    objects are wrapped (dist._ThreadedWork) - the scheduling code is the one that runs over RCCL."""
    res = _run_paged(3, dst=dst, polled=True)
    want = [pg for _, mine, _, _ in res for pg in mine]
    assert res[dst][2] == want and len(want) == 5 + 13
    assert res[dst][3] == sum(len(p) for p in want)
    assert all(res[r][2] == [] for r in range(3) if r != dst)


@pytest.mark.parametrize("polled", [False, True])
def test_paged_concat_root_rings_follow_the_byte_budget(polled):
    """root_ring_bytes = 2 500 with two senders and 1 000-byte pages: one receive buffer per sender instead of ring_slots = 3 - the
    stream is the same (the worker checks stats["root_slots_per_sender"] against dist.root_ring_slots)"""
    from genomicsdb_amd import dist as gdist
    assert gdist.root_ring_slots(3, 1000, 3, 2500) == 1 and gdist.root_ring_slots(3, 1000, 3, 4000) == 2
    assert gdist.root_ring_slots(8, 1 << 30, 3, 8 << 30) == 1 and gdist.root_ring_slots(8, 256 << 20, 3, 8 << 30) == 3
    assert gdist.root_ring_slots(8, 1 << 30, 3, 0) == 1          # never below one buffer per sender
    res = _run_paged(3, polled=polled, root_ring_bytes=2500)
    want = [pg for _, mine, _, _ in res for pg in mine]
    assert res[0][2] == want


@pytest.mark.parametrize("polled", [False, True])
def test_paged_concat_a_failing_sink_does_not_strand_the_senders(polled):
    """the sink raises at its 8th page: the root keeps receiving (and dropping) until every rank has closed its stream, then
    re-raises; the senders finish normally instead of hanging in wait()"""
    res = _run_paged(3, polled=polled, fail_at=7)
    assert res[0][3] == -1 and len(res[0][2]) == 7
    assert res[1][3] == 0 and res[2][3] == sum(len(p) for p in res[2][1]) > 0


@pytest.mark.parametrize("polled", [False, True])
def test_paged_concat_an_oversize_page_header_does_not_strand_the_senders(polled):
    """the root was given a smaller page_bytes (400) than the senders use (1 000): the first header above it is an error of the
    ROOT's call, but the root keeps receiving - the oversize pages into buffers of their own - and dropping until every rank has closed
    its stream; the senders finish normally, then the root raises"""
    res = _run_paged(3, polled=polled, root_page_bytes=400)
    assert res[0][3] == -2
    assert res[1][3] == 0 and res[2][3] == sum(len(p) for p in res[2][1]) > 0


def test_paged_concat_to_another_root_keeps_rank_order():
    """dst = 2 of 3: the stream is still rank 0, rank 1, rank 2 (the root's own pages in their place, not first)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_paged_worker, args=(r, 3, port, q, 2)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = [pg for _, mine, _, _ in res for pg in mine]
    assert res[2][2] == want and res[0][2] == [] and res[1][2] == []
    assert res[2][3] == sum(len(p) for p in want)


def test_equi_partition_prints_what_the_reference_prints():
    """ColumnHistogramOperator::equi_partition_and_print_bins (variant_operations.cc:769-796) on a histogram small enough to do by hand:
    bins of 100 columns from 1000, counts 5 1 1 1 8 0 0 4 -> total 20; 2 parts of 10.0: bins 0-4 (16 >= 10 after the 5th), then 5-7 (4);
    4 parts of 5.0: [0], [1-4] (11), [5-7] (4)"""
    import numpy as np
    from genomicsdb_amd import dist as gdist
    counts = np.array([5, 1, 1, 1, 8, 0, 0, 4], dtype=np.uint64)
    parts, text = gdist.equi_partition(counts, 1000, 100, 2)
    assert text == "Total 20 #bins 2 count/bins 10.0\n1000,1499,16\n1500,1799,4\n\n"
    assert parts == [(1000, 1499, 16), (1500, 1799, 4)]
    parts, text = gdist.equi_partition(counts, 1000, 100, 4)
    assert text == "Total 20 #bins 4 count/bins 5.0\n1000,1099,5\n1100,1499,11\n1500,1799,4\n\n"
    with pytest.raises(ValueError):
        gdist.equi_partition(counts, 1000, 100, 8)
    # every rank gets a partition, the partitions tile the histogram's columns in order
    for world in (2, 3, 4, 5):
        spans = [gdist.balanced_partition(counts, 1000, 100, r, world) for r in range(world)]
        assert spans[0][0] == 1000 and spans[-1][1] == 1799
        assert all(spans[i][1] + 1 == spans[i + 1][0] for i in range(world - 1)), spans


def test_bench_concat_leg_accounts_for_every_byte():
    """`bench.py --gpus 3 --concat --dry-run`: the ranks' (synthetic) pages go through dist.paged_concat to rank 0; the line carries
    the bytes, the rate and how long ranks were blocked"""
    import subprocess
    env = dict(os.environ, GDBAMD_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-run", "--concat", "--interval-bp", "1000"],
                       capture_output=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][0])
    c = out["concat"]
    assert c["ranks"] == 3 and c["bytes"] == sum((r + 1) * 8 * (1 << 20) for r in range(3)) and c["pages"] == 8 * 3
    assert c["GBps"] > 0 and c["root_blocked_s"] >= 0 and c["max_sender_blocked_s"] >= 0 and c["ordered"] is True


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it becomes two ranks (torch.distributed.run, 127.0.0.1) and rank 0
    reports n_gpus = 2; --dry-run keeps the device out of it (no GPU in the CPU suite), the backend is gloo"""
    import subprocess
    env = dict(os.environ, GDBAMD_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--interval-bp", "1000"],
                       capture_output=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_reporting"] == 2 and out["dry_run"] is True and out["value"] is None
    assert out["columns_all_ranks"] == 2000 and abs(out["max_over_ranks_s"] - 0.002) < 1e-9
