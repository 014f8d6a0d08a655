"""GPU parity tests of the whole-genome shape (BASELINE.json configs[3]): columns are a flattened genome of several contigs
(VidMapper::get_contig_location / get_next_contig_location, vid_mapper.cc:240-304; switch_contig, broad_combined_gvcf.cc:
903-909), query intervals and column partitions cross contig boundaries or begin exactly at a contig's offset, CHROM / POS /
END are contig-relative, and the partitions' outputs are concatenated page by page in column order."""
import json
import os
import subprocess
import sys

import pytest

import helpers

pytestmark = pytest.mark.gpu

GENOME = [("1", 0, 6000), ("2", 6000, 900), ("3", 6900, 5000), ("X", 11900, 2500), ("Y", 14400, 1700), ("MT", 16100, 400)]
END = 16500


@pytest.fixture(scope="module")
def gdb():
    import genomicsdb_amd
    from genomicsdb_amd import _lib
    assert _lib.lib().gdb_mi355_device_count() > 0, "no HIP device"
    return genomicsdb_amd


def _chrom_runs(body):
    out = []
    for l in body.split(b"\n"):
        c = l.split(b"\t", 1)[0]
        if l and (not out or out[-1] != c):
            out.append(c)
    return out


@pytest.mark.parametrize("n_samples,opts", [(180, {}), (64, {"produce_GT_field": True}), (1100, {"sites_only_query": True})])
def test_device_matches_oracle_across_contig_boundaries(gdb, tmp_path, n_samples, opts):
    from genomicsdb_amd import synth
    g = synth.Generator(n_samples, 0, END, contigs=GENOME)
    cells, _ = g.chunk_bytes(END)
    qb, qe = 2000, END - 200
    q = helpers.synth_query(tmp_path, n_samples, qb, qe, contigs=GENOME)
    q.update(opts)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    assert _chrom_runs(want) == [b"1", b"2", b"3", b"X", b"Y", b"MT"]
    e = gdb.CombineEngine(q)
    e.stage_cells(cells)
    e.set_reference(0, synth.reference(0, END + 16))
    body, st = e.run_interval(qb, qe, arena_bytes=1 << 20)
    assert st.num_records == nrec
    assert body == want
    # the header names every contig of the vid mapping that the template lacks, in vid order
    for name, _, ln in GENOME[1:]:
        assert b"##contig=<ID=%s,length=%d>" % (name.encode(), ln) in e.header
    # pieces that end exactly on a contig's last column / begin on its first one
    pieces = [(qb, 5999), (6000, 6899), (6900, 11899), (11900, 14399), (14400, qe)]
    assert b"".join(e.run_interval(b, en, arena_bytes=1 << 22)[0] for b, en in pieces) == \
        helpers.oracle_run_synth(dict(q, query_column_ranges=[[list(p) for p in pieces]]), cells, synth.SEED, with_header=False)[0]
    e.close()


def test_bcf_and_windowed_streaming_across_contigs(gdb, tmp_path, monkeypatch):
    """BCF2: the record's CHROM is the contig's index in the header dictionary and POS / END are contig-relative; then the same
    array streamed through HBM one begin column per staging window (every interval alive at a contig's end crosses a carry-over;
    none may leak into the next contig)"""
    from genomicsdb_amd import synth
    N = 90
    g = synth.Generator(N, 0, END, contigs=GENOME)
    cells, _ = g.chunk_bytes(END)
    q = helpers.synth_query(tmp_path, N, 5000, 15000, contigs=GENOME)
    q["reference_genome"] = _fasta(tmp_path, synth)
    want, nrec, _ = helpers.oracle_run(q, cells)
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    assert s.read() == want
    s.close()
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True)
    bcf = s.read()
    s.close()
    assert helpers.bcf_stream_to_text(bcf) == want
    monkeypatch.setenv("GDBAMD_STAGE_BUDGET_BYTES", "1")
    s = gdb.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
    assert s.read() == want
    s.close()


def _fasta(tmp_path, synth):
    p = tmp_path / "genome.fa"
    with open(p, "wb") as f:
        for name, off, ln in GENOME:
            f.write(b">" + name.encode() + b"\n")
            seq = synth.reference(off, ln)
            for i in range(0, ln, 60):
                f.write(seq[i:i + 60] + b"\n")
    return str(p)


def test_reference_vid_offsets_beyond_int32(gdb, tmp_path):
    """the real table (tests/inputs/vid.json: contigs 1-22, X, Y, MT): columns around the 22 | X boundary are > 2^31"""
    from genomicsdb_amd import synth
    ctg = synth.genome_contigs(os.path.join(helpers.GOLDEN, "inputs", "vid.json"))
    x_off = dict((n, o) for n, o, _ in ctg)["X"]
    assert x_off > 2**31
    N, B, L = 120, x_off - 2500, 5000
    g = synth.Generator(N, B, L, contigs=ctg)
    cells, _ = g.chunk_bytes(B + L)
    q = helpers.synth_query(tmp_path, N, B + 100, B + L - 100, contigs=ctg)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, synth.SEED, with_header=False)
    assert _chrom_runs(want) == [b"22", b"X"]
    assert want.split(b"\nX\t", 1)[1].split(b"\t", 1)[0] == b"1"          # first record of X is at POS 1
    e = gdb.CombineEngine(q)
    e.stage_cells(cells)
    e.set_reference(B, synth.reference(B, L + 16))
    body, st = e.run_interval(B + 100, B + L - 100, arena_bytes=1 << 24)
    assert st.num_records == nrec and body == want
    eb = gdb.CombineEngine(q, is_bcf=True)
    eb.stage_cells(cells)
    eb.set_reference(B, synth.reference(B, L + 16))
    bcf, _ = eb.run_interval(B + 100, B + L - 100, arena_bytes=1 << 24)
    import struct
    import bcf2text
    h = bcf2text.Header(e.header.decode())
    at, lines = 0, []
    while at < len(bcf):
        l_shared, l_indiv = struct.unpack_from("<II", bcf, at)
        lines.append(bcf2text.record_to_text(h, bcf[at:at + 8 + l_shared + l_indiv], helpers.format_float))
        at += 8 + l_shared + l_indiv
    assert ("\n".join(lines) + "\n").encode() == want
    e.close()
    eb.close()


@pytest.mark.parametrize("world", [2, 3])
def test_partitions_per_rank_and_paged_concat(world):
    """c4 end to end at test size: `world` ranks (gloo: they share the one GPU of the box), one column partition each - one of
    them beginning exactly at a contig's offset -, pages of at most 64 KiB concatenated on rank 0 through a ring of three
    buffers, byte-identical to the oracle run over the same partitioning"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, GDBAMD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(helpers.ROOT, "tests", "tools", "c4_sanity.py"), "150", str(64 << 10)],
                       capture_output=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout.decode()[-1500:], r.stderr.decode()[-3000:])
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["ranks"] == world and len(out["partitions"]) == world
    assert out["contigs_in_order"] == ["1", "2", "3", "X", "Y", "MT"]
    assert out["pages"] > 2 * world and out["max_page"] <= 64 << 10
    assert any(b in (o for _, o, _ in GENOME) for b, _ in out["partitions"][1:])


def test_c4_in_miniature_100000_rows_8_partitions():
    """BASELINE.json configs[3] as ONE workload at test size: 100 000 samples x a whole (scaled-down) genome of six contigs x 8 column
    partitions on 8 ranks (gloo: they share the box's GPU), the partitions cut by the reference's recipe - ColumnHistogramOperator on the
    device + equi_partition_and_print_bins -, every rank's pages (4.5 MB of text per record) through the ordered paged concat to rank 0,
    byte-identical there to the oracle run over the same partitioning; the partitions' cell counts are balanced"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, GDBAMD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(helpers.ROOT, "tests", "tools", "c4_sanity.py"), "100000", str(64 << 20), "60"],
                       capture_output=True, timeout=1500, env=env)
    assert r.returncode == 0, (r.stdout.decode()[-1500:], r.stderr.decode()[-3000:])
    out = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert out["ok"] and out["ranks"] == 8 and len(out["partitions"]) == 8
    assert out["contigs_in_order"] == ["1", "2", "3", "X", "Y", "MT"]
    assert out["max_page"] <= 64 << 20 and out["pages"] >= 16                        # a record is 4.5 - 22 MB of text: a few per page
    per, ideal = out["cells_per_partition"], out["cells"] / 8.0
    assert sum(per) == out["cells"]
    # the greedy cut closes a partition with the bin that takes it over total / P: no partition exceeds the ideal by more than one bin
    assert max(per[:-1]) <= ideal + out["largest_bin"]


def test_partitions_from_the_histogram_are_within_5_percent(gdb, tmp_path):
    """genome-mode input with many columns per partition: 300 samples, 16 500 columns, 8 partitions from the device histogram (bins of 10
    columns): every partition's cell count within 5 % of total / 8; the histogram itself equals numpy's over the cells' begin columns"""
    import struct
    import numpy as np
    from genomicsdb_amd import synth, dist as gdist
    N = 300
    g = synth.Generator(N, 0, END, contigs=GENOME)
    cells, _ = g.chunk_bytes(END)
    q = helpers.synth_query(tmp_path, N, 0, END - 1, contigs=GENOME)
    e = gdb.CombineEngine(q)
    e.stage_cells(cells)
    cols, off = [], 0
    while off < len(cells):
        _, col, sz = struct.unpack_from("<qqQ", cells, off)
        cols.append(col); off += sz
    for b0, b1, bin_size in ((0, END - 1, 10), (2000, 9999, 100), (0, 4_000_000_000, 1_000_000)):
        got = e.column_histogram(b0, b1, bin_size)
        idx = np.array([0 if c <= b0 else (len(got) - 1 if c >= b1 else (c - b0) // bin_size) for c in cols])
        want = np.bincount(idx, minlength=len(got)).astype(np.uint64)
        assert (got == want).all(), (b0, b1, bin_size)
    counts = e.column_histogram(0, END - 1, 10)
    spans = [gdist.balanced_partition(counts, 0, 10, r, 8) for r in range(8)]
    assert spans[0][0] == 0 and all(spans[i][1] + 1 == spans[i + 1][0] for i in range(7))
    per = [int(counts[b // 10:(e_ + 1) // 10].sum()) for b, e_ in spans]
    assert sum(per) == len(cols)
    assert all(abs(p - len(cols) / 8.0) <= 0.05 * len(cols) / 8.0 for p in per), per
    e.close()


def test_gt_mpi_gather_produce_histogram(gdb, tmp_path):
    """the reference's `gt_mpi_gather --produce-histogram` (tools/src/gt_mpi_gather.cc:404-411, :600): cells per 100 columns of [0, 4e9) and
    the equal-load partitions for 128 ... 2 ranks, in the text of ColumnHistogramOperator::equi_partition_and_print_bins; counted on the GPU,
    checked against the same arithmetic in Python over the cells' begin columns"""
    import struct
    from golden_cases import CASES
    case = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    _, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, _ = helpers.query_json(callsets, vid, ov, mode)
    ws = tmp_path / "ws"
    (ws / "t0_1_2").mkdir(parents=True)
    (ws / "t0_1_2" / "cells.bin").write_bytes(cells)
    q["workspace"] = str(ws); q["array"] = "t0_1_2"
    qf = tmp_path / "query.json"
    qf.write_text(json.dumps(q))
    r = subprocess.run([os.path.join(helpers.ROOT, "genomicsdb_amd", "gt_mpi_gather"), "-j", str(qf), "--produce-histogram"], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    cols, off = [], 0
    while off < len(cells):
        _, col, sz = struct.unpack_from("<qqQ", cells, off)
        cols.append(col); off += sz
    nbins = 4_000_000_000 // 100 + 1
    hist = {}
    for c in cols:
        b = 0 if c <= 0 else (nbins - 1 if c >= 4_000_000_000 else c // 100)
        hist[b] = hist.get(b, 0) + 1
    want = ""
    for parts in (128, 64, 32, 16, 8, 4, 2):
        per = len(cols) / parts
        want += "Total %d #bins %d count/bins %.1f\n" % (len(cols), parts, per)
        i, keys = 0, sorted(hist)
        while i < nbins:
            cur, j = 0, i
            while cur < per and j < nbins:      # (skipping over empty bins in one step: they add nothing)
                nxt = next((k for k in keys if k >= j), None)
                if nxt is None:
                    j = nbins
                    break
                cur += hist[nxt]
                j = nxt + 1
            want += "%d,%d,%d\n" % (i * 100, j * 100 - 1, cur)
            i = j
        want += "\n"
    assert r.stdout.decode() == want


@pytest.mark.parametrize("interval,total", [((12150, 12250), 2), ((12144, 17384), 5), ((12300, 17000), 0)])
def test_produce_histogram_counts_the_cells_of_the_query_intervals(gdb, tmp_path, interval, total):
    """the reference hands ColumnHistogramOperator the cells of the QUERY's column intervals (iterate_over_cells(ad, query_config, op),
    tools/src/gt_mpi_gather.cc:404-411): the cells that begin inside an interval and the intervals that began in front of it and reach
    its begin.  t0 / t1 / t2: reference blocks [12140, 12294] and [12144, 12276], three cells at 17384.  [12150, 12250]: the two blocks
    reach in (2); [12144, 17384]: four cells begin inside, the block from 12140 reaches in (5); [12300, 17000]: nothing - and the
    partition lines are then just the header (the reference's loop would not terminate on an empty histogram)."""
    from golden_cases import CASES
    case = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    _, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, _ = helpers.query_json(callsets, vid, ov, mode)
    ws = tmp_path / "ws"
    (ws / "t0_1_2").mkdir(parents=True)
    (ws / "t0_1_2" / "cells.bin").write_bytes(cells)
    q["workspace"] = str(ws); q["array"] = "t0_1_2"
    q["query_column_ranges"] = [[[interval[0], interval[1]]]]
    qf = tmp_path / "query.json"
    qf.write_text(json.dumps(q))
    r = subprocess.run([os.path.join(helpers.ROOT, "genomicsdb_amd", "gt_mpi_gather"), "-j", str(qf), "--produce-histogram"], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    blocks = r.stdout.decode().split("\n\n")
    first = blocks[0].split("\n")
    assert first[0] == "Total %d #bins 128 count/bins %.1f" % (total, total / 128.0)
    if total == 0:
        assert len(first) == 1
    else:
        assert sum(int(l.split(",")[2]) for l in first[1:]) == total


def test_rccl_paged_concat_one_rank(gdb, tmp_path):
    """the same concat over the "nccl" backend (= RCCL) with the pages in HBM; one GPU here, so one rank: device buffers and
    the page pull are exercised, ordering across ranks by the gloo runs above and tests/test_multi_rank_cpu.py"""
    import socket
    import torch
    import torch.distributed as dist
    from genomicsdb_amd import dist as gdist, synth
    N = 100
    g = synth.Generator(N, 0, END, contigs=GENOME)
    cells, _ = g.chunk_bytes(END)
    q = helpers.synth_query(tmp_path, N, 0, END - 1, contigs=GENOME)
    e = gdb.CombineEngine(q)
    e.stage_cells(cells)
    e.set_reference(0, synth.reference(0, END + 16))
    want, _ = e.run_interval(0, END - 1, arena_bytes=1 << 30)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        got = []
        total = gdist.gather_interval_paged(e, 0, END - 1, lambda t: got.append(bytes(t.cpu().numpy().tobytes())), page_bytes=1 << 18)
        assert total == len(want) and b"".join(got) == want and len(got) > 4
    finally:
        dist.destroy_process_group()
    e.close()


def test_bench_gpus_2_reports_two_ranks():
    """`python bench.py --gpus 2` launches its own two ranks (they share this box's GPU under gloo) and reports n_gpus 2 with
    the records of both partitions"""
    env = dict(os.environ, GDBAMD_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--samples", "100", "--interval-bp", "40000", "--window-bp", "20000", "--steps", "2", "--warmup", "1", "--arena-mb", "256",
            "--no-cpu-baseline", "--no-stream"]
    one = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", "1"] + args, capture_output=True, timeout=600, env=env)
    two = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2"] + args, capture_output=True, timeout=600, env=env)
    assert one.returncode == 0 and two.returncode == 0, two.stderr.decode()[-3000:]
    o1 = json.loads([l for l in one.stdout.decode().splitlines() if l.startswith("{")][-1])
    o2 = json.loads([l for l in two.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert o1["n_gpus"] == 1 and o2["n_gpus"] == 2
    r1, r2 = o1["value"] * o1["ms_per_step"], o2["value"] * o2["ms_per_step"]     # records per step (x 1e3)
    assert 1.8 < r2 / r1 < 2.2
