"""Differential fuzz on the GPU box: random synthetic configurations, HIP path vs the CPU oracle, byte for byte.
usage: python tests/tools/fuzz.py [ncases] [seed0]"""
import random, sys, tempfile, time
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import genomicsdb_amd, helpers
from genomicsdb_amd import synth

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
t00 = time.time()
for case in range(ncases):
    rnd = random.Random(seed0 + case)
    N = rnd.choice([1, 2, 3, 7, 31, 64, 65, 100, 130, 200, 333, 1500])
    L = rnd.randint(300, 3500) if N < 150 else rnd.randint(300, 1500) if N < 1000 else rnd.randint(100, 300)
    B = 10_000_000 + rnd.randint(0, 5) * 1000
    gseed = rnd.randint(1, 10**6)
    dense = None
    if rnd.random() < 0.3:
        dense = (B + 50 * rnd.randint(1, 5), 50 * rnd.randint(1, 4), 50, rnd.choice([3, 20, 70]))
    off = rnd.randint(0, 400)
    qb, qe = B + off, B + off + L - 1
    opts = {}
    if rnd.random() < 0.3: opts["produce_GT_field"] = True
    if rnd.random() < 0.2: opts["produce_GT_with_min_PL_value_for_spanning_deletions"] = True; opts["produce_GT_field"] = True
    if rnd.random() < 0.15: opts["sites_only_query"] = True
    if rnd.random() < 0.2: opts["max_diploid_alt_alleles_that_can_be_genotyped"] = rnd.choice([1, 2, 5, 64])
    arena = rnd.choice([1 << 12, 1 << 16, 1 << 20, 1 << 26])
    tmp = tempfile.mkdtemp()
    rs_scale = rnd.choice([None, None, 1.0, 2.0, 10.0])      # coarse rank sums: tied medians, zeros of both signs
    # generator modes: overlapping intervals of one sample (the scan's overlap override), FILTER ids on the variant cells (one id:
    # unions of equal ids across samples), ID tokens (sorted union, one or two tokens per call; the schema gains an ID attribute)
    modes = {}
    if rnd.random() < 0.4: modes["overlap_permille"] = rnd.choice([30, 150, 400])
    if rnd.random() < 0.3: modes["filter_permille"] = rnd.choice([100, 500]); opts["produce_FILTER_field"] = True
    if "filter_permille" in modes and rnd.random() < 0.5:     # a second id: unions of different ids, written in libstdc++'s set order
        modes["filter2_permille"] = rnd.choice([100, 400]); modes["filter_id"] = rnd.choice([0, 1]); modes["filter_id2"] = 1 - modes["filter_id"]
    with_id = rnd.random() < 0.3
    if with_id: modes["id_permille"] = rnd.choice([100, 600]); modes["with_id"] = True
    # genome mode: the columns of the array are cut into contigs of random lengths (one boundary every few hundred columns), so
    # the query interval and the pieces below cross contig ends; CHROM / POS / END turn contig-relative
    contigs = None
    if rnd.random() < 0.35:
        contigs, at, i = [], 0, 0
        span = B + off + L + 2500
        first = B - rnd.randint(0, 3) * 1000          # the first boundary may lie before, at or behind the array's first column
        while at < span + 10:
            ln = first if i == 0 and first > 0 else rnd.choice([1, 2, 37, 150, 400, 1500, 5000])
            contigs.append(("c%d" % i, at, ln)); at += ln; i += 1
        if rnd.random() < 0.5:
            contigs = contigs[::-1]                   # vid order need not be offset order
    g = synth.Generator(N, B, off + L + 2500, seed=gseed, dense=dense, rank_sum_scale=rs_scale, contigs=contigs, **modes)
    cells, nc = g.chunk_bytes(B + off + L + 2500)
    q = helpers.synth_query(tmp, N, qb, qe, with_id=with_id, contigs=contigs)
    q.update(opts)
    if with_id and rnd.random() < 0.5: q["id_union_order"] = "unordered_set"     # (round 5: the ID union of a Release build of the reference)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, gseed, with_header=False)
    eng = genomicsdb_amd.CombineEngine(q)
    # stage in 1-3 parts
    nparts = rnd.choice([1, 1, 2, 3])
    if nparts == 1:
        eng.stage_cells(cells)
    else:
        import struct, ctypes
        offs = []; o = 0
        while o < len(cells):
            offs.append(o); o += struct.unpack_from("<Q", cells, o + 16)[0]
        cuts = sorted(set([0] + [offs[len(offs) * i // nparts] for i in range(1, nparts)] + [len(cells)]))
        eng.stage_cells_begin()
        for a, b in zip(cuts[:-1], cuts[1:]):
            part = cells[a:b]
            buf = ctypes.create_string_buffer(part, len(part))
            eng.stage_cells_append(ctypes.addressof(buf), len(part))
        eng.stage_cells_end()
    eng.set_reference(B, synth.reference(B, off + L + 4096, seed=gseed))
    got, st = eng.run_interval(qb, qe, arena_bytes=arena)
    ok = got == want and st.num_records == nrec
    if not ok and os.environ.get("FUZZ_VERBOSE"):
        k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1)
        print("  whole interval: got %d bytes, want %d, first difference at %d: %r / %r" % (len(got), len(want), k, got[max(0, k - 60):k + 30], want[max(0, k - 60):k + 30]))
    if rnd.random() < 0.3:   # the same interval in pieces cut before cell begins: byte-identical by construction
        maxc = rnd.choice([25, 100, 400])
        pieces, cur = [], qb
        while cur <= qe:
            pe = eng.split_point(cur, qe, maxc)
            body, _ = eng.run_interval(cur, pe, arena_bytes=arena)
            pieces.append(body)
            cur = pe + 1
        if ok and b"".join(pieces) != want and os.environ.get("FUZZ_VERBOSE"):
            j = b"".join(pieces)
            k = next((i for i in range(min(len(j), len(want))) if j[i] != want[i]), -1)
            print("  pieces (max %d columns, %d pieces): got %d bytes, want %d, first difference at %d: %r / %r" % (maxc, len(pieces), len(j), len(want), k, j[max(0, k - 80):k + 40], want[max(0, k - 80):k + 40]))
        ok = ok and b"".join(pieces) == want
        if ok and len(pieces) > 1 and rnd.random() < 0.6:   # (round 5) the same pieces, 2-3 of them in flight at a time on lane pipelines that share the fragment
            cur, ivs = qb, []
            while cur <= qe:
                pe = eng.split_point(cur, qe, maxc)
                ivs.append((cur, pe)); cur = pe + 1
            res = eng.run_intervals(ivs, arena_bytes=max(arena, 1 << 12), lanes=rnd.choice([2, 3]), fetch=True)
            ok = b"".join(r[0] for r in res) == want
            if not ok: print("  lanes: %d intervals differ from the one-at-a-time pieces" % len(ivs))
    eng.close()
    if ok and rnd.random() < 0.25:   # the same interval as BGZF blocks deflated on the device: the inflated stream is the text
        import zlib
        ez = genomicsdb_amd.CombineEngine(q, output_format="z")
        ez.stage_cells(cells)
        ez.set_reference(B, synth.reference(B, off + L + 4096, seed=gseed))
        zbody, _ = ez.run_interval(qb, qe, arena_bytes=max(arena, 1 << 16))
        ez.close()
        out, rest = [], zbody
        while rest:
            d = zlib.decompressobj(31)
            out.append(d.decompress(rest)); rest = d.unused_data
        ok = b"".join(out) == want
        if not ok:
            j = b"".join(out)
            k = next((i for i in range(min(len(j), len(want))) if j[i] != want[i]), -1)
            print("  BGZF check: inflated %d bytes, text %d bytes, %d blocks, first difference at %d: %r / %r" % (len(j), len(want), len(out), k, j[max(0, k - 20):k + 20], want[max(0, k - 20):k + 20]))
    if ok and rnd.random() < 0.2:    # the same interval as BCF2 records, decoded by the tests' own reader back to text
        import struct
        eb = genomicsdb_amd.CombineEngine(q, output_format="bu")
        eb.stage_cells(cells)
        eb.set_reference(B, synth.reference(B, off + L + 4096, seed=gseed))
        bbody, _ = eb.run_interval(qb, qe, arena_bytes=max(arena, 1 << 16))
        htext = eb.header
        eb.close()
        if not htext.endswith(b"\x00"): htext += b"\x00"
        stream = b"BCF\x02\x02" + struct.pack("<I", len(htext)) + htext + bbody
        dec = b"".join(l + b"\n" for l in helpers.bcf_stream_to_text(stream).split(b"\n") if l and not l.startswith(b"#"))
        ok = dec == want
        if not ok:
            k = next((i for i in range(min(len(dec), len(want))) if dec[i] != want[i]), -1)
            print("  BCF2 check: decoded %d bytes, text %d bytes, first difference at %d: %r / %r" % (len(dec), len(want), k, dec[max(0, k - 60):k + 30], want[max(0, k - 60):k + 30]))
    if ok and rnd.random() < 0.25:   # the same array streamed through HBM in windows of a random staging budget (carry-over of the live
        import ctypes                  # intervals on the device, the next window staged by the prefetch thread while this one computes)
        budget = rnd.choice([1, 3000, 40_000, 400_000])
        os.environ["GDBAMD_STAGE_BUDGET_BYTES"] = str(budget)
        try:
            ew = genomicsdb_amd.CombineEngine(q)
            buf = ctypes.create_string_buffer(cells, len(cells))
            ew.open_memory_cells((ctypes.addressof(buf), len(cells)))
            ew.set_reference(B, synth.reference(B, off + L + 4096, seed=gseed))
            parts, cur, nwin = [], qb, 0
            while cur <= qe:
                lo, hi = ew.cover(cur)
                nwin += 1
                end = min(qe, hi)
                body, _ = ew.run_interval(cur, end, arena_bytes=arena)
                parts.append(body)
                cur = end + 1
            ew.close()
        finally:
            del os.environ["GDBAMD_STAGE_BUDGET_BYTES"]
        ok = b"".join(parts) == want
        if not ok:
            j = b"".join(parts)
            k = next((i for i in range(min(len(j), len(want))) if j[i] != want[i]), -1)
            print("  windowed (budget %d, %d windows): got %d bytes, want %d, first difference at %d: %r / %r" % (budget, nwin, len(j), len(want), k, j[max(0, k - 60):k + 30], want[max(0, k - 60):k + 30]))
    if not ok:
        bad += 1
        print("MISMATCH case %d: N=%d L=%d B=%d off=%d seed=%d dense=%s rs_scale=%s opts=%s modes=%s contigs=%s arena=%d parts=%d records %d/%d" % (case, N, L, B, off, gseed, dense, rs_scale, opts, modes, (len(contigs) if contigs else None), arena, nparts, st.num_records, nrec), flush=True)
print("fuzz: %d cases, %d mismatches, %.0f s" % (ncases, bad, time.time() - t00))
sys.exit(1 if bad else 0)
