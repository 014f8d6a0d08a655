"""BASELINE.json configs[4] at its stated size: 50 000 samples, a 100 kb dense region in which every sample starts an insertion
from a pool of K = 64 alleles every 50 columns (2 000 hot sites of 50 000 calls, ~66 merged alleles, PL vectors of 2 278
genotypes).  The interval is worked off in pieces (engine.split_point), the pages stay in HBM.
usage (GPU box): python tests/tools/c5_full.py [samples] [bp] [piece_bp] [K]"""
import sys, os, tempfile, time, json
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
import genomicsdb_amd, helpers
from genomicsdb_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
PIECE = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
K = int(sys.argv[4]) if len(sys.argv) > 4 else 64
B = 10_000_000
tmp = tempfile.mkdtemp()
q = helpers.synth_query(tmp, N, B, B + L - 1)
q["max_diploid_alt_alleles_that_can_be_genotyped"] = 64
eng = genomicsdb_amd.CombineEngine(q)
t0 = time.time()
g = synth.Generator(N, B, L + 3000, dense=(B, L, 50, K))
eng.stage_cells_begin()
col, ncells, nbytes_all = B, 0, 0
while col < B + L + 3000:
    col = min(B + L + 3000, col + 4000)      # (a chunk stays below 4 GB: the size walk on the device counts in 32 bits)
    ptr, nbytes, nc = g.next_chunk(col)
    eng.stage_cells_append(ptr, nbytes); ncells += nc; nbytes_all += nbytes
eng.stage_cells_end()
eng.set_reference(B, synth.reference(B, L + 8000))
t_stage = time.time() - t0
def run(LANES):
    tot = {"records": 0, "bytes": 0, "remap": 0, "ms_sweep": 0.0, "ms_site": 0.0, "ms_size": 0.0, "ms_write": 0.0, "pieces": 0, "heavy": 0}
    def account(st):
        tot["records"] += st.num_records; tot["bytes"] += st.bytes_out; tot["remap"] += st.num_remap_elements; tot["heavy"] += st.num_heavy_incidences
        tot["ms_sweep"] += st.ms_sweep; tot["ms_site"] += st.ms_site; tot["ms_size"] += st.ms_size; tot["ms_write"] += st.ms_write; tot["pieces"] += 1
    cur, qe = B, B + L - 1
    if LANES > 1:
        pieces = []
        while cur <= qe:
            pe = eng.split_point(cur, qe, PIECE)
            pieces.append((cur, pe)); cur = pe + 1
        eng.run_intervals(pieces[:LANES], arena_bytes=16 << 30, lanes=LANES)          # every lane once, untimed: it adopts and classifies the fragment, sizes its buffers
        t1 = time.time()
        for st in eng.run_intervals(pieces, arena_bytes=16 << 30, lanes=LANES):
            account(st)
    else:
        t1 = time.time()
        while cur <= qe:
            pe = eng.split_point(cur, qe, PIECE)
            _, st = eng.run_interval(cur, pe, arena_bytes=64 << 30, fetch=False)
            account(st)
            cur = pe + 1
            if time.time() - t1 > 900:
                print("time limit: stopped at column", cur); break
    dt = time.time() - t1
    dev = (tot["ms_sweep"] + tot["ms_site"] + tot["ms_size"] + tot["ms_write"]) * 1e-3
    print(json.dumps({"what": "c5: %d samples x %d bp dense region, K = %d, %d-column pieces" % (N, L, K, PIECE), "lanes": LANES, "cells": ncells, "cell_bytes": nbytes_all, "stage_s": t_stage,
                      "covered_bp": cur - B, "pieces": tot["pieces"], "records": tot["records"], "heavy_incidences": tot["heavy"], "bytes_out": tot["bytes"], "wall_s": dt, "device_s": dev,
                      "phase_ms": {k: tot[k] for k in ("ms_sweep", "ms_site", "ms_size", "ms_write")}, "positions_per_s": tot["records"] / dt, "GBps_out": tot["bytes"] / dt / 1e9,
                      "remap_elements": tot["remap"], "remap_elements_per_s": tot["remap"] / dt, "remap_elements_per_s_device": tot["remap"] / max(dev, 1e-9),
                      "site_pass_PL_read_GBps": tot["remap"] * 4 / max(tot["ms_site"] * 1e-3, 1e-9) / 1e9}))


# C5_LANES="1,3": the same staged array once per lane count (pieces in flight at a time, gdbamd_engine_run_intervals; > 1: the phase times overlap)
for lanes in [int(x) for x in os.environ.get("C5_LANES", "1").split(",")]:
    run(lanes)
