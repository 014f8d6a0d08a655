"""where the staging time goes: generator vs the engine's stage_cells_append (host size walk + H2D + device parse)"""
import sys, os, tempfile, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import genomicsdb_amd, helpers
from genomicsdb_amd import synth
N, B, L = 1000, 10_000_000, 2_000_000
tmp = tempfile.mkdtemp()
q = helpers.synth_query(tmp, N, B, B + L - 1)
eng = genomicsdb_amd.CombineEngine(q)
g = synth.Generator(N, B, L)
eng.stage_cells_begin()
tg = ta = 0.0; nb = 0; nc_tot = 0
col = B
while col < B + L:
    col += 1_000_000
    t0 = time.time(); ptr, nbytes, nc = g.next_chunk(col); t1 = time.time()
    eng.stage_cells_append(ptr, nbytes); t2 = time.time()
    tg += t1 - t0; ta += t2 - t1; nb += nbytes; nc_tot += nc
t0 = time.time(); eng.stage_cells_end(); te = time.time() - t0
print("cells %d bytes %.2f GB: generator %.2f s, append %.2f s (%.2f GB/s, %.1f M cells/s), finish %.2f s" % (nc_tot, nb / 1e9, tg, ta, nb / 1e9 / ta, nc_tot / 1e6 / ta, te))
