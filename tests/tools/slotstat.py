import sys, os, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import genomicsdb_amd, helpers
from genomicsdb_amd import synth
N, B, L = 1000, 10_000_000, 100_000
tmp = tempfile.mkdtemp()
q = helpers.synth_query(tmp, N, B, B + L - 1)
eng = genomicsdb_amd.CombineEngine(q)
g = synth.Generator(N, B, L + 3000)
ptr, nbytes, nc = g.next_chunk(B + L + 3000)
eng.stage_cells_begin(); eng.stage_cells_append(ptr, nbytes); eng.stage_cells_end()
eng.set_reference(B, synth.reference(B, L + 8000))
_, st = eng.run_interval(B, B + L - 1, arena_bytes=8 << 30, fetch=False)
ns = st.num_text_slots
print("slots", ns, "types", st.num_record_types, "pool bytes", st.text_pool_bytes, "overflow bytes", st.text_pool_bytes - ns * 128, "records", st.num_records, "bytes_out", st.bytes_out)
