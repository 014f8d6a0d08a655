"""record types / text slots of one window: how many record types an interval has and what the entry text table costs
usage (GPU box): python tests/tools/type_stats.py [samples] [bp]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import genomicsdb_amd, helpers
from genomicsdb_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
B = 10_000_000
q = helpers.synth_query(tempfile.mkdtemp(), N, B, B + L - 1)
g = synth.Generator(N, B, L + 3000)
ptr, nbytes, nc = g.next_chunk(B + L + 3000)
e = genomicsdb_amd.CombineEngine(q)
e.stage_cells_begin(); e.stage_cells_append(ptr, nbytes); e.stage_cells_end()
e.set_reference(B, synth.reference(B, L + 4096))
for it in range(4):
    _, st = e.run_interval(B, B + L - 1, arena_bytes=40 << 30, fetch=False)
    print("run %d phases ms: sweep %.1f site %.1f size %.1f write %.1f (kernel %.2f)" % (it, st.ms_sweep, st.ms_site, st.ms_size, st.ms_write, st.ms_write_kernel_avg))
T = st.num_heavy_incidences
print("N %d L %d: records %d cells_in_window %d heavy incidences %d (%.1f per record) types %d text slots %d (%.2f per cell) pool %.2f GB bytes_out %.2f GB"
      % (N, L, st.num_records, st.num_cells_in_window, T, T / max(1, st.num_records), st.num_record_types, st.num_text_slots, st.num_text_slots / max(1, st.num_cells_in_window),
         st.text_pool_bytes / 1e9, st.bytes_out / 1e9))
print("phases ms: sweep %.1f site %.1f size %.1f write %.1f (kernel %.2f)" % (st.ms_sweep, st.ms_site, st.ms_size, st.ms_write, st.ms_write_kernel_avg))
