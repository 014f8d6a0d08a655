"""fragment file: raw vs DEFLATE tiles inflated on the device - sizes, save and load times for one c2 window
usage (GPU box): python tests/tools/frag_bench.py [samples] [bp]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import genomicsdb_amd
from genomicsdb_amd import synth
import helpers
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
B = 10_000_000
tmp = tempfile.mkdtemp(prefix="gdbamd_frag_")
q = helpers.synth_query(tmp, N, B, B + L - 1)
g = synth.Generator(N, B, L)
e = genomicsdb_amd.CombineEngine(q)
e.stage_cells_begin()
ptr, nbytes, nc = g.next_chunk(B + L)
e.stage_cells_append(ptr, nbytes)
e.stage_cells_end()
t = time.time(); e.save_fragment(os.path.join(tmp, "raw.gdbamd")); t_raw = time.time() - t
t = time.time(); e.save_fragment(os.path.join(tmp, "z.gdbamd"), compress=True); t_z = time.time() - t
e.close()
sizes = {n: os.path.getsize(os.path.join(tmp, n)) for n in ("raw.gdbamd", "z.gdbamd")}
print("cells %d (%.2f GB of reference cells); raw file %.2f GB (saved in %.1f s), compressed %.2f GB = %.2f of raw (saved in %.1f s)"
      % (nc, nbytes / 1e9, sizes["raw.gdbamd"] / 1e9, t_raw, sizes["z.gdbamd"] / 1e9, sizes["z.gdbamd"] / sizes["raw.gdbamd"], t_z))
for name in ("raw.gdbamd", "z.gdbamd", "raw.gdbamd", "z.gdbamd"):
    e = genomicsdb_amd.CombineEngine(q)
    torch.cuda.synchronize(); t = time.time()
    e.load_fragment(os.path.join(tmp, name))
    torch.cuda.synchronize(); dt = time.time() - t
    print("load %-10s %.3f s = %.1f GB/s of file, %.1f GB/s of columns" % (name, dt, sizes[name] / dt / 1e9, sizes["raw.gdbamd"] / dt / 1e9))
    e.close()
