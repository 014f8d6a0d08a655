"""debug helper: run one golden case through the GPU stream and diff with the oracle"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import helpers
from golden_cases import CASES
import genomicsdb_amd

name = sys.argv[1]
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
case = [c for c in CASES if c[0] == name][0]
_, callsets, vid, ov, golden, mode = case
cells = helpers.cells_for(callsets, vid)
q, pb = helpers.query_json(callsets, vid, ov, mode)
want, _, _ = helpers.oracle_run(q, cells, partition_begin=pb)
s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=cap)
got = s.read()
print(name, "match" if got == want else "MISMATCH", len(got), len(want))
if got != want:
    a = got.decode(errors="replace").split("\n"); b = want.decode().split("\n")
    for x, y in zip(a, b):
        if x != y and not x.startswith("##"):
            print("GOT ", x[:400]); print("WANT", y[:400])
