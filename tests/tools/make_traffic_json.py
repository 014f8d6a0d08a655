"""profiles/traffic.json from a tests/tools/prof_traffic.sh run: per-launch HBM bytes of the dominant kernel, the configuration
it was measured on and the hash of the kernel sources (bench.py prints the figure only while all three still match).
usage: python tests/tools/make_traffic_json.py gpurun_out/<dir>/traffic_by_kernel.json [samples window_bp arena_mb]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
src = json.load(open(sys.argv[1]))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
arena = (int(sys.argv[4]) if len(sys.argv) > 4 else 49152) << 20
out = {"samples": N, "window_bp": W, "arena_bytes": arena, "kernel_source_hash": bench.kernel_source_hash(),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace only (tests/tools/prof_traffic.sh); FETCH_SIZE doubled (gfx950), KiB -> bytes",
       "k_assemble_write": src["k_assemble_write"]}
for k in ("k_assemble_size", "k_slots_light<0>", "k_site_size", "k_cell_ranges", "k_cells_scatter", "k_cells_measure"):
    if k in src:
        out[k] = src[k]
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out["k_assemble_write"]))
