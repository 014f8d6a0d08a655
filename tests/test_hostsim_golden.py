"""The GDB_HD kernel bodies (same source the HIP kernels compile), run serially on the CPU, against the
reference goldens.  Cases the device path does not support yet must fail loudly (UnsupportedOnDevice)."""
import pytest

import helpers
from golden_cases import CASES

# golden cases whose configuration the device path rejects today, with the reason it reports
DEVICE_UNSUPPORTED = {}   # every C++-path golden of the reference that the oracle covers now runs on the device path


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_hostsim_matches_reference_golden(case):
    name, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    if name in DEVICE_UNSUPPORTED:
        # either the plan builder rejects the configuration or a kernel raises an error bit (-> exception in the product)
        try:
            _, errbits = helpers.hostsim_run(q, cells)
        except RuntimeError as e:
            assert "UnsupportedOnDevice" in str(e)
            return
        assert errbits != 0
        return
    txt, errbits = helpers.hostsim_run(q, cells)
    assert errbits == 0
    assert txt == helpers.golden_text(golden)


def test_unqueried_rows_still_split_intervals():
    """The reference's scan iterates ALL array rows and closes the current interval at every cell begin before it checks whether
    the row is queried (query_variants.cc:478-507).  A query for samples 0 and 2 therefore gets the interval splits caused by
    sample 1's cells: staging keeps those cells as position-only boundary markers."""
    case = [c for c in CASES if c[0] == "t0_1_2_vcf_at_0"][0]
    _, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    q["query_row_ranges"] = [{"range_list": [{"low": 0, "high": 0}, {"low": 2, "high": 2}]}]
    want, nrec, _ = helpers.oracle_run(q, cells, partition_begin=pb)
    got, errbits = helpers.hostsim_run(q, cells)
    assert errbits == 0 and got == want
    assert b"END=12144" in want    # the split at 12145 comes from the unqueried sample's cell


def test_median_selection_is_the_c_library_selection():
    """The reference takes medians with std::nth_element and prints the selected float: with -0 and +0 tied at the middle the
    printed sign depends on the permutation libstdc++'s introselect leaves.  The restatement the kernels use must leave the
    same permutation as the library for random, tie-heavy, sorted and adversarial (depth-limit -> heap select) inputs."""
    import ctypes
    import numpy as np
    lib = helpers.hostsim_lib()
    lib.hostsim_nth_element_same.restype = ctypes.c_int
    lib.hostsim_nth_element_same.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
    rng = np.random.default_rng(7)

    def check(a, nth=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        nth = len(a) // 2 if nth is None else nth
        assert lib.hostsim_nth_element_same(a.ctypes.data, len(a), nth, -1) == 1, (len(a), nth)
        for depth in (0, 1, 3):        # the library's loop with a tiny depth budget: its heap-select branch
            assert lib.hostsim_nth_element_same(a.ctypes.data, len(a), nth, depth) == 1, (len(a), nth, depth)

    for n in list(range(1, 40)) + [63, 64, 65, 100, 257, 1000, 4097, 20000]:
        for rep in range(6):
            check(rng.standard_normal(n))
            z = rng.choice(np.array([-0.0, 0.0, -0.5, 0.25, 1.0], dtype=np.float32), size=n)          # many tied zeros of both signs
            check(z)
            check(np.round(rng.standard_normal(n), 1) + np.float32(0.0) * rng.choice([-1.0, 1.0], size=n))
            check(rng.standard_normal(n), nth=int(rng.integers(0, n)))
        check(np.arange(n)); check(np.arange(n)[::-1]); check(np.zeros(n)); check(np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]))


ASA_ATTRIBUTES = ["END", "REF", "ALT", "BaseQRankSum", "ClippingRankSum", "MQRankSum", "ReadPosRankSum", "MQ", "RAW_MQ", "MQ0", "DP",
                  "GT", "GQ", "SB", "AD", "PL", "PGT", "PID", "MIN_DP", "DP_FORMAT", "FILTER", "AS_RAW_MQ", "AS_RAW_MQRankSum"]


def test_queried_composite_field_is_flattened():
    """A query that names a composite field (AS_RAW_MQRankSum: tuple of bins and counts) gets its flattened tuple elements at the
    end of the attribute list (VariantQueryConfig::flatten_composite_fields, variant_query_config.cc:187-229; the attribute list
    is run.py's asa_vcf_attributes).  Oracle and kernel bodies agree, and the record of the golden is in the output."""
    cells = helpers.cells_for("t0_1_2_all_asa.json", "vid_all_asa.json")
    q, pb = helpers.query_json("t0_1_2_all_asa.json", "vid_all_asa.json", {"query_column_ranges": [{"range_list": [{"low": 0, "high": 1000000000}]}]}, "query")
    q["attributes"] = ASA_ATTRIBUTES
    want, nrec, _ = helpers.oracle_run(q, cells, partition_begin=pb)
    got, errbits = helpers.hostsim_run(q, cells)
    assert errbits == 0 and got == want
    assert b"AS_RAW_MQ=8.000,10.000,9.000|28.000,16.000,18.000,14.000|31.000|32.000,12.000,13.000,14.000;AS_RAW_MQRankSum=|0.600,6,0.800,2,0.900,15|0.100,2,0.600,7|" in want


def _asa_cells(rng, n_samples, n_sites, lookalike=False):
    """synthetic cells for vid_all_asa.json: every site has a shared pool of ALT alleles, each sample takes some of them and
    carries random per-allele vectors / histograms (empty vectors, NaN and repeated bins included)"""
    import struct
    import vcf2cells
    import os
    fields, contigs = vcf2cells.load_vid(os.path.join(helpers.GOLDEN, "inputs", "vid_all_asa.json"))
    attrs, info_fields, fmt_fields = vcf2cells.schema_attributes(fields)
    cells = []
    for s in range(n_sites):
        col = 1000 + 10 * s
        pool = ["A", "C", "T", "GA"][: 1 + int(rng.integers(0, 4))]
        for row in range(n_samples):
            if rng.random() < 0.15:
                continue
            k = 1 + int(rng.integers(0, len(pool)))
            alts = list(rng.permutation(pool)[:k]) + ["<NON_REF>"]
            nall = len(alts) + 1

            def vec(lo, hi, as_int=False):
                n = int(rng.integers(lo, hi + 1))
                out = []
                for _ in range(n):
                    if rng.random() < 0.1:
                        out.append("NaN" if not as_int else "")
                    elif as_int:
                        out.append(str(int(rng.integers(0, 50))))
                    else:
                        out.append("%.2f" % (rng.integers(-4000, 4000) / 100.0))
                return out
            mq = "|".join(",".join(vec(0, 4)) for _ in range(nall)) if rng.random() < 0.85 else None
            def hist():
                n = int(rng.integers(0, 4))
                toks = []
                for _ in range(n):
                    toks += ["%.1f" % (int(rng.integers(-5, 6)) / 10.0) if rng.random() > 0.1 else "NaN", str(int(rng.integers(1, 20)))]
                return ",".join(toks)
            rs = "|".join(hist() for _ in range(nall)) if rng.random() < 0.85 else None
            if lookalike:
                # the counts of one allele, as consecutive int32: [1, 0, 5000, 0, 64, 0, 7, 7] = the bytes of a cell header
                # (row 1, column 5000, cell size 64) in the middle of a cell's payload
                fake = "0.1,1,0.2,0,0.3,5000,0.4,0,0.5,64,0.6,0,0.7,7,0.8,7"
                rs = "|".join([fake] + [hist() for _ in range(nall - 1)])
            info_d = {"AS_RAW_MQ": mq, "AS_RAW_MQRankSum": rs}
            alt_ser = "|".join("&" if a == "<NON_REF>" else a for a in alts)
            body = struct.pack("<q", col)
            body += struct.pack("<i", 1) + b"G"
            body += struct.pack("<i", len(alt_ser)) + alt_ser.encode()
            body += struct.pack("<I", vcf2cells.TILEDB_NULL_FLOAT_BITS)
            body += struct.pack("<i", 0)
            for f in info_fields:
                body += vcf2cells.encode_values(f, info_d.get(f.vcf_name), len(alts))
            for f in fmt_fields:
                if f.vcf_name == "GT":
                    body += vcf2cells.encode_gt(f, "0/1")
                else:
                    body += vcf2cells.encode_values(f, None, len(alts))
            cells.append((row, col, struct.pack("<qqQ", row, col, 24 + len(body)) + body))
    cells.sort(key=lambda t: (t[1], t[0]))
    return b"".join(c[2] for c in cells)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_allele_specific_annotations_fuzz(seed):
    """random per-allele vectors and histograms over sites with different ALT subsets per sample: the streaming reducers of
    gdb_asa.hpp (allele-LUT lookup per element, next-larger-bin selection) against the oracle's remapped blobs, vectors and
    std::map (remap_allele_specific_annotations, compute_valid_element_wise_sum_2D_vector, histogram_sum)"""
    import numpy as np
    rng = np.random.default_rng(seed)
    cells = _asa_cells(rng, 3, 40)
    q, pb = helpers.query_json("t0_1_2_all_asa.json", "vid_all_asa.json", {}, "load")
    want, nrec, _ = helpers.oracle_run(q, cells, partition_begin=pb)
    got, errbits = helpers.hostsim_run(q, cells)
    assert errbits == 0
    assert nrec >= 30 and want.count(b"AS_RAW_MQ=") > 10 and want.count(b"AS_RAW_MQRankSum=") > 10
    assert got == want


def test_fixed3_matches_printf_on_every_class_of_float():
    """allele-specific annotations print with std::fixed << setprecision(3) (genomicsdb_multid_vector_field / stringify_2D_vector);
    the device does it with integer arithmetic on the float's mantissa: checked against C's "%.3f" of the same float - ties,
    denormals, sums beyond 2^53 (exact integers of up to 39 digits), infinities and NaNs included"""
    import ctypes
    import random
    import struct
    lib = helpers.hostsim_lib()
    lib.hostsim_fixed3.argtypes = [ctypes.c_float, ctypes.c_char_p, ctypes.c_uint64]
    libc = ctypes.CDLL(None)
    libc.snprintf.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_double]
    rnd = random.Random(3)
    bits = [0, 0x80000000, 1, 0x007FFFFF, 0x00800000, 0x3A83126F, 0x3F000000, 0x3A03126F, 0x7F7FFFFF, 0xFF7FFFFF, 0x7F800000, 0xFF800000,
            0x7FC00000, 0x5A000000, 0x5A800001, 0x5F000000, 0x7E967699, 0x4B000000, 0x4B7FFFFF, 0x3F8020C5, 0x3B03126F, 0x3A83126E]
    bits += [rnd.getrandbits(32) for _ in range(20000)]
    bits += [struct.unpack("<I", struct.pack("<f", k / 2000.0))[0] for k in range(-3000, 3000)]      # exact ties at the third decimal
    for b in bits:
        f = struct.unpack("<f", struct.pack("<I", b))[0]
        mine = ctypes.create_string_buffer(128)
        n = lib.hostsim_fixed3(ctypes.c_float(f), mine, 128)
        want = ctypes.create_string_buffer(128)
        libc.snprintf(want, 128, b"%.3f", ctypes.c_double(f))
        w = want.value
        if f != f:
            w = b"-nan" if b >> 31 else b"nan"      # (a NaN's sign survives the float -> double conversion; glibc prints it)
        assert n >= 0 and mine.value == w, (hex(b), mine.value, w)
