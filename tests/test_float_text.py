"""Every float has a text: put_float (gdb_core.hpp, the body the device runs, compiled for the host in tests/hostsim) against what
the reference prints - htslib's kputd: its six-digit rule inside [0.0001, 999999], else the sign and printf("%g") of the magnitude
(broad_combined_gvcf.cc:374-429 hands every value on, none is refused).  "%g" comes from the C library here: 4 million random bit
patterns, every exponent with a spread of mantissas, all boundaries, NaN / Inf, and the oracle's format_float on the same values."""
import ctypes
import os
import struct

import numpy as np

import helpers

HOSTSIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim", "libhostsim.so")


def _lib():
    lib = ctypes.CDLL(HOSTSIM)
    lib.hostsim_put_float_check.restype = ctypes.c_int64
    lib.hostsim_put_float_check.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64]
    lib.hostsim_put_float.argtypes = [ctypes.c_float, ctypes.c_char_p, ctypes.c_uint64]
    return lib


def _check(lib, bits):
    bits = np.ascontiguousarray(bits, dtype=np.uint32)
    mine, libc = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
    bad = lib.hostsim_put_float_check(bits.ctypes.data, len(bits), mine, libc, 64)
    assert bad < 0, "bits 0x%08x: put_float %r, C library %r" % (int(bits[bad]), mine.value, libc.value)


def _text(lib, f):
    buf = ctypes.create_string_buffer(64)
    lib.hostsim_put_float(ctypes.c_float(f), buf, 64)
    return buf.value.decode()


def test_random_bit_patterns_against_the_c_library():
    lib = _lib()
    rng = np.random.default_rng(20260929)
    _check(lib, rng.integers(0, 1 << 32, size=4_000_000, dtype=np.uint64).astype(np.uint32))


def test_every_exponent_and_the_boundaries():
    lib = _lib()
    pats = []
    mans = [0, 1, 2, 3, 0x400000, 0x7FFFFF, 0x7FFFFE, 0x555555, 0x2AAAAA, 0x100000, 0x0FFFFF, 0x19999A, 0x666666]
    for e in range(0, 255):
        for m in mans:
            for sign in (0, 1):
                pats.append((sign << 31) | (e << 23) | m)
    # subnormals, a dense walk around every power of ten and around the style switches (1e-4, 999999 / 1e6) and rounding ties
    for k in range(0, 64):
        pats += [k, 0x7FFFFF - k, 0x800000 + k]
    for p in range(-45, 39):
        b = struct.unpack("<I", struct.pack("<f", float("1e%d" % p) if -46 < p < 39 else 0.0))[0]
        pats += [(b + d) & 0xFFFFFFFF for d in range(-40, 41)]
    for v in (1e-4, 9.9999994e-5, 9.99995e-5, 999999.0, 999999.06, 999999.44, 999999.5, 1e6, 1000005.0, 1000015.0, 1.5e-5, 2.5e-5, 1.2345675e-7, 16777216.0, 3.4028235e38,
              9.2e18, 9.3e18, 1.8446744e19, 5e-324, 1.17549435e-38):
        b = struct.unpack("<I", struct.pack("<f", np.float32(v)))[0]
        pats += [(b + d) & 0xFFFFFFFF for d in range(-8, 9)]
    _check(lib, np.array(pats, dtype=np.uint64).astype(np.uint32))


def test_nan_inf_and_known_answers():
    lib = _lib()
    assert _text(lib, float("inf")) == "inf" and _text(lib, float("-inf")) == "-inf"
    nan_pos = struct.unpack("<f", struct.pack("<I", 0x7FC00000))[0]
    assert _text(lib, nan_pos) == "nan"
    for v, want in ((5e-5, "5e-05"), (1.5e-7, "1.5e-07"), (1e-4, "0.0001"), (9.9999994e-5, "0.0001"), (999999.06, "999999"), (1e6, "1e+06"), (1234567.0, "1.23457e+06"),
                    (3.4028235e38, "3.40282e+38"), (1.4e-45, "1.4013e-45"), (-2.5e-10, "-2.5e-10"), (1e20, "1e+20"), (0.5, "0.5"), (-0.0, "-0"), (8.0, "8.0")):
        assert _text(lib, v) == want, (v, _text(lib, v), want)


def test_oracle_and_kernel_body_print_the_same():
    """the checker's format_float (snprintf-based) and the product's put_float on 200 000 random patterns plus the specials"""
    lib = _lib()
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 1 << 32, size=200_000, dtype=np.uint64).astype(np.uint32)
    for b in list(bits[:20000]) + [0x7F800000, 0xFF800000, 0x7FC00000, 0x00000001, 0x80000001, 0x38D1B717, 0x38D1B716, 0x497423F0, 0x497423F1]:
        f = struct.unpack("<f", struct.pack("<I", int(b)))[0]
        if int(b) in (0x7F800001, 0x7F800002):    # BCF2 "missing" / "end of vector": never reach the formatter
            continue
        if f != f and (int(b) >> 31):             # the C library prints the sign bit of a NaN; np/ctypes may not preserve it
            continue
        assert helpers.format_float(f) == _text(lib, f), hex(int(b))


# ---- the same through the whole path: cells whose floats are tiny / huge / subnormal / inf / NaN --------------------------------
def _stress_case(tmp_path, n_samples=120, L=1500, qual_op="sum"):
    """synthetic cells of which a quarter of the variant calls carry floats outside [1e-4, 999999]; QUAL combined (sum, so that NaN and
    inf stay order-independent), rank sums / MQ by median, RAW_MQ by sum - the reference's defaults (known_field_info.cc:239-308)"""
    import json
    from genomicsdb_amd import synth
    B = 10_000_000
    g = synth.Generator(n_samples, B, L + 2500, seed=11, float_stress_permille=250)
    cells, _ = g.chunk_bytes(B + L + 2500)
    tpl = json.load(open(os.path.join(helpers.GOLDEN, "inputs", "vid.json")))
    tpl["fields"]["QUAL"] = {"type": "float", "VCF_field_combine_operation": qual_op}
    tp = os.path.join(str(tmp_path), "vid_template_qual_%s.json" % qual_op)
    json.dump(tpl, open(tp, "w"))
    vp, cp = synth.write_metadata(str(tmp_path), n_samples, tp)
    q = {"vid_mapping_file": vp, "callset_mapping_file": cp, "vcf_header_filename": os.path.join(helpers.GOLDEN, "inputs", "template_vcf_header.vcf"),
         "attributes": helpers.VCF_ATTRIBUTES_ORDER + ["QUAL"], "query_column_ranges": [[[B + 10, B + L]]]}
    return B, cells, q


def _has_the_unusual_texts(body):
    quals = [l.split(b"\t")[5] for l in body.split(b"\n") if l]
    infos = b"\n".join(l.split(b"\t")[7] for l in body.split(b"\n") if l)
    assert any(b"e-" in x for x in quals) and any(b"e+" in x for x in quals) and b"inf" in quals and b"nan" in quals, sorted(set(quals))[:30]
    assert b"e-" in infos and b"e+" in infos


def test_kernel_bodies_print_unusual_floats_like_the_oracle(tmp_path):
    B, cells, q = _stress_case(tmp_path)
    want, nrec, _ = helpers.oracle_run(q, cells, with_header=False)
    _has_the_unusual_texts(want)
    got, err = helpers.hostsim_run(q, cells, with_header=False, rows_per_chunk=16, records_per_run=5)
    assert err == 0 and got == want


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("is_bcf", [False, True])
def test_device_prints_unusual_floats_like_the_oracle(tmp_path, is_bcf):
    """QUAL and the INFO reducers with values below 1e-4, above 2^63, subnormal, inf and NaN: no GDB_ERR_FLOAT_RANGE any more, the
    text is the reference's (kputd + "%g"); as BCF2 the values are binary and decode to the same text"""
    import struct as st_
    import bcf2text
    import genomicsdb_amd
    from genomicsdb_amd import synth
    B, cells, q = _stress_case(tmp_path, n_samples=700, L=2500)
    want, nrec, _ = helpers.oracle_run_synth(q, cells, 11, with_header=False)
    _has_the_unusual_texts(want)
    e = genomicsdb_amd.CombineEngine(q, is_bcf=is_bcf)
    e.stage_cells(cells)
    e.set_reference(B, synth.reference(B, 2500 + 2500 + 16, seed=11))
    body, stats = e.run_interval(B + 10, B + 2500, arena_bytes=1 << 20)
    hdr = e.header
    e.close()
    assert stats.num_records == nrec
    if is_bcf:
        h = bcf2text.Header(genomicsdb_amd.CombineEngine(q).header.decode()) if not hdr.startswith(b"##") else bcf2text.Header(hdr.decode())
        at, lines = 0, []
        while at < len(body):
            l_shared, l_indiv = st_.unpack_from("<II", body, at)
            lines.append(bcf2text.record_to_text(h, body[at:at + 8 + l_shared + l_indiv], helpers.format_float))
            at += 8 + l_shared + l_indiv
        body = ("\n".join(lines) + "\n").encode()
    assert body == want
