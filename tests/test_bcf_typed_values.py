"""BCF2 typed values, byte for byte.  The BCF streams of this build are otherwise validated by decoding them with the tests' own reader
(tests/tools/bcf2text.py): a misreading of the format common to encoder and decoder would pass.  These known answers are written down
from the BCFv2.2 specification (section 6.3.3 "Type encoding": a descriptor byte = element count in the high nibble - 15 = "a typed
integer with the real count follows" - and the type in the low nibble: 1 int8, 2 int16, 3 int32, 5 float, 7 char; little endian;
missing / end-of-vector = 0x80 / 0x81, 0x8000 / 0x8001, 0x80000000 / 0x80000001; an absent value is a descriptor 0x00) and from the
width rule htslib applies in bcf_enc_vint (vcf.c; vcf.h: BCF_MIN_BT_INT8 = -120, BCF_MAX_BT_INT8 = 127, BCF_MIN_BT_INT16 = -32760,
BCF_MAX_BT_INT16 = 32767: the eight values below each type's minimum are reserved), which the reference inherits (vcf_adapter.cc:475-509)."""
import ctypes

import helpers

MISSING, EOV = -2**31, -2**31 + 1


def _vint(vals):
    lib = helpers.hostsim_lib()
    a = (ctypes.c_int32 * max(1, len(vals)))(*vals)
    out = (ctypes.c_uint8 * 4096)()
    n = lib.hostsim_bcf_enc_vint(a, len(vals), out, 4096)
    return bytes(out[:n])


def test_single_integers_take_the_narrowest_type_that_is_not_reserved():
    assert _vint([1]) == bytes([0x11, 0x01])
    assert _vint([0]) == bytes([0x11, 0x00])
    assert _vint([-1]) == bytes([0x11, 0xFF])
    assert _vint([127]) == bytes([0x11, 0x7F])
    assert _vint([128]) == bytes([0x12, 0x80, 0x00])
    assert _vint([-120]) == bytes([0x11, 0x88])
    assert _vint([-121]) == bytes([0x12, 0x87, 0xFF])            # -121 .. -128 are int8's reserved values
    assert _vint([32767]) == bytes([0x12, 0xFF, 0x7F])
    assert _vint([32768]) == bytes([0x13, 0x00, 0x80, 0x00, 0x00])
    assert _vint([-32760]) == bytes([0x12, 0x08, 0x80])
    assert _vint([-32761]) == bytes([0x13, 0x07, 0x80, 0xFF, 0xFF])
    assert _vint([2147483647]) == bytes([0x13, 0xFF, 0xFF, 0xFF, 0x7F])


def test_vectors_share_one_type_chosen_over_their_real_values():
    assert _vint([]) == bytes([0x00])
    assert _vint([1, 2, 3]) == bytes([0x31, 1, 2, 3])
    assert _vint([0, 300]) == bytes([0x22, 0x00, 0x00, 0x2C, 0x01])
    assert _vint([1, MISSING, EOV]) == bytes([0x31, 0x01, 0x80, 0x81])                        # the two sentinels do not widen the type
    assert _vint([300, MISSING, EOV]) == bytes([0x32, 0x2C, 0x01, 0x00, 0x80, 0x01, 0x80])
    assert _vint([70000, MISSING]) == bytes([0x23, 0x70, 0x11, 0x01, 0x00, 0x00, 0x00, 0x00, 0x80])
    assert _vint([MISSING]) == bytes([0x11, 0x80])
    assert _vint([EOV]) == bytes([0x11, 0x81])
    assert _vint([MISSING, MISSING]) == bytes([0x21, 0x80, 0x80])
    assert _vint([-5, 5]) == bytes([0x21, 0xFB, 0x05])


def test_counts_of_15_and_more_overflow_into_a_typed_integer():
    assert _vint([0] * 14) == bytes([0xE1]) + bytes(14)
    assert _vint([0] * 15) == bytes([0xF1, 0x11, 0x0F]) + bytes(15)
    assert _vint([7] * 127) == bytes([0xF1, 0x11, 0x7F]) + bytes([7]) * 127
    assert _vint([7] * 128) == bytes([0xF1, 0x12, 0x80, 0x00]) + bytes([7]) * 128
    assert _vint([1000] * 20) == bytes([0xF2, 0x11, 0x14]) + bytes([0xE8, 0x03]) * 20
    lib = helpers.hostsim_lib()
    out = (ctypes.c_uint8 * 16)()
    for size, typ, want in ((0, 7, [0x07]), (4, 7, [0x47]), (14, 5, [0xE5]), (15, 7, [0xF7, 0x11, 0x0F]), (16, 7, [0xF7, 0x11, 0x10]), (300, 7, [0xF7, 0x12, 0x2C, 0x01]),
                            (40000, 1, [0xF1, 0x13, 0x40, 0x9C, 0x00, 0x00]), (2, 5, [0x25])):
        n = lib.hostsim_bcf_enc_size(size, typ, out, 16)
        assert n == len(want) and list(out[:n]) == want, (size, typ)
