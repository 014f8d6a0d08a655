"""JSON and (b)gzip readers.  Until round 3 the oracle read its files through the product's csrc/common/mini_json.hpp and gz_text.hpp: a
misreading there would have been common to checker and checked.  Now the oracle has readers of its own (oracle/oracle_json.hpp, written
independently); here BOTH pairs are compared with Python's json and gzip modules on every fixture file of the tree and on hostile hand-made
inputs - the oracle's through liboracle_gvcf.so, the product's through the kernel-body harness tests/hostsim."""
import ctypes
import glob
import gzip
import json
import os

import pytest

import helpers

FIXTURES = os.path.join(helpers.GOLDEN, "inputs")


def _dump_py(v):
    if v is None:
        return "n"
    if v is True:
        return "t"
    if v is False:
        return "f"
    if isinstance(v, int):
        return "i%d" % v
    if isinstance(v, float):
        return "d%s" % ("%.17g" % v)
    if isinstance(v, str):
        return "s" + v.encode("utf-8").hex()
    if isinstance(v, _Obj):
        return "{" + "".join(k.encode("utf-8").hex() + ":" + _dump_py(x) + "," for k, x in v.pairs) + "}"
    if isinstance(v, list):
        return "[" + "".join(_dump_py(x) + "," for x in v) + "]"
    raise TypeError(type(v))


class _Obj:
    def __init__(self, pairs):
        self.pairs = pairs


def _native(which, kind):
    """(function, free) of the oracle's / the product's reader"""
    if which == "oracle":
        lib = helpers.oracle_lib()
        return getattr(lib, "oracle_json_dump" if kind == "json" else "oracle_gz_read_all"), lib.oracle_free
    lib = helpers.hostsim_lib()
    return getattr(lib, "hostsim_json_dump" if kind == "json" else "hostsim_gz_read_all"), lib.hostsim_free


def _dump_one(which, text):
    fn, free = _native(which, "json")
    fn.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_uint64]
    free.argtypes = [ctypes.c_void_p]
    out, n = ctypes.c_void_p(), ctypes.c_uint64()
    err = ctypes.create_string_buffer(512)
    rc = fn(text if isinstance(text, bytes) else text.encode("utf-8"), ctypes.byref(out), ctypes.byref(n), err, 512)
    if rc != 0:
        raise ValueError(err.value.decode())
    s = ctypes.string_at(out.value, n.value).decode()
    free(out)
    return s


def _dump_native(text):
    """both readers must succeed and agree (or both refuse)"""
    res = []
    for which in ("oracle", "product"):
        try:
            res.append(_dump_one(which, text))
        except ValueError as e:
            res.append(e)
    if isinstance(res[0], ValueError) or isinstance(res[1], ValueError):
        assert isinstance(res[0], ValueError) and isinstance(res[1], ValueError), res
        raise res[0]
    assert res[0] == res[1]
    return res[0]


JSON_FILES = sorted(glob.glob(os.path.join(FIXTURES, "*.json")) + glob.glob(os.path.join(FIXTURES, "callsets", "*.json")))


def test_there_are_fixture_files():
    assert len(JSON_FILES) >= 15


@pytest.mark.parametrize("path", JSON_FILES, ids=[os.path.basename(p) for p in JSON_FILES])
def test_mini_json_agrees_with_python_json_on_fixture(path):
    text = open(path, "rb").read()
    want = _dump_py(json.loads(text.decode("utf-8"), object_pairs_hook=_Obj))
    assert _dump_native(text) == want


@pytest.mark.parametrize("doc", [
    '{"a": [1, -2, 3.5, 1e3, -0.25E-2, 9223372036854775807, -9223372036854775808], "b": {"c": null, "d": true, "e": false}}',
    '{"esc": "tab\\there \\"quoted\\" back\\\\slash \\/ nl\\n \\u00e9 \\u4e2d", "": "", "k k": " "}',
    '  [ ]  ', '{}', '[[[[]]],{"x":[{}]}]', '"just a string"', '-17', '0.1', '[1,\n\t2 ,\r\n 3]',
    '{"dup": 1, "dup": 2, "z": [true,false,null]}',
    '{"query_column_ranges": [{"range_list": [{"low": 0, "high": 1000000000}]}], "big": 1234567890123456789}',
])
def test_mini_json_agrees_with_python_json_on_hand_made_documents(doc):
    assert _dump_native(doc) == _dump_py(json.loads(doc, object_pairs_hook=_Obj))


@pytest.mark.parametrize("doc", ['{"a": 1,}', '[1 2]', '{"a" 1}', '{"a": tru}', '"open', '{"a": 1} x', '', '[', '{"a":}'])
def test_mini_json_refuses_what_python_refuses(doc):
    with pytest.raises(ValueError):
        json.loads(doc)
    with pytest.raises(ValueError):
        _dump_native(doc)


def _gz_native(path):
    got = []
    for which in ("oracle", "product"):
        fn, free = _native(which, "gz")
        fn.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_uint64]
        free.argtypes = [ctypes.c_void_p]
        out, n = ctypes.c_void_p(), ctypes.c_uint64()
        err = ctypes.create_string_buffer(512)
        rc = fn(os.fsencode(path), ctypes.byref(out), ctypes.byref(n), err, 512)
        if rc != 0:
            raise ValueError(err.value.decode())
        got.append(ctypes.string_at(out.value, n.value))
        free(out)
    assert got[0] == got[1]
    return got[0]


GZ_FILES = sorted(glob.glob(os.path.join(FIXTURES, "**", "*.gz"), recursive=True))


@pytest.mark.parametrize("path", GZ_FILES, ids=[os.path.basename(p) for p in GZ_FILES])
def test_gz_text_agrees_with_python_gzip_on_fixture(path):
    assert len(GZ_FILES) >= 10
    with gzip.open(path, "rb") as f:     # (multi-member files = BGZF blocks are read to the end by both)
        want = f.read()
    assert _gz_native(path) == want


def test_gz_text_plain_multi_member_and_empty(tmp_path):
    p = tmp_path / "plain.txt"
    p.write_bytes(b"not compressed\nline 2\n")
    assert _gz_native(str(p)) == b"not compressed\nline 2\n"
    m = tmp_path / "multi.gz"
    m.write_bytes(gzip.compress(b"first member\n") + gzip.compress(b"") + gzip.compress(b"third\n" * 5000, compresslevel=1))
    assert _gz_native(str(m)) == b"first member\n" + b"third\n" * 5000
    e = tmp_path / "empty.gz"
    e.write_bytes(gzip.compress(b""))
    assert _gz_native(str(e)) == b""


def test_the_oracle_includes_nothing_of_the_product():
    """the dependency direction the round-3 review asked for: oracle/ builds from its own files alone"""
    odir = os.path.join(helpers.ROOT, "oracle")
    for fn in os.listdir(odir):
        if fn.endswith((".hpp", ".cc", ".h")) or fn == "Makefile":
            text = open(os.path.join(odir, fn)).read()
            assert "genomicsdb_amd/csrc" not in text and '#include "mini_json.hpp"' not in text and '#include "gz_text.hpp"' not in text, fn
