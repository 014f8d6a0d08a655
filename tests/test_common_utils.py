"""The oracle and the product share two format-agnostic utilities (csrc/common/mini_json.hpp, gz_text.hpp): a parsing bug there
would be common-mode - invisible to every product-vs-oracle comparison.  Here both are checked against independent
implementations (Python's json and gzip modules) on every fixture file of the tree and on hostile hand-made inputs."""
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


def _dump_native(text):
    lib = helpers.oracle_lib()
    lib.oracle_json_dump.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_uint64]
    out, n = ctypes.c_void_p(), ctypes.c_uint64()
    err = ctypes.create_string_buffer(512)
    rc = lib.oracle_json_dump(text if isinstance(text, bytes) else text.encode("utf-8"), ctypes.byref(out), ctypes.byref(n), err, 512)
    if rc != 0:
        raise ValueError(err.value.decode())
    s = ctypes.string_at(out.value, n.value).decode()
    lib.oracle_free(out)
    return s


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
    lib = helpers.oracle_lib()
    lib.oracle_gz_read_all.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_uint64]
    out, n = ctypes.c_void_p(), ctypes.c_uint64()
    err = ctypes.create_string_buffer(512)
    rc = lib.oracle_gz_read_all(os.fsencode(path), ctypes.byref(out), ctypes.byref(n), err, 512)
    if rc != 0:
        raise ValueError(err.value.decode())
    b = ctypes.string_at(out.value, n.value)
    lib.oracle_free(out)
    return b


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
