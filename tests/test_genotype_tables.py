"""The reference's genotype-order known answers (GTest genotype_ordering.*, src/test/cpp/src/test_non_diploid_mapper.cc:113-476,
committed as data in tests/golden/genotype_tables.json) against BOTH the CPU oracle's restatement of
remap_data_based_on_genotype_general and the device's own enumeration (the GDB_HD functions, compiled by g++ in hostsim)."""
import ctypes
import json
import os
from math import comb

import pytest

import helpers

TABLES = json.load(open(os.path.join(helpers.GOLDEN, "genotype_tables.json")))["cases"]


def genotype_index(combo):
    """VCF genotype index of an allele combination given as letters: sum_i C(a_i + i, i + 1) over the sorted alleles"""
    a = sorted(ord(c) - ord("A") for c in combo)
    return sum(comb(x + i, i + 1) for i, x in enumerate(a))


def enumerate_genotypes(num_alleles, ploidy):
    out = []

    def rec(prefix, max_allele):
        if len(prefix) == ploidy:
            out.append("".join(chr(ord("A") + x) for x in sorted(prefix)))
            return
        for x in range(max_allele + 1):
            rec(prefix + [x], x)
    rec([], num_alleles - 1)
    return sorted(out, key=genotype_index)


@pytest.mark.parametrize("case", TABLES, ids=[c["name"] for c in TABLES])
def test_genotype_order_known_answers(case):
    na, ploidy = case["num_merged_alleles"], case["ploidy"]
    merged = case["merged_genotypes"]
    # the table itself is in VCF order
    assert merged == enumerate_genotypes(na, ploidy)
    assert [genotype_index(g) for g in merged] == list(range(len(merged)))
    lut = case["input_to_merged"]
    if lut is None:
        m2i = list(range(na))
        n_in = 5
    else:
        m2i = [-1] * na
        for i, m in lut.items():
            m2i[m] = int(i)
        n_in = len(lut)
    nr_exists = bool(case["NON_REF_exists"]) and lut is not None and len(lut) == 3     # the 3-allele call carries <NON_REF> as its last allele
    nr_in = n_in - 1 if nr_exists else -1
    want_inputs = case["input_genotypes"] or merged
    want = [-1 if g == "." else genotype_index(g) for g in want_inputs]
    # (a) the oracle's restatement of the reference algorithm
    lib = helpers.oracle_lib()
    arr = (ctypes.c_int64 * na)(*m2i)
    out = (ctypes.c_int64 * 64)()
    n = lib.oracle_genotype_map(arr, na, 1 if nr_exists else 0, ploidy, 10**6, out, 64)
    assert n == len(merged)
    assert list(out[:n]) == want
    # (b) the device's enumeration (kernel bodies compiled for the host)
    hs = helpers.hostsim_lib()
    hs.hostsim_genotype_map.argtypes = [ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    arr32 = (ctypes.c_int32 * na)(*m2i)
    out2 = (ctypes.c_int64 * 64)()
    n2 = hs.hostsim_genotype_map(arr32, na, nr_in, ploidy, out2, 64)
    assert n2 == len(merged)
    assert list(out2[:n2]) == want
