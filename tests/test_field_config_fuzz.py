"""One extra INFO / FORMAT field in random vid configurations (tests/tools/field_fuzz.py): length A / R / G / fixed / VAR, int / float,
every combine operation the vid mapper accepts - the oracle against the kernel bodies under g++ here, against the device in the GPU
suite.  (Written after a hand-derived known answer found that scalar reducers over A-length INFO fields read the stored element 0:
tests/test_info_scalar_allele_fields.py.)"""
import os
import sys

import pytest

import helpers

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
import field_fuzz


@pytest.mark.parametrize("cls,seeds", [("INFO", range(0, 36)), ("FORMAT", range(0, 24))])
def test_random_field_configurations_oracle_and_kernel_bodies(tmp_path, cls, seeds):
    seen = set()
    for seed in seeds:
        d = tmp_path / ("%s%d" % (cls, seed))
        d.mkdir()
        cells, q, what = field_fuzz.inputs(seed, cls, str(d))
        seen.add(what)
        txt, nrec, _ = helpers.oracle_run(q, cells, with_header=False)
        got, err = helpers.hostsim_run(q, cells, with_header=False)
        assert err == 0 and got == txt, (seed, what)
        assert (b"XF=" in txt) if cls == "INFO" else (b":XF" in txt or b"XF:" in txt), (seed, what)
    assert len(seen) >= (12 if cls == "INFO" else 8)


@pytest.mark.gpu
@pytest.mark.parametrize("cls,seeds", [("INFO", range(100, 112)), ("FORMAT", range(100, 108))])
def test_random_field_configurations_device(tmp_path, cls, seeds):
    import genomicsdb_amd
    for seed in seeds:
        d = tmp_path / ("%s%d" % (cls, seed))
        d.mkdir()
        cells, q, what = field_fuzz.inputs(seed, cls, str(d))
        want, _, _ = helpers.oracle_run(q, cells)
        s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
        got = s.read()
        s.close()
        assert got == want, (seed, what)
        s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True)
        bcf = s.read()
        s.close()
        assert helpers.bcf_stream_to_text(bcf) == want, (seed, what)


def _fixture_cases():
    from golden_cases import CASES
    return [c for c in CASES if c[5] == "load" and not c[3] and "as_array" not in c[0]]


def _fixture_inputs(seed, tmp_path):
    loads = _fixture_cases()
    name, callsets, vid_name, ov, golden, mode = loads[seed % len(loads)]
    d = tmp_path / ("s%d" % seed)
    d.mkdir()
    cells, vp, cp, what = field_fuzz.inputs_on_fixture(seed, callsets, vid_name, str(d))
    q, _ = helpers.query_json(callsets, vid_name, ov, mode)
    q["vid_mapping_file"] = vp
    q["callset_mapping_file"] = cp
    return cells, q, name + ": " + what


def test_an_extra_field_on_every_fixture_oracle_and_kernel_bodies(tmp_path):
    """haploid / triploid calls, spanning deletions, overlapping intervals, files of several samples: three random field configurations
    per callset fixture of the reference"""
    assert len(_fixture_cases()) >= 12
    for seed in range(3 * len(_fixture_cases())):
        cells, q, what = _fixture_inputs(seed, tmp_path)
        txt, nrec, _ = helpers.oracle_run(q, cells)
        got, err = helpers.hostsim_run(q, cells)
        assert err == 0 and got == txt, (seed, what)


@pytest.mark.gpu
def test_an_extra_field_on_every_fixture_device(tmp_path):
    import genomicsdb_amd
    for seed in range(200, 200 + 2 * len(_fixture_cases())):
        cells, q, what = _fixture_inputs(seed, tmp_path)
        want, _, _ = helpers.oracle_run(q, cells)
        s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20)
        got = s.read()
        s.close()
        assert got == want, (seed, what)
        s = genomicsdb_amd.GenomicsDBQueryStream(query_json=q, cells=cells, buffer_capacity=1 << 20, is_bcf=True)
        bcf = s.read()
        s.close()
        assert helpers.bcf_stream_to_text(bcf) == want, (seed, what)
