"""Pins the CPU oracle to the reference's own golden outputs (tests/golden_outputs of the
reference, copied verbatim under tests/golden/outputs): byte-identical, header included."""
import pytest

import helpers
from golden_cases import CASES


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_reference_golden(case):
    name, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    txt, nrec, _ = helpers.oracle_run(q, cells, partition_begin=pb)
    assert txt == helpers.golden_text(golden)


@pytest.mark.parametrize("case", [c for c in CASES if c[5] == "query"], ids=[c[0] for c in CASES if c[5] == "query"])
def test_oracle_batched_output_identical(case):
    """'-p 128': the operator overflows after (almost) every record and the scan resumes through
    the scan state (reference tests/run.py:938 'batched_vcf')."""
    name, callsets, vid, ov, golden, mode = case
    cells = helpers.cells_for(callsets, vid)
    q, pb = helpers.query_json(callsets, vid, ov, mode)
    txt, _, _ = helpers.oracle_run(q, cells, partition_begin=pb, buffer_limit=128)
    assert txt == helpers.golden_text(golden)
