"""The CPU oracle restatement must reproduce, byte for byte, what the unmodified reference wrote
(tests/golden/*.out, produced by tests/golden/make_golden.py)."""
import os

import pytest

import oracle_cli
from golden_util import align_columns
from cases import CASES

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_reproduces_reference_output(case):
    geno = os.path.join(GOLD, case["fixture"] + ".geno.gz")
    out = os.path.join(GOLD, case["name"] + ".out")
    argv = [a.format(geno=geno, dir=GOLD, out=out) for a in case["argv"]]
    got = oracle_cli.run(case["tool"], argv)
    with open(out) as f:
        want = f.read()
    assert align_columns(got, want) == want
