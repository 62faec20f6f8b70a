"""A short fixed-seed campaign of the randomised differential checker (tests/fuzz_parity.py): random scheme / ring / chain /
level / batch / dispatcher switches / operation, every output word against the oracle."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [1, 2])
def test_random_cases_match_the_oracle(seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "--seconds", "25", "--seed", str(seed)],
                       capture_output=True, text=True, timeout=600)
    print(p.stdout[-2000:])
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    summary = json.loads(p.stdout.strip().splitlines()[-1])
    assert summary["cases"] >= 10
