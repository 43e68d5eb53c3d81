"""A short randomized parity run (tools/fuzz_gpu.py; beside the seeded oracle sweep of test_gpu_fuzz.py): sizes, alignments and value distributions outside the fixed lists
of the other test files — both scan entry points, both digit widths of the sort, group-by, exclusive scan, reduce and
both join forms against numpy."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("seed", [11, 12])
def test_randomized_parity(seed):
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz_gpu.py"), "6", str(seed)], capture_output=True,
                       text=True, timeout=300, cwd=str(ROOT))
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
