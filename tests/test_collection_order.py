"""The driver runs `pytest tests -x -q -m gpu`: a failing bench-contract run or fuzz sweep must never be able to stop it in
front of a parity test (round 3 lost its whole parity run that way).  Checked here, on the CPU, by collecting the GPU suite:
every test that launches bench.py or a fuzz sweep is collected behind every other one."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_parity_tests_are_collected_before_fuzz_and_bench_tests():
    out = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests"), "--collect-only", "-q", "-m", "gpu"],
                         capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    files = [line.split("::", 1)[0] for line in out.stdout.splitlines() if "::" in line]
    assert len(files) > 150
    rank = lambda f: 2 if "bench" in f else (1 if "fuzz" in f else 0)
    ranks = [rank(f) for f in files]
    assert ranks == sorted(ranks), "a bench / fuzz test is collected in front of a parity test"
    assert ranks.count(2) >= 5 and ranks.count(1) >= 10 and ranks[0] == 0
