"""GPU: the bench's work-list at 2^12 rows produces the same proof elements (the `digest` of the quotient's piece commitments and of the
evaluations at x) whatever the order of work: upstream's quotient order / 8 sub-cosets, interpreter / compiled evaluator, one stream."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--logn", "12", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", *flags],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["metric"] == "constraints/sec" and line["config"]["rows_per_step"] == 1 << 12
    return line["digest"]


@pytest.mark.gpu
def test_work_list_digest_is_independent_of_the_order_of_work():
    base = _run()
    assert len(base["h_commitments"]) == 16 and len(base["evals_at_x"]) == 16
    for flags in (("--quotient-parts", "8"), ("--quotient-parts", "2"), ("--expr-kernel", "never"), ("--expr-kernel", "always"), ("--serial",),
                  ("--expr-limbs", "32"), ("--ipa", "virtual")):
        assert _run(*flags) == base, flags
