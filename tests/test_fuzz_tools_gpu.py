"""A bounded slice of every round-5 fuzzer (tools/fuzz_*.py) inside the suite the driver runs: 30 seeds x {one shard, three sliced shards} each, against
the oracle, through rfx_select / the operators with the standalone host model.  tools/fuzz_null_tuples.py is what found the checksum defect of rounds 2-5
(profiles/r05_fuzz_extra.txt); a regression there must show in GPUTEST_r*.json, not only when somebody runs the tool."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FUZZERS = ["fuzz_null_tuples", "fuzz_select_extremes", "fuzz_update_group", "fuzz_operators"]
MODES = {"one_shard": {}, "three_sliced_shards": {"RFX_SHARDS": "3", "RFX_EXEC_SLICE_SHARDS": "1"},
         # round 6: the reproducible sums with two limbs answer within the fuzzers' own f64 tolerance (one limb does not: its cell is absolute)
         "two_limbs": {"RFX_DETERMINISTIC": "2"}, "two_limbs_three_sliced_shards": {"RFX_DETERMINISTIC": "2", "RFX_SHARDS": "3", "RFX_EXEC_SLICE_SHARDS": "1"}}


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("tool", FUZZERS)
def test_fuzzer_slice(built, tool, mode):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if mode.startswith("two_limbs") and tool in ("fuzz_operators", "fuzz_update_group"):
        pytest.skip("no grouped f64 sums through rfx_select in this tool")
    lo = {"one_shard": 1000, "three_sliced_shards": 2000, "two_limbs": 3000, "two_limbs_three_sliced_shards": 4000}[mode]  # (seed ranges the builder's own long runs did not start from)
    env = dict(os.environ)
    env.pop("RFX_VALIDATE", None)
    env.pop("RFX_DETERMINISTIC", None)
    env.update(MODES[mode])
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool + ".py"), str(lo), str(lo + 30)], env=env, capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    tail = (p.stdout[-3000:] + p.stderr[-1500:])
    assert p.returncode == 0, tail
    m = re.search(r"done (\d+) seeds,(?: \d+ calls,)? (\d+) handed back.*?, (\d+) failures", p.stdout)
    assert m, tail
    assert int(m.group(1)) == 30 and int(m.group(3)) == 0, tail
    if tool != "fuzz_update_group":  # (update over shards is the host's own built-in there: handed back by design until joins / update run sharded)
        assert int(m.group(2)) == 0, tail
