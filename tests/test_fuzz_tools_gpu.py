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
MODES = {"one_shard": {}, "three_sliced_shards": {"RFX_SHARDS": "3", "RFX_EXEC_SLICE_SHARDS": "1"}}


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("tool", FUZZERS)
def test_fuzzer_slice(built, tool, mode):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    lo = 1000 if mode == "one_shard" else 2000  # (seed ranges the builder's own long runs did not start from)
    env = dict(os.environ, **MODES[mode])
    env.pop("RFX_VALIDATE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool + ".py"), str(lo), str(lo + 30)], env=env, capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    tail = (p.stdout[-3000:] + p.stderr[-1500:])
    assert p.returncode == 0, tail
    m = re.search(r"done (\d+) seeds,(?: \d+ calls,)? (\d+) handed back.*?, (\d+) failures", p.stdout)
    assert m, tail
    assert int(m.group(1)) == 30 and int(m.group(3)) == 0, tail
    if tool != "fuzz_update_group":  # (update over shards is the host's own built-in there: handed back by design until joins / update run sharded)
        assert int(m.group(2)) == 0, tail
