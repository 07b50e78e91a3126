"""The RCCL exchange of BASELINE config 4 on ONE GPU: a fresh child process brings up a world-size-1 `nccl` (= RCCL) group and runs
bench.Exchange (grouped async all_gather of int16 tokens / PSD / band features / keep mask + histogram all_reduce), the
gather_* / allreduce_hist helpers with their world-size-1 short cut switched off (equal, padded and rejection-thinned shards),
Scaler.fit(distributed=True) and the EMA step's all-reduce.  The world-2 gloo tests (tests/test_dist_gloo.py) cover rank ORDER on
CPU; this covers the RCCL branch itself (communicator, RCCL's stream against the compute stream, device buffers)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_exchange_world_size_one():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # a child process of its own (started, not exec'ed into): it makes its first GPU call after RCCL's environment is in place
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tests", "rccl_world1_child.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, f"child failed ({r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    assert lines, r.stdout[-2000:]
    res = json.loads(lines[-1][len("RESULT "):])
    bad = [k for k, v in res.items() if v is not True]
    assert not bad, f"failed checks: {bad} ({res})"
    assert len(res) >= 12
