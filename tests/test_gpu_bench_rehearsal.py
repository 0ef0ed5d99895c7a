"""The N > 1 path of bench.py on the one-GPU box: two ranks (gloo, sharing the GPU) shard the targets with shard_ranges, screen
their blocks through pcr_screen_device, all-gather the bitset words, and rank 0 checks the reassembled words and the coverage
against the SAME pairs screened on the unsharded set through pcr_select_words + pcr_amplify -- the check every N > 1 run of the
bench performs before its clock starts.  Weak (C2 blocks) and strong (one C5 set cut in two) modes.  Run with `-m gpu`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config,port", [("C2", 29541), ("C5", 29542)])
def test_two_rank_rehearsal_verifies_sharded_equals_unsharded(config, port):
    env = dict(os.environ, PCRAMP_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", config, "--scale", "0.04",
           "--steps", "24", "--warmup", "4", "--gather-every", "5"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # the JSON line is alone on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == ("strong" if config == "C5" else "weak")
    v = d["config"]["sharded_equals_unsharded"]
    assert v["mode"] == "whole set" and v["checked_targets"] == d["config"]["targets_total"] and v["amplification_calls_set"] > 0
    assert d["value"] > 0 and "REHEARSAL" in d["data"]
