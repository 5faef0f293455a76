"""GPU: the multi-rank launch path of bench.py (utterance batches shard across ranks with no data-path
collective, models.py:361-362 processes one utterance at a time) on a ONE-GPU box: two ranks started by torch.distributed.run as
fresh child processes, both on GPU 0 (AS_BENCH_TEST_ONE_GPU=1: gloo for the barrier and the max-over-ranks).  Checks the JSON line
of the weak-scaling launch and C4's sharded-vs-single-rank result."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(extra, timeout=900):
    env = dict(os.environ, AS_BENCH_TEST_ONE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-utts", "0"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                             # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_ranks_weak_scaling_line(cuda):
    line = launch(["--steps", "6", "--warmup", "2", "--no-extras"])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 6
    assert line["config"]["global_batch"] == 64 and line["config"]["frames_per_step"] == 2 * 32 * 200
    assert line["value"] > 0 and abs(line["value"] - line["config"]["frames_per_step"] * 6 / (line["ms_per_step"] * 6e-3)) < 1e-6 * line["value"]


def test_two_ranks_c4_global_batch_sharded(cuda):
    line = launch(["--steps", "3", "--warmup", "1", "--global-batch", "64"])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    c4 = line["c4_shard_check"]
    assert c4["utterances"] == 64 and c4["shards"] == 2
    assert c4["max_abs_sharded_vs_single_rank"] <= 5e-5, c4
