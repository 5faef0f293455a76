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


def test_bench_gpus_2_starts_its_own_ranks(cuda):
    """`python3 bench.py --gpus 2` with NO launcher around it (the way the driver runs `--gpus 1`): the script itself starts the two
    ranks as a fresh child process group before it touches the GPU and relays rank 0's single line and exit code."""
    env = dict(os.environ, AS_BENCH_TEST_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-extras", "--cpu-utts", "0"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 6 and line["value"] > 0
    assert line["config"]["global_batch"] == 64


def test_two_ranks_weak_scaling_line(cuda):
    line = launch(["--steps", "6", "--warmup", "2", "--no-extras"])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 6
    assert line["config"]["global_batch"] == 64 and line["config"]["frames_per_step"] == 2 * 32 * 200
    assert line["value"] > 0 and abs(line["value"] - line["config"]["frames_per_step"] * 6 / (line["ms_per_step"] * 6e-3)) < 1e-6 * line["value"]


def test_two_ranks_c4_global_batch_sharded(cuda, tmp_path):
    """BASELINE config C4 at its stated size: ONE ragged batch of 256 utterances, sharded (here over two ranks sharing the one GPU; the
    8-GPU form is the driver's to launch), merged on rank 0.  Against rank 0 running all 256 alone -- and three of the merged
    utterances (shortest, median, longest) against the CPU oracle with the same forced durations: 1e-4, the north-star bound."""
    import numpy as np
    import torch
    import bench
    from artspeech_amd import synth
    from artspeech_amd.weights import DEFAULT_STATS, fold_state_dict, load_distribution
    from oracle import acoustic
    dump = str(tmp_path / "c4.npz")
    line = launch(["--steps", "3", "--warmup", "1", "--global-batch", "256", "--c4-dump", dump])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    c4 = line["c4_shard_check"]
    assert c4["utterances"] == 256 and c4["shards"] == 2
    assert c4["max_abs_sharded_vs_single_rank"] <= 5e-5, c4
    host, _ = bench.make_inputs(None, 256, vary=True)
    W = fold_state_dict(synth.synth_state_dict(512, 64, seed=bench.WEIGHT_SEED))
    dist = load_distribution(DEFAULT_STATS)
    z = np.load(dump)
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 16)))
    for i in z["idx"]:
        i = int(i)
        ref = acoustic.forward_test(W, torch.from_numpy(host["tokens"][i]), torch.from_numpy(host["mel"][i]), torch.from_numpy(host["f0"][i]),
                                    torch.from_numpy(host["ema"][i]), dist, forced_dur=host["forced"][i])["mel"].numpy()
        got = z[f"mel_{i}"]
        assert got.shape == ref.shape, (i, got.shape, ref.shape)
        d = float(np.abs(got - ref).max())
        assert d <= 1e-4, (i, d)
