"""bench.py launched both ways the multi-GPU measurement may launch it - plain `python bench.py --gpus 2` (the script
starts its own ranks) and under torch.distributed.run (one process per rank) - with two ranks sharing the one GPU of the test machine: MGEN_REHEARSE_ON_ONE_GPU=1 puts both on device 0 and routes the
gradient exchange through gloo (RCCL refuses two ranks on one device). Checks the control flow of the N > 1 path -
gradient hooks and buckets, barriers, the max-over-ranks time, one JSON line from rank 0 - not its speed.

The file name sorts before the other GPU tests on purpose: the child processes are started while this process has
not touched the GPU yet."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("launcher", ["plain", "torchrun"])
def test_two_rank_bench_prints_one_json_line(launcher):
    env = dict(os.environ, MGEN_REHEARSE_ON_ONE_GPU="1", MGEN_BENCH_WATCHDOG="150")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--seq",
            "512", "--no-cpu-baseline"]
    if launcher == "plain":
        cmd = [sys.executable] + tail
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port)] + tail
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and abs(out["value"] - 2 * 2 * 512 * 2 / (out["ms_per_step"] * 2e-3)) < 0.01 * out["value"]
    assert out["comm"]["world_size"] == 2 and out["comm"]["ranks_seen"] == 2
    assert out["roofline"]["achieved"] > 0
    assert 0 < out["final_loss"] < 13.0


def test_two_ranks_average_like_one_rank_on_the_whole_batch(tmp_path):
    import torch
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "grad.pt")
    env = dict(os.environ, MGEN_REHEARSE_ON_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "ddp_worker.py"), out]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    got = torch.load(out, weights_only=True)
    from mgen import MidiLM
    from mgen.config import LMConfig
    cfg = LMConfig(vocab_size=1000, hidden_size=256, intermediate_size=512, num_hidden_layers=3,
                   num_attention_heads=2, num_key_value_heads=2, head_dim=128)
    m = MidiLM(cfg, device="cuda:0", seed=0)
    ids = got["ids"].cuda()
    m.zero_grad()
    m(input_ids=ids, labels=ids).loss.backward()     # both rows, equal token counts: mean of the two ranks' losses
    want = m.flat_grad.cpu()
    err = float((got["grad"] - want).abs().max() / want.abs().max())
    assert err < 2e-2, err
    # bucket order of a step: the tied head's weight gradient first (final at the start of backward, a buffer of its
    # own), the layers, and last the embedding lookup's rows - only the rows [0, max id + 1) any rank can have touched
    b = got["buckets"].tolist()
    assert b[0] == cfg.vocab_size * cfg.hidden_size * 4
    assert b[-1] == (int(got["ids"].max()) + 1) * cfg.hidden_size * 4 and b[-1] < b[0] // 2
