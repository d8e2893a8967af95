"""bench.py's --gpus flag on a machine without GPUs: it must refuse to run fewer ranks than asked for (n_gpus in the
JSON line always equals --gpus) and must refuse a launcher whose WORLD_SIZE disagrees with the flag."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ, **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MGEN_REHEARSE_ON_ONE_GPU"):
        if k not in env:
            e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True,
                          text=True, timeout=120)


def test_world_size_must_agree_with_gpus_flag():
    r = _run(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_more_gpus_than_visible_is_an_error():
    import torch
    n = torch.cuda.device_count()
    r = _run(["--gpus", str(n + 2)])
    assert r.returncode != 0 and "visible" in r.stderr
    assert "{" not in r.stdout
