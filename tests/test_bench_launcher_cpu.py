"""bench.py's own launcher: `python bench.py --gpus N` with no torchrun around it must start N ranks itself.  There is no GPU in the
build container, so "started" means: every rank got as far as the GPU check and says which rank of how many it is."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env["HIP_VISIBLE_DEVICES"] = ""      # also on a GPU box this test exercises the launcher only
    env["CUDA_VISIBLE_DEVICES"] = ""
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=timeout)


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode != 0
    assert "exec -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1" in r.stderr
    # the first rank to fail takes the job down (torchrun SIGTERMs the other), so at least one rank reports; both name the world size
    assert "of 2: bench.py needs a GPU" in r.stderr, r.stderr[-2000:]
    assert r.stdout.strip() == ""        # no JSON line from a run that did not measure anything


def test_gpus_1_does_not_relaunch():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "1"])
    assert r.returncode != 0
    assert "torch.distributed.run" not in r.stderr
    assert "rank 0 of 1: bench.py needs a GPU" in r.stderr


def test_world_size_mismatch_is_still_an_error():
    r = _run(["--gpus", "4", "--steps", "2"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0
    assert "--gpus 4 but WORLD_SIZE=2" in r.stderr
