"""`python bench.py --gpus N` must run N ranks (reference launcher: tools/dist_train.sh:9-19 spawns
--nproc_per_node=$GPUS).  CPU / gloo dry run of the launcher path: self-spawn under torch.distributed.run, rendezvous on
127.0.0.1, flat-bucket exchange overlapped with backward, barrier + max-over-ranks timing, ONE JSON line with n_gpus = N."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-launcher", "--steps", "3", "--warmup", "1",
                          *extra], capture_output=True, text=True, timeout=300, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout          # exactly one JSON line, printed by rank 0
    return json.loads(lines[0])


def test_gpus_flag_self_spawns_n_ranks():
    rec = _run(["--gpus", "2"])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["exchange"]["buckets"] >= 2 and rec["exchange"]["bytes_per_step"] > 0
    assert rec["exchange"]["mode"] == "all_reduce"        # gloo has no reduce-scatter: RCCL runs rs_ag


def test_single_process_default():
    rec = _run([])
    assert rec["n_gpus"] == 1 and rec["exchange"]["bytes_per_step"] == 0
