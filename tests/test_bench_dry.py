"""bench.py's multi-rank plumbing without a GPU (`--dry`: gloo on the CPU, no kernels): the self-launcher, the driver's launcher form,
the LPT assignment, the per-rank stats gather, barriers + max-over-ranks timing and the ONE JSON line of rank 0."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _check(stdout, world):
    lines = [l for l in stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # exactly one line on stdout
    assert len(lines[0]) < 8192, len(lines[0])         # the driver's record keeps ~9 KB of stdout (a 23 KB line did not parse)
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["dry"] is True and line["steps"] == 3 and line["warmup"] == 1
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in line, key
    owned = sorted(t for r in line["per_rank"] for t in r["snapshots"])
    assert owned == list(range(line["config"]["snapshots"]))          # every snapshot on exactly one rank
    assert [r["rank"] for r in line["per_rank"]] == list(range(world))
    assert line["config"]["assignment"] == [r["snapshots"] for r in line["per_rank"]]


def test_compact_record_of_a_full_run_fits_the_driver():
    """bench.compact_record on the largest record this script has produced (round 5's 23 KB line, kept under profiles/): the stdout line
    stays under bench.MAX_LINE_BYTES and still carries the contract's keys, `roofline` and `cpu_baseline`."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))
    full["per_rank_ms"] = [{"snapshot_branches_ms": 20.123, "exchange_exposed_ms": 1.5, "temporal_head_ms": 2.25, "rank": r} for r in range(8)]
    rec = bench.compact_record(full, os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    text = json.dumps(rec, separators=(",", ":"))
    assert len(text) < bench.MAX_LINE_BYTES <= 4096, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert rec[key] == full[key], key
    assert rec["roofline"]["bound"] in ("hbm", "mfma") and rec["roofline"]["frac"] == full["roofline"]["frac"]
    assert rec["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"] and rec["cpu_baseline"]["kind"] in ("port", "reference")
    assert rec["config"]["name"] == "synthetic-1m" and "model" not in rec["config"]
    assert rec["detail_file"] == os.path.join("gpurun_out", "bench_detail.json")


def test_bench_self_launch_two_ranks_dry():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--workload", "tiny", "--steps", "3", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    _check(p.stdout, 2)


def test_bench_under_the_drivers_launcher_dry():
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ..."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry", "--workload", "enron-like", "--steps", "3", "--warmup", "1"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    _check(p.stdout, 3)
