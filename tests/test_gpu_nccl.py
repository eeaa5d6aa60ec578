"""The sharded path under RCCL (backend "nccl") on real GPUs: world size 1 always, 2 when the box has two GPUs.
The gloo tests (tests/test_snapshot_parallel_gloo.py) cover N > 1 on CPU with the oracle injected; here the HIP kernels and
RCCL's collectives run: sharded forward must equal the unsharded HIP forward bit for bit (rows are independent sequences in
every kernel), gradients to fp32 summation order."""
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [1, 2])
def test_sharded_forward_and_backward_under_rccl(world):
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs" % world)
    from _nccl_worker import run
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(run, args=(world, _free_port(), results), nprocs=world, join=True)
    assert len(results) == world
    for rank in range(world):
        rep = results[rank]
        for exchange in ("all_to_all", "all_gather"):
            assert rep[exchange + "_fwd_bitwise"], (rank, exchange, rep)
            assert rep[exchange + "_train_fwd_bitwise"], (rank, exchange, rep)
            assert rep[exchange + "_gather_bitwise"], (rank, exchange, rep)
            assert rep[exchange + "_grad_rel_err"] <= (1e-6 if world == 1 else 2e-5), (rank, exchange, rep)


def test_bench_single_rank_sharded_path_matches_plain(tmp_path):
    """bench.py --gpus 1 through the RCCL code path (CTGCN_FORCE_DIST=1) reports the same workload and a step time close to the
    plain single-GPU run: the N = 1 point of a scaling curve is the BENCH number."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = []
    for force in ("0", "1"):
        env = dict(os.environ, CTGCN_FORCE_DIST=force, MASTER_PORT=str(_free_port()))
        detail = str(tmp_path / ("detail%s.json" % force))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "tiny", "--steps", "3", "--warmup", "1",
                              "--no-cpu-baseline", "--no-extras", "--no-pmc", "--detail-file", detail], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        json_lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(json_lines) == 1 and out.stdout.strip().splitlines()[-1] == json_lines[0], out.stdout[-500:]   # ONE line, and it is the last
        # the driver keeps ~9 KB of stdout: round 5's 23 KB line reached it truncated and unparsed
        assert len(json_lines[0]) < 4096, len(json_lines[0])
        compact = json.loads(json_lines[0])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                    "config", "roofline", "cpu_baseline", "detail_file"):
            assert key in compact, key
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "algorithmic_bytes_per_launch"):
            assert key in compact["roofline"], key
        full = json.load(open(detail))
        assert full["ms_per_step"] == compact["ms_per_step"] and full["value"] == compact["value"]
        lines.append(full)
    a, b = lines
    assert a["config"]["aggregated_edges_per_step"] == b["config"]["aggregated_edges_per_step"]
    assert a["config"]["K_per_snapshot"] == b["config"]["K_per_snapshot"]
    assert a["n_gpus"] == b["n_gpus"] == 1
