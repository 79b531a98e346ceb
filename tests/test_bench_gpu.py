"""bench.py itself on the GPU box, in the shapes the driver runs it — so that the driver's suite exercises what its bench run will meet.

* FTK_BENCH_FORCE_DIST=1 at world size 1 takes the N > 1 code path: process group on RCCL, one all-gather per step, the K steps captured
  in one HIP graph with the gather on a side stream, and the `config5_sharded` object (BASELINE configs[4]: features sharded over the
  ranks, torch.distributed AND the C ABI's own ncclAllGather) — VERDICT r4 item 3.
* the default N = 1 line carries `roofline`, `cpu_baseline` and the `real_images` rows (the reference's example pair), every row
  bit-identical to the oracle — VERDICT r4 item 1.
The bench process is a CHILD of the test process (one GPU process at a time beside pytest itself)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    res = subprocess.run([sys.executable, BENCH, *argv], env=e, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return lines[0]


def test_forced_dist_run_carries_config5_sharded_through_rccl():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    line = _run(["--steps", "5", "--warmup", "2", "--config5-total", "40000"], FTK_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["scaling"] == "weak"
    assert "all-gather" in line["config"]["parallelism"]
    assert line["parity"]["bit_identical"] and line["parity"]["last_launch_bit_identical"]
    c5 = line["config5_sharded"]
    assert "error" not in c5, c5
    assert c5["backend"] == "nccl" and c5["rccl_ranks"] == 1 and c5["total_features"] == 40000 and c5["features_per_rank"] == 40000
    t = c5["torch"]
    assert t["gathered_equals_unsharded_bitwise"] is True and t["ms_per_step"] > 0 and t["all_gather_us"] > 0 and t["kernel_us"] > 0
    nat = c5["native_comm"]
    assert "error" not in nat and "skipped" not in nat, nat
    assert nat["gathered_equals_unsharded_bitwise"] is True and nat["rccl_ranks"] == 1 and nat["ms_per_step"] > 0


def test_default_line_carries_roofline_cpu_baseline_and_real_image_rows():
    line = _run(["--steps", "20", "--warmup", "5", "--no-configs-leg", "--no-tree-leg", "--no-upload-leg"])
    assert line["n_gpus"] == 1 and line["parity"]["bit_identical"]
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
    real = line["real_images"]
    assert "skipped" not in real, real
    assert len(real["rows"]) == 18 and real["all_bit_identical"], {k: v["bit_identical"] for k, v in real["rows"].items()}
    for key, row in real["rows"].items():
        assert row["ms_per_step"] > 0 and row["host_call_ms"] > row["ms_per_step"] * 0.5, (key, row)
    assert "config5_sharded" not in line  # N = 1 without the forced process group: no collective anywhere
