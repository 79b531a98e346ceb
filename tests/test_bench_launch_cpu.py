"""`python bench.py --gpus N` must produce an N-rank measurement on its own and must never mislabel one
(VERDICT r2 item 1).  No device work here: --dry-launch runs the rank plumbing over gloo on the CPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_plain_gpus_2_starts_two_ranks_by_itself():
    # exactly how the driver invokes a run when no launcher wraps it: no WORLD_SIZE in the environment
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch"], env=_env(),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1, res.stdout  # rank 0 only
    line = lines[0]
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["ranks_counted"] == 2
    assert line["all_gather_ok"] is True and line["self_launched"] is True


def test_three_ranks_under_a_launcher_environment():
    # under torchrun the process IS one rank: emulate the launcher with three children carrying its variables
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, BENCH, "--gpus", "3", "--dry-launch"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=_env(WORLD_SIZE="3", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
             for r in range(3)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    lines = [l for o in outs for l in _json_lines(o[0])]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 3 and lines[0]["self_launched"] is False


def test_world_and_gpus_mismatch_is_refused():
    # the round-2 bug: WORLD_SIZE=1 with --gpus 8 ran ONE GPU and printed n_gpus 1
    for world, gpus in (("1", "8"), ("2", "1"), ("4", "2")):
        res = subprocess.run([sys.executable, BENCH, "--gpus", gpus, "--dry-launch"], env=_env(WORLD_SIZE=world, RANK="0", LOCAL_RANK="0"),
                             capture_output=True, text=True, timeout=120)
        assert res.returncode != 0 and "refusing" in res.stderr
        assert _json_lines(res.stdout) == []


def test_a_failing_rank_ends_the_whole_launch_with_its_code():
    # no GPU in this container: both children refuse to run, the parent relays a non-zero code and prints no line
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=_env(HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert _json_lines(res.stdout) == []


def test_two_ranks_rehearse_config5_strong_scaling_bookkeeping():
    # BASELINE.json configs[4]'s shape (200 000 features sharded over the ranks) through the SAME ShardedKlt objects and step loop as
    # bench.py --shard-total on GPUs, over gloo with N > 1: blocks, capacity, packed shards, two result slots,
    # the global kMaxTrackPointsNumber cutting into a block, unpacking in global order
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch", "--shard-total", "200000"], env=_env(),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _json_lines(res.stdout)[0]
    sh = line["sharded"]
    assert line["n_gpus"] == 2 and sh["total"] == 200000 and sh["capacity"] == 100000 and sh["packed_bytes"] == 900000
    assert sh["blocks_cover_the_list"] and sh["gathered_equals_unsharded"] and sh["all_ranks_agree"]
    # an odd total and three ranks: ragged blocks, padded capacity
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, BENCH, "--gpus", "3", "--steps", "2", "--dry-launch", "--shard-total", "100003"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=_env(WORLD_SIZE="3", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
             for r in range(3)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    sh = [l for o in outs for l in _json_lines(o[0])][0]["sharded"]
    assert sh["capacity"] == 33335 and sh["gathered_equals_unsharded"] and sh["all_ranks_agree"]


CONFIG5_KEYS = {"workload", "total_features", "features_per_rank", "packed_bytes_per_rank", "rccl_ranks", "backend", "steps", "torch", "native_comm"}
CONFIG5_TORCH_KEYS = {"ms_per_step", "features_per_s", "ms_per_step_plain_loop", "kernel_us", "all_gather_us", "graph", "gathered_equals_unsharded_bitwise",
                      "tracked_fraction", "what"}


def test_two_ranks_produce_the_config5_sharded_object_without_extra_flags():
    # `bench.py --gpus N` (N > 1) with the DRIVER's arguments must carry BASELINE configs[4] — 200 000 features sharded
    # over the ranks, one all-gather per step — in its one line.  The dry launch runs the same config5_sharded_leg() the GPU run calls
    # (step loop, slots, timing reductions, gathered == unsharded), over gloo around a stand-in tracker: schema and bookkeeping at 2 ranks.
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-launch"], env=_env(), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    c5 = _json_lines(res.stdout)[0]["config5_sharded"]
    assert set(c5) == CONFIG5_KEYS and set(c5["torch"]) == CONFIG5_TORCH_KEYS
    assert c5["total_features"] == 200000 and c5["features_per_rank"] == 100000 and c5["rccl_ranks"] == 2 and c5["packed_bytes_per_rank"] == 900000
    assert c5["torch"]["gathered_equals_unsharded_bitwise"] is True
    assert c5["torch"]["ms_per_step"] > 0 and c5["torch"]["kernel_us"] > 0 and c5["torch"]["all_gather_us"] > 0
    assert abs(c5["torch"]["features_per_s"] - 200000 / (c5["torch"]["ms_per_step"] * 1e-3)) < 1e-3 * c5["torch"]["features_per_s"]
