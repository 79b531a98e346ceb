"""Host-side logic that needs no GPU: input normalisation, bit packing, feature sharding and the
world_size-2 result exchange over gloo."""
import os
import socket
import subprocess
import sys

import numpy as np

from feature_tracker_amd import dist as FD
from feature_tracker_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_early_returns_need_no_device(ftk):
    klt = ftk.OpticalFlowAffineKlt()
    ok, c, s = klt.TrackFeatures(np.zeros((4, 4), np.uint8), np.zeros((4, 4), np.uint8), np.zeros((0, 2), np.float32))
    assert ok is False  # optical_flow.cpp:30
    assert klt.OpticalFlowMethodName() == "Affine-Klt" and ftk.OpticalFlowLssdKlt().OpticalFlowMethodName() == "Lssd-Klt"
    m = ftk.BriefMatcher()
    assert m.options().kMaxValidDescriptorDistance == 0.0 and m.options().kMaxValidPredictRowDistance == 40
    ok, _ = m.ForceMatch(np.zeros((3, 8), np.uint8), np.zeros((0, 8), np.uint8))
    assert ok is False  # descriptor_matcher.h:58
    ok, _ = m.NearbyMatch(np.zeros((3, 8), np.uint8), np.zeros((2, 8), np.uint8), np.zeros((2, 2)), np.zeros((2, 2)))
    assert ok is False  # descriptor_matcher.h:95


def test_pack_brief_layout(ftk):
    rs = np.random.RandomState(1)
    for n_bits in (1, 31, 32, 33, 200, 256):
        bits = rs.randint(0, 2, size=(17, n_bits)).astype(np.uint8)
        words = ftk.pack_brief(bits)
        assert words.shape == (17, (n_bits + 31) // 32) and words.dtype == np.uint32
        assert np.array_equal(words, synth.pack_bits(bits))
        for i in (0, 16):
            for b in range(n_bits):
                assert (int(words[i, b // 32]) >> (b % 32)) & 1 == bits[i, b]
        # Hamming distance survives the packing
        d_bits = int((bits[0] != bits[1]).sum())
        d_words = sum(bin(int(a) ^ int(b)).count("1") for a, b in zip(words[0], words[1]))
        assert d_bits == d_words
    assert ftk.pack_brief(np.zeros((5, 0), np.uint8)).shape == (5, 1)


def test_shard_bounds_partition():
    for n in (0, 1, 7, 2000, 200000, 200003):
        for world in (1, 2, 3, 8):
            spans = [FD.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) <= FD.shard_capacity(n, world) if n else True


def test_pack_unpack_roundtrip():
    import torch
    n, world = 37, 4
    cap = FD.shard_capacity(n, world)
    uv = torch.arange(n * 2, dtype=torch.float32).view(n, 2)
    st = (torch.arange(n) % 5).to(torch.uint8)
    chunks = []
    for r in range(world):
        b, e = FD.shard_bounds(n, world, r)
        buf = torch.zeros(FD.packed_bytes(cap), dtype=torch.uint8)
        puv, pst = FD.pack_views(buf, cap)
        puv[: e - b] = uv[b:e]
        pst[: e - b] = st[b:e]
        chunks.append(buf)
    guv, gst = FD.unpack_gathered(torch.cat(chunks), n, world)
    assert torch.equal(guv, uv) and torch.equal(gst, st)


_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["FTK_ROOT"])
import numpy as np, torch, torch.distributed as dist
from feature_tracker_amd import dist as FD, synth
from tests import oracle_lib, scenes
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
ref_levels, cur_levels = scenes.scene(160, 120, 2)
n = 101
uv = scenes.features(n, 160, 120, half=4)
b, e = FD.shard_bounds(n, world, rank)
cap = FD.shard_capacity(n, world)
# stand-in for the device kernel on this rank's shard: the oracle (this is a test of the sharding / exchange path)
class OracleTracker:
    max_track_points = n
    def track(self, ref_uv, cur_in, st_in, cur_out, st_out, iters, max_track_points=None):
        cap = self.max_track_points if max_track_points is None else max_track_points
        ok, c, st, it = oracle_lib.klt_track_pyramid("basic", ref_levels, cur_levels, ref_uv.numpy(), cur_in.numpy(), st_in.numpy(),
                                                     method="fast", half=4, max_points=cap)
        cur_out.copy_(torch.from_numpy(c)); st_out.copy_(torch.from_numpy(st))
sharded = FD.ShardedKlt(OracleTracker(), n, "cpu", world, rank)
assert (sharded.begin, sharded.end) == (b, e)
t_uv = torch.from_numpy(uv)
guv, gst = sharded.track(t_uv, t_uv.clone(), torch.zeros(n, dtype=torch.uint8))
ok, c_all, st_all, _ = oracle_lib.klt_track_pyramid("basic", ref_levels, cur_levels, uv, method="fast", half=4, max_points=n)
assert np.array_equal(guv.numpy().view(np.uint32), c_all.view(np.uint32)), "gathered uv differs from the unsharded run"
assert np.array_equal(gst.numpy(), st_all)
# kMaxTrackPointsNumber is a GLOBAL cap (basic_klt.cpp:9): with cap < n only features [0, cap) are tracked, whichever rank holds them
for cap in (0, 30, 60, 100):
    capped = OracleTracker(); capped.max_track_points = cap
    guv, gst = FD.ShardedKlt(capped, n, "cpu", world, rank).track(t_uv, t_uv.clone(), torch.zeros(n, dtype=torch.uint8))
    ok, c_cap, st_cap, _ = oracle_lib.klt_track_pyramid("basic", ref_levels, cur_levels, uv, method="fast", half=4, max_points=cap)
    assert np.array_equal(guv.numpy().view(np.uint32), c_cap.view(np.uint32)) and np.array_equal(gst.numpy(), st_cap), f"global cap {cap}"
    assert (gst.numpy()[cap:] == 0).all() and np.array_equal(guv.numpy()[cap:], uv[cap:])
# descriptor matcher: ref rows sharded, candidates replicated, one all-gather of the index shards
ref_bits, cur_bits, _ = synth.make_descriptors(53, 40, n_bits=64, flips=5)
fref, fcur, _ = synth.make_float_descriptors(53, 40, dim=32)
rs = np.random.RandomState(3)
cur_uv = rs.uniform(0, 100, (40, 2)).astype(np.float32); pred_uv = rs.uniform(0, 100, (53, 2)).astype(np.float32)
def hamming(ref_rows, cur_rows, pred, cuv, idx):
    if pred is None:
        ok, out = oracle_lib.force_match(ref_rows.numpy(), cur_rows.numpy(), 20.0, idx.numpy())
    else:
        ok, out = oracle_lib.nearby_match(ref_rows.numpy(), cur_rows.numpy(), pred.numpy(), cuv.numpy(), 20.0, 60, 60, idx.numpy())
    idx.copy_(torch.from_numpy(out))
def cosine(ref_rows, cur_rows, pred, cuv, idx):
    ok, out = oracle_lib.match_float(ref_rows.numpy(), cur_rows.numpy(), 0.3, None if pred is None else pred.numpy(), None if cuv is None else cuv.numpy(),
                                     60, 60, idx.numpy())
    idx.copy_(torch.from_numpy(out))
stale = torch.arange(53, dtype=torch.int32) + 500
sm = FD.ShardedMatcher(hamming, 53, "cpu", world, rank)
got = sm.match_all(torch.from_numpy(ref_bits), torch.from_numpy(cur_bits), index_pairs=stale)
ok, want = oracle_lib.force_match(ref_bits, cur_bits, 20.0, stale.numpy())
assert np.array_equal(got.numpy(), want), "sharded ForceMatch differs from the unsharded run"
got = sm.match_all(torch.from_numpy(ref_bits), torch.from_numpy(cur_bits), torch.from_numpy(pred_uv), torch.from_numpy(cur_uv))
ok, want = oracle_lib.nearby_match(ref_bits, cur_bits, pred_uv, cur_uv, 20.0, 60, 60)
assert np.array_equal(got.numpy(), want), "sharded NearbyMatch differs from the unsharded run"
sc = FD.ShardedMatcher(cosine, 53, "cpu", world, rank)
got = sc.match_all(torch.from_numpy(fref), torch.from_numpy(fcur))
ok, want = oracle_lib.match_float(fref, fcur, 0.3)
assert np.array_equal(got.numpy(), want) and (want >= 0).sum() > 10, "sharded cosine ForceMatch differs from the unsharded run"
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_gloo_exchange(tmp_path):
    """N > 1 path on CPU: features (tracker) and reference rows (matchers) sharded over 2 ranks, one all-gather
    of the result shards, results identical to the unsharded runs."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FTK_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {rank} ok" in out


def test_python_quaternion_helpers_match_oracle(oracle):
    """The host-side algebra of DirectMethod's world-frame overload (tracker._quat_*) is the oracle's, bit for bit."""
    from feature_tracker_amd import tracker as T
    rs = np.random.RandomState(5)
    for _ in range(200):
        a = rs.standard_normal(4).astype(np.float32)
        b = rs.standard_normal(4).astype(np.float32)
        v = (rs.standard_normal(3) * 10).astype(np.float32)
        assert np.array_equal(T._quat_mul(a, b).view(np.uint32), oracle.quat_mul(a, b).view(np.uint32))
        assert np.array_equal(T._quat_rotate(a, v).view(np.uint32), oracle.quat_rotate(a, v).view(np.uint32))
        assert np.array_equal(T._quat_inverse(a).view(np.uint32), oracle.quat_inverse(a).view(np.uint32))
    assert np.array_equal(T._quat_inverse(np.zeros(4, np.float32)), np.zeros(4, np.float32))


def test_direct_method_early_returns_need_no_device(ftk):
    dm = ftk.DirectMethod()
    assert dm.options().kMaxConvergeStep == 1e-6 and dm.options().kMethod == "direct" and dm.options().kMaxTrackPointsNumber == 500

    class FakePyramid:  # level() is all the early returns look at
        def __init__(self, n):
            self._n = n

        def level(self):
            return self._n
    ok, *_ = dm.TrackFeatures(FakePyramid(3), FakePyramid(3), [1, 1, 0, 0], np.zeros((0, 3)), np.zeros((0, 2), np.float32))
    assert ok is False  # direct_method_tracker.cpp:38
    ok, *_ = dm.TrackFeatures(FakePyramid(3), FakePyramid(4), [1, 1, 0, 0], np.zeros((2, 3)), np.zeros((2, 2), np.float32))
    assert ok is False  # :39
    m = ftk.CosineMatcher()
    ok, _ = m.ForceMatch(np.zeros((3, 8), np.float32), np.zeros((0, 8), np.float32))
    assert ok is False  # descriptor_matcher.h:58


def test_argument_errors_are_raised_before_any_device_call(ftk):
    import pytest

    class FakePyramid:
        def level(self):
            return 2
    m = ftk.CosineMatcher()
    with pytest.raises(ValueError):
        m.ForceMatch(np.zeros((3, 8), np.float32), np.zeros((4, 16), np.float32))  # descriptor lengths differ
    dm = ftk.DirectMethod()
    with pytest.raises(ValueError):
        dm.TrackFeatures(FakePyramid(), FakePyramid(), [1, 1, 0, 0], np.zeros((2, 3), np.float32), np.zeros((5, 2), np.float32))  # fewer points than features
    ok, idx = ftk.CosineMatcher().NearbyMatch(np.zeros((3, 8), np.float32), np.zeros((2, 8), np.float32), np.zeros((2, 2)), np.zeros((2, 2)))
    assert ok is False  # descriptor_matcher.h:95 — pred size != ref size


# ---- the C++ layer's multi-GPU rendezvous (host/src/device_runtime.cpp, SharedComm): no device needed ----

COMM_CLI = os.path.join(ROOT, "feature_tracker_amd", "host", "build", "comm_id_cli")
LAUNCHER_VARS = ("FTK_WORLD_SIZE", "FTK_RANK", "FTK_COMM_ID_FILE", "FTK_COMM_NONCE", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID")


def _comm_cli(args, timeout=30, **env):
    assert os.path.exists(COMM_CLI), "host layer not built (python -c 'import __graft_entry__ as g; g.build()')"
    base = {k: v for k, v in os.environ.items() if k not in LAUNCHER_VARS}
    return subprocess.run([COMM_CLI, *args], capture_output=True, text=True, timeout=timeout, env=dict(base, **env))


def test_sharded_mode_needs_an_explicit_ftk_opt_in():
    # a torchrun-style environment alone (data-parallel job: every rank tracks DIFFERENT frames) must NOT switch the trackers to
    # the sharded path, nor make them fail
    assert _comm_cli(["optin"], WORLD_SIZE="8", RANK="3", LOCAL_RANK="3", MASTER_PORT="29500").stdout.strip() == "off"
    assert _comm_cli(["optin"]).stdout.strip() == "off"
    assert _comm_cli(["optin"], FTK_WORLD_SIZE="1").stdout.strip() == "off"
    # FTK_WORLD_SIZE > 1 is explicit but incomplete without the id file
    assert _comm_cli(["optin"], FTK_WORLD_SIZE="2", FTK_RANK="1").stdout.startswith("error ")
    # the id file is the opt-in; launcher variables then serve as defaults, FTK_* win
    assert _comm_cli(["optin"], FTK_COMM_ID_FILE="/tmp/x", WORLD_SIZE="8", RANK="3").stdout.strip() == "on 3 8"
    assert _comm_cli(["optin"], FTK_COMM_ID_FILE="/tmp/x", WORLD_SIZE="8", RANK="3", FTK_WORLD_SIZE="2", FTK_RANK="1").stdout.strip() == "on 1 2"
    assert _comm_cli(["optin"], FTK_COMM_ID_FILE="/tmp/x").stdout.strip() == "on 0 1"
    assert _comm_cli(["optin"], FTK_COMM_ID_FILE="/tmp/x", FTK_WORLD_SIZE="2", FTK_RANK="2").stdout.startswith("error ")


def test_stale_id_file_is_not_mistaken_for_this_launch(tmp_path):
    path = str(tmp_path / "rccl_id.bin")
    # an earlier launch (nonce A) left its file behind; a reader of launch B must not take it ...
    assert _comm_cli(["publish", path, "17"], FTK_COMM_NONCE="A").returncode == 0
    late = _comm_cli(["await", path, "300"], FTK_COMM_NONCE="B")
    assert late.returncode == 2 and "another launch" in late.stderr
    # ... nor a round-2 style file (the bare 128 bytes), nor a truncated one
    with open(path, "wb") as f:
        f.write(bytes(128))
    assert _comm_cli(["await", path, "200"], FTK_COMM_NONCE="B").returncode == 2
    # launch B's rank 0 replaces the file while a reader is already waiting: the reader gets B's id, not A's
    assert _comm_cli(["publish", path, "17"], FTK_COMM_NONCE="A").returncode == 0
    base = {k: v for k, v in os.environ.items() if k not in LAUNCHER_VARS}
    reader = subprocess.Popen([COMM_CLI, "await", path, "20000"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(base, FTK_COMM_NONCE="B"))
    import time
    time.sleep(0.5)
    assert reader.poll() is None  # still waiting: A's file does not satisfy it
    assert _comm_cli(["publish", path, "42"], FTK_COMM_NONCE="B").returncode == 0
    out, err = reader.communicate(timeout=30)
    assert reader.returncode == 0 and out.strip() == "42", err
    # the nonce falls back to what torchrun exports — which is the SAME on every default launch ("none", 29500: ADVICE r3), so
    # under a launcher (LOCAL_RANK / TORCHELASTIC_RUN_ID present) the launching process's identity (pid @ start time) and the
    # elastic restart count are part of it: equal for the ranks of one launch, different for the next launch
    assert _comm_cli(["nonce"], MASTER_PORT="29512").stdout.strip() == "MASTER_PORT=29512"
    assert _comm_cli(["nonce"], MASTER_PORT="29512", FTK_COMM_NONCE="n").stdout.strip() == "FTK_COMM_NONCE=n"
    assert _comm_cli(["nonce"]).stdout.strip() == ""
    mine = _comm_cli(["nonce"], MASTER_PORT="29500", TORCHELASTIC_RUN_ID="none", LOCAL_RANK="1").stdout.strip()
    assert mine.startswith("TORCHELASTIC_RUN_ID=none;parent=%d@" % os.getpid()), mine
    assert _comm_cli(["nonce"], MASTER_PORT="29500", TORCHELASTIC_RUN_ID="none", LOCAL_RANK="0").stdout.strip() == mine  # a sibling rank
    assert _comm_cli(["nonce"], MASTER_PORT="29500", LOCAL_RANK="0").stdout.strip().startswith("MASTER_PORT=29500;parent=%d@" % os.getpid())
    restarted = _comm_cli(["nonce"], MASTER_PORT="29500", TORCHELASTIC_RUN_ID="none", LOCAL_RANK="1", TORCHELASTIC_RESTART_COUNT="1").stdout.strip()
    assert restarted == mine + ";restart=1"
    # a per-rank WRAPPER in between (it was started with the rank's LOCAL_RANK: `torchrun --no-python wrapper.sh`, a per-rank
    # `rocprofv3 -- python3 ...`) is looked through — the launcher is the nearest ancestor that was not itself started as a rank, the
    # same for every rank (taking the parent gave every wrapped rank its own nonce and a 120 s time-out)
    wrapped = subprocess.run(["sh", "-c", COMM_CLI + " nonce; true"], capture_output=True, text=True, timeout=30,
                             env=dict(base, MASTER_PORT="29500", TORCHELASTIC_RUN_ID="none", LOCAL_RANK="1")).stdout.strip()
    assert wrapped == mine
    # the same launcher variables under ANOTHER launching process (a shell that was not started as a rank) give another nonce
    other = subprocess.run(["sh", "-c", "LOCAL_RANK=1 " + COMM_CLI + " nonce; true"], capture_output=True, text=True, timeout=30,
                           env=dict(base, MASTER_PORT="29500", TORCHELASTIC_RUN_ID="none")).stdout.strip()
    assert other.startswith("TORCHELASTIC_RUN_ID=none;parent=") and other != mine
    # a nonce longer than the id file's 64-byte field (a UUID run id + parent + restart) is stored as head + hash of the WHOLE string:
    # two launches that differ only behind the 63rd character are still told apart, and the time-out names both nonces
    long_a, long_b = "u" * 70 + ";restart=0", "u" * 70 + ";restart=1"
    assert _comm_cli(["publish", path, "17"], FTK_COMM_NONCE=long_a).returncode == 0
    miss = _comm_cli(["await", path, "300"], FTK_COMM_NONCE=long_b)
    assert miss.returncode == 2 and "expects the nonce" in miss.stderr and "FTK_COMM_NONCE" in miss.stderr
    hit = _comm_cli(["await", path, "300"], FTK_COMM_NONCE=long_a)
    assert hit.returncode == 0 and hit.stdout.strip() == "17"
    # and whatever its nonce, a file written long before this process started is a leftover (ranks start within two minutes)
    assert _comm_cli(["publish", path, "99"], FTK_COMM_NONCE="B").returncode == 0
    old = time.time() - 600
    os.utime(path, (old, old))
    aged = _comm_cli(["await", path, "300"], FTK_COMM_NONCE="B")
    assert aged.returncode == 2, aged.stdout
    assert _comm_cli(["publish", path, "42"], FTK_COMM_NONCE="B").returncode == 0
    # no temporary files are left next to the id file
    assert sorted(os.listdir(tmp_path)) == ["rccl_id.bin"]


def test_vector_bool_word_conversions_match_the_per_bit_definition():
    """host/compat/bit_words.h moves std::vector<bool> descriptors a word at a time (libstdc++ internals on a little-endian host,
    per-bit loops elsewhere): every length 0..300, random contents, dirty padding — against the per-bit definition, on the CPU."""
    exe = os.path.join(ROOT, "feature_tracker_amd", "host", "build", "bit_words_selftest")
    assert os.path.exists(exe), "host layer not built (python -c 'import __graft_entry__ as g; g.build()')"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
