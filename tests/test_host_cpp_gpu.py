"""GPU tests of the C++ host layer (feature_tracker_amd/host): the reference's class names driven
through small CLI programs, compared with the oracle bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

from feature_tracker_amd import synth
from tests import scenes

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "feature_tracker_amd", "host", "build")
DATA = os.path.join(ROOT, "tests", "data", "optical_flow")
METHOD_ID = {"inverse": 0, "direct": 1, "fast": 2}


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, np.uint8).tobytes())


def write_features(path, uv, pred=None, status=None):
    with open(path, "w") as f:
        for i in range(len(uv)):
            line = f"{float(uv[i, 0]).hex()} {float(uv[i, 1]).hex()}"
            if pred is not None:
                line += f" {float(pred[i, 0]).hex()} {float(pred[i, 1]).hex()} {int(status[i])}"
            f.write(line + "\n")


def run_track_cli(tmp_path, model, method, levels, half, ref, cur, uv, pred=None, status=None, max_points=100000, prior=None, lum=False, env=None):
    exe = os.path.join(BUILD, "track_cli")
    assert os.path.exists(exe), "host layer not built (python -c 'import __graft_entry__ as g; g.build()')"
    write_pgm(tmp_path / "ref.pgm", ref)
    write_pgm(tmp_path / "cur.pgm", cur)
    write_features(tmp_path / "f.txt", uv, pred, status)
    cmd = [exe, model, str(METHOD_ID[method]), str(levels), str(half), str(half), str(tmp_path / "ref.pgm"), str(tmp_path / "cur.pgm"),
           str(tmp_path / "f.txt"), str(max_points)]
    pr = np.eye(2, dtype=np.float32) if prior is None else np.asarray(prior, np.float32)
    cmd += [float(x).hex() for x in pr.reshape(4)] + [str(int(lum))]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=None if env is None else dict(os.environ, **env))
    assert res.returncode == 0, res.stdout + res.stderr
    lines = res.stdout.strip().splitlines()
    while lines and not lines[0].startswith("ok "):  # RCCL prints a version banner on stdout when a communicator is made
        lines.pop(0)
    assert lines and lines[0].startswith("ok 1"), res.stdout[:500]
    rows = [l.split() for l in lines[1:]]
    uvb = np.array([[int(r[0], 16), int(r[1], 16)] for r in rows], dtype=np.uint32)
    return uvb.view(np.float32), np.array([int(r[2]) for r in rows], np.uint8), np.array([int(r[3]) for r in rows], np.uint32), lines[0]


@pytest.mark.parametrize("model", ["basic", "affine", "lssd"])
@pytest.mark.parametrize("method", ["inverse", "direct", "fast"])
def test_cpp_pyramid_tracking_matches_oracle(tmp_path, oracle, model, method):
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "easy", "similarity")
    uv = scenes.features(150, 320, 240, half=5)
    c, s, it, head = run_track_cli(tmp_path, model, method, 3, 5, ref_levels[0], cur_levels[0], uv)
    ok, oc, os_, oit = oracle.klt_track_pyramid(model, ref_levels, cur_levels, uv, method=method, half=5, max_points=100000)
    assert np.array_equal(s, os_)
    assert np.array_equal(c.view(np.uint32), oc.view(np.uint32))
    assert np.array_equal(it, oit)
    assert {"basic": "Basic-Klt", "affine": "Affine-Klt", "lssd": "Lssd-Klt"}[model] in head


@pytest.mark.parametrize("model", ["basic", "affine", "lssd"])
def test_cpp_single_image_overload_with_prediction(tmp_path, oracle, model):
    ref_levels, cur_levels = scenes.scene(320, 240, 1, "easy", "similarity")
    uv = scenes.features(100, 320, 240, half=6)
    pred = uv + np.float32([2.0, -1.0])
    status = (np.arange(100) % 4).astype(np.uint8)
    prior = np.float32([[1.01, 0.02], [-0.02, 0.99]])
    c, s, it, _ = run_track_cli(tmp_path, model, "fast", 0, 6, ref_levels[0], cur_levels[0], uv, pred, status, max_points=90, prior=prior, lum=True)
    ok, oc, os_, oit = oracle.klt_track_single(model, ref_levels[0], cur_levels[0], uv, pred, status, prior=prior, consider_luminance=True,
                                               method="fast", half=6, max_points=90)
    assert np.array_equal(s, os_) and np.array_equal(c.view(np.uint32), oc.view(np.uint32))


def write_descriptors(path, ref, cur, ref_uv, cur_uv):
    with open(path, "w") as f:
        f.write(f"{len(ref)} {len(cur)} {ref.shape[1]}\n")
        for d, uv in ((ref, ref_uv), (cur, cur_uv)):
            for i in range(len(d)):
                bits = "".join("1" if b else "0" for b in d[i]) or "-"
                f.write(f"{bits} {float(uv[i, 0]).hex()} {float(uv[i, 1]).hex()}\n")


@pytest.mark.parametrize("mode", ["force", "nearby"])
def test_cpp_brief_matcher_matches_oracle(tmp_path, oracle, mode):
    ref, cur, perm = synth.make_descriptors(300, 420, flips=20)
    rs = np.random.RandomState(4)
    cur_uv = np.stack([rs.uniform(0, 640, 420), rs.uniform(0, 480, 420)], axis=1).astype(np.float32)
    ref_uv = np.stack([rs.uniform(0, 640, 300), rs.uniform(0, 480, 300)], axis=1).astype(np.float32)
    ref_uv[perm[:300] % 300] = cur_uv[:300] + 5.0
    write_descriptors(tmp_path / "d.txt", ref, cur, ref_uv, cur_uv)
    exe = os.path.join(BUILD, "match_cli")
    res = subprocess.run([exe, mode, "60", "50", "40", str(tmp_path / "d.txt")], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = res.stdout.strip().splitlines()
    assert lines[0] == "ok 1" and lines[1] == "ok2 1"
    rows = [l.split() for l in lines[2:]]
    idx = np.array([int(r[0]) for r in rows], np.int32)
    st = np.array([int(r[1]) for r in rows], np.uint8)
    if mode == "force":
        ok, oidx = oracle.force_match(ref, cur, 60.0)
    else:
        ok, oidx = oracle.nearby_match(ref, cur, ref_uv, cur_uv, 60.0, max_col=50, max_row=40)
    assert np.array_equal(idx, oidx)
    omatched, ost = oracle.fill_matched_pixels(oidx, cur_uv)
    assert np.array_equal(st, ost)
    got = np.array([[int(r[2], 16), int(r[3], 16)] for r in rows], dtype=np.uint32).view(np.float32)
    assert np.array_equal(got[ost == 1], omatched[ost == 1])


@pytest.mark.parametrize("mode,dim", [("force", 256), ("nearby", 256), ("force", 128), ("nearby", 128)])
def test_cpp_float_matcher_matches_oracle(tmp_path, oracle, mode, dim):
    """DescriptorMatcher<SuperpointDescriptorType / DiskDescriptorType> with the reference's cosine ComputeDistance: the
    probe must recognise it, the call must run on the device, and the indices must be the oracle's."""
    ref, cur, perm = synth.make_float_descriptors(300, 420, dim=dim, noise=0.3)
    rs = np.random.RandomState(4)
    cur_uv = np.stack([rs.uniform(0, 640, 420), rs.uniform(0, 480, 420)], axis=1).astype(np.float32)
    ref_uv = np.stack([rs.uniform(0, 640, 300), rs.uniform(0, 480, 300)], axis=1).astype(np.float32)
    ref_uv[perm[:300] % 300] = cur_uv[:300] + 5.0
    with open(tmp_path / "d.bin", "wb") as f:
        f.write(struct.pack("<ii", 300, 420))
        for a in (ref, cur, ref_uv, cur_uv):
            f.write(np.ascontiguousarray(a, np.float32).tobytes())
    exe = os.path.join(BUILD, "match_float_cli")
    res = subprocess.run([exe, mode, str(dim), "0.1", "50", "40", str(tmp_path / "d.bin")], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = res.stdout.strip().splitlines()
    assert lines[0] == "ok 1" and lines[1] == "ok2 1" and lines[2] == "device 1", lines[:3]
    rows = [l.split() for l in lines[3:]]
    idx = np.array([int(r[0]) for r in rows], np.int32)
    st = np.array([int(r[1]) for r in rows], np.uint8)
    if mode == "force":
        ok, oidx = oracle.match_float(ref, cur, 0.1)
    else:
        ok, oidx = oracle.match_float(ref, cur, 0.1, ref_uv, cur_uv, max_col=50, max_row=40)
    assert np.array_equal(idx, oidx)
    assert (oidx >= 0).sum() > 50
    omatched, ost = oracle.fill_matched_pixels(oidx, cur_uv)
    assert np.array_equal(st, ost)
    got = np.array([[int(r[2], 16), int(r[3], 16)] for r in rows], dtype=np.uint32).view(np.float32)
    assert np.array_equal(got[ost == 1], omatched[ost == 1])


MISSING_DROPIN = ("feature_tracker_amd/host/build/dropin/ is missing: the reference's own caller programs are compiled unchanged by build() "
                  "(scripts/check_dropin.sh) in the container where /root/reference is mounted and travel to the GPU box as build artefacts.  A box "
                  "without them has NOT exercised the drop-in boundary, so this is a failure, not a skip (VERDICT r4 item 9).")


@pytest.mark.parametrize("prog,expect", [("test_optical_flow", r"tracked|cost time"), ("test_descriptor_matcher_brief", r"tracked features (\d+) / (\d+)"),
                                         ("test_descriptor_matcher_superpoint", r"tracked features (\d+) / (\d+)"),
                                         ("test_descriptor_matcher_disk", r"tracked features (\d+) / (\d+)")])
def test_reference_callers_run_unchanged(tmp_path, prog, expect):
    """The reference's own test programs, compiled UNCHANGED against this repo's headers / libraries by
    scripts/check_dropin.sh (only possible where the reference is mounted; the binaries travel as build artefacts),
    run headless on the GPU with the reference's example images at the relative path they hard-code."""
    import re
    exe = os.path.join(BUILD, "dropin", prog)
    if not os.path.exists(exe):
        pytest.fail(MISSING_DROPIN)
    os.makedirs(tmp_path / "example", exist_ok=True)
    os.symlink(DATA, tmp_path / "example" / "optical_flow")
    os.makedirs(tmp_path / "build", exist_ok=True)
    res = subprocess.run([exe], cwd=tmp_path / "build", capture_output=True, text=True, timeout=180)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out[-2000:]
    m = re.search(expect, out)
    assert m, out[-2000:]
    if m.groups():
        assert int(m.group(2)) >= 50 and int(m.group(1)) >= 10, out[-1000:]  # matches found on the real image pair


def test_reference_direct_method_program_matches_oracle(tmp_path, oracle):
    """The reference's own test_direct_method.cpp, compiled unchanged (scripts/check_dropin.sh), on its own example
    frames: 300 std::rand() points with stereo depth, 5 frames, 5-level pyramids.  The poses it prints must be the
    oracle's for the same inputs (the program prints ~6 significant digits; the ABI-level tests compare bit for bit)."""
    import ctypes
    import re
    from PIL import Image
    exe = os.path.join(BUILD, "dropin", "test_direct_method")
    if not os.path.exists(exe):
        pytest.fail(MISSING_DROPIN)
    data = os.path.join(ROOT, "tests", "data", "direct_method")
    os.makedirs(tmp_path / "example", exist_ok=True)
    os.symlink(data, tmp_path / "example" / "direct_method")
    os.makedirs(tmp_path / "build", exist_ok=True)
    res = subprocess.run([exe], cwd=tmp_path / "build", capture_output=True, text=True, timeout=300)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out[-2000:]
    got = re.findall(r"q_rc \[wxyz\]\[([^\]]+)\], p_rc \[([^\]]+)\]", out)
    assert len(got) == 5, out[-2000:]
    # the same inputs, rebuilt here: glibc rand() from its default seed (test_direct_method.cpp:45-49)
    fx = fy = np.float32(718.856)
    cx, cy, baseline = np.float32(607.1928), np.float32(185.2157), np.float32(0.573)
    left = np.array(Image.open(os.path.join(data, "left.png")).convert("L"))
    disp = np.array(Image.open(os.path.join(data, "disparity.png")).convert("L"))
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    uv = np.zeros((300, 2), np.float32)
    depth = np.zeros(300, np.float32)
    with np.errstate(all="ignore"):
        for i in range(300):
            # Vec2(std::rand() % cols, std::rand() % rows): g++ evaluates constructor arguments right to left
            v = libc.rand() % left.shape[0]
            u = libc.rand() % left.shape[1]
            uv[i] = (u, v)
            depth[i] = np.float32(fx * baseline) / np.float32(int(disp[v, u]))
        p_w = np.stack([(uv[:, 0] - cx) / fx * depth, (uv[:, 1] - cy) / fy * depth, np.float32(1.0) * depth], axis=1).astype(np.float32)
    ref_levels = synth.build_pyramid(left, 5)
    q, p, cur_uv, status = np.float32([1, 0, 0, 0]), np.zeros(3, np.float32), None, None
    with np.errstate(all="ignore"):
        for k in range(5):
            cur = np.array(Image.open(os.path.join(data, f"00000{k + 1}.png")).convert("L"))
            # world frame == reference camera frame here (q_ref = I, p_ref = 0), so the world overload reduces to the camera one
            ok, cur_uv, q, p, status, _ = oracle.direct_track(ref_levels, synth.build_pyramid(cur, 5), [fx, fy, cx, cy], p_w, uv, cur_uv, q, p, None,
                                                              max_points=500)
            gq = np.array([float(x) for x in got[k][0].split(",")])
            gp = np.array([float(x) for x in got[k][1].split(",")])
            assert np.allclose(gq, q, rtol=2e-4, atol=2e-6), (k, gq, q)
            assert np.allclose(gp, p, rtol=2e-4, atol=2e-6), (k, gp, p)
    assert abs(p[2]) > 0.5  # the car drove forward over the five frames


def test_demo_on_reference_example_images():
    """The reference's own example pair (752x480 PNGs): Harris corners -> pyramids -> 3 trackers; most corners must track."""
    exe = os.path.join(BUILD, "demo_optical_flow")
    res = subprocess.run([exe, os.path.join(DATA, "ref_image.png"), os.path.join(DATA, "cur_image.png"), "4", "6", "2"],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    import re
    hits = re.findall(r"(Basic|Affine|Lssd) klt pass 1: ok 1, (\d+) / (\d+) tracked", res.stdout)
    assert len(hits) == 3, res.stdout
    for name, tracked, total in hits:
        assert int(total) >= 100 and int(tracked) >= 0.6 * int(total), res.stdout


def test_real_images_python_vs_oracle(ftk, oracle):
    """Parity on the reference's real example images through the Python binding, all three models, fast method (the tests' default)."""
    from PIL import Image
    ref = np.array(Image.open(os.path.join(DATA, "ref_image.png")))
    cur = np.array(Image.open(os.path.join(DATA, "cur_image.png")))
    assert ref.shape == (480, 752) and ref.dtype == np.uint8
    ref_levels, cur_levels = synth.build_pyramid(ref, 4), synth.build_pyramid(cur, 4)
    uv = synth.make_features(300, 752, 480, seed=3, half=6)
    for model, cls in (("basic", ftk.OpticalFlowBasicKlt), ("affine", ftk.OpticalFlowAffineKlt), ("lssd", ftk.OpticalFlowLssdKlt)):
        for method in ("fast", "inverse"):
            klt = cls()
            klt.options().kMethod = method
            ok, c, s = klt.TrackFeatures(ftk.ImagePyramid.from_host_levels(ref_levels), ftk.ImagePyramid.from_host_levels(cur_levels), uv)
            ok2, oc, os_, oit = oracle.klt_track_pyramid(model, ref_levels, cur_levels, uv, method=method, half=6)
            assert np.array_equal(s, os_), (model, method)
            assert np.array_equal(c.view(np.uint32), oc.view(np.uint32)), (model, method)


def test_cpp_image_pyramid_built_on_device_lazy_host_levels_and_threads():
    """ImagePyramid::CreateImagePyramid (inside the reference's timed region, test_optical_flow.cpp:69-73) builds levels >= 1 in
    HBM from one upload of level 0; host copies appear only when read and equal the box mean; a frame written in place into the
    aliased level-0 buffer is noticed; two threads with their own trackers on shared const pyramids get the serial result."""
    exe = os.path.join(BUILD, "pyramid_cli")
    assert os.path.exists(exe), "host layer not built"
    res = subprocess.run([exe, os.path.join(DATA, "ref_image.png"), os.path.join(DATA, "cur_image.png"), "4"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    out = dict(l.split() for l in res.stdout.strip().splitlines() if len(l.split()) == 2)
    for key in ("built_on_device", "tracked_without_host_levels", "levels_equal_box_mean", "same_after_host_read", "in_place_overwrite_noticed",
                "two_threads_equal_serial"):
        assert out.get(key) == "1", res.stdout
    assert res.stdout.strip().endswith("PASS")


@pytest.mark.parametrize("variant", ["nearmiss", "nooffload"])
def test_cpp_matcher_honours_a_virtual_distance_that_only_looks_like_hamming(tmp_path, oracle, variant):
    """DescriptorMatcher offloads only a RECOGNISED distance.  'nearmiss' agrees with Hamming on random pairs (the probe) and
    answers 1000 where fewer than 25 bits differ — i.e. exactly on the true matches; the reference would call the virtual
    for every pair, find nothing under the threshold for those rows, and so must this.  'nooffload' turns the offload off."""
    ref, cur, perm = synth.make_descriptors(300, 420, flips=20)
    rs = np.random.RandomState(4)
    cur_uv = np.stack([rs.uniform(0, 640, 420), rs.uniform(0, 480, 420)], axis=1).astype(np.float32)
    ref_uv = np.stack([rs.uniform(0, 640, 300), rs.uniform(0, 480, 300)], axis=1).astype(np.float32)
    write_descriptors(tmp_path / "d.txt", ref, cur, ref_uv, cur_uv)
    exe = os.path.join(BUILD, "match_cli")
    res = subprocess.run([exe, "force", "60", "50", "40", str(tmp_path / "d.txt"), variant], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = res.stdout.strip().splitlines()
    assert lines[0] == "ok 1"
    idx = np.array([int(l.split()[0]) for l in lines[2:]], np.int32)
    ok, hamming_idx = oracle.force_match(ref, cur, 60.0)
    if variant == "nooffload":
        assert np.array_equal(idx, hamming_idx)
    else:
        # host semantics of the near-miss distance: pairs closer than 25 bits are worth 1000 (> threshold); every other pair of
        # this set differs in ~128 bits, so nothing matches — while the plain Hamming answer matches almost every row
        d = (ref[:, None, :] != cur[None, :, :]).sum(axis=2).astype(np.float32)
        d[d < 25] = 1000.0
        want = np.where((d < 60).any(axis=1), d.argmin(axis=1), -1).astype(np.int32)
        assert np.array_equal(idx, want)
        assert (hamming_idx >= 0).sum() > 250 and (idx == -1).all()


@pytest.mark.parametrize("model,method", [("basic", "inverse"), ("lssd", "fast")])
def test_cpp_tracker_as_one_rank_of_a_communicator(tmp_path, oracle, model, method):
    """The C++ OpticalFlow classes pick the multi-GPU path up from the environment (device_runtime.h, SharedComm): with
    FTK_COMM_ID_FILE set this process creates the RCCL communicator (world size 1 here: one GPU per test box), tracks its block
    and goes through ncclAllGather + the scatter; the result must be the single-GPU one, bit for bit, incl. a global cap."""
    ref_levels, cur_levels = scenes.scene(320, 240, 3, "easy", "similarity")
    uv = scenes.features(203, 320, 240, half=5)
    env = {"FTK_WORLD_SIZE": "1", "FTK_RANK": "0", "FTK_COMM_ID_FILE": str(tmp_path / "rccl_id.bin")}
    with open(tmp_path / "rccl_id.bin", "wb") as f:
        f.write(b"\x55" * 128)  # a stale file of an earlier launch under the same path: rank 0 replaces it
    c, s, it, head = run_track_cli(tmp_path, model, method, 3, 5, ref_levels[0], cur_levels[0], uv, max_points=150, env=env)
    assert not os.path.exists(tmp_path / "rccl_id.bin")  # rank 0 published the unique id and removed the file once the communicator existed
    ok, oc, os_, oit = oracle.klt_track_pyramid(model, ref_levels, cur_levels, uv, method=method, half=5, max_points=150)
    assert np.array_equal(s, os_) and np.array_equal(c.view(np.uint32), oc.view(np.uint32)) and np.array_equal(it, oit)
