"""The CMake entry a maintainer of the reference would use ("CMake") configures, builds the C-ABI library
through feature_tracker_amd/csrc/Makefile and links a caller against the reference's target names.  The list of translation
units lives in that Makefile only: a second list in CMakeLists.txt went stale once (klt_fast_kernels.hip, ftk_comm.cpp and
ftk_build_info.cpp were missing from it), which this test would have caught."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cmake_lists_no_kernel_sources_of_its_own():
    text = open(os.path.join(ROOT, "CMakeLists.txt")).read()
    code = "\n".join(line for line in text.splitlines() if not line.lstrip().startswith("#"))
    assert not re.search(r"\b\w+\.hip\b", code), "CMakeLists.txt must not keep its own list of .hip sources (csrc/Makefile is the one list)"
    assert "make" in code.lower() and "libftk_hip.so" in code


def test_every_kernel_source_is_in_the_makefile():
    csrc = os.path.join(ROOT, "feature_tracker_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    srcs = re.search(r"^SRCS\s*:=\s*(.*)$", mk, flags=re.M).group(1).split()
    on_disk = sorted(f for f in os.listdir(csrc) if f.endswith(".hip") or (f.endswith(".cpp") and f != "ftk_build_info.cpp"))
    assert sorted(srcs) == on_disk
    objs = re.search(r"^OBJS\s*:=\s*(.*)$", mk, flags=re.M).group(1).split()
    assert sorted(objs) == sorted([os.path.splitext(s)[0] + ".o" for s in srcs] + ["ftk_build_info.o"])


@pytest.mark.skipif(shutil.which("cmake") is None or shutil.which("make") is None, reason="cmake / make not installed")
def test_cmake_configures_builds_and_links_a_caller(tmp_path):
    build = str(tmp_path / "cm")
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    cfg = subprocess.run(["cmake", "-S", ROOT, "-B", build] + gen, capture_output=True, text=True, timeout=300)
    assert cfg.returncode == 0, cfg.stdout[-2000:] + cfg.stderr[-2000:]
    # the library (up to date after build(): make only checks) and one of this repo's drivers against the reference's target names
    out = subprocess.run(["cmake", "--build", build, "--target", "track_cli"], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    exe = os.path.join(build, "track_cli")
    assert os.path.exists(exe)
    needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout if shutil.which("readelf") else "libftk_hip.so"
    assert "libftk_hip.so" in needed
