"""The reference's two tests compiled against the C++ mirror of its API (include/srack.hpp, tests/cpp/test_mirror.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_mirror")
LIBDIR = os.path.join(ROOT, "s-rack_amd")


@pytest.fixture(scope="module")
def mirror_bin():
    src = os.path.join(ROOT, "tests", "cpp", "test_mirror.cpp")
    deps = [src, os.path.join(ROOT, "include", "srack.hpp"), os.path.join(ROOT, "include", "srack_hip.h")]
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-o", BIN, src, "-L" + LIBDIR, "-lsrack_hip", "-Wl,-rpath," + LIBDIR,
                        "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return BIN


def test_topological_sort_cpp(mirror_bin):
    r = subprocess.run([mirror_bin, "topo"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "topological_sort ok" in r.stdout


@pytest.mark.gpu
def test_produces_440_cpp(mirror_bin):
    r = subprocess.run([mirror_bin, "dco"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "produces_440 ok" in r.stdout


@pytest.mark.gpu
def test_one_rank_mix_comm_cpp(mirror_bin):
    """srack_dist_unique_id / init / comm_count / reduce_mix / destroy from a compiled C++ host (include/srack.hpp: MixComm)."""
    r = subprocess.run([mirror_bin, "dist"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "one_rank_mix_comm ok" in r.stdout
