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


@pytest.mark.gpu
@pytest.mark.parametrize("voices", [1, 300, 5000])
def test_audio_callback_cpp(mirror_bin, voices):
    """The reference's audio callback (src/main.rs:59-90) transliterated onto the C++ mirror: `execute` once per buffer_size frames
    whenever the staging buffers run dry, interleave into a `data` slice of whatever length the device asks for — bit for bit what
    one long render gives (1 voice: the reference's own case; 300 and 5000: per-voice pitch, the flagship kernel with its co-scheduled
    control block, i.e. a tick session from the second execute on)."""
    r = subprocess.run([mirror_bin, "callback", str(voices)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "audio_callback ok" in r.stdout
