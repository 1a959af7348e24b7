"""The host-thread pool (csrc/host_threads.hpp) hands jobs to spinning workers without its mutex (a sequence-lock publication) and to sleeping
ones through the condition variable: a native stress of both paths, compiled here with g++ (CPU only)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("threads", [2, 4])
def test_host_threads_stress(tmp_path, threads):
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "ht_stress")
    flags = ["-O2", "-std=c++17", "-pthread"]
    if os.environ.get("ATLAS_TEST_TSAN") == "1":
        flags = ["-O1", "-g", "-fsanitize=thread", "-std=c++17", "-pthread"]
    subprocess.run([cxx, *flags, os.path.join(ROOT, "tests", "native", "host_threads_stress.cpp"), "-o", exe], check=True, timeout=300)
    env = dict(os.environ, ATLAS_HOST_THREADS=str(threads), ATLAS_HOST_SPIN_US="200")
    p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "bad 0" in p.stdout
    assert "ThreadSanitizer" not in p.stderr
