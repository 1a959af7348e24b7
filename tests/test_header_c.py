"""CPU: include/atlas_hip.h is valid C99 and C++17, and a plain-C caller compiles and links against the library."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "atlas_hip.h")


def test_header_is_c99_and_cxx17():
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", HDR],
                ["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", HDR]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


import pytest


@pytest.mark.parametrize("name", ["prove_dot", "commit_open"])
def test_c_example_links(tmp_path, name):
    import jolt_atlas_amd as A
    libdir = os.path.dirname(A.LIB_PATH)
    exe = str(tmp_path / name)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name + ".c"),
                        "-L", libdir, "-latlas_hip", "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # no GPU here: the program must fail loudly through the library's error path, not crash
    run = subprocess.run([exe, "4"], capture_output=True, text=True)
    assert run.returncode in (0, 1)
    if run.returncode == 1:
        assert "atlas_init" in run.stderr
