"""CPU: the host-side field multiplication of the product (csrc/host_field.hpp, used by the transcript thread to finish every
round polynomial) — unrolled CIOS against the product-then-reduce form kept beside it, on 2.2 million operand pairs."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cios_matches_sos(tmp_path):
    exe = tmp_path / "check_host_field"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", ROOT, os.path.join(ROOT, "tools", "check_host_field.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches 0" in out.stdout
