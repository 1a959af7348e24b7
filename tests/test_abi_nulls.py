"""CPU: every C-ABI entry point survives an all-NULL / all-zero call (bad-argument or no-device error codes, never
a crash).  Runs in a child process so that a segfault is a test failure, not the end of the session."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, re, sys
sys.path.insert(0, %r)
import jolt_atlas_amd as A
if %d:
    A.init(0)
src = open(%r).read()
src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
protos = re.findall(r"\b([a-z_ ]*?[a-z_0-9]+\s*\**)\s*(atlas_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S)
skip = {"atlas_init", "atlas_shutdown"}
n = 0
for ret, name, params in protos:
    if name in skip:
        continue
    params = " ".join(params.split())
    args = []
    if params and params != "void":
        for p in params.split(","):
            p = p.strip()
            if "*" in p or "[" in p or re.search(r"_t\s+\w+$", p) and not re.search(r"\b(size_t|uint\d+_t|int\d+_t)\b", p):
                args.append(C.c_void_p(None))
            elif re.search(r"\b(size_t|uint64_t)\b", p):
                args.append(C.c_size_t(0))
            else:
                args.append(C.c_int(0))
    f = getattr(A.lib, name)
    f.restype = C.c_void_p if "*" in ret else (C.c_size_t if "size_t" in ret else C.c_int)
    f.argtypes = None
    print(name, flush=True)
    f(*args)
    n += 1
print("CALLED", n)
'''


import pytest


@pytest.mark.gpu
def test_all_entry_points_survive_null_arguments_with_a_device():
    """same with the library initialised: the argument checks behind NEED_INIT are the ones that answer"""
    _run(1)


def test_all_entry_points_survive_null_arguments():
    _run(0)


def _run(init):
    code = CHILD % (ROOT, init, os.path.join(ROOT, "include", "atlas_hip.h"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0, f"crashed in {lines[-1] if lines else '?'} (rc {r.returncode}): {r.stderr[-400:]}"
    m = re.match(r"CALLED (\d+)", lines[-1])
    assert m and int(m.group(1)) >= 90
