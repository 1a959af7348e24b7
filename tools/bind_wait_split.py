"""Review item 3: where the first bind pass of a 2^22 instance spends its time, on the device's clock (ATLAS_DEV_STAMPS=1).  Proves a few
2^22 degree-2 instances, dumps the stamps of the last one and prints per data pass: entry -> challenge seen (waiting) -> work done -> mailed.
    ATLAS_DEV_STAMPS=1 python tools/bind_wait_split.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import jolt_atlas_amd as A
from oracle import orc

A.init(0)
n = int(os.environ.get("LOG_N", "22"))
L = orc.random_fr(1 << n, 1); R = orc.random_fr(1 << n, 2)
claim = None
A.lib.atlas_rt_stamps_dump.argtypes = [C.c_char_p]
for rep in range(4):
    pl, pr = A.MultilinearPolynomial.from_fr(L), A.MultilinearPolynomial.from_fr(R)
    prover = A.EinsumDotProver(pl, pr, None, A.EQ_NONE, 0, 0)
    t = A.Blake2bTranscript(b"bind_wait_split")
    if claim is None:
        claim = orc.dot_claim(L, R, None, A.EQ_NONE, 0, 0)
    A.sync()
    assert A.lib.atlas_rt_stamps_dump(b"/tmp/discard.txt") == 0
    A.Sumcheck.prove(prover, claim[0], t, n)
    prover.free()
    assert A.lib.atlas_rt_stamps_dump(b"/tmp/bind_stamps.txt") == 0
EV = {1: "entry", 2: "challenge", 5: "mailed", 6: "work_done"}
dev = {}
for line in open("/tmp/bind_stamps.txt"):
    k, ev, tag, tt = line.split()
    if k != "D": continue
    dev.setdefault(int(tag), {}).setdefault(EV.get(int(ev), ev), int(tt) / 100.0)
tags = sorted(dev, key=lambda g: min(dev[g].values()))
t0 = min(min(d.values()) for d in dev.values())
print("# 2^%d degree-2 instance, workgroup 0 of each data pass (us since the first stamp; the round-0 pass waits for nothing and is not stamped at entry)" % n)
print("%8s %9s %9s %9s %9s | %8s %8s %8s" % ("mail tag", "entry", "challenge", "work_done", "mailed", "waiting", "working", "mailing"))
for g in tags:
    d = dev[g]
    f = lambda k: ("%9.1f" % (d[k] - t0)) if k in d else "        -"
    w = d.get("challenge", float("nan")) - d.get("entry", float("nan"))
    k = d.get("work_done", float("nan")) - d.get("challenge", d.get("entry", float("nan")))
    m = d.get("mailed", float("nan")) - d.get("work_done", float("nan"))
    print("%8d %s %s %s %s | %8.1f %8.1f %8.1f" % (g, f("entry"), f("challenge"), f("work_done"), f("mailed"), w, k, m))
