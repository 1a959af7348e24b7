"""Split a round's time on the DEVICE's clock (ATLAS_DEV_STAMPS=1; csrc/channel.hip.h: ch_stamp): prove one Einsum node, dump the stamps
(atlas_rt_stamps_dump) and print, per mail tag (= one launch of one lane in one round), when its first workgroup entered, saw the
challenge, began the reduction of the partial rows, had them all, and mailed — all relative to the mail of the round before.
    ATLAS_DEV_STAMPS=1 python tools/dev_stamps.py [node_einsum|node_relu|node_mul]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import build_graphs as BG
import jolt_atlas_amd as A
from jolt_atlas_amd import graph as GG

name = sys.argv[1] if len(sys.argv) > 1 else "node_einsum"
A.init(0)
nodes, outputs, inputs = getattr(BG, name)()
nv = BG.max_vars(nodes)
srs = A.SRS.generate(np.array([0x1234567, 0, 0, 0], dtype=np.uint64), 1 << nv)
srs.precompute()
G = GG.Graph(nodes, outputs)
for rep in range(3):
    G.prove(srs, inputs)
    A.lib.atlas_rt_stamps_dump.argtypes = [C.c_char_p]
    path = "/tmp/stamps_%d.txt" % rep
    assert A.lib.atlas_rt_stamps_dump(path.encode()) == 0
EV = {1: "entry", 2: "challenge", 3: "reduce_begin", 4: "rows_in", 5: "mailed", 6: "work_done"}
dev = {}
order = []
for line in open(path):
    k, ev, tag, t = line.split()
    if k != "D": continue
    ev, tag, t = int(ev), int(tag), int(t)
    if tag not in dev: dev[tag] = {}; order.append(tag)
    dev[tag].setdefault(EV.get(ev, str(ev)), t / 100.0)          # us
order.sort(key=lambda g: min(dev[g].values()))
t0 = min(min(d.values()) for d in dev.values())
print("# %s: %d launches with stamps; times in us since the first stamp" % (name, len(order)))
print("%10s %9s %9s %9s %9s %9s %9s | %8s %8s" % ("tag", "entry", "chall", "work", "red_beg", "rows_in", "mailed", "ch->mail", "period"))
prev_mail = None
for g in order:
    d = dev[g]
    f = lambda k: ("%9.1f" % (d[k] - t0)) if k in d else "        -"
    span = (d["mailed"] - d["challenge"]) if "mailed" in d and "challenge" in d else float("nan")
    per = (d["mailed"] - prev_mail) if "mailed" in d and prev_mail else float("nan")
    print("%10d %s %s %s %s %s %s | %8.1f %8.1f" % (g, f("entry"), f("challenge"), f("work_done"), f("reduce_begin"), f("rows_in"), f("mailed"), span, per))
    if "mailed" in d: prev_mail = d["mailed"]
