"""Where the IOP of a whole proof goes: per operator (ATLAS_GRAPH_TRACE=1: wall clock of every prove_node, device synchronised per node)
and per stage inside the operator flows (ATLAS_GRAPH_TRACE=2: the marks of the flows, device synchronised per mark).  Diagnosis only — the
synchronisations make the totals larger than the untraced proof.   usage: python tools/gpt2_by_operator.py [graph=gpt2] [level=2]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPS = ("Input Constant Identity Add Sub Mul Square Cube And Iff ReLU Einsum Reshape MoveAxis Broadcast Slice Concat Sum ScalarConstDiv Div MeanOfSquares "
       "Rsqrt SoftmaxLastAxis Tanh GatherLarge GatherSmall Erf Sigmoid Neg IsNan Clamp Sin Cos").split()


def run(graph, level, trace):
    env = dict(os.environ, ATLAS_GRAPH_TRACE=str(trace), ATLAS_GRAPH_VERIFY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_graph.py"), graph, str(level), "2"], env=env, capture_output=True, text=True)
    return p.stdout, p.stderr


def main():
    graph = sys.argv[1] if len(sys.argv) > 1 else "gpt2"
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    out, err = run(graph, level, 1)
    print("# %s: per operator (ATLAS_GRAPH_TRACE=1, last of two proofs)" % graph)
    rows = re.findall(r"\[atlas graph\] op\s+(\d+)\s+x(\d+)\s+([\d.]+) ms\s+\(([\d.]+) ms each\)", err)
    n_ops = len(set(r[0] for r in rows))
    rows = rows[-n_ops:]
    tot = sum(float(r[2]) for r in rows)
    print("%-18s %6s %10s %9s %6s" % ("operator", "nodes", "ms", "ms each", "share"))
    for op, n, ms, each in sorted(rows, key=lambda r: -float(r[2])):
        print("%-18s %6s %10.1f %9.3f %5.1f%%" % (OPS[int(op)], n, float(ms), float(each), 100 * float(ms) / tot))
    print("%-18s %6d %10.1f" % ("total", sum(int(r[1]) for r in rows), tot))
    print(out.strip().splitlines()[-1][:600])
    out2, err2 = run(graph, level, 2)
    # marks: "[atlas graph]   node N what   x ms"; attribute to the operator of node N through the per-node op list printed by level 1? the marks carry the
    # stage name only, so aggregate by stage name (names are unique per flow)
    marks = re.findall(r"\[atlas graph\]\s+node (\d+) op (\d+) \| (.+?) \|\s+([\d.]+) ms", err2)
    half = len(marks) // 2
    agg = collections.OrderedDict()
    for node, op, what, ms in marks[half:]:
        e = agg.setdefault("%s / %s" % (OPS[int(op)], what.strip()), [0.0, 0])
        e[0] += float(ms); e[1] += 1
    print("\n# %s: per stage inside the operator flows (ATLAS_GRAPH_TRACE=2: device synchronised at every mark; second proof)" % graph)
    print("%-52s %6s %10s %9s" % ("operator / stage", "count", "ms", "ms each"))
    for what, (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print("%-52s %6d %10.1f %9.3f" % (what, n, ms, ms / n))
    print(out2.strip().splitlines()[-1][:600])


if __name__ == "__main__":
    main()
