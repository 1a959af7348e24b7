"""Stage split of atlas_prove_graph_sharded over WORLD ranks (processes sharing this box's GPU): usage  python tools/time_sharded.py [graph=gpt2] [world=2] [reps=3]
(WHOLE_TABLE=1: every rank holds the whole fixed-base table instead of its point range; ATLAS_REDUCTION_REPLICATED=1: every rank steps every member of the opening-reduction sumcheck, the round-4 behaviour)."""
import json, os, subprocess, sys, textwrap
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
graph = sys.argv[1] if len(sys.argv) > 1 else "gpt2"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
child = textwrap.dedent(f"""
    import json, os, sys, time
    sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, "tools"))
    import numpy as np
    import build_graphs as BG
    import jolt_atlas_amd as A
    from jolt_atlas_amd import sharded, graph as GG
    rank, world = int(sys.argv[1]), {world}
    A.init(0)
    nodes, outputs, inputs = getattr(BG, {graph!r})() if hasattr(BG, {graph!r}) else BG.tiny(layers=2)
    nv = BG.max_vars(nodes)
    srs = A.SRS.generate(np.array([0x1234567, 0, 0, 0], dtype=np.uint64), 1 << nv)
    if nv >= 16:
        if world > 1 and not os.environ.get("WHOLE_TABLE"): srs.precompute_range(rank * ((1 << nv) // world), (1 << nv) // world)   # this rank's point range of the table
        else: srs.precompute()
    G = GG.Graph(nodes, outputs)
    A.device_memory(reset_peak=True)
    grp = sharded.ShardGroup(sys.argv[2], world, rank) if world > 1 else None
    best = None
    for rep in range({reps}):
        t0 = time.time()
        proof, state, tm = G.prove(srs, inputs, group=grp) if grp else G.prove(srs, inputs)
        tm = dict(tm); tm["wall_ms"] = 1e3 * (time.time() - t0); tm["state"] = state.hex()[:16]
        if best is None or tm["total_ms"] < best["total_ms"]: best = tm
    best["peak_GB"] = A.device_memory()[1] / 2 ** 30
    print("RANK", rank, json.dumps({{k: (round(v, 2) if isinstance(v, float) else v) for k, v in best.items() if k.endswith("_ms") or k in ("state", "peak_GB")}}))
    if grp: grp.close()
""")
path = "/tmp/time_sharded_child.py"
open(path, "w").write(child)
name = f"/atlas_ts_{os.getpid()}"
procs = [subprocess.Popen([sys.executable, path, str(r), name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
for p in procs:
    o, e = p.communicate(timeout=1200)
    print(o.strip() if p.returncode == 0 else ("FAILED: " + e[-1500:]))
