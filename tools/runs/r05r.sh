mkdir -p gpurun_out
python bench.py --no-pmc > gpurun_out/r05r_bench.json 2> gpurun_out/r05r_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05r_bench.json').read().strip().splitlines()[-1])
print("step", d["ms_per_step"], "frac", d["roofline"]["frac"], "whole", d["roofline"].get("whole_step_frac"))
print("msm", {k:v for k,v in d.get("msm",{}).items() if k in ("ms_per_msm","sort_and_fold_ms")})
for k,v in d.get("prove_graph",{}).items():
    if isinstance(v,dict): print(k, {kk:v.get(kk) for kk in ("total_ms","iop_ms","commit_ms","reduction_ms","hyperkzg_ms","verified","proof_sha16")})
print("nodes", {k:(v.get("iop_ms"), v.get("prove_graph_ms")) for k,v in d.get("node_graphs",{}).items()})
print("node", d.get("node"))
PY
