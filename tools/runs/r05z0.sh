mkdir -p gpurun_out
ATLAS_PROF=1 ATLAS_TRACE=1 ATLAS_GRAPH_VERIFY=0 python tools/time_graph.py gpt2 2 2 > gpurun_out/r05z0_prof.txt 2>&1
grep -E "atlas prof|prove_reduced_openings|batched_prove \(|onehot pool" gpurun_out/r05z0_prof.txt | awk '/Model::trace/{c++} c==2' | grep -v "node loop" | awk '/node loop \(iop\)/{skip=1} /prove_reduced_openings:/{skip=0} !skip' | head -60 | cut -c1-260
