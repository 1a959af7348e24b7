for n in node_relu node_add; do
echo "=== $n"; ATLAS_GRAPH_TRACE=2 python tools/time_graph.py $n 2 3 2>&1 | grep -E "^node|\| |total_ms" | tail -24 | cut -c1-200
done
