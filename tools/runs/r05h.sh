DIMS=16,64,1024,14 REPS=3 ATLAS_TRACE=1 ATLAS_TRACE_ROUNDS=1 python tools/time_node.py 2>&1 | grep -v "^\[atlas trace\]   round" | tail -75
