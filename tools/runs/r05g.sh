ATLAS_TRACE=1 ATLAS_TRACE_ROUNDS=1 python tools/time_clamp_rounds.py 2>&1 | tail -90
