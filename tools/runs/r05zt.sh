for v in 13 14 15 16; do echo "--- ATLAS_BOOL_SPLIT_LOG=$v"; ATLAS_BOOL_SPLIT_LOG=$v LOG_T=16 python tools/trace_ra_small.py 2>&1 | grep booleanity | tail -2; done
for v in 12 13 14; do echo "--- ATLAS_RA_SPLIT_MIN=$v"; ATLAS_RA_SPLIT_MIN=$v LOG_T=16 python tools/trace_ra_small.py 2>&1 | grep ra_virtual | tail -2; done
