#!/bin/bash
# The Einsum node with every device allocation pre-filled (ATLAS_POOL_POISON): the transcript state must not depend on the fill.
cd "${GRAFT_REPO_ROOT:-.}"
echo clean; PRINT_STATE=1 REPS=2 python tools/time_node.py 2>&1 | tail -2
for b in 165 255 1; do echo poison $b; ATLAS_POOL_POISON=$b REPS=2 PRINT_STATE=1 timeout 120 python tools/time_node.py 2>&1 | tail -2; done
