# the lazy path forced down to 2^15: the cached digests of (16, 15) exercise it (the H rows with None fall back, the rows cut from the lookups do not: host lookups are not packed -> use a test of the H path without None)
ATLAS_RA_LAZY_LOG=15 timeout 600 python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import jolt_atlas_amd as A
from jolt_atlas_amd import instances as I
import importlib.util
spec = importlib.util.spec_from_file_location("full", "tests/test_gpu_full_size.py"); T = importlib.util.module_from_spec(spec); spec.loader.exec_module(T)
A.init(0)
d, log_T = 16, 15
log_k, _, _, r_cycle, claim, log_K, lookups, r_address = T.ra_large_inputs(d, log_T)
want = T._oracle_digest(f"ra_large2[{d}-{log_T}]", None)
Hl = [((lookups >> np.uint64(log_k * (d - 1 - i))) & np.uint64(15)).astype(np.int32) for i in range(d)]
inst = I.ra_virtual(Hl, log_k, r_address.reshape(d, log_k, 4), r_cycle)
t_g = A.Blake2bTranscript(b"ra_large2")
rows_g, ch_g = inst.prove(claim, t_g)
print("lazy at 2^15:", "OK" if T._digest(rows_g, ch_g, t_g.state) == want else "MISMATCH")
inst.free()
PY
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_ra.py -q -m gpu -x -k "not lazy" 2>&1 | tail -3
