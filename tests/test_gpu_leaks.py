"""HBM does not leak: free memory after many construct / prove / free cycles of every handle family stays where it
was after the first cycle (the arenas the library keeps on purpose are allocated by then)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_bytes():
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_no_hbm_growth_over_repeated_cycles(atlas):
    from oracle import orc
    from jolt_atlas_amd import instances as I, reduced
    A = atlas
    n = 14
    rng = np.random.default_rng(1)
    srs = A.SRS.generate(orc.random_fr(1, 1)[0], 1 << n)
    L, R = orc.random_fr(1 << n, 2), orc.random_fr(1 << n, 3)
    pt = orc.random_fr(n, 4)
    idx64 = rng.integers(0, 1 << 32, size=1 << n, dtype=np.uint64)
    H = [rng.integers(0, 16, size=1 << n).astype(np.int32) for _ in range(8)]
    chunks = orc.random_fr(8 * 4, 5).reshape(8, 4, 4)
    claim = orc.random_fr(1, 6)[0]
    c128 = [int(x) for x in rng.integers(1, 1 << 62, size=n)]

    def cycle():
        pl, pr = A.MultilinearPolynomial.from_fr(L), A.MultilinearPolynomial.from_fr(R)
        prover = A.EinsumDotProver(pl, pr, None, 0, 0, 0)
        A.Sumcheck.prove(prover, claim, A.Blake2bTranscript(b"l"), n)
        prover.free()
        p = A.MultilinearPolynomial.from_fr(L)
        srs.msm(p)
        A.HyperKZG.open(srs, p, c128, A.Blake2bTranscript(b"l"))
        srs.msm_small(rng.integers(-100, 100, size=1 << n, dtype=np.int32))
        for make in (lambda: I.dense_opening(p.clone(), pt),
                     lambda: I.ra_virtual(H, 4, chunks, pt),
                     lambda: I.booleanity(np.zeros((8, 16, 4), dtype=np.uint64), H, 4, orc.random_fr(8, 7), orc.random_fr(4, 8), pt),
                     lambda: I.ps_shout_relu(idx64, 32, pt, claim),
                     lambda: I.ps_shout_ult(idx64, pt, claim),
                     lambda: I.identity_range_check(idx64 & np.uint64(0xFFFF), 16, 4, pt),
                     lambda: I.onehot_opening(H[0], 4, orc.random_fr(4, 9), pt),
                     lambda: I.elementwise(I.EW_MUL, [p, p], pt),
                     lambda: I.softmax_instance(I.SM_EXP_SUM, p, None, 4, n - 4, orc.random_fr(4, 10))):
            inst = make()
            inst.prove(claim, A.Blake2bTranscript(b"l"))
            inst.free()
        ops = [dict(poly=p, point=pt, claim=claim), dict(k=H[0][: 1 << (n - 4)], log_K=4, r_address=orc.random_fr(4, 11),
                                                         r_cycle=orc.random_fr(n - 4, 12), claim=claim)]
        reduced.prove_reduced_openings(ops, srs, A.Blake2bTranscript(b"l"))
        p.free()
        A.sync()

    cycle(); cycle()
    base = _free_bytes()
    for _ in range(15):
        cycle()
    grown = base - _free_bytes()
    assert grown < (8 << 20), f"HBM shrank by {grown / 2**20:.1f} MiB over 15 cycles"
    srs.free()
