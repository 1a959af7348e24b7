"""Seeded differential fuzzing of the device paths against the oracle over random small shapes and degenerate inputs
(all-zero operands, all-None one-hot rows, extreme lookup indices, one-variable instances)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
import os
OFF = int(os.environ.get("ATLAS_FUZZ_OFFSET", "0"))        # shift every seed: a different corpus per run


def _rows_equal(a, b):
    return len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))


def _special_fr(orc, n, rng, seed):
    """random field elements with zeros, ones and p - 1 mixed in"""
    v = orc.random_fr(n, seed)
    kinds = rng.integers(0, 8, size=n)
    one, m1 = orc.from_ints([1])[0], orc.from_ints([-1])[0]
    v[kinds == 0] = 0
    v[kinds == 1] = one
    v[kinds == 2] = m1
    return v


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_elementwise_and_softmax(atlas, seed):
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import instances as I
    A = atlas
    rng = np.random.default_rng(1000 + seed + OFF)
    n_ops = {0: 2, 1: 2, 2: 1, 3: 1, 4: 3, 5: 2, 6: 1, 7: 4, 8: 5, 9: int(rng.choice([2, 4, 8])), 10: 3, 11: int(rng.integers(1, 9)), 12: 3}
    for _ in range(6):
        op = int(rng.integers(0, 13))
        n_vars = int(rng.integers(1, 9))
        n = 1 << n_vars
        ops = [_special_fr(orc, n, rng, int(rng.integers(1 << 30))) for _ in range(n_ops[op])]
        if rng.random() < 0.2:
            ops[0][:] = 0
        k = {8: 2, 10: 1, 12: 1, 11: n_ops[11]}.get(op, 0)
        consts = orc.random_fr(k, int(rng.integers(1 << 30))) if k else None
        r = orc.random_fr(n_vars, int(rng.integers(1 << 30)))
        claim = orc.random_fr(1, int(rng.integers(1 << 30)))[0]
        o = OR.elementwise(op, ops, r, consts)
        t_o = orc.new_transcript(b"fz")
        rows_o, ch_o = o.prove(claim, t_o)
        polys = [A.MultilinearPolynomial.from_fr(v) for v in ops]
        inst = I.elementwise(op, polys, r, consts)
        t_g = A.Blake2bTranscript(b"fz")
        rows_g, ch_g = inst.prove(claim, t_g)
        assert ch_g == ch_o and _rows_equal(rows_g, rows_o) and t_g.state == t_o.state_bytes(), (op, n_vars)
        assert np.array_equal(np.stack(inst.final_claims()), o.finals())
        inst.free()
        for p_ in polys:
            p_.free()
    for _ in range(4):
        kind = int(rng.integers(0, 4))
        log_K = 0 if kind == 3 else int(rng.integers(1, 5))
        log_N = int(rng.integers(1, 6))
        n, K = 1 << (log_K + log_N), 1 << log_K
        a = _special_fr(orc, n, rng, int(rng.integers(1 << 30)))
        b = None
        if kind == 1:
            b = _special_fr(orc, n, rng, int(rng.integers(1 << 30)))
        if kind == 2:
            b = _special_fr(orc, K, rng, int(rng.integers(1 << 30)))
        r = orc.random_fr(log_K + (log_N if kind == 2 else 0), 5) if kind != 3 else None
        claim = orc.random_fr(1, 6)[0]
        o = OR.softmax(kind, a, b, log_K, log_N, r)
        t_o = orc.new_transcript(b"fz")
        rows_o, ch_o = o.prove(claim, t_o)
        pa = A.MultilinearPolynomial.from_fr(a)
        pb = A.MultilinearPolynomial.from_fr(b) if b is not None else None
        inst = I.softmax_instance(kind, pa, pb, log_K, log_N, r)
        t_g = A.Blake2bTranscript(b"fz")
        rows_g, ch_g = inst.prove(claim, t_g)
        assert ch_g == ch_o and _rows_equal(rows_g, rows_o) and t_g.state == t_o.state_bytes(), (kind, log_K, log_N)
        assert np.array_equal(np.stack(inst.final_claims()), o.finals())
        inst.free(); pa.free()
        if pb is not None:
            pb.free()


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_ra_family_and_lookups(atlas, seed):
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import instances as I
    A = atlas
    rng = np.random.default_rng(2000 + seed + OFF)
    for _ in range(3):
        d, log_k, log_T = int(rng.integers(1, 17)), int(rng.choice([1, 2, 4, 8])), int(rng.integers(1, 10))
        T, K = 1 << log_T, 1 << log_k
        H = [rng.integers(0, K, size=T).astype(np.int32) for _ in range(d)]
        for h in H:
            mode = rng.integers(0, 5)
            if mode == 0:
                h[:] = -1                                  # an all-None row
            elif mode == 1:
                h[rng.random(T) < 0.5] = -1
            elif mode == 2:
                h[:] = K - 1
        chunks = orc.random_fr(d * log_k, int(rng.integers(1 << 30))).reshape(d, log_k, 4)
        r_cycle = orc.random_fr(log_T, int(rng.integers(1 << 30)))
        claim = orc.random_fr(1, 3)[0]
        t_o = orc.new_transcript(b"fz")
        rows_o, ch_o = OR.ra_virtual(H, log_k, chunks, r_cycle).prove(claim, t_o)
        inst = I.ra_virtual(H, log_k, chunks, r_cycle)
        t_g = A.Blake2bTranscript(b"fz")
        rows_g, ch_g = inst.prove(claim, t_g)
        assert ch_g == ch_o and _rows_equal(rows_g, rows_o) and t_g.state == t_o.state_bytes(), ("ra", d, log_k, log_T)
        inst.free()
        r_address, gammas = orc.random_fr(log_k, 7), orc.random_fr(d, 9)
        G = OR.ra_G(H, log_k, r_cycle)
        zero = orc.fr_array(1)[0]
        t_o = orc.new_transcript(b"fz")
        rows_o, ch_o = OR.booleanity(G, H, log_k, gammas, r_address, r_cycle).prove(zero, t_o)
        inst = I.booleanity(G, H, log_k, gammas, r_address, r_cycle)
        t_g = A.Blake2bTranscript(b"fz")
        rows_g, ch_g = inst.prove(zero, t_g)
        assert ch_g == ch_o and _rows_equal(rows_g, rows_o) and t_g.state == t_o.state_bytes(), ("bool", d, log_k, log_T)
        inst.free()
    for _ in range(3):
        log_T = int(rng.integers(1, 9))
        T = 1 << log_T
        pat = int(rng.integers(0, 4))
        idx = rng.integers(0, 1 << 32, size=T, dtype=np.uint64)
        if pat == 0:
            idx[:] = 0
        elif pat == 1:
            idx[:] = (1 << 32) - 1
        elif pat == 2:
            idx[:] = 1 << 31
        r_node, gamma, claim = orc.random_fr(log_T, 5), orc.random_fr(1, 6)[0], orc.random_fr(1, 7)[0]
        which = int(rng.integers(0, 4))
        if which == 0:
            o, g_ = OR.ps_relu(idx, 32, r_node, gamma), I.ps_shout_relu(idx, 32, r_node, gamma)
        elif which == 1:
            bound, sym = int(rng.integers(1, 31)), bool(rng.integers(0, 2))
            o, g_ = OR.ps_clamp(idx, 32, bound, sym, r_node, gamma), I.ps_shout_clamp(idx, 32, bound, sym, r_node, gamma)
        elif which == 2:
            sh = int(rng.integers(0, 32))
            o, g_ = OR.ps_rshift(idx, 32, sh, r_node, gamma), I.ps_shout_rshift(idx, 32, sh, r_node, gamma)
        else:
            idx64 = idx | (rng.integers(0, 1 << 32, size=T, dtype=np.uint64) << np.uint64(32))
            o, g_ = OR.ps_ult(idx64, r_node, gamma), I.ps_shout_ult(idx64, r_node, gamma)
        t_o = orc.new_transcript(b"fz")
        rows_o, ch_o = o.prove(claim, t_o)
        t_g = A.Blake2bTranscript(b"fz")
        rows_g, ch_g = g_.prove(claim, t_g)
        assert ch_g == ch_o and _rows_equal(rows_g, rows_o) and t_g.state == t_o.state_bytes(), ("ps", which, pat, log_T)
        g_.free()


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_msm_and_openings(atlas, seed):
    from oracle import orc, orc_ra as OR, orc_batched as OB
    from jolt_atlas_amd import instances as I
    A = atlas
    rng = np.random.default_rng(3000 + seed + OFF)
    n_max = 1 << 11
    tau = orc.random_fr(1, 99)[0]
    srs = A.SRS.generate(tau, n_max)
    ref = srs.download()
    for _ in range(4):
        n = int(rng.integers(1, n_max + 1))                       # any length, not only powers of two
        sc = _special_fr(orc, n, rng, int(rng.integers(1 << 30)))
        assert orc.g1_eq(srs.msm(sc), orc.msm(ref[:n], sc)), ("msm", n)
        dt = [np.uint8, np.uint16, np.uint32, np.uint64, np.int32, np.int64][int(rng.integers(0, 6))]
        info = np.iinfo(dt)
        hi = [1, 15, int(info.max)][int(rng.integers(0, 3))]
        small = rng.integers(max(info.min, -hi), hi, size=n, dtype=dt, endpoint=True)
        assert orc.g1_eq(srs.msm_small(small), OB.msm_small(ref[:n], small)), ("msm_small", n, dt)
    for _ in range(3):
        n = int(rng.integers(1, 12))
        poly, pt = _special_fr(orc, 1 << n, rng, int(rng.integers(1 << 30))), orc.random_fr(n, 2)
        claim = orc.evaluate(poly, pt)
        t_o = orc.new_transcript(b"fz")
        rows_o, ch_o = OR.dense_opening(poly, pt).prove(claim, t_o)
        inst = I.dense_opening(A.MultilinearPolynomial.from_fr(poly), pt)
        t_g = A.Blake2bTranscript(b"fz")
        rows_g, ch_g = inst.prove(claim, t_g)
        assert ch_g == ch_o and _rows_equal(rows_g, rows_o) and t_g.state == t_o.state_bytes(), ("dense", n)
        inst.free()
        log_K, log_T = int(rng.integers(1, 9)), int(rng.integers(1, 10))
        idx = rng.integers(0, 1 << log_K, size=1 << log_T).astype(np.int32)
        if rng.random() < 0.3:
            idx[:] = -1
        ra_, rc_ = orc.random_fr(log_K, 5), orc.random_fr(log_T, 6)
        t_o = orc.new_transcript(b"fz")
        rows_o, ch_o = OR.onehot_opening(idx, log_K, ra_, rc_).prove(claim, t_o)
        inst = I.onehot_opening(idx, log_K, ra_, rc_)
        t_g = A.Blake2bTranscript(b"fz")
        rows_g, ch_g = inst.prove(claim, t_g)
        assert ch_g == ch_o and _rows_equal(rows_g, rows_o) and t_g.state == t_o.state_bytes(), ("onehot", log_K, log_T)
        inst.free()
    srs.free()


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_batched_mixes(atlas, seed):
    """BatchedSumcheck::prove over random mixes of instance families with different round counts and degrees."""
    from oracle import orc, orc_ra as OR, orc_batched as OB
    from jolt_atlas_amd import instances as I
    A = atlas
    rng = np.random.default_rng(4000 + seed + OFF)
    P = A.MultilinearPolynomial.from_fr
    n_inst = int(rng.integers(2, 6))
    gpu, oc, claims, keep = [], [], [], []
    for _ in range(n_inst):
        fam = int(rng.integers(0, 6))
        s = int(rng.integers(1 << 30))
        claim = orc.random_fr(1, s)[0]
        if fam == 0:                                   # element-wise operator
            op = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 9]))
            n_ops = {0: 2, 1: 2, 2: 1, 3: 1, 4: 3, 5: 2, 6: 1, 7: 4, 9: 2}[op]
            nv = int(rng.integers(1, 8))
            ops = [orc.random_fr(1 << nv, s + i) for i in range(n_ops)]
            r = orc.random_fr(nv, s + 9)
            polys = [P(v) for v in ops]; keep += polys
            gpu.append(I.elementwise(op, polys, r)); oc.append(OB.ra_instance(OR.elementwise(op, ops, r), claim))
        elif fam == 1:                                 # softmax family
            kind = int(rng.integers(0, 4))
            log_K = 0 if kind == 3 else int(rng.integers(1, 4)); log_N = int(rng.integers(1, 5))
            a = orc.random_fr(1 << (log_K + log_N), s)
            b = orc.random_fr(1 << (log_K + log_N), s + 1) if kind == 1 else orc.random_fr(1 << log_K, s + 1) if kind == 2 else None
            r = orc.random_fr(log_K + (log_N if kind == 2 else 0), s + 2) if kind != 3 else None
            pa = P(a); pb = P(b) if b is not None else None; keep += [pa] + ([pb] if pb is not None else [])
            gpu.append(I.softmax_instance(kind, pa, pb, log_K, log_N, r)); oc.append(OB.ra_instance(OR.softmax(kind, a, b, log_K, log_N, r), claim))
        elif fam == 2:                                 # dense opening
            nv = int(rng.integers(1, 9))
            poly, pt = orc.random_fr(1 << nv, s), orc.random_fr(nv, s + 1)
            gpu.append(I.dense_opening(P(poly), pt)); oc.append(OB.ra_instance(OR.dense_opening(poly, pt), claim))
        elif fam == 3:                                 # hamming weight (host-only instance)
            d, log_k = int(rng.integers(1, 5)), int(rng.choice([1, 2, 4]))
            G, gp = orc.random_fr(d << log_k, s).reshape(d, 1 << log_k, 4), orc.random_fr(d, s + 1)
            gpu.append(I.hamming_weight(G, log_k, gp)); oc.append(OB.ra_instance(OR.hamming(G, log_k, gp), claim))
        elif fam == 4:                                 # RaVirtual
            d, log_k, log_T = int(rng.integers(1, 6)), 2, int(rng.integers(1, 7))
            H = [rng.integers(-1, 1 << log_k, size=1 << log_T).astype(np.int32) for _ in range(d)]
            chunks, rc = orc.random_fr(d * log_k, s).reshape(d, log_k, 4), orc.random_fr(log_T, s + 1)
            gpu.append(I.ra_virtual(H, log_k, chunks, rc)); oc.append(OB.ra_instance(OR.ra_virtual(H, log_k, chunks, rc), claim))
        else:                                          # the dot prover through its adapter
            nv = int(rng.integers(1, 9))
            L, R = orc.random_fr(1 << nv, s), orc.random_fr(1 << nv, s + 1)
            claim = orc.dot_claim(L, R)[0]
            gpu.append(A.EinsumDotProver(P(L), P(R))); oc.append(OB.dot_instance(L, R, claim))
        claims.append(claim)
    t_o = orc.new_transcript(b"fz_batched")
    rows_o, ch_o, _ = OB.batched_prove(oc, t_o)
    t_g = A.Blake2bTranscript(b"fz_batched")
    rows_g, ch_g = A.BatchedSumcheck.prove(gpu, claims, t_g)
    assert ch_g == ch_o and _rows_equal(rows_g, rows_o) and t_g.state == t_o.state_bytes()
    for x in gpu:
        x.free()
    for p_ in keep:
        p_.free()


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_hyperkzg_open_degenerate_polynomials(atlas, seed):
    """HyperKZG::open of zero, constant, one-nonzero-coefficient and random polynomials against the oracle (points at
    infinity in the transcript, all-zero MSM vectors)."""
    from oracle import orc
    A = atlas
    rng = np.random.default_rng(5000 + seed + OFF)
    tau = orc.random_fr(1, 11)[0]
    for _ in range(3):
        ell = int(rng.integers(1, 10))
        n = 1 << ell
        kind = int(rng.integers(0, 4))
        poly = orc.random_fr(n, int(rng.integers(1 << 30)))
        if kind == 0:
            poly[:] = 0
        elif kind == 1:
            poly[:] = poly[0]
        elif kind == 2:
            keep = poly[int(rng.integers(0, n))].copy(); poly[:] = 0; poly[int(rng.integers(0, n))] = keep
        pt = [int.from_bytes(rng.bytes(16), "little") for _ in range(ell)]
        srs = A.SRS.generate(tau, n)
        ref = srs.download()
        t_o = orc.new_transcript(b"fz_open")
        com_o, w_o, v_o = orc.hyperkzg_open(ref, poly, pt, t_o)
        p = A.MultilinearPolynomial.from_fr(poly)
        t_g = A.Blake2bTranscript(b"fz_open")
        com_g, w_g, v_g = A.HyperKZG.open(srs, p, pt, t_g)
        assert np.array_equal(v_g, v_o), (ell, kind)
        for a, b in list(zip(com_g, com_o)) + list(zip(w_g, w_o)):
            assert orc.g1_eq(a, b), (ell, kind)
        assert t_g.state == t_o.state_bytes(), (ell, kind)
        assert orc.g1_eq(A.HyperKZG.commit(srs, p), orc.msm(ref, poly))
        p.free(); srs.free()
