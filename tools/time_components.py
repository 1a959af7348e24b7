"""Wall-clock timings of the hot-path components beyond the headline sumcheck, at SURVEY §8(d)
sizes, through the C-ABI (inputs resident where the API allows; each figure says what it includes).
Usage (GPU box): python tools/time_components.py [--out profiles/r01c_components.json]"""
import argparse
import json
import sys
import time

import numpy as np

import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import jolt_atlas_amd as A  # noqa: E402
from jolt_atlas_amd import instances as I, rlc  # noqa: E402


def timed(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    A.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); A.sync(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--log-n", type=int, default=22)
    ap.add_argument("--log-t", type=int, default=20)
    a = ap.parse_args()
    A.init(0)
    out = {}
    n = 1 << a.log_n
    rng = np.random.default_rng(1)
    srs = A.SRS.generate(A.random_fr(1, 1)[0], n)

    # ---- MSM variants (SURVEY §8d M(n) a/b/c)
    fr = A.MultilinearPolynomial.from_fr(A.random_fr(n, 2))
    out["msm_fr_ms"] = timed(lambda: srs.msm(fr))
    u20 = rng.integers(0, 1 << 20, size=n, dtype=np.uint32)
    out["msm_u32_lt_2^20_ms_incl_h2d"] = timed(lambda: srs.msm_small(u20))
    i14 = A.MultilinearPolynomial.from_i32(rng.integers(-(1 << 14), 1 << 14, size=n, dtype=np.int32))
    out["msm_i32_pm2^14_resident_ms"] = timed(lambda: srs.msm(i14))
    u8 = rng.integers(0, 256, size=n, dtype=np.uint8)
    out["msm_u8_ms_incl_h2d"] = timed(lambda: srs.msm_small(u8))
    T = n >> 4
    oh = (rng.integers(0, 16, size=T, dtype=np.uint32) * T + np.arange(T, dtype=np.uint32)).astype(np.uint32)
    out["onehot_commit_T=n/16_ms_incl_h2d"] = timed(lambda: srs.sum_indexed(oh))
    fr.free(); i14.free()

    # ---- build_materialized_rlc: 2 dense Fr + 1 dense i32 of 2^log_n, 8 one-hot polys K=16, T=n/16
    d1, d2 = A.MultilinearPolynomial.from_fr(A.random_fr(n, 3)), A.MultilinearPolynomial.from_fr(A.random_fr(n, 4))
    d3 = A.MultilinearPolynomial.from_i32(rng.integers(-(1 << 14), 1 << 14, size=n, dtype=np.int32))
    co = A.random_fr(16, 5)
    ohs = [(rng.integers(0, 16, size=T, dtype=np.int32), 16, co[3 + j]) for j in range(8)]

    def do_rlc():
        j = rlc.build_materialized_rlc([(d1, co[0]), (d2, co[1]), (d3, co[2])], ohs)
        j.free()
    out["rlc_3dense_8onehot_ms_incl_index_h2d"] = timed(do_rlc)
    out["rlc_bytes_algorithmic"] = int(n * (32 + 32 + 4) + n * 32 + 8 * T * (4 + 64))
    d1.free(); d2.free(); d3.free()

    # ---- host-stepped instances at T = 2^log_t
    Tt = 1 << a.log_t
    for d in (8, 16):
        H = [rng.integers(0, 16, size=Tt, dtype=np.int32) for _ in range(d)]
        chunks = A.random_fr(d * 4, 6).reshape(d, 4, 4); rc = A.random_fr(a.log_t, 7)
        gam = A.random_fr(d, 8); radr = A.random_fr(4, 9)

        def ra_once():
            inst = I.ra_virtual(H, 4, chunks, rc)
            t0 = time.perf_counter()
            inst.prove(A.random_fr(1, 1)[0], A.Blake2bTranscript(b"t"))
            dt = time.perf_counter() - t0
            inst.free()
            return dt
        ra_once()
        out[f"ra_virtual_d{d}_T2^{a.log_t}_prove_ms"] = 1e3 * float(np.median([ra_once() for _ in range(3)]))

        if d == 16:                                        # the same at T = 2^16 and 2^18 (where the split product of d = 16 starts to pay: tools/time_ra_split.py)
            for lt in (16, 18):
                Hs = [h[:1 << lt] for h in H]; rcs = A.random_fr(lt, 7)

                def ra_small():
                    inst = I.ra_virtual(Hs, 4, chunks, rcs)
                    t0 = time.perf_counter()
                    inst.prove(A.random_fr(1, 1)[0], A.Blake2bTranscript(b"t"))
                    dt = time.perf_counter() - t0
                    inst.free()
                    return dt
                ra_small()
                out[f"ra_virtual_d16_T2^{lt}_prove_ms"] = 1e3 * float(np.median([ra_small() for _ in range(5)]))

        G = np.zeros((d, 16, 4), dtype=np.uint64)

        def bool_once():
            inst = I.booleanity(G, H, 4, gam, radr, rc)
            t0 = time.perf_counter()
            inst.prove(np.zeros(4, dtype=np.uint64), A.Blake2bTranscript(b"t"))
            dt = time.perf_counter() - t0
            inst.free()
            return dt
        bool_once()
        out[f"booleanity_d{d}_T2^{a.log_t}_prove_ms"] = 1e3 * float(np.median([bool_once() for _ in range(3)]))

    # ---- opening reduction + evaluation reduction
    pol = A.random_fr(n, 10); pt = A.random_fr(a.log_n, 11)

    def dense_once():
        inst = I.dense_opening(A.MultilinearPolynomial.from_fr(pol), pt)
        A.sync()
        t0 = time.perf_counter()
        inst.prove(A.random_fr(1, 1)[0], A.Blake2bTranscript(b"t"))
        dt = time.perf_counter() - t0
        inst.free()
        return dt
    dense_once()
    out[f"dense_opening_2^{a.log_n}_prove_ms"] = 1e3 * float(np.median([dense_once() for _ in range(3)]))
    idx = rng.integers(0, 16, size=Tt, dtype=np.int32)

    def oh_once():
        t0 = time.perf_counter()
        inst = I.onehot_opening(idx, 4, A.random_fr(4, 1), A.random_fr(a.log_t, 2))
        inst.prove(A.random_fr(1, 1)[0], A.Blake2bTranscript(b"t"))
        dt = time.perf_counter() - t0
        inst.free()
        return dt
    oh_once()
    out[f"onehot_opening_K16_T2^{a.log_t}_new+prove_ms"] = 1e3 * float(np.median([oh_once() for _ in range(3)]))
    m20 = A.MultilinearPolynomial.from_fr(A.random_fr(Tt, 12))
    pts = A.random_fr(3 * a.log_t, 13).reshape(3, a.log_t, 4)
    cl = A.random_fr(3, 14)
    out[f"eval_reduction_N3_2^{a.log_t}_ms"] = timed(lambda: I.eval_reduction_prove(m20, pts, cl, A.Blake2bTranscript(b"t")))
    m20.free()

    # ---- batched sumcheck of 4 dot instances 2^20 (host-stepped driver) vs fused single-instance driver
    def batched_once():
        P = A.MultilinearPolynomial.from_fr
        insts = [A.EinsumDotProver(P(pol[:Tt]), P(pol[Tt:2 * Tt])) for _ in range(4)]
        A.sync()
        t0 = time.perf_counter()
        A.BatchedSumcheck.prove(insts, [np.zeros(4, dtype=np.uint64)] * 4, A.Blake2bTranscript(b"t"))
        dt = time.perf_counter() - t0
        for x in insts:
            x.free()
        return dt
    batched_once()
    out[f"batched_4xdot_2^{a.log_t}_prove_ms"] = 1e3 * float(np.median([batched_once() for _ in range(3)]))

    # ---- prefix-suffix Shout (ReLU, X_LEN = 32) and the identity range check at T = 2^log_t
    act = (rng.integers(-(1 << 14), 1 << 14, size=Tt, dtype=np.int64) & 0xffffffff).astype(np.uint64)
    rn = A.random_fr(a.log_t, 21); gm = A.random_fr(1, 22)[0]

    def ps_once():
        t0 = time.perf_counter()
        inst = I.ps_shout_relu(act, 32, rn, gm)
        t1 = time.perf_counter()
        inst.prove(A.random_fr(1, 1)[0], A.Blake2bTranscript(b"t"))
        t2 = time.perf_counter()
        inst.free()
        return t1 - t0, t2 - t1
    ps_once()
    r = [ps_once() for _ in range(3)]
    out[f"ps_shout_relu32_T2^{a.log_t}_new_ms_incl_h2d"] = 1e3 * float(np.median([x[0] for x in r]))
    out[f"ps_shout_relu32_T2^{a.log_t}_prove_ms"] = 1e3 * float(np.median([x[1] for x in r]))

    def idrc_once():
        inst = I.identity_range_check(act & np.uint64(0xffff), 16, 8, rn)
        t1 = time.perf_counter()
        inst.prove(A.random_fr(1, 1)[0], A.Blake2bTranscript(b"t"))
        t2 = time.perf_counter()
        inst.free()
        return t2 - t1
    idrc_once()
    out[f"identity_rc_logK16_T2^{a.log_t}_prove_ms"] = 1e3 * float(np.median([idrc_once() for _ in range(3)]))

    # ---- prove_reduced_openings: 16 one-hot polynomials (K = 16, T = 2^16, one r_cycle) + 1 dense 2^20, SRS 2^20
    from jolt_atlas_amd import reduced
    srs20 = A.SRS.generate(A.random_fr(1, 1)[0], 1 << 20)
    rcy = A.random_fr(16, 31)
    ops = [dict(poly=A.MultilinearPolynomial.from_fr(A.random_fr(1 << 20, 32)), point=A.random_fr(20, 33), claim=A.random_fr(1, 34)[0])]
    for q in range(16):
        ops.append(dict(k=rng.integers(0, 16, size=1 << 16, dtype=np.int32), log_K=4, r_address=A.random_fr(4, 35 + q), r_cycle=rcy,
                        claim=A.random_fr(1, 60 + q)[0]))
    out["prove_reduced_openings_16onehot_T2^16_1dense_2^20_ms"] = timed(lambda: reduced.prove_reduced_openings(ops, srs20, A.Blake2bTranscript(b"t")), reps=3)
    print(json.dumps(out, indent=1))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
