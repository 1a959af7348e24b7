"""CPU: oracle/ra.c (Gruen split-eq, product grid + Toom interpolation, expanding table) against
the naive full-table Python models in oracle/pymodel/ra.py."""
import numpy as np
import pytest

from oracle import orc, orc_ra as OR
from oracle.pymodel import field as F, ra as PR, sumcheck as PS
from oracle.pymodel.transcript import Blake2bTranscript


def _rand(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return [int.from_bytes(rng.bytes(40), "little") % F.FR for _ in range(n)]


def _indices(d, T, K, seed, none_frac=0.15):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(d):
        a = rng.integers(0, K, size=T).astype(np.int32)
        a[rng.random(T) < none_frac] = -1
        out.append(a)
    return out


def _prove_py(model, claim, label):
    t = Blake2bTranscript(label)
    rows, rs, raw, last = PS.prove(model, claim, t)
    return rows, raw, t


@pytest.mark.parametrize("d,log_k,log_T", [(1, 2, 3), (2, 4, 4), (3, 4, 5), (4, 2, 1), (8, 4, 3), (16, 4, 2), (5, 3, 6)])
def test_ra_virtual_oracle_matches_naive_model(d, log_k, log_T):
    T, K = 1 << log_T, 1 << log_k
    H = _indices(d, T, K, 7 * d + log_T)
    chunks = [_rand(log_k, 100 + i) for i in range(d)]
    r_cycle = _rand(log_T, 55)
    model = PR.RaVirtualModel([list(map(int, h)) for h in H], chunks, r_cycle)
    claim = model.input_claim()
    rows_p, raw_p, tp = _prove_py(model, claim, b"ra_virtual")
    inst = OR.ra_virtual(H, log_k, np.stack([orc.from_ints(c) for c in chunks]), orc.from_ints(r_cycle))
    to = orc.new_transcript(b"ra_virtual")
    rows_o, raw_o = inst.prove(orc.from_ints([claim])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state


@pytest.mark.parametrize("d,log_k,log_T", [(1, 1, 1), (2, 4, 3), (3, 2, 4), (4, 4, 2), (8, 4, 1)])
def test_booleanity_oracle_matches_naive_model(d, log_k, log_T):
    T, K = 1 << log_T, 1 << log_k
    H = _indices(d, T, K, 11 * d + log_T)
    gammas = [g >> 130 for g in _rand(d, 3)]          # challenge-sized values
    r_address, r_cycle = _rand(log_k, 4), _rand(log_T, 5)
    model = PR.BooleanityModel([list(map(int, h)) for h in H], log_k, gammas, r_address, r_cycle)
    rows_p, raw_p, tp = _prove_py(model, 0, b"booleanity")
    G = OR.ra_G(H, log_k, orc.from_ints(r_cycle))
    inst = OR.booleanity(G, H, log_k, orc.from_ints(gammas), orc.from_ints(r_address), orc.from_ints(r_cycle))
    to = orc.new_transcript(b"booleanity")
    rows_o, raw_o = inst.prove(orc.from_ints([0])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state


@pytest.mark.parametrize("d,log_k", [(1, 1), (2, 4), (8, 4), (3, 8)])
def test_hamming_oracle_matches_naive_model(d, log_k):
    K = 1 << log_k
    G = [_rand(K, 20 + i) for i in range(d)]
    gp = _rand(d, 9)
    claim = sum(g * sum(row) for g, row in zip(gp, G)) % F.FR
    model = PR.HammingModel(G, gp)
    rows_p, raw_p, tp = _prove_py(model, claim, b"hamming")
    inst = OR.hamming(np.stack([orc.from_ints(g) for g in G]), log_k, orc.from_ints(gp))
    to = orc.new_transcript(b"hamming")
    rows_o, raw_o = inst.prove(orc.from_ints([claim])[0], to)
    assert raw_o == raw_p
    assert [orc.to_ints(r) for r in rows_o] == rows_p
    assert bytes(to.state) == tp.state


def test_degenerate_all_none_indices_trim_to_zero_polys():
    d, log_k, log_T = 2, 2, 2
    H = [np.full(4, -1, dtype=np.int32) for _ in range(d)]
    chunks = [_rand(log_k, 1), _rand(log_k, 2)]
    r_cycle = _rand(log_T, 3)
    model = PR.RaVirtualModel([list(map(int, h)) for h in H], chunks, r_cycle)
    rows_p, raw_p, tp = _prove_py(model, 0, b"zero")
    inst = OR.ra_virtual(H, log_k, np.stack([orc.from_ints(c) for c in chunks]), orc.from_ints(r_cycle))
    to = orc.new_transcript(b"zero")
    rows_o, raw_o = inst.prove(orc.from_ints([0])[0], to)
    assert [orc.to_ints(r) for r in rows_o] == rows_p and raw_o == raw_p
    assert all(r == [0] for r in rows_p)
