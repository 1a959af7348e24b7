"""Lookup indices cut on the device from operand tensors (compute_lookup_indices_from_operands) and passed as device
memory to the read-raf / one-hot constructors: same indices as the host restatement, same proofs as with host arrays."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _interleave(x, y):
    out = np.zeros(len(x), dtype=np.uint64)
    for b in range(32):
        out |= ((x.astype(np.uint64) >> np.uint64(b)) & np.uint64(1)) << np.uint64(2 * b + 1)
        out |= ((y.astype(np.uint64) >> np.uint64(b)) & np.uint64(1)) << np.uint64(2 * b)
    return out


def _download_u64(A, dev):
    import ctypes as C
    host = np.zeros(dev.n, dtype=np.uint64)
    # a device poly wrapper would do; the runtime copy is enough for a test
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), dev.ptr, C.c_size_t(8 * dev.n), C.c_int(2)) == 0
    return host


@pytest.mark.parametrize("log_T", [3, 12])
def test_indices_from_operands_and_device_resident_constructors(atlas, log_T):
    from oracle import orc, orc_ra as OR
    from jolt_atlas_amd import instances as I
    A = atlas
    T = 1 << log_T
    rng = np.random.default_rng(log_T)
    x = rng.integers(-(1 << 31), 1 << 31, size=T, dtype=np.int64).astype(np.int32)
    y = rng.integers(-(1 << 31), 1 << 31, size=T, dtype=np.int64).astype(np.int32)
    x[0], y[0] = -1, 0
    tx, ty = A.TensorI32(x), A.TensorI32(y)
    d_un = I.DeviceU64.from_operands(tx)
    d_bin = I.DeviceU64.from_operands(tx, ty)
    un = x.astype(np.uint32).astype(np.uint64)
    bn = _interleave(x.astype(np.uint32), y.astype(np.uint32))
    assert np.array_equal(_download_u64(A, d_un), un)
    assert np.array_equal(_download_u64(A, d_bin), bn)
    r_node, gamma, claim = orc.random_fr(log_T, 5), orc.random_fr(1, 6)[0], orc.random_fr(1, 7)[0]

    def same(make_host, make_dev, label):
        a, b = make_host(), make_dev()
        ta, tb = A.Blake2bTranscript(label), A.Blake2bTranscript(label)
        ra, ca = a.prove(claim, ta); rb, cb = b.prove(claim, tb)
        assert ca == cb and ta.state == tb.state and all(np.array_equal(u, v) for u, v in zip(ra, rb))
        assert all(np.array_equal(u, v) for u, v in zip(a.final_claims(), b.final_claims()))
        a.free(); b.free()
        return ra, ca, ta

    rows, ch, t = same(lambda: I.ps_shout_relu(un, 32, r_node, gamma), lambda: I.ps_shout_relu(d_un, 32, r_node, gamma), b"relu")
    t_o = orc.new_transcript(b"relu")
    rows_o, ch_o = OR.ps_relu(un, 32, r_node, gamma).prove(claim, t_o)
    assert ch == ch_o and t.state == t_o.state_bytes() and all(np.array_equal(u, v) for u, v in zip(rows, rows_o))
    same(lambda: I.ps_shout_ult(bn, r_node, gamma), lambda: I.ps_shout_ult(d_bin, r_node, gamma), b"ult")
    r_addr, r_cyc = orc.random_fr(32, 8), orc.random_fr(log_T, 9)
    same(lambda: I.ra_virtual_from_lookups(un, 32, 4, r_addr, r_cyc), lambda: I.ra_virtual_from_lookups(d_un, 32, 4, r_addr, r_cyc), b"ra")
    up = I.DeviceU64.upload(un)
    same(lambda: I.ps_shout_rshift(un, 32, 3, r_node, gamma), lambda: I.ps_shout_rshift(up, 32, 3, r_node, gamma), b"rs")
    for d in (d_un, d_bin, up):
        d.free()
    tx.free(); ty.free()
