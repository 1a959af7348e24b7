"""Blake2b Fiat-Shamir transcript, restated from
/root/reference/joltworks/src/transcripts/blake2b.rs (line numbers below).

TEST INFRASTRUCTURE ONLY (oracle).  Uses hashlib.blake2b(digest_size=32) — an
implementation of RFC 7693 independent of both the C oracle and the device code.
"""
import hashlib

from . import field as F


def _h(*parts):
    h = hashlib.blake2b(digest_size=32)
    for p in parts:
        h.update(p)
    return h.digest()


class Blake2bTranscript:
    def __init__(self, label: bytes):
        # blake2b.rs:81-100 : state = H(label || zero-pad to 32), n_rounds = 0
        assert len(label) < 33
        self.state = _h(label + b"\0" * (32 - len(label)))
        self.n_rounds = 0
        self.state_history = [self.state]

    # blake2b.rs:31-37
    def _prefix(self):
        return self.state + b"\0" * 28 + self.n_rounds.to_bytes(4, "big")

    # blake2b.rs:64-78
    def _update(self, new_state):
        self.state = new_state
        self.n_rounds += 1
        self.state_history.append(new_state)

    # blake2b.rs:109-122
    def append_message(self, msg: bytes):
        assert len(msg) < 33
        self._update(_h(self._prefix(), msg + b"\0" * (32 - len(msg))))

    # blake2b.rs:124-128
    def append_bytes(self, b: bytes):
        self._update(_h(self._prefix(), b))

    # blake2b.rs:130-136
    def append_u64(self, x: int):
        self._update(_h(self._prefix(), b"\0" * 24 + int(x).to_bytes(8, "big")))

    # blake2b.rs:138-146
    def append_scalar(self, a: int):
        self.append_bytes(F.fr_be_bytes(a))

    # blake2b.rs:158-164
    def append_scalars(self, xs):
        self.append_message(b"begin_append_vector")
        for x in xs:
            self.append_scalar(x)
        self.append_message(b"end_append_vector")

    # blake2b.rs:166-187 ; pt = None (identity) or (x, y) affine ints
    def append_point(self, pt):
        if pt is None:
            self.append_bytes(b"\0" * 64)
            return
        x, y = pt
        self._update(_h(self._prefix(), int(x).to_bytes(32, "big"), int(y).to_bytes(32, "big")))

    # blake2b.rs:189-195
    def append_points(self, pts):
        self.append_message(b"begin_append_vector")
        for p in pts:
            self.append_point(p)
        self.append_message(b"end_append_vector")

    # blake2b.rs:148-156 with T = G1Affine: serialize_uncompressed = x_le || y_le
    # (infinity flag = bit 6 of the last byte), then fully reversed.
    def append_g1_serializable(self, pt):
        if pt is None:
            buf = bytearray(64)
            buf[63] |= 0x40
        else:
            buf = bytearray(int(pt[0]).to_bytes(32, "little") + int(pt[1]).to_bytes(32, "little"))
        self.append_bytes(bytes(buf[::-1]))

    # blake2b.rs:57-62
    def challenge_bytes32(self):
        d = _h(self._prefix())
        self._update(d)
        return d

    # blake2b.rs:197-202 : first 16 digest bytes, reversed, read big-endian == LE u128
    def challenge_u128(self):
        return int.from_bytes(self.challenge_bytes32()[:16], "little")

    # blake2b.rs:209-215 : reversed then from_le_bytes_mod_order == BE integer
    def challenge_scalar(self):
        return int.from_bytes(self.challenge_bytes32()[:16], "big") % F.FR

    def challenge_vector(self, n):
        return [self.challenge_scalar() for _ in range(n)]

    # blake2b.rs:224-231
    def challenge_scalar_powers(self, n):
        q = self.challenge_scalar()
        out = [1] * n
        for i in range(1, n):
            out[i] = out[i - 1] * q % F.FR
        return out

    # blake2b.rs:233-238 ; returns the raw (unmasked) u128
    def challenge_u128_optimized(self):
        return self.challenge_u128()

    def challenge_scalar_optimized(self):
        return F.challenge_to_fr(self.challenge_u128())

    def challenge_vector_optimized(self, n):
        return [self.challenge_scalar_optimized() for _ in range(n)]
