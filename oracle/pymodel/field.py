"""BN254 scalar field Fr / base field Fq as plain Python integers.

TEST INFRASTRUCTURE ONLY (oracle).  Independent big-int model written from the
public BN254 parameters; used to pin the C oracle (oracle/*.c) and to generate the
golden fixtures under tests/golden/.  Nothing in the product path imports this.

Conventions restated from the reference (paths relative to /root/reference):
  * Fr memory layout = 4 x u64 little-endian limbs of the Montgomery residue
    a*R mod r, R = 2^256            (joltworks/src/field/ark.rs:20-29 transmutes
                                     MontConfig::R / R2 into Fr).
  * serialize = 32-byte little-endian canonical integer (ark-serialize), the
    transcript reverses it to big-endian (transcripts/blake2b.rs:138-146).
  * MontU128Challenge: 125-bit value c stored as limbs [0,0,lo,hi]
    (field/challenge/mont_ark_u128.rs:51-62) and used *as a Montgomery residue*
    (`from_bigint_unchecked` + `mul_hi_bigint_u128`, macros.rs:274-283), i.e. the
    field element is c*2^128*R^-1 = c*2^-128 mod r.  CHALLENGE_MODE selects the
    alternative reading (value c*2^128 mod r) flagged in SURVEY.md App. A.2.
"""

FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617
FQ = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R256 = 1 << 256
MASK64 = (1 << 64) - 1

FR_R = R256 % FR
FR_R2 = (R256 * R256) % FR
FR_RINV = pow(R256, -1, FR)
FQ_R = R256 % FQ
FQ_RINV = pow(R256, -1, FQ)

# "mont" = limbs are the Montgomery residue (default, see module doc);
# "plain" = limbs are the canonical integer c<<128.
CHALLENGE_MODE = "mont"


def to_mont(a, p=FR):
    return (a * R256) % p


def from_mont(m, p=FR):
    return (m * pow(R256, -1, p)) % p


def limbs64(x):
    return [(x >> (64 * i)) & MASK64 for i in range(4)]


def from_limbs64(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def fr_le_bytes(a):
    """ark-serialize of Fr: canonical integer, 32 bytes little-endian."""
    return int(a % FR).to_bytes(32, "little")


def fr_be_bytes(a):
    return int(a % FR).to_bytes(32, "big")


def challenge_mask(c128):
    """MontU128Challenge::new: keep the low 125 bits (mont_ark_u128.rs:55)."""
    return c128 & ((1 << 128) - 1 >> 3)


def challenge_to_fr(c128):
    """Field value of MontU128Challenge::from(c128)."""
    c = challenge_mask(c128)
    big = c << 128  # limbs [0,0,lo,hi]
    if CHALLENGE_MODE == "mont":
        return (big * FR_RINV) % FR
    return big % FR


def challenge_mont_limbs(c128):
    """The 4xu64 Montgomery limbs a device kernel multiplies by."""
    return limbs64(to_mont(challenge_to_fr(c128)))


def inv(a, p=FR):
    return pow(a, -1, p)


def from_i64(v):
    """JoltField::from_i64 (field/ark.rs:127-150): negative -> -(|v|)."""
    return v % FR
