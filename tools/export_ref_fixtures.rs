//! export_ref_fixtures — pins this repository's oracle against a run of the reference (ICME-Lab/jolt-atlas).
//!
//! NOT part of the product and NOT compiled in this repository (the build image has no Rust toolchain, so this
//! file has never been through rustc: expect to fix a `use` path or two).  A maintainer with the reference
//! checked out copies it to `joltworks/examples/export_ref_fixtures.rs` and runs
//!
//!     cargo run --release -p joltworks --example export_ref_fixtures > ref_fixtures.json
//!
//! then drops the file at `tests/golden/ref_fixtures.json` in this repository.  `tests/test_reference_fixtures.py`
//! replays every section through the C oracle (CPU) and through libatlas_hip.so (GPU) and compares bytes; with that
//! file present the three encoding inferences of SURVEY.md App. A (challenge = limbs [0,0,lo,hi] taken as the
//! Montgomery residue, ark-serialize flag bits, transcript byte order) stop being inferences.
//!
//! Everything is printed as hex of `serialize_compressed` bytes (Fr: 32 B LE canonical; G1Affine: 32 B), so the
//! reader needs no knowledge of arkworks' in-memory layout.  Sections:
//!   transcript      every Blake2bTranscript operation the path uses, with the 32-byte state after each
//!   challenge_to_fr MontU128Challenge::from(x) -> Fr for 16 values
//!   sumcheck        the loop of Sumcheck::prove (sumcheck.rs:565-599) over a degree-2 dot product, built from the
//!                   reference's own primitives (sumcheck_evals, UniPoly::from_evals_and_hint, compress,
//!                   append_to_transcript, challenge_scalar_optimized, evaluate, bind_parallel HighToLow)
//!   hyperkzg        SRS slice, commitment, HyperKZG::open proof bytes and transcript state for ell = 4
use ark_bn254::{Bn254, Fr};
use ark_ec::CurveGroup;
use ark_serialize::CanonicalSerialize;
use ark_std::{UniformRand, Zero};
use joltworks::field::challenge::MontU128Challenge;
use joltworks::field::JoltField;
use joltworks::poly::commitment::commitment_scheme::CommitmentScheme;
use joltworks::poly::commitment::hyperkzg::{HyperKZG, HyperKZGProverKey, HyperKZGSRS, HyperKZGVerifierKey};
use joltworks::poly::multilinear_polynomial::{
    BindingOrder, MultilinearPolynomial, PolynomialBinding, PolynomialEvaluation,
};
use joltworks::poly::unipoly::UniPoly;
use joltworks::transcripts::{AppendToTranscript, Blake2bTranscript, Transcript};
use rand_core::SeedableRng;

fn hex(bytes: &[u8]) -> String {
    bytes.iter().map(|b| format!("{b:02x}")).collect()
}
fn ser<T: CanonicalSerialize>(x: &T) -> String {
    let mut v = Vec::new();
    x.serialize_compressed(&mut v).unwrap();
    hex(&v)
}
fn u128_hex(x: u128) -> String {
    hex(&x.to_le_bytes())
}
/// the u128 a MontU128Challenge carries (already masked to 125 bits): value() = [0, 0, lo, hi]
fn chal_u128(c: &MontU128Challenge<Fr>) -> u128 {
    let v = c.value();
    (v[2] as u128) | ((v[3] as u128) << 64)
}

fn main() {
    let mut rng = rand_chacha::ChaCha20Rng::seed_from_u64(0xA71A5);
    println!("{{");

    // ---- transcript
    let mut ops: Vec<String> = Vec::new();
    let mut t = Blake2bTranscript::new(b"ref_fixture");
    ops.push(format!("{{\"op\":\"new\",\"arg\":\"{}\",\"state\":\"{}\"}}", hex(b"ref_fixture"), hex(&t.state)));
    t.append_message(b"hello");
    ops.push(format!("{{\"op\":\"append_message\",\"arg\":\"{}\",\"state\":\"{}\"}}", hex(b"hello"), hex(&t.state)));
    t.append_u64(0xdeadbeef12345678);
    ops.push(format!("{{\"op\":\"append_u64\",\"arg\":\"{}\",\"state\":\"{}\"}}", hex(&0xdeadbeef12345678u64.to_le_bytes()), hex(&t.state)));
    let s0 = Fr::rand(&mut rng);
    t.append_scalar(&s0);
    ops.push(format!("{{\"op\":\"append_scalar\",\"arg\":\"{}\",\"state\":\"{}\"}}", ser(&s0), hex(&t.state)));
    let sv: Vec<Fr> = (0..3).map(|_| Fr::rand(&mut rng)).collect();
    t.append_scalars::<Fr>(&sv);
    ops.push(format!(
        "{{\"op\":\"append_scalars\",\"arg\":\"{}\",\"state\":\"{}\"}}",
        sv.iter().map(ser).collect::<Vec<_>>().join(""),
        hex(&t.state)
    ));
    let g = (ark_bn254::G1Projective::rand(&mut rng)).into_affine();
    t.append_point(&ark_bn254::G1Projective::from(g));
    ops.push(format!("{{\"op\":\"append_point\",\"arg\":\"{}\",\"state\":\"{}\"}}", ser(&g), hex(&t.state)));
    let c = t.challenge_u128();
    ops.push(format!("{{\"op\":\"challenge_u128\",\"out\":\"{}\",\"state\":\"{}\"}}", u128_hex(c), hex(&t.state)));
    let cs: Fr = t.challenge_scalar();
    ops.push(format!("{{\"op\":\"challenge_scalar\",\"out\":\"{}\",\"state\":\"{}\"}}", ser(&cs), hex(&t.state)));
    let co = t.challenge_scalar_optimized::<Fr>();
    let co_f: Fr = co.into();
    ops.push(format!(
        "{{\"op\":\"challenge_scalar_optimized\",\"out_u128\":\"{}\",\"out\":\"{}\",\"state\":\"{}\"}}",
        u128_hex(chal_u128(&co)),
        ser(&co_f),
        hex(&t.state)
    ));
    println!("\"transcript\":[{}],", ops.join(","));

    // ---- MontU128Challenge -> Fr
    let mut rows: Vec<String> = Vec::new();
    let fixed: [u128; 6] = [0, 1, 2, 4, u128::MAX, (1u128 << 125) - 1];
    for i in 0..16 {
        let x: u128 = if i < 6 { fixed[i] } else { (u128::from(rand_core::RngCore::next_u64(&mut rng)) << 64) | u128::from(rand_core::RngCore::next_u64(&mut rng)) };
        let f: Fr = MontU128Challenge::<Fr>::from(x).into();
        rows.push(format!("{{\"u128\":\"{}\",\"fr\":\"{}\"}}", u128_hex(x), ser(&f)));
    }
    // and the product the binds use: challenge * field element (field/challenge/macros.rs:274-283)
    let a = Fr::rand(&mut rng);
    let ch = MontU128Challenge::<Fr>::from(0x0123456789abcdef_fedcba9876543210u128);
    let prod: Fr = a * ch;
    println!(
        "\"challenge_to_fr\":[{}],\"challenge_mul\":{{\"a\":\"{}\",\"u128\":\"{}\",\"product\":\"{}\"}},",
        rows.join(","),
        ser(&a),
        u128_hex(0x0123456789abcdef_fedcba9876543210u128),
        ser(&prod)
    );

    // ---- the loop of Sumcheck::prove over sum_x L(x) R(x), HighToLow
    let n = 6usize;
    let lv: Vec<Fr> = (0..1 << n).map(|_| Fr::rand(&mut rng)).collect();
    let rv: Vec<Fr> = (0..1 << n).map(|_| Fr::rand(&mut rng)).collect();
    let claim: Fr = lv.iter().zip(rv.iter()).map(|(x, y)| *x * *y).sum();
    let mut l = MultilinearPolynomial::from(lv.clone());
    let mut r = MultilinearPolynomial::from(rv.clone());
    let mut tr = Blake2bTranscript::new(b"synthetic_sc");
    tr.append_scalar(&claim); // sumcheck.rs:573-574
    let mut prev = claim;
    let mut round_rows: Vec<String> = Vec::new();
    let mut chals: Vec<String> = Vec::new();
    for _round in 0..n {
        let half = l.len() / 2;
        let (mut e0, mut e2) = (Fr::zero(), Fr::zero());
        for i in 0..half {
            let le = l.sumcheck_evals(i, 2, BindingOrder::HighToLow); // evaluations at 0 and 2
            let re = r.sumcheck_evals(i, 2, BindingOrder::HighToLow);
            e0 += le[0] * re[0];
            e2 += le[1] * re[1];
        }
        let poly = UniPoly::from_evals_and_hint(prev, &[e0, e2]);
        let comp = poly.compress();
        comp.append_to_transcript(&mut tr);
        let r_j = tr.challenge_scalar_optimized::<Fr>();
        prev = poly.evaluate(&r_j);
        l.bind_parallel(r_j, BindingOrder::HighToLow);
        r.bind_parallel(r_j, BindingOrder::HighToLow);
        round_rows.push(format!("\"{}\"", ser(&comp)));
        chals.push(format!("\"{}\"", u128_hex(chal_u128(&r_j))));
    }
    println!(
        "\"sumcheck\":{{\"n\":{},\"left\":\"{}\",\"right\":\"{}\",\"claim\":\"{}\",\"compressed_polys\":[{}],\"challenges\":[{}],\"final_left\":\"{}\",\"final_right\":\"{}\",\"final_claim\":\"{}\",\"state\":\"{}\"}},",
        n,
        lv.iter().map(ser).collect::<Vec<_>>().join(""),
        rv.iter().map(ser).collect::<Vec<_>>().join(""),
        ser(&claim),
        round_rows.join(","),
        chals.join(","),
        ser(&l.get_bound_coeff(0)),
        ser(&r.get_bound_coeff(0)),
        ser(&prev),
        hex(&tr.state)
    );

    // ---- HyperKZG: commit + open at ell = 4 (hyperkzg/tests.rs:19-110 is the shape)
    let ell = 4usize;
    let mut srs_rng = rand_chacha::ChaCha20Rng::seed_from_u64(0);
    let srs = HyperKZGSRS::<Bn254>::setup(&mut srs_rng, 1 << ell);
    let (pk, _vk): (HyperKZGProverKey<Bn254>, HyperKZGVerifierKey<Bn254>) = srs.trim(1 << ell);
    let pv: Vec<Fr> = (0..1 << ell).map(|_| Fr::rand(&mut rng)).collect();
    let poly = MultilinearPolynomial::from(pv.clone());
    let point: Vec<MontU128Challenge<Fr>> = (0..ell)
        .map(|_| MontU128Challenge::<Fr>::from((u128::from(rand_core::RngCore::next_u64(&mut rng)) << 64) | u128::from(rand_core::RngCore::next_u64(&mut rng))))
        .collect();
    let eval = poly.evaluate(&point);
    let com = HyperKZG::<Bn254>::commit(&poly, &pk).0;
    let mut tr = Blake2bTranscript::new(b"TestEval");
    let proof = HyperKZG::<Bn254>::open(&pk, &poly, &point, &mut tr).unwrap();
    println!(
        "\"hyperkzg\":{{\"ell\":{},\"g1_powers\":\"{}\",\"poly\":\"{}\",\"point\":[{}],\"eval\":\"{}\",\"commitment\":\"{}\",\"proof\":\"{}\",\"state\":\"{}\"}}",
        ell,
        pk.kzg_pk.g1_powers().iter().map(ser).collect::<Vec<_>>().join(""),
        pv.iter().map(ser).collect::<Vec<_>>().join(""),
        point.iter().map(|c| format!("\"{}\"", u128_hex(chal_u128(c)))).collect::<Vec<_>>().join(","),
        ser(&eval),
        ser(&com),
        ser(&proof),
        hex(&tr.state)
    );
    println!("}}");
}
