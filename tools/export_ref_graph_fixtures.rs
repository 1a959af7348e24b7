//! export_ref_graph_fixtures — pins the OPERATOR COMPOSITIONS and the ONNXProof container against a run of the reference
//! (ICME-Lab/jolt-atlas): whole `ONNXProof::prove` runs over small models built with the reference's own `ModelBuilder`.
//!
//! NOT part of the product and NOT compiled in this repository (no Rust toolchain in the build image: this file has never been
//! through rustc — expect to fix a `use` path or two).  With the reference checked out:
//!
//!     cp tools/export_ref_graph_fixtures.rs <reference>/jolt-atlas-core/examples/
//!     cargo run --release -p jolt-atlas-core --example export_ref_graph_fixtures > ref_graph_fixtures.json
//!
//! then drop the file at `tests/golden/ref_graph_fixtures.json`.  `tests/test_reference_fixtures.py` proves every model of the file
//! with the CPU oracle composition (`oracle/graph.py`) and, on a GPU, with `atlas_prove_graph`, over the SRS powers the file carries,
//! and compares PROOF BYTES (`serialize_proof`, proof_serialization.rs:285-296) — one `cargo run` then pins fused-rescale Einsum / Mul,
//! ReLU, Add, the clamp lookup, the one-hot checks, NodeEvalReduction, the reduced openings and the container's tag encodings.
//!
//! Each model appears twice: built through `ModelBuilder` (what the reference proves) and as the node list in this repository's
//! graph vocabulary (`jolt-atlas_amd/graph.py`: what the library is given).  `pretty` is the reference's own print of the model,
//! to see at a glance whether the two descriptions drifted apart.
use ark_bn254::{Bn254, Fr};
use ark_serialize::CanonicalSerialize;
use atlas_onnx_tracer::{model::{test::ModelBuilder, Model}, tensor::Tensor};
use jolt_atlas_core::onnx_proof::{
    proof_serialization::serialize_proof, AtlasProverPreprocessing, AtlasSharedPreprocessing, AtlasVerifierPreprocessing, ONNXProof,
};
use joltworks::{poly::commitment::hyperkzg::HyperKZG, transcripts::Blake2bTranscript};
use rand::{rngs::StdRng, Rng, SeedableRng};

fn hex(bytes: &[u8]) -> String {
    bytes.iter().map(|b| format!("{b:02x}")).collect()
}
fn ser<T: CanonicalSerialize>(x: &T) -> String {
    let mut v = Vec::new();
    x.serialize_compressed(&mut v).unwrap();
    hex(&v)
}
fn ints(v: &[i32]) -> String {
    format!("[{}]", v.iter().map(|x| x.to_string()).collect::<Vec<_>>().join(","))
}
fn rnd(rng: &mut StdRng, n: usize, lim: i32) -> Vec<i32> {
    (0..n).map(|_| rng.gen_range(-lim..lim)).collect()
}

/// prove, verify (the reference accepts its own proof), print one JSON object
fn emit(name: &str, model: Model, inputs: Vec<Tensor<i32>>, nodes_json: &str, outputs_json: &str, last: bool) {
    let pretty = model.pretty_print();
    let pp = AtlasSharedPreprocessing::preprocess(model);
    let ppp = AtlasProverPreprocessing::<Fr, HyperKZG<Bn254>>::new(pp);
    let (proof, io, debug_info) = ONNXProof::<Fr, Blake2bTranscript, HyperKZG<Bn254>>::prove(&ppp, &inputs);
    let vpp = AtlasVerifierPreprocessing::<Fr, HyperKZG<Bn254>>::from(&ppp);
    proof.verify(&vpp, &io, debug_info).expect("the reference rejects its own proof");
    let bytes = serialize_proof(&proof).expect("serialize_proof");
    let g1: Vec<String> = ppp.generators.kzg_pk.g1_powers().iter().map(|p| format!("\"{}\"", ser(p))).collect();
    println!("  {{\"name\": \"{name}\",");
    println!("   \"pretty\": {:?},", pretty);
    println!("   \"nodes\": {nodes_json},");
    println!("   \"outputs\": {outputs_json},");
    println!("   \"inputs\": [{}],", inputs.iter().map(|t| ints(t.data())).collect::<Vec<_>>().join(","));
    println!("   \"output\": {},", ints(io.outputs[0].data()));
    println!("   \"srs_g1\": [{}],", g1.join(","));
    println!("   \"proof\": \"{}\"}}{}", hex(&bytes), if last { "" } else { "," });
}

fn main() {
    let mut rng = StdRng::seed_from_u64(0xA71A5);
    println!("{{\"graphs\": [");

    // 1. ReLU over 16 activations: the unary prefix-suffix lookup + one-hot checks, output claim, NodeEvalReduction, reduced openings
    {
        let mut b = ModelBuilder::new();
        let x = b.input(vec![16]);
        let y = b.relu(x);
        b.mark_output(y);
        let inp = Tensor::new(Some(&rnd(&mut rng, 16, 1 << 12)), &[16]).unwrap();
        emit("relu16", b.build(), vec![inp],
             r#"[{"idx":0,"op":"Input","inputs":[],"dims":[16]},{"idx":1,"op":"ReLU","inputs":[0],"dims":[16]}]"#, "[1]", false);
    }
    // 2. Add of an input and a constant: the 64-bit saturating clamp lookup over the accumulation
    {
        let c = rnd(&mut rng, 16, 1 << 20);
        let mut b = ModelBuilder::new();
        let x = b.input(vec![4, 4]);
        let k = b.constant(Tensor::new(Some(&c), &[4, 4]).unwrap());
        let y = b.add(x, k);
        b.mark_output(y);
        let inp = Tensor::new(Some(&rnd(&mut rng, 16, 1 << 20)), &[4, 4]).unwrap();
        let nodes = format!(r#"[{{"idx":0,"op":"Input","inputs":[],"dims":[4,4]}},{{"idx":1,"op":"Constant","inputs":[],"dims":[4,4],"data":{}}},{{"idx":2,"op":"Add","inputs":[0,1],"dims":[4,4]}}]"#, ints(&c));
        emit("add4x4", b.build(), vec![inp], &nodes, "[2]", false);
    }
    // 3. fused-rescale Einsum (mk,kn->mn at MODEL_SCALE) followed by ReLU: remainder advice, clamp lookup, contraction sumcheck,
    //    remainder range check, both one-hot checks
    {
        let w = rnd(&mut rng, 8 * 16, 1 << 12);
        let mut b = ModelBuilder::new();
        let x = b.input(vec![4, 8]);
        let k = b.constant(Tensor::new(Some(&w), &[8, 16]).unwrap());
        let y = b.einsum("mk,kn->mn", vec![x, k], vec![4, 16]);
        let z = b.relu(y);
        b.mark_output(z);
        let inp = Tensor::new(Some(&rnd(&mut rng, 32, 1 << 14)), &[4, 8]).unwrap();
        let nodes = format!(r#"[{{"idx":0,"op":"Input","inputs":[],"dims":[4,8]}},{{"idx":1,"op":"Constant","inputs":[],"dims":[8,16],"data":{}}},{{"idx":2,"op":"Einsum","inputs":[0,1],"dims":[4,16],"layout":"mk,kn->mn","scale":14,"shape":[4,8,16]}},{{"idx":3,"op":"ReLU","inputs":[2],"dims":[4,16]}}]"#, ints(&w));
        emit("einsum_relu", b.build(), vec![inp], &nodes, "[3]", false);
    }
    // 4. Mul with fused rescale of an input by a constant
    {
        let c = rnd(&mut rng, 32, 1 << 14);
        let mut b = ModelBuilder::new();
        let x = b.input(vec![4, 8]);
        let k = b.constant(Tensor::new(Some(&c), &[4, 8]).unwrap());
        let y = b.mul(x, k);
        b.mark_output(y);
        let inp = Tensor::new(Some(&rnd(&mut rng, 32, 1 << 14)), &[4, 8]).unwrap();
        let nodes = format!(r#"[{{"idx":0,"op":"Input","inputs":[],"dims":[4,8]}},{{"idx":1,"op":"Constant","inputs":[],"dims":[4,8],"data":{}}},{{"idx":2,"op":"Mul","inputs":[0,1],"dims":[4,8],"scale":14}}]"#, ints(&c));
        emit("mul4x8", b.build(), vec![inp], &nodes, "[2]", true);
    }
    println!("]}}");
}
