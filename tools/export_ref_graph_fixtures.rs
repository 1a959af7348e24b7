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
//! ReLU, Add, the clamp lookup, the one-hot checks, NodeEvalReduction, the reduced openings and the container's tag encodings, and (round 5)
//! SoftmaxLastAxis, Tanh, GatherSmall, Div (also over one element), Rsqrt and a LayerNorm-shaped chain (Sum, ScalarConstDiv, Broadcast, Sub,
//! MeanOfSquares, Rsqrt, Mul) — eleven models, the operator set of the nanoGPT / GPT-2 graphs.
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
        emit("mul4x8", b.build(), vec![inp], &nodes, "[2]", false);
    }
    // ---- round 5: the operator families a transformer adds (review of round 4, item 7): one `cargo run` then also pins SoftmaxLastAxis,
    // Tanh, GatherSmall, Div, Rsqrt, MeanOfSquares / Sum / ScalarConstDiv, the ReductionFlow::Custom order and the one-element Div.
    // 5. SoftmaxLastAxis over [2, 4, 8] logits (ops/softmax_last_axis/mod.rs:177-262): the 3 F auxiliary scalars, the four batched stages
    {
        let mut b = ModelBuilder::new();
        let x = b.input(vec![2, 4, 8]);
        let y = b.softmax_last_axis(x);
        b.mark_output(y);
        let inp = Tensor::new(Some(&rnd(&mut rng, 64, 1 << 15)), &[2, 4, 8]).unwrap();
        emit("softmax2x4x8", b.build(), vec![inp],
             r#"[{"idx":0,"op":"Input","inputs":[],"dims":[2,4,8]},{"idx":1,"op":"SoftmaxLastAxis","inputs":[0],"dims":[2,4,8],"scale":14}]"#, "[1]", false);
    }
    // 6. Tanh over 16 activations spanning the clamp bound (ops/tanh.rs + activation_clamped/mod.rs): clamp lookup, small-table read, one batch of one-hot checks
    {
        let mut b = ModelBuilder::new();
        let x = b.input(vec![4, 4]);
        let y = b.tanh(x);
        b.mark_output(y);
        let inp = Tensor::new(Some(&rnd(&mut rng, 16, 1 << 18)), &[4, 4]).unwrap();
        emit("tanh4x4", b.build(), vec![inp],
             r#"[{"idx":0,"op":"Input","inputs":[],"dims":[4,4]},{"idx":1,"op":"Tanh","inputs":[0],"dims":[4,4],"scale":14}]"#, "[1]", false);
    }
    // 7. GatherSmall: 8 indices into a 16 x 4 dictionary (ops/gather/small.rs): ONE committed one-hot polynomial over all address bits
    {
        let dict = rnd(&mut rng, 16 * 4, 1 << 14);
        let idx: Vec<i32> = (0..8).map(|_| rng.gen_range(0..16)).collect();
        let mut b = ModelBuilder::new();
        let i = b.input(vec![8]);
        let d = b.constant(Tensor::new(Some(&dict), &[16, 4]).unwrap());
        let y = b.gather(d, i, 0, vec![8, 4]);
        b.mark_output(y);
        let inp = Tensor::new(Some(&idx), &[8]).unwrap();
        let nodes = format!(r#"[{{"idx":0,"op":"Input","inputs":[],"dims":[8]}},{{"idx":1,"op":"Constant","inputs":[],"dims":[16,4],"data":{}}},{{"idx":2,"op":"GatherSmall","inputs":[1,0],"dims":[8,4],"axis":0,"dict_len":16}}]"#, ints(&dict));
        emit("gather8of16", b.build(), vec![inp], &nodes, "[2]", false);
    }
    // 8. Div of an input by a positive constant tensor (ops/div.rs, ReductionFlow::Custom): sumcheck at a fresh point, eval reduction, ULT range check
    {
        let den: Vec<i32> = (0..16).map(|_| rng.gen_range(1..(1 << 10))).collect();
        let mut b = ModelBuilder::new();
        let x = b.input(vec![4, 4]);
        let k = b.constant(Tensor::new(Some(&den), &[4, 4]).unwrap());
        let y = b.div(x, k);
        b.mark_output(y);
        let inp = Tensor::new(Some(&rnd(&mut rng, 16, 1 << 20)), &[4, 4]).unwrap();
        let nodes = format!(r#"[{{"idx":0,"op":"Input","inputs":[],"dims":[4,4]}},{{"idx":1,"op":"Constant","inputs":[],"dims":[4,4],"data":{}}},{{"idx":2,"op":"Div","inputs":[0,1],"dims":[4,4]}}]"#, ints(&den));
        emit("div4x4", b.build(), vec![inp], &nodes, "[2]", false);
    }
    // 9. Rsqrt of positive inputs (ops/rsqrt.rs): the committed quotient, its two range checks in one batch
    {
        let mut b = ModelBuilder::new();
        let x = b.input(vec![4, 4]);
        let y = b.rsqrt(x);
        b.mark_output(y);
        let v: Vec<i32> = (0..16).map(|_| rng.gen_range(1..(1 << 20))).collect();
        let inp = Tensor::new(Some(&v), &[4, 4]).unwrap();
        emit("rsqrt4x4", b.build(), vec![inp],
             r#"[{"idx":0,"op":"Input","inputs":[],"dims":[4,4]},{"idx":1,"op":"Rsqrt","inputs":[0],"dims":[4,4],"scale":14}]"#, "[1]", false);
    }
    // 10. Div over ONE element (ops/div.rs:93-160, the scalar branch: the quotient is the only committed polynomial, no range check) inside a
    //     graph that gets there and back: [2,2] -> Sum -> Sum -> s; 70000 / s; Broadcast to [1,2]; Add of a constant
    {
        let mut b = ModelBuilder::new();
        let x = b.input(vec![2, 2]);
        let s1 = b.sum(x, vec![1], vec![2, 1]);
        let s = b.sum(s1, vec![0], vec![1, 1]);
        let k = b.constant(Tensor::new(Some(&[70000]), &[1, 1]).unwrap());
        let q = b.div(k, s);
        let qb = b.broadcast(q, vec![1, 2]);
        let c = b.constant(Tensor::new(Some(&[3, -4]), &[1, 2]).unwrap());
        let y = b.add(qb, c);
        b.mark_output(y);
        let inp = Tensor::new(Some(&[5000, 6000, 7000, 8000]), &[2, 2]).unwrap();
        emit("div1", b.build(), vec![inp],
             r#"[{"idx":0,"op":"Input","inputs":[],"dims":[2,2]},{"idx":1,"op":"Sum","inputs":[0],"dims":[2,1],"axes":[1]},{"idx":2,"op":"Sum","inputs":[1],"dims":[1,1],"axes":[0]},{"idx":3,"op":"Constant","inputs":[],"dims":[1,1],"data":[70000]},{"idx":4,"op":"Div","inputs":[3,2],"dims":[1,1]},{"idx":5,"op":"Broadcast","inputs":[4],"dims":[1,2]},{"idx":6,"op":"Constant","inputs":[],"dims":[1,2],"data":[3,-4]},{"idx":7,"op":"Add","inputs":[5,6],"dims":[1,2]}]"#, "[7]", false);
    }
    // 11. a LayerNorm-shaped chain over [4, 8] (what handlers/ emit for it): mean = Sum / count, centred x, MeanOfSquares, Rsqrt, Mul by the broadcast
    //     reciprocal — Sum, ScalarConstDiv, Broadcast, Sub, MeanOfSquares, Rsqrt, Mul in one transcript
    {
        let mut b = ModelBuilder::new();
        let x = b.input(vec![4, 8]);
        let s = b.sum(x, vec![1], vec![4, 1]);
        let m = b.scalar_const_div(s, 8);
        let mb = b.broadcast(m, vec![4, 8]);
        let c = b.sub(x, mb);
        let v = b.mean_of_squares(c, vec![1], vec![4, 1]);
        let r = b.rsqrt(v);
        let rb = b.broadcast(r, vec![4, 8]);
        let y = b.mul(c, rb);
        b.mark_output(y);
        let inp = Tensor::new(Some(&rnd(&mut rng, 32, 1 << 15)), &[4, 8]).unwrap();
        emit("layernorm4x8", b.build(), vec![inp],
             r#"[{"idx":0,"op":"Input","inputs":[],"dims":[4,8]},{"idx":1,"op":"Sum","inputs":[0],"dims":[4,1],"axes":[1]},{"idx":2,"op":"ScalarConstDiv","inputs":[1],"dims":[4,1],"divisor":8},{"idx":3,"op":"Broadcast","inputs":[2],"dims":[4,8]},{"idx":4,"op":"Sub","inputs":[0,3],"dims":[4,8]},{"idx":5,"op":"MeanOfSquares","inputs":[4],"dims":[4,1],"axes":[1],"scale":14,"count":8},{"idx":6,"op":"Rsqrt","inputs":[5],"dims":[4,1],"scale":14},{"idx":7,"op":"Broadcast","inputs":[6],"dims":[4,8]},{"idx":8,"op":"Mul","inputs":[4,7],"dims":[4,8],"scale":14}]"#, "[8]", true);
    }
    println!("]}}");
}
