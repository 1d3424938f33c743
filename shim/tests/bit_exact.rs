//! The parity test that pins the GPU prover to the reference: the proof bytes of `prove_hip_with_rs(.., r, s)` must equal
//! `Proof::from_bellman(&bellman::groth16::create_proof(bcs, &params.0, r, s))` -- the CPU path behind prover.rs:80 with the
//! same (r, s).  Mirrors /root/reference/fawkes-crypto/tests/bellman_groth16.rs (poseidon merkle proof, depth 32:
//! BASELINE configs[0]).  Needs cargo, an MI355X and libfawkes_hip.so:  cargo test --features heavy_tests
//! COULD NOT BE RUN in the build image (no Rust toolchain): unverified.
#![cfg(feature = "heavy_tests")]
use borsh::BorshSerialize;
use fawkes_crypto::{
    backend::bellman_groth16::{engines::Bn256, prover::Proof, setup::setup, verifier, BellmanCS},
    circuit::cs::{WitnessCS, CS},
    circuit::num::CNum,
    circuit::poseidon::{c_poseidon_merkle_proof_root, CMerkleProof},
    core::signal::Signal,
    core::sizedvec::SizedVec,
    engines::bn256::Fr,
    ff_uint::Num,
    native::poseidon::{poseidon_merkle_proof_root, MerkleProof, PoseidonParams},
    rand::{thread_rng, Rng},
};
use fawkes_crypto_hip::{forget, prove, prove_hip_with_rs, prove_with_rs, HipProver};

fn circuit<C: CS>(public: CNum<C>, secret: (CNum<C>, CMerkleProof<C, 32>)) {
    let poseidon_params = PoseidonParams::<C::Fr>::new(3, 8, 53);
    let res = c_poseidon_merkle_proof_root(&secret.0, &secret.1, &poseidon_params);
    res.assert_eq(&public);
}

#[test]
fn hip_proof_bytes_equal_bellman_create_proof() {
    let params = setup::<Bn256, _, _, _>(circuit);
    let hip = HipProver::new(&[0], &params);

    let mut rng = thread_rng();
    let poseidon_params = PoseidonParams::<Fr>::new(3, 8, 53);
    let leaf: Num<Fr> = rng.gen();
    let sibling = (0..32).map(|_| rng.gen()).collect::<SizedVec<_, 32>>();
    let path = (0..32).map(|_| rng.gen()).collect::<SizedVec<bool, 32>>();
    let proof = MerkleProof { sibling, path };
    let root = poseidon_merkle_proof_root(leaf, &proof, &poseidon_params);
    let (r, s): (Num<Fr>, Num<Fr>) = (rng.gen(), rng.gen());

    // GPU
    let (inputs, got) = prove_hip_with_rs(&params, &hip, &root, &(leaf, proof.clone()), circuit, r, s);

    // CPU reference: the same witness through bellman's create_proof with the same (r, s)
    let ref rcs = params.get_witness_rcs();
    let signal_pub = <CNum<WitnessCS<Fr>> as Signal<_>>::alloc(rcs, Some(&root));
    signal_pub.inputize();
    let signal_sec = <(CNum<WitnessCS<Fr>>, CMerkleProof<WitnessCS<Fr>, 32>) as Signal<_>>::alloc(rcs, Some(&(leaf, proof.clone())));
    circuit(signal_pub, signal_sec);
    let bcs = BellmanCS::<Bn256, WitnessCS<Fr>>::new(rcs.clone());
    let want = Proof::<Bn256>::from_bellman(
        &bellman::groth16::create_proof(bcs, &params.0,
            fawkes_crypto::backend::bellman_groth16::num_to_bellman_fp(r),
            fawkes_crypto::backend::bellman_groth16::num_to_bellman_fp(s)).unwrap());

    assert_eq!(got.try_to_vec().unwrap(), want.try_to_vec().unwrap(), "GPU proof bytes differ from bellman's");

    // the sharded form of the same call: two ranks (the same GPU named twice works on a one-GPU machine; [0, 1] on a node)
    let hip2 = HipProver::new(&[0, 0], &params);
    let (_, got2) = prove_hip_with_rs(&params, &hip2, &root, &(leaf, proof.clone()), circuit, r, s);
    assert_eq!(got2.try_to_vec().unwrap(), want.try_to_vec().unwrap(), "2-rank GPU proof bytes differ from bellman's");
    assert!(verifier::verify(&params.get_vk(), &got, &inputs), "Verifier result should be true");   // tests/bellman_groth16.rs:45-46

    // the reference's OWN signature (prover.rs:63-68): no prover argument, the resident state comes from the process-wide cache
    // keyed by `params` and FK_DEVICES.  `prove_with_rs` is its deterministic twin; `prove` draws r, s like create_random_proof.
    drop(hip); drop(hip2);
    let (_, got3) = prove_with_rs(&params, &root, &(leaf, proof.clone()), circuit, r, s);
    assert_eq!(got3.try_to_vec().unwrap(), want.try_to_vec().unwrap(), "prove_with_rs (cached prover) differs from bellman's");
    let (inputs4, got4) = prove(&params, &root, &(leaf, proof.clone()), circuit);                   // exactly tests/bellman_groth16.rs:43
    assert!(verifier::verify(&params.get_vk(), &got4, &inputs4), "Verifier result should be true");
    forget(&params);
}
