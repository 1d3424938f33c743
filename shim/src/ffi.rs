//! `extern "C"` mirror of include/fawkes_hip.h -- only the entry points the shim uses.
//! Layouts are `#[repr(C)]` images of the C structs; every function returns FK_OK (0) or an error code.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct fk_ctx { _p: [u8; 0] }
#[repr(C)] pub struct fk_key { _p: [u8; 0] }
#[repr(C)] pub struct fk_r1cs_dev { _p: [u8; 0] }
#[repr(C)] pub struct fk_gates { _p: [u8; 0] }
#[repr(C)] pub struct fk_multi { _p: [u8; 0] }
#[repr(C)] pub struct fk_multi_key { _p: [u8; 0] }
#[repr(C)] pub struct fk_multi_r1cs { _p: [u8; 0] }

pub const FK_OK: c_int = 0;
pub const FK_PROOF_BYTES: usize = 256;
pub const FK_Z_EQUAL_SPLIT: f64 = -1.0;
pub const FK_GATES_BROTLI: c_int = 1;

/// `fk_key_desc` (fawkes_hip.h): host pointers to the raw little-endian Montgomery images of bellman's
/// `Parameters { vk, h, l, a, b_g1, b_g2 }` (mod.rs:139), i.e. `into_raw_uncompressed_le` per point (group.rs:57-66,97-103).
#[repr(C)]
pub struct fk_key_desc {
    pub m: u64, pub num_input: u32, pub num_aux: u32,
    pub alpha_g1: *const u8, pub beta_g1: *const u8, pub delta_g1: *const u8,
    pub beta_g2: *const u8, pub delta_g2: *const u8,
    pub h: *const u8, pub n_h: u64,
    pub l: *const u8, pub n_l: u64,
    pub a: *const u8, pub n_a: u64,
    pub b_g1: *const u8, pub b_g2: *const u8, pub n_b: u64,
    pub shard_index: u32, pub shard_count: u32,
    pub z_frac_lo: f64, pub z_frac_hi: f64,
}

extern "C" {
    pub fn fk_init(device_id: c_int, out: *mut *mut fk_ctx) -> c_int;
    pub fn fk_free(ctx: *mut fk_ctx);
    pub fn fk_last_error(ctx: *const fk_ctx) -> *const c_char;
    pub fn fk_key_load(ctx: *mut fk_ctx, desc: *const fk_key_desc, out: *mut *mut fk_key) -> c_int;
    pub fn fk_key_free(ctx: *mut fk_ctx, key: *mut fk_key);
    // the circuit half of Parameters: brotli gate blob -> resident constraint system (decoded once, not per proof)
    pub fn fk_gates_decode(ctx: *mut fk_ctx, blob: *const u8, len: usize, format: c_int, num_gates: u32, num_input: u32,
                           num_aux: u32, out: *mut *mut fk_gates) -> c_int;
    pub fn fk_gates_free(gates: *mut fk_gates);
    pub fn fk_r1cs_load_gates(ctx: *mut fk_ctx, gates: *const fk_gates, out: *mut *mut fk_r1cs_dev) -> c_int;
    pub fn fk_r1cs_free(ctx: *mut fk_ctx, r1cs: *mut fk_r1cs_dev);
    // witness in -> 256-byte Borsh proof out
    pub fn fk_prove_r1cs(ctx: *mut fk_ctx, key: *const fk_key, r1cs: *const fk_r1cs_dev, z: *const u64,
                         r: *const u64, s: *const u64, out_proof: *mut u8, timings: *mut c_void) -> c_int;
    // pipelined form: the upload of proof k+1's witness runs underneath proof k (pinned buffers: fk_host_alloc)
    pub fn fk_host_alloc(ctx: *mut fk_ctx, bytes: usize, hptr: *mut *mut c_void) -> c_int;
    pub fn fk_host_free(ctx: *mut fk_ctx, hptr: *mut c_void) -> c_int;
    pub fn fk_prove_r1cs_submit(ctx: *mut fk_ctx, key: *const fk_key, r1cs: *const fk_r1cs_dev, z: *const u64,
                                r: *const u64, s: *const u64, ticket: *mut c_int) -> c_int;
    pub fn fk_prove_r1cs_wait(ctx: *mut fk_ctx, ticket: c_int, out_proof: *mut u8, timings: *mut c_void) -> c_int;

    // ---- N GPUs of one node behind one call (one process, a worker thread per GPU, exchanges inside the library).
    // fk_init_devices(1, [d]) is the single-GPU prover; the proof bytes do not depend on N.
    pub fn fk_init_devices(n_devices: c_int, device_ids: *const c_int, out: *mut *mut fk_multi) -> c_int;
    pub fn fk_multi_free(multi: *mut fk_multi);
    pub fn fk_multi_last_error(multi: *const fk_multi) -> *const c_char;
    // first contact with a node: out[i * N + j] = FK_PEER_* for copies into rank i's device from rank j's; one verified, timed pull per ordered pair
    pub fn fk_multi_topology(multi: *const fk_multi, out: *mut i32) -> c_int;
    pub fn fk_multi_preflight(multi: *mut fk_multi, bytes: usize, gbps: *mut f64, status: *mut i32, host_events_out: *mut c_int) -> c_int;
    pub fn fk_multi_key_load(multi: *mut fk_multi, desc: *const fk_key_desc, out: *mut *mut fk_multi_key) -> c_int;
    pub fn fk_multi_key_free(multi: *mut fk_multi, key: *mut fk_multi_key);
    pub fn fk_multi_r1cs_load_gates(multi: *mut fk_multi, gates: *const fk_gates, out: *mut *mut fk_multi_r1cs) -> c_int;
    pub fn fk_multi_r1cs_free(multi: *mut fk_multi, r1cs: *mut fk_multi_r1cs);
    pub fn fk_multi_prove_r1cs(multi: *mut fk_multi, key: *const fk_multi_key, r1cs: *const fk_multi_r1cs, z: *const u64,
                               r: *const u64, s: *const u64, out_proof: *mut u8, timings: *mut c_void) -> c_int;
    pub fn fk_multi_prove_r1cs_submit(multi: *mut fk_multi, key: *const fk_multi_key, r1cs: *const fk_multi_r1cs, z: *const u64,
                                      r: *const u64, s: *const u64, ticket: *mut c_int) -> c_int;
    pub fn fk_multi_prove_r1cs_wait(multi: *mut fk_multi, ticket: c_int, out_proof: *mut u8, timings: *mut c_void) -> c_int;

    // verifier (verifier.rs:75-81): vk = fawkes' Borsh `VK`, inputs = Montgomery Fr without the leading ONE, proof = Borsh `Proof`.
    // fk_verify is host code (ctx may be null); the batch form judges `count` proofs of one key on the GPU, accept[i] = 1 / 0
    // (a proof that does not decode is that proof's rejection, the call still returns 0).
    pub fn fk_verify(ctx: *mut fk_ctx, vk: *const u8, vk_len: usize, inputs: *const u64, n_inputs: u32, proof: *const u8, accept: *mut c_int) -> c_int;
    pub fn fk_verify_batch_dev(ctx: *mut fk_ctx, vk: *const u8, vk_len: usize, inputs: *const u64, n_inputs: u32, proofs: *const u8,
                               count: u32, accept: *mut u8) -> c_int;
}
