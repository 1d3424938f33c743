//! Drop-in for `fawkes_crypto::backend::bellman_groth16::prover::prove`
//! (/root/reference/fawkes-crypto/src/backend/bellman_groth16/prover.rs:63-90) with the one heavy call,
//! `bellman::groth16::create_random_proof(bcs, &params.0, rng)` (prover.rs:80), replaced by libfawkes_hip.so.
//!
//! Everything above that line (running the circuit closure on `WitnessCS` to obtain the witness, prover.rs:69-76) and
//! below it (`Proof`, the const-tracker assertion, the public inputs, prover.rs:82-89) is the reference's own code path.
//!
//! UNCOMPILED: the build image has no Rust toolchain (see README.md).  The C ABI underneath is exercised by the
//! repository's test-suite through the ctypes mirror fawkes-crypto_amd/api.py.
pub mod ffi;

use std::collections::HashMap;
use std::ffi::CStr;
use std::ptr;
use std::sync::{Arc, Mutex, OnceLock};

use borsh::BorshDeserialize;
use fawkes_crypto::backend::bellman_groth16::{
    bellman_fp_to_num, engines::Engine, osrng::OsRng, prover::Proof, Parameters,
};
use fawkes_crypto::circuit::cs::{WitnessCS, CS};
use fawkes_crypto::circuit::lc::Index;
use fawkes_crypto::core::signal::Signal;
use fawkes_crypto::ff_uint::Num;

use bellman::pairing::{CurveAffine, RawEncodable};

/// A proving key and its constraint system resident in the HBM of one or several MI355X of one node: built once per
/// `Parameters`.  With N devices every GPU keeps 1/N of each key array and a replica of the constraint system; a proof is
/// still ONE call (`fk_multi_prove_r1cs`): the library shards the evaluation of a, b, c, the quotient and the five
/// multi-scalar multiplications and moves the data between the GPUs itself (csrc/multi.hip).
pub struct HipProver {
    multi: *mut ffi::fk_multi,
    key: *mut ffi::fk_multi_key,
    r1cs: *mut ffi::fk_multi_r1cs,
}

unsafe impl Send for HipProver {}
// The C ABI allows one proof at a time per fk_multi (SURVEY 8(b) "Threading"): the process-wide cache below hands a prover out
// behind a Mutex, so sharing the handle between threads is sound.
unsafe impl Sync for HipProver {}

fn last_error(multi: *const ffi::fk_multi) -> String {
    unsafe { CStr::from_ptr(ffi::fk_multi_last_error(multi)).to_string_lossy().into_owned() }
}

/// `into_raw_uncompressed_le` per point (the call group.rs:59,97 already uses), the identity as zeros (group.rs:55)
fn raw_points<G: CurveAffine + RawEncodable>(v: &[G], width: usize) -> Vec<u8> {
    let mut out = Vec::with_capacity(v.len() * width);
    for p in v {
        if p.is_zero() { out.extend(std::iter::repeat(0u8).take(width)); }
        else { out.extend_from_slice(p.into_raw_uncompressed_le().as_ref()); }
    }
    out
}

impl HipProver {
    /// Uploads `params.0` (bellman's key) and decodes `params.2` (the brotli gate blob, setup.rs:25-32) into the resident
    /// constraint system -- ONCE, instead of once per proof (cs.rs:243-245).
    /// `device_ids`: the GPUs to prove on (`&[0]` = one GPU; 1, 2, 4 or 8 of them shard the quotient as well as the
    /// multiplications).
    pub fn new<E: Engine>(device_ids: &[i32], params: &Parameters<E>) -> Self {
        let bp = &params.0;
        let num_input = bp.vk.ic.len() as u32;                     // includes the constant ONE (cs.rs:111)
        let num_aux = bp.l.len() as u32;
        let m = (params.1 as u64 + num_input as u64).next_power_of_two();
        let (h, l, a) = (raw_points(&bp.h, 64), raw_points(&bp.l, 64), raw_points(&bp.a, 64));
        let (b1, b2) = (raw_points(&bp.b_g1, 64), raw_points(&bp.b_g2, 128));
        let vk1 = raw_points(&[bp.vk.alpha_g1, bp.vk.beta_g1, bp.vk.delta_g1], 64);
        let vk2 = raw_points(&[bp.vk.beta_g2, bp.vk.delta_g2], 128);
        let desc = ffi::fk_key_desc {
            m, num_input, num_aux,
            alpha_g1: vk1.as_ptr(), beta_g1: vk1[64..].as_ptr(), delta_g1: vk1[128..].as_ptr(),
            beta_g2: vk2.as_ptr(), delta_g2: vk2[128..].as_ptr(),
            h: h.as_ptr(), n_h: bp.h.len() as u64, l: l.as_ptr(), n_l: bp.l.len() as u64,
            a: a.as_ptr(), n_a: bp.a.len() as u64,
            b_g1: b1.as_ptr(), b_g2: b2.as_ptr(), n_b: bp.b_g1.len() as u64,
            shard_index: 0, shard_count: 1, z_frac_lo: ffi::FK_Z_EQUAL_SPLIT, z_frac_hi: ffi::FK_Z_EQUAL_SPLIT,
        };
        // Everything acquired so far is owned by a guard whose Drop releases it: an `assert!` below that fails (no usable GPU, out of HBM, a
        // malformed gate blob) unwinds through it, so a failed build leaves NOTHING behind in HBM or host memory (each failed build
        // used to leak the key shards and the decoded gate arrays for the life of the process).
        struct Building { multi: *mut ffi::fk_multi, key: *mut ffi::fk_multi_key, gates: *mut ffi::fk_gates, r1cs: *mut ffi::fk_multi_r1cs }
        impl Drop for Building {
            fn drop(&mut self) {
                unsafe {
                    if !self.gates.is_null() { ffi::fk_gates_free(self.gates); }
                    if !self.multi.is_null() {
                        if !self.r1cs.is_null() { ffi::fk_multi_r1cs_free(self.multi, self.r1cs); }
                        if !self.key.is_null() { ffi::fk_multi_key_free(self.multi, self.key); }
                        ffi::fk_multi_free(self.multi);
                    }
                }
            }
        }
        let mut b = Building { multi: ptr::null_mut(), key: ptr::null_mut(), gates: ptr::null_mut(), r1cs: ptr::null_mut() };
        unsafe {
            let rc = ffi::fk_init_devices(device_ids.len() as i32, device_ids.as_ptr(), &mut b.multi);
            assert!(rc == ffi::FK_OK, "fk_init_devices: no usable MI355X (there is no CPU fallback)");
            // shard g of every key array goes to device_ids[g] (desc.shard_* are ignored by the multi-GPU loader)
            let rc = ffi::fk_multi_key_load(b.multi, &desc, &mut b.key);
            assert!(rc == ffi::FK_OK, "fk_multi_key_load: {}", last_error(b.multi));
            // the gate blob is decoded once on the host (ctx = NULL), then uploaded to every GPU
            let rc = ffi::fk_gates_decode(ptr::null_mut(), params.2.as_ptr(), params.2.len(), ffi::FK_GATES_BROTLI, params.1, num_input, num_aux, &mut b.gates);
            assert!(rc == ffi::FK_OK, "fk_gates_decode failed ({}): {}", rc, CStr::from_ptr(ffi::fk_last_error(ptr::null())).to_string_lossy());
            let rc = ffi::fk_multi_r1cs_load_gates(b.multi, b.gates, &mut b.r1cs);
            ffi::fk_gates_free(b.gates);
            b.gates = ptr::null_mut();
            assert!(rc == ffi::FK_OK, "fk_multi_r1cs_load_gates: {}", last_error(b.multi));
            // finished: ownership moves to the prover (whose own Drop frees the same three handles)
            let out = HipProver { multi: b.multi, key: b.key, r1cs: b.r1cs };
            b.multi = ptr::null_mut(); b.key = ptr::null_mut(); b.r1cs = ptr::null_mut();
            out
        }
    }

    /// `create_proof(circuit, params, r, s)` on the GPU: z = values_input ++ values_aux, r and s as `Num<Fr>`
    /// (4 x u64 Montgomery limbs, the in-memory image the C ABI reads: ff-uint/src/num/mod.rs:21-23, mod.rs:105-137).
    fn prove_bytes<Fr: fawkes_crypto::ff_uint::PrimeField>(&self, z: &[Num<Fr>], r: &Num<Fr>, s: &Num<Fr>) -> [u8; ffi::FK_PROOF_BYTES] {
        let mut out = [0u8; ffi::FK_PROOF_BYTES];
        let rc = unsafe {
            ffi::fk_multi_prove_r1cs(self.multi, self.key, self.r1cs, z.as_ptr() as *const u64, r as *const _ as *const u64,
                                     s as *const _ as *const u64, out.as_mut_ptr(), ptr::null_mut())
        };
        // the reference `.unwrap()`s bellman's SynthesisError at prover.rs:80: a non-zero status panics here as well
        assert!(rc == ffi::FK_OK, "fk_multi_prove_r1cs ({}): {}", rc, last_error(self.multi));
        out
    }
}

impl Drop for HipProver {
    fn drop(&mut self) {
        unsafe { ffi::fk_multi_r1cs_free(self.multi, self.r1cs); ffi::fk_multi_key_free(self.multi, self.key); ffi::fk_multi_free(self.multi); }
    }
}

/// prover.rs:69-76 and 83-87 -- everything `prove` does on the host around `create_random_proof`: allocates the signals in a `WitnessCS`,
/// runs the circuit closure (which fills in the assignment), and returns (z = values_input ++ values_aux in the variable order of
/// cs.rs:255-268, the public inputs without the leading ONE, whether every cached constant was consumed: the assertion of prover.rs:83,
/// which the callers make AFTER the proof like the reference).  Takes no lock and touches no GPU: a panic in the caller's closure
/// (a witness that cannot be computed) leaves no shared state behind.
fn host_witness<'a, E: Engine, Pub: Signal<WitnessCS<'a, E::Fr>>, Sec: Signal<WitnessCS<'a, E::Fr>>, C: Fn(Pub, Sec)>(
    params: &'a Parameters<E>,
    input_pub: &Pub::Value,
    input_sec: &Sec::Value,
    circuit: C,
) -> (Vec<Num<E::Fr>>, Vec<Num<E::Fr>>, bool) {
    let ref rcs = params.get_witness_rcs();                                  // prover.rs:69
    let signal_pub = Pub::alloc(rcs, Some(input_pub));
    signal_pub.inputize();
    let signal_sec = Sec::alloc(rcs, Some(input_sec));
    circuit(signal_pub, signal_sec);                                         // prover.rs:74: fills WitnessCS

    let cs = rcs.borrow();
    // `Num<Fr>` is #[repr(transparent)] over the limbs: the vector is the byte image the C ABI reads
    let mut z: Vec<Num<E::Fr>> = Vec::with_capacity(cs.num_input() + cs.num_aux());
    for i in 0..cs.num_input() as u32 { z.push(cs.get_value(Index::Input(i)).unwrap()); }
    for j in 0..cs.num_aux() as u32 { z.push(cs.get_value(Index::Aux(j)).unwrap()); }
    let mut inputs = Vec::with_capacity(cs.num_input());
    for i in 1..cs.num_input() as u32 { inputs.push(cs.get_value(Index::Input(i)).unwrap()); }  // prover.rs:84-87
    (z, inputs, cs.const_tracker_index == cs.const_tracker.len())
}

/// Deterministic twin of `prove_hip`: r, s given -- what `bellman::groth16::create_proof(circuit, params, r, s)` takes.
/// Same signature as prover.rs:63-68 plus the resident prover and (r, s).
pub fn prove_hip_with_rs<'a, E: Engine, Pub: Signal<WitnessCS<'a, E::Fr>>, Sec: Signal<WitnessCS<'a, E::Fr>>, C: Fn(Pub, Sec)>(
    params: &'a Parameters<E>,
    hip: &HipProver,
    input_pub: &Pub::Value,
    input_sec: &Sec::Value,
    circuit: C,
    r: Num<E::Fr>,
    s: Num<E::Fr>,
) -> (Vec<Num<E::Fr>>, Proof<E>) {
    let (z, inputs, tracker_consumed) = host_witness::<E, Pub, Sec, C>(params, input_pub, input_sec, circuit);
    let bytes = hip.prove_bytes(&z, &r, &s);
    // the 256 bytes ARE fawkes' Borsh `Proof` (prover.rs:39-60: a, b, c; canonical little-endian coordinates)
    let proof = Proof::<E>::try_from_slice(&bytes).expect("proof bytes");
    assert!(tracker_consumed, "not all cached data used");                  // prover.rs:83
    (inputs, proof)
}

/// `prover::prove` with the GPU behind it: r, s drawn exactly as `create_random_proof` draws them (prover.rs:78-80:
/// `Fr::rand` over fawkes' OsRng; the accepted limbs are the Montgomery representation).
pub fn prove_hip<'a, E: Engine, Pub: Signal<WitnessCS<'a, E::Fr>>, Sec: Signal<WitnessCS<'a, E::Fr>>, C: Fn(Pub, Sec)>(
    params: &'a Parameters<E>,
    hip: &HipProver,
    input_pub: &Pub::Value,
    input_sec: &Sec::Value,
    circuit: C,
) -> (Vec<Num<E::Fr>>, Proof<E>) {
    use bellman::pairing::ff::Field;
    let ref mut rng = OsRng::new();
    let r = <<E::BE as bellman::pairing::ff::ScalarEngine>::Fr as Field>::rand(rng);
    let s = <<E::BE as bellman::pairing::ff::ScalarEngine>::Fr as Field>::rand(rng);
    prove_hip_with_rs(params, hip, input_pub, input_sec, circuit, bellman_fp_to_num(r), bellman_fp_to_num(s))
}

// ------------------------------------------------------------------------------------------------------------------------------
// `prove` with EXACTLY the reference's signature (prover.rs:63-68): call sites do not change at all.
//
// The resident state the GPU path needs (key shards, fixed-base levels, the decoded constraint system: seconds to build, 10^2 GB
// at 2^25) cannot be rebuilt per call, and the reference's signature has no place to pass it -- so it lives in a process-wide
// cache keyed by the `Parameters` the caller passes (its address plus a fingerprint of its contents) and the device list.
// Devices: the environment variable FK_DEVICES, a comma-separated list of HIP device ids ("0", "0,1,2,3,4,5,6,7"); default "0".
// `prove_hip(params, &hip, ..)` above stays for callers that want to own the prover (several keys, explicit lifetime, eviction).

type CacheKey = (usize, u32, usize, usize, [u8; 64], Vec<i32>);

/// One entry of the cache: the prover of a key, or the marker that some thread is building it.  The map's own lock is held only to find or
/// insert a slot; the BUILD happens under the slot's lock, so (a) exactly one thread builds a given key while the others wait on that slot and
/// then share the result -- N first calls for the same `Parameters` do not run N key loads, level derivations and 61 GB gate decodes side by side
/// on the same GPU, each planning against HBM the others are about to take -- and (b) proofs of OTHER keys go on meanwhile.
pub struct Slot { prover: Mutex<Option<HipProver>> }

fn cache() -> &'static Mutex<HashMap<CacheKey, Arc<Slot>>> {
    static CACHE: OnceLock<Mutex<HashMap<CacheKey, Arc<Slot>>>> = OnceLock::new();
    CACHE.get_or_init(|| Mutex::new(HashMap::new()))
}

/// FK_DEVICES="0,1,2,3" -> [0, 1, 2, 3]; unset or empty -> [0].  Panics on anything that is not a list of integers (a typo must
/// not silently fall back to one GPU).
pub fn devices_from_env() -> Vec<i32> {
    match std::env::var("FK_DEVICES") {
        Ok(v) if !v.trim().is_empty() => v.split(',').map(|t| t.trim().parse::<i32>().expect("FK_DEVICES: comma-separated HIP device ids")).collect(),
        _ => vec![0],
    }
}

fn cache_key<E: Engine>(params: &Parameters<E>, devices: &[i32]) -> CacheKey {
    let bp = &params.0;
    let mut tag = [0u8; 64];
    tag.copy_from_slice(bp.vk.delta_g1.into_raw_uncompressed_le().as_ref());    // the key's delta: differs between any two setups
    (params as *const _ as usize, params.1, bp.h.len(), bp.l.len(), tag, devices.to_vec())
}

/// A panic while a lock was held (a failed key load, a failed proof) poisons a std Mutex; the state behind ours stays usable -- a slot
/// holds a finished prover or None, and the C library clears whatever a failed call had in flight -- so a poisoned lock is simply taken over.
/// (The reference's `prove` has no shared state and survives a caught panic; so does this one.)
fn lock_ignoring_poison<T>(m: &Mutex<T>) -> std::sync::MutexGuard<'_, T> {
    m.lock().unwrap_or_else(|e| e.into_inner())
}

/// The cached prover of (`params`, FK_DEVICES), built on first use, handed out LOCKED (the C ABI allows one proof at a time per prover).
/// A `Parameters` value that is dropped should be `forget`-ed: the cache cannot see a drop, and HBM is released only when the entry goes.
/// The map's lock is held only while the slot is looked up or inserted.  The first thread to lock an empty slot builds the prover inside it
/// (seconds to minutes; may panic -- no usable GPU, out of HBM); threads that want the same key meanwhile wait on the slot and find it
/// filled.  A build that panics unwinds with the slot still empty (`HipProver::new` releases what it had acquired), the poisoned slot lock
/// is taken over by the next caller, and that caller builds again -- a failed build never leaves a half-made or a leaked prover behind.
pub fn with_resident_prover<E: Engine, R>(params: &Parameters<E>, f: impl FnOnce(&HipProver) -> R) -> R {
    let devices = devices_from_env();
    let key = cache_key(params, &devices);
    let slot = lock_ignoring_poison(cache()).entry(key).or_insert_with(|| Arc::new(Slot { prover: Mutex::new(None) })).clone();
    let mut guard = lock_ignoring_poison(&slot.prover);
    if guard.is_none() {
        *guard = Some(HipProver::new(&devices, params));      // may panic: the slot stays None, nothing else is locked
    }
    f(guard.as_ref().unwrap())
}

/// Drops the cached prover(s) of `params` (all device lists): frees the key shards, levels and constraint system in HBM.
pub fn forget<E: Engine>(params: &Parameters<E>) {
    let probe = cache_key(params, &[]);
    lock_ignoring_poison(cache()).retain(|k, _| !(k.0 == probe.0 && k.1 == probe.1 && k.2 == probe.2 && k.3 == probe.3 && k.4 == probe.4));
}

/// `fawkes_crypto::backend::bellman_groth16::prover::prove` -- the SAME signature, argument meaning, return value and panics
/// (prover.rs:63-90) -- with `create_random_proof` (prover.rs:80) running on the MI355X named by FK_DEVICES.
/// Replace `use fawkes_crypto::backend::bellman_groth16::prover::prove;` by `use fawkes_crypto_hip::prove;`: nothing else changes.
pub fn prove<'a, E: Engine, Pub: Signal<WitnessCS<'a, E::Fr>>, Sec: Signal<WitnessCS<'a, E::Fr>>, C: Fn(Pub, Sec)>(
    params: &'a Parameters<E>,
    input_pub: &Pub::Value,
    input_sec: &Sec::Value,
    circuit: C,
) -> (Vec<Num<E::Fr>>, Proof<E>) {
    use bellman::pairing::ff::Field;
    let ref mut rng = OsRng::new();
    let r = <<E::BE as bellman::pairing::ff::ScalarEngine>::Fr as Field>::rand(rng);
    let s = <<E::BE as bellman::pairing::ff::ScalarEngine>::Fr as Field>::rand(rng);
    prove_with_rs(params, input_pub, input_sec, circuit, bellman_fp_to_num(r), bellman_fp_to_num(s))
}

/// Deterministic twin of `prove` (r, s given: bellman's `create_proof(circuit, params, r, s)`), for byte-exact comparisons.
/// The caller's circuit closure runs BEFORE the prover's lock is taken: witness generation of one thread overlaps with the proof of
/// another, and a closure that panics has locked nothing.
pub fn prove_with_rs<'a, E: Engine, Pub: Signal<WitnessCS<'a, E::Fr>>, Sec: Signal<WitnessCS<'a, E::Fr>>, C: Fn(Pub, Sec)>(
    params: &'a Parameters<E>,
    input_pub: &Pub::Value,
    input_sec: &Sec::Value,
    circuit: C,
    r: Num<E::Fr>,
    s: Num<E::Fr>,
) -> (Vec<Num<E::Fr>>, Proof<E>) {
    let (z, inputs, tracker_consumed) = host_witness::<E, Pub, Sec, C>(params, input_pub, input_sec, circuit);
    let bytes = with_resident_prover(params, |hip| hip.prove_bytes(&z, &r, &s));      // the only stretch that holds the prover
    let proof = Proof::<E>::try_from_slice(&bytes).expect("proof bytes");
    assert!(tracker_consumed, "not all cached data used");                  // prover.rs:83
    (inputs, proof)
}
