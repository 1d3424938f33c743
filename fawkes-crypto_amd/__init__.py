"""
fawkes-crypto_amd -- MI355X (gfx950) Groth16 proving backend for fawkes-crypto.

Host-side mirror (ctypes over the C ABI in include/fawkes_hip.h) of the reference interface
`fawkes_crypto::backend::bellman_groth16::{Parameters, prover::{Proof, prove}, group::{G1Point, G2Point}}`
(/root/reference/fawkes-crypto/src/backend/bellman_groth16/{mod.rs:139-175, prover.rs:13-90, group.rs:13-122}).

The directory name carries a hyphen (the reference's crate name); import it as `fawkes_crypto_amd`
(the sibling shim package at the repo root) -- both names load this file.

There is NO CPU fallback: if libfawkes_hip.so is missing or no GPU is visible, `Context()` raises.
"""
from .api import (  # noqa: F401
    FkError, Context, MultiContext, Parameters, Proof, VK, G1Point, G2Point, R1cs, prove, prove_with_rs, lib_path, load_library,
    FK_MSM_RESULT_BYTES, FK_PROOF_BYTES, EXPORTED_SYMBOLS, build_library,
    verify, verify_batch, vk_to_borsh, synthesize, sample_fr,
)
from . import parallel  # noqa: F401
from . import params_io  # noqa: F401
