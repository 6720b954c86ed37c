"""zerokit_amd -- MI355X-native RLN proving backend behind zerokit's `rln::public` / C-FFI surface.

Host-side mirror (Python over ctypes) of the reference's operator interface for the proving hot path:
  zerokit_amd.public.RLN        <-> rln::public::RLN          (/root/reference/rln/src/public.rs:65-771)
  zerokit_amd.hashers           <-> rln::hashers              (/root/reference/rln/src/hashers.rs)
  zerokit_amd.batch.BatchProver <-> (extension) n x generate_zk_proof_with_rs (protocol/proof.rs:753-777)
  zerokit_amd.batch.PoseidonTree<-> utils FullMerkleTree       (utils/src/merkle_tree/full_merkle_tree.rs)
All compute goes through zerokit_amd/lib/librln.so (HIP kernels for gfx950); nothing here computes.
"""
import os as _os

# seven HIP streams are kept busy by the prover; ROCclr's default of 4 hardware queues makes streams share a queue and
# serialise (see csrc/common.cpp).  Read at HIP runtime initialisation, so set before torch / HIP is first used.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from ._native import LIB_PATH, NativeMissing, RLNError, lib  # noqa: F401

__all__ = ["lib", "LIB_PATH", "NativeMissing", "RLNError"]
