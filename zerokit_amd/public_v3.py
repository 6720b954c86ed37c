"""Mirror of the V3 API (`RLNV3` / `RLNBuilder`, /root/reference/rln/src/public.rs:774-996; witness and value
enums protocol/witness.rs:940-1267, protocol/proof.rs:896-1155) over the `ffi_rln_v3_*` C ABI of include/rln.h.
Field elements are Python ints; errors are RLNError with the text the ABI returned."""
import ctypes as C

from ._native import CFr, RLNError, VecSize, lib
from .public import (_cfr, _err, _ok_bool, _ok_ptr, _take_bytes, _take_cfr, _take_vec_bool, _take_vec_cfr, _vec_bool,
                     _vec_cfr, _vec_u8)


def _res_vec(res, take):
    if res.err.ptr:
        raise RLNError(_err(res.err))
    return take(res.ok)


def _res_cfr(res):
    return _take_cfr(C.cast(_ok_ptr(res), C.POINTER(CFr)))


class _Handle:
    _free = None

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            getattr(lib(), self._free)(self._h)
            self._h = None

    def _ref(self):
        return C.byref(self._h)


class RLNWitnessInputV3(_Handle):
    """RLNWitnessInputV3::{new_single, new_multi} (witness.rs:1016-1107)"""
    _free = "ffi_rln_v3_witness_input_free"

    @classmethod
    def new_single(cls, identity_secret, user_message_limit, message_id, path_elements, identity_path_index, x,
                   external_nullifier):
        pe, _k1 = _vec_cfr(path_elements)
        pi, _k2 = _vec_u8(bytes(identity_path_index))
        return cls(_ok_ptr(lib().ffi_rln_v3_witness_input_new_single(
            C.byref(_cfr(identity_secret)), C.byref(_cfr(user_message_limit)), C.byref(_cfr(message_id)),
            C.byref(pe), C.byref(pi), C.byref(_cfr(x)), C.byref(_cfr(external_nullifier)))))

    @classmethod
    def new_multi(cls, identity_secret, user_message_limit, message_ids, path_elements, identity_path_index, x,
                  external_nullifier, selector_used):
        mi, _k0 = _vec_cfr(message_ids)
        pe, _k1 = _vec_cfr(path_elements)
        pi, _k2 = _vec_u8(bytes(identity_path_index))
        su, _k3 = _vec_bool(selector_used)
        return cls(_ok_ptr(lib().ffi_rln_v3_witness_input_new_multi(
            C.byref(_cfr(identity_secret)), C.byref(_cfr(user_message_limit)), C.byref(mi), C.byref(pe), C.byref(pi),
            C.byref(_cfr(x)), C.byref(_cfr(external_nullifier)), C.byref(su))))

    identity_secret = property(lambda s: _take_cfr(lib().ffi_rln_v3_witness_input_get_identity_secret(s._ref())))
    user_message_limit = property(lambda s: _take_cfr(lib().ffi_rln_v3_witness_input_get_user_message_limit(s._ref())))
    x = property(lambda s: _take_cfr(lib().ffi_rln_v3_witness_input_get_x(s._ref())))
    external_nullifier = property(lambda s: _take_cfr(lib().ffi_rln_v3_witness_input_get_external_nullifier(s._ref())))
    path_elements = property(lambda s: _take_vec_cfr(lib().ffi_rln_v3_witness_input_get_path_elements(s._ref())))
    message_id = property(lambda s: _res_cfr(lib().ffi_rln_v3_witness_input_get_message_id(s._ref())))
    message_ids = property(lambda s: _res_vec(lib().ffi_rln_v3_witness_input_get_message_ids(s._ref()), _take_vec_cfr))
    selector_used = property(
        lambda s: _res_vec(lib().ffi_rln_v3_witness_input_get_selector_used(s._ref()), _take_vec_bool))

    @property
    def identity_path_index(self):
        v = lib().ffi_rln_v3_witness_input_get_identity_path_index(self._ref())
        b = C.string_at(v.ptr, v.len)
        lib().ffi_vec_u8_free(v)
        return list(b)

    def to_bytes_le(self):
        return _take_bytes(lib().ffi_rln_v3_witness_to_bytes_le(self._ref()))

    def to_bytes_be(self):
        return _take_bytes(lib().ffi_rln_v3_witness_to_bytes_be(self._ref()))

    @classmethod
    def from_bytes_le(cls, b):
        v, _k = _vec_u8(b)
        return cls(_ok_ptr(lib().ffi_bytes_le_to_rln_v3_witness(C.byref(v))))

    @classmethod
    def from_bytes_be(cls, b):
        v, _k = _vec_u8(b)
        return cls(_ok_ptr(lib().ffi_bytes_be_to_rln_v3_witness(C.byref(v))))

    def to_partial(self):
        return RLNPartialWitnessInputV3(C.c_void_p(lib().ffi_rln_v3_witness_to_partial_witness(self._ref())))


class RLNPartialWitnessInputV3(_Handle):
    """RLNPartialWitnessInputV3 (witness.rs:1269-1318)"""
    _free = "ffi_rln_v3_partial_witness_input_free"

    @classmethod
    def new(cls, identity_secret, user_message_limit, path_elements, identity_path_index):
        pe, _k1 = _vec_cfr(path_elements)
        pi, _k2 = _vec_u8(bytes(identity_path_index))
        return cls(_ok_ptr(lib().ffi_rln_v3_partial_witness_input_new(
            C.byref(_cfr(identity_secret)), C.byref(_cfr(user_message_limit)), C.byref(pe), C.byref(pi))))

    identity_secret = property(
        lambda s: _take_cfr(lib().ffi_rln_v3_partial_witness_input_get_identity_secret(s._ref())))
    user_message_limit = property(
        lambda s: _take_cfr(lib().ffi_rln_v3_partial_witness_input_get_user_message_limit(s._ref())))
    path_elements = property(
        lambda s: _take_vec_cfr(lib().ffi_rln_v3_partial_witness_input_get_path_elements(s._ref())))

    def to_bytes_le(self):
        return _take_bytes(lib().ffi_rln_v3_partial_witness_to_bytes_le(self._ref()))

    def to_bytes_be(self):
        return _take_bytes(lib().ffi_rln_v3_partial_witness_to_bytes_be(self._ref()))

    @classmethod
    def from_bytes_le(cls, b):
        v, _k = _vec_u8(b)
        return cls(_ok_ptr(lib().ffi_bytes_le_to_rln_v3_partial_witness(C.byref(v))))

    @classmethod
    def from_bytes_be(cls, b):
        v, _k = _vec_u8(b)
        return cls(_ok_ptr(lib().ffi_bytes_be_to_rln_v3_partial_witness(C.byref(v))))


class RLNProofValuesV3(_Handle):
    """RLNProofValuesV3 (proof.rs:896-960)"""
    _free = "ffi_rln_v3_proof_values_free"

    root = property(lambda s: _take_cfr(lib().ffi_rln_v3_proof_values_get_root(s._ref())))
    x = property(lambda s: _take_cfr(lib().ffi_rln_v3_proof_values_get_x(s._ref())))
    external_nullifier = property(
        lambda s: _take_cfr(lib().ffi_rln_v3_proof_values_get_external_nullifier(s._ref())))
    y = property(lambda s: _res_cfr(lib().ffi_rln_v3_proof_values_get_y(s._ref())))
    nullifier = property(lambda s: _res_cfr(lib().ffi_rln_v3_proof_values_get_nullifier(s._ref())))
    ys = property(lambda s: _res_vec(lib().ffi_rln_v3_proof_values_get_ys(s._ref()), _take_vec_cfr))
    nullifiers = property(lambda s: _res_vec(lib().ffi_rln_v3_proof_values_get_nullifiers(s._ref()), _take_vec_cfr))
    selector_used = property(
        lambda s: _res_vec(lib().ffi_rln_v3_proof_values_get_selector_used(s._ref()), _take_vec_bool))

    def to_bytes_le(self):
        return _take_bytes(lib().ffi_rln_v3_proof_values_to_bytes_le(self._ref()))

    def to_bytes_be(self):
        return _take_bytes(lib().ffi_rln_v3_proof_values_to_bytes_be(self._ref()))

    @classmethod
    def from_bytes_le(cls, b):
        v, _k = _vec_u8(b)
        return cls(_ok_ptr(lib().ffi_bytes_le_to_rln_v3_proof_values(C.byref(v))))

    @classmethod
    def from_bytes_be(cls, b):
        v, _k = _vec_u8(b)
        return cls(_ok_ptr(lib().ffi_bytes_be_to_rln_v3_proof_values(C.byref(v))))

    def recover_secret(self, other) -> int:
        """RecoverSecret (proof.rs:973-1140): same-mode and cross-mode"""
        return _res_cfr(lib().ffi_rln_v3_recover_id_secret(self._ref(), other._ref()))


class RLNProofV3(_Handle):
    """RLNProofV3 { proof, values } (proof.rs:1143-1155)"""
    _free = "ffi_rln_v3_proof_free"

    @property
    def values(self):
        return RLNProofValuesV3(C.c_void_p(lib().ffi_rln_v3_proof_get_values(self._ref())))

    def to_bytes_le(self):
        return _take_bytes(lib().ffi_rln_v3_proof_to_bytes_le(self._ref()))

    def to_bytes_mixed(self):
        return _take_bytes(lib().ffi_rln_v3_proof_to_bytes_mixed(self._ref()))

    @classmethod
    def from_bytes_le(cls, b):
        v, _k = _vec_u8(b)
        return cls(_ok_ptr(lib().ffi_bytes_le_to_rln_v3_proof(C.byref(v))))

    @classmethod
    def from_bytes_mixed(cls, b):
        v, _k = _vec_u8(b)
        return cls(_ok_ptr(lib().ffi_bytes_mixed_to_rln_v3_proof(C.byref(v))))


class PartialProofV3(_Handle):
    _free = "ffi_rln_v3_partial_proof_free"

    def to_bytes_le(self):
        return _take_bytes(lib().ffi_rln_v3_partial_proof_to_bytes_le(self._ref()))

    @classmethod
    def from_bytes_le(cls, b):
        v, _k = _vec_u8(b)
        return cls(_ok_ptr(lib().ffi_bytes_le_to_rln_v3_partial_proof(C.byref(v))))


def compute_id_secret(share1, share2) -> int:
    return _res_cfr(lib().ffi_rln_v3_compute_id_secret(C.byref(_cfr(share1[0])), C.byref(_cfr(share1[1])),
                                                       C.byref(_cfr(share2[0])), C.byref(_cfr(share2[1]))))


class RLNV3(_Handle):
    """RLNV3<State, ArkGroth16Backend> (public.rs:774-955); build with RLNV3.stateless(...) / RLNV3.stateful(...),
    the two arms of RLNBuilder (public.rs:957-996)."""
    _free = "ffi_rln_v3_free"

    @classmethod
    def stateless(cls, zkey: bytes = None, graph: bytes = None):
        if zkey is None:
            h = lib().ffi_rln_v3_new_stateless_default()
            if not h:
                raise RLNError("no HIP device")
            return cls(C.c_void_p(h))
        z, _k1 = _vec_u8(zkey)
        g, _k2 = _vec_u8(graph)
        return cls(_ok_ptr(lib().ffi_rln_v3_new_stateless(C.byref(z), C.byref(g))))

    @classmethod
    def stateful(cls, tree_depth=20, zkey: bytes = None, graph: bytes = None, tree="full"):
        """tree: "full" | "optimal" | "pm" (one device tree behind all three)"""
        L = lib()
        if zkey is None:
            h = {"full": L.ffi_rln_v3_new_with_full_merkle_tree_default,
                 "optimal": L.ffi_rln_v3_new_with_optimal_merkle_tree_default,
                 "pm": L.ffi_rln_v3_new_with_pm_tree_default}[tree]()
            if not h:
                raise RLNError("no HIP device")
            return cls(C.c_void_p(h))
        z, _k1 = _vec_u8(zkey)
        g, _k2 = _vec_u8(graph)
        if tree == "pm":
            return cls(_ok_ptr(L.ffi_rln_v3_new_with_pm_tree(tree_depth, C.byref(z), C.byref(g), b"")))
        fn = L.ffi_rln_v3_new_with_full_merkle_tree if tree == "full" else L.ffi_rln_v3_new_with_optimal_merkle_tree
        return cls(_ok_ptr(fn(tree_depth, C.byref(z), C.byref(g))))

    # ---- proofs
    def generate_proof(self, witness: RLNWitnessInputV3) -> RLNProofV3:
        return RLNProofV3(_ok_ptr(lib().ffi_rln_v3_generate_proof(self._ref(), witness._ref())))

    def generate_proof_with_rs(self, witness: RLNWitnessInputV3, r, s) -> RLNProofV3:
        return RLNProofV3(_ok_ptr(lib().ffi_rln_v3_generate_proof_with_rs(self._ref(), witness._ref(),
                                                                         C.byref(_cfr(r)), C.byref(_cfr(s)))))

    def verify(self, proof: RLNProofV3, x) -> bool:
        return _ok_bool(lib().ffi_rln_v3_verify(self._ref(), proof._ref(), C.byref(_cfr(x))))

    def verify_with_roots(self, proof: RLNProofV3, x, roots) -> bool:
        v, _k = _vec_cfr(list(roots))
        return _ok_bool(lib().ffi_rln_v3_verify_with_roots(self._ref(), proof._ref(), C.byref(v), C.byref(_cfr(x))))

    def generate_partial_proof(self, partial_witness: RLNPartialWitnessInputV3) -> PartialProofV3:
        return PartialProofV3(_ok_ptr(lib().ffi_rln_v3_generate_partial_proof(self._ref(), partial_witness._ref())))

    def finish_proof(self, partial: PartialProofV3, witness: RLNWitnessInputV3) -> RLNProofV3:
        return RLNProofV3(_ok_ptr(lib().ffi_rln_v3_finish_proof(self._ref(), partial._ref(), witness._ref())))

    # ---- tree (public.rs:811-896)
    def set_leaf(self, index, leaf):
        _ok_bool(lib().ffi_rln_v3_set_leaf(self._ref(), index, C.byref(_cfr(leaf))))

    def get_leaf(self, index):
        return _res_cfr(lib().ffi_rln_v3_get_leaf(self._ref(), index))

    def set_next_leaf(self, leaf):
        _ok_bool(lib().ffi_rln_v3_set_next_leaf(self._ref(), C.byref(_cfr(leaf))))

    def set_leaves_from(self, index, leaves):
        v, _k = _vec_cfr(list(leaves))
        _ok_bool(lib().ffi_rln_v3_set_leaves_from(self._ref(), index, C.byref(v)))

    def init_tree_with_leaves(self, leaves):
        v, _k = _vec_cfr(list(leaves))
        _ok_bool(lib().ffi_rln_v3_init_tree_with_leaves(self._ref(), C.byref(v)))

    def atomic_operation(self, index, leaves, indices):
        v, _k = _vec_cfr(list(leaves))
        arr = (C.c_size_t * max(len(indices), 1))(*indices)
        iv = VecSize(C.cast(arr, C.POINTER(C.c_size_t)), len(indices), len(indices))
        _ok_bool(lib().ffi_rln_v3_atomic_operation(self._ref(), index, C.byref(v), C.byref(iv)))

    def delete_leaf(self, index):
        _ok_bool(lib().ffi_rln_v3_delete_leaf(self._ref(), index))

    def leaves_set(self):
        return int(lib().ffi_rln_v3_leaves_set(self._ref()))

    def get_root(self):
        return _take_cfr(lib().ffi_rln_v3_get_root(self._ref()))

    def get_merkle_proof(self, index):
        res = lib().ffi_rln_v3_get_merkle_proof(self._ref(), index)
        if not res.ok:
            raise RLNError(_err(res.err))
        from ._native import MerkleProof
        mp = C.cast(res.ok, C.POINTER(MerkleProof)).contents
        elems = [int.from_bytes(bytes(mp.path_elements.ptr[i].le), "little") for i in range(mp.path_elements.len)]
        bits = list(C.string_at(mp.path_index.ptr, mp.path_index.len))
        lib().ffi_rln_v3_merkle_proof_free(res.ok)
        return elems, bits

    def set_metadata(self, metadata: bytes):
        v, _k = _vec_u8(metadata)
        _ok_bool(lib().ffi_rln_v3_set_metadata(self._ref(), C.byref(v)))

    def get_metadata(self):
        return _take_bytes(lib().ffi_rln_v3_get_metadata(self._ref()))

    def flush(self):
        _ok_bool(lib().ffi_rln_v3_flush(self._ref()))
