"""Mirror of `rln::public::RLN` (/root/reference/rln/src/public.rs:65-771) over the zerokit C ABI
(include/rln.h).  Same method names, argument meaning and error behaviour; field elements are Python ints,
errors are RLNError carrying the string the ABI returned (the reference stringifies RLNError the same way,
rln/src/ffi/ffi_rln.rs:52-55)."""
import ctypes as C

from ._native import (CFr, CResultPtr, MerkleProof, RLNError, VecBool, VecCFr, VecSize, VecString, VecU8, lib)


_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _canonical(x) -> bytes:
    """A CFr of the C ABI only ever holds a canonical value (ffi_bytes_*_to_cfr rejects anything else, and
    validate_witness / the duplicate-id checks compare raw bytes), so the Python side enforces the same rule."""
    x = int(x)
    if not 0 <= x < _R:
        raise RLNError("Non-canonical field element: value is not in [0, r-1]")
    return x.to_bytes(32, "little")


def _cfr(x: int) -> CFr:
    c = CFr()
    C.memmove(c.le, _canonical(x), 32)
    return c


def _int(p) -> int:
    return int.from_bytes(bytes(p.contents.le), "little")


def _take_cfr(p) -> int:
    v = _int(p)
    lib().ffi_cfr_free(C.cast(p, C.c_void_p))
    return v


def _vec_cfr(vals):
    arr = (CFr * max(len(vals), 1))()
    for i, v in enumerate(vals):
        C.memmove(arr[i].le, _canonical(v), 32)
    return VecCFr(C.cast(arr, C.POINTER(CFr)), len(vals), len(vals)), arr


def _vec_u8(b: bytes):
    arr = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) if len(b) else b"\0")
    return VecU8(C.cast(arr, C.POINTER(C.c_uint8)), len(b), len(b)), arr


def _vec_bool(vals):
    arr = (C.c_bool * max(len(vals), 1))(*[bool(v) for v in vals])
    return VecBool(C.cast(arr, C.POINTER(C.c_bool)), len(vals), len(vals)), arr


def _take_vec_cfr(v: VecCFr):
    out = [int.from_bytes(bytes(v.ptr[i].le), "little") for i in range(v.len)]
    lib().ffi_vec_cfr_free(v)
    return out


def _take_vec_bool(v: VecBool):
    out = [bool(v.ptr[i]) for i in range(v.len)]
    lib().ffi_vec_bool_free(v)
    return out


def _err(v: VecU8) -> str:
    s = C.string_at(v.ptr, v.len).decode("utf-8", "replace") if v.ptr else ""
    if v.ptr:
        lib().ffi_c_string_free(v)
    return s


def _ok_ptr(res: CResultPtr):
    if not res.ok:
        raise RLNError(_err(res.err))
    return C.c_void_p(res.ok)


def _ok_bool(res):
    if res.err.ptr:
        raise RLNError(_err(res.err))
    return bool(res.ok)


def _take_bytes(res):
    if res.err.ptr:
        raise RLNError(_err(res.err))
    b = C.string_at(res.ok.ptr, res.ok.len)
    lib().ffi_vec_u8_free(res.ok)
    return b


class RLNWitnessInput:
    """RLNWitnessInput::new_single (protocol/witness.rs:78-108)"""

    def __init__(self, identity_secret, user_message_limit, message_id, path_elements, identity_path_index, x,
                 external_nullifier, _handle=None):
        if _handle is not None:
            self._h = _handle
            return
        pe, _k1 = _vec_cfr(path_elements)
        pi, _k2 = _vec_u8(bytes(identity_path_index))
        self._h = _ok_ptr(lib().ffi_rln_witness_input_new_single(
            C.byref(_cfr(identity_secret)), C.byref(_cfr(user_message_limit)), C.byref(_cfr(message_id)),
            C.byref(pe), C.byref(pi), C.byref(_cfr(x)), C.byref(_cfr(external_nullifier))))

    @classmethod
    def new_multi(cls, identity_secret, user_message_limit, message_ids, path_elements, identity_path_index, x,
                  external_nullifier, selector_used):
        """RLNWitnessInput::new_multi (protocol/witness.rs:117-180)"""
        mi, _k0 = _vec_cfr(message_ids)
        pe, _k1 = _vec_cfr(path_elements)
        pi, _k2 = _vec_u8(bytes(identity_path_index))
        su, _k3 = _vec_bool(selector_used)
        return cls(0, 0, 0, [], [], 0, 0, _handle=_ok_ptr(lib().ffi_rln_witness_input_new_multi(
            C.byref(_cfr(identity_secret)), C.byref(_cfr(user_message_limit)), C.byref(mi), C.byref(pe),
            C.byref(pi), C.byref(_cfr(x)), C.byref(_cfr(external_nullifier)), C.byref(su))))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ffi_rln_witness_input_free(self._h)
            self._h = None

    @property
    def version_byte(self):
        return lib().ffi_rln_witness_input_get_version_byte(C.byref(self._h))

    def to_bigint_json(self) -> str:
        """rln_witness_to_bigint_json (protocol/witness.rs:317-366)"""
        return _take_bytes(lib().ffi_rln_witness_to_bigint_json(C.byref(self._h))).decode()

    @property
    def message_ids(self):
        return _take_vec_cfr(lib().ffi_rln_witness_input_get_message_ids(C.byref(self._h)))

    @property
    def selector_used(self):
        return _take_vec_bool(lib().ffi_rln_witness_input_get_selector_used(C.byref(self._h)))

    def to_bytes_le(self):
        return _take_bytes(lib().ffi_rln_witness_to_bytes_le(C.byref(self._h)))

    def to_bytes_be(self):
        return _take_bytes(lib().ffi_rln_witness_to_bytes_be(C.byref(self._h)))

    @classmethod
    def from_bytes_le(cls, b):
        v, _k = _vec_u8(b)
        return cls(0, 0, 0, [], [], 0, 0, _handle=_ok_ptr(lib().ffi_bytes_le_to_rln_witness(C.byref(v))))

    @classmethod
    def from_bytes_be(cls, b):
        v, _k = _vec_u8(b)
        return cls(0, 0, 0, [], [], 0, 0, _handle=_ok_ptr(lib().ffi_bytes_be_to_rln_witness(C.byref(v))))

    @property
    def x(self):
        return _take_cfr(lib().ffi_rln_witness_input_get_x(C.byref(self._h)))

    @property
    def path_elements(self):
        v = lib().ffi_rln_witness_input_get_path_elements(C.byref(self._h))
        out = [int.from_bytes(bytes(v.ptr[i].le), "little") for i in range(v.len)]
        lib().ffi_vec_cfr_free(v)
        return out


class RLNProofValues:
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ffi_rln_proof_values_free(self._h)
            self._h = None

    @classmethod
    def from_bytes_le(cls, b):
        v, _k = _vec_u8(b)
        return cls(_ok_ptr(lib().ffi_bytes_le_to_rln_proof_values(C.byref(v))))

    @classmethod
    def from_bytes_be(cls, b):
        v, _k = _vec_u8(b)
        return cls(_ok_ptr(lib().ffi_bytes_be_to_rln_proof_values(C.byref(v))))

    root = property(lambda s: _take_cfr(lib().ffi_rln_proof_values_get_root(C.byref(s._h))))
    x = property(lambda s: _take_cfr(lib().ffi_rln_proof_values_get_x(C.byref(s._h))))
    external_nullifier = property(
        lambda s: _take_cfr(lib().ffi_rln_proof_values_get_external_nullifier(C.byref(s._h))))

    @property
    def y(self):
        r = lib().ffi_rln_proof_values_get_y(C.byref(self._h))
        return _take_cfr(C.cast(_ok_ptr(r), C.POINTER(CFr)))

    @property
    def nullifier(self):
        r = lib().ffi_rln_proof_values_get_nullifier(C.byref(self._h))
        return _take_cfr(C.cast(_ok_ptr(r), C.POINTER(CFr)))

    @property
    def version_byte(self):
        return lib().ffi_rln_proof_values_get_version_byte(C.byref(self._h))

    def _vec(self, fn, take):
        r = fn(C.byref(self._h))
        if r.err.ptr:
            raise RLNError(_err(r.err))
        return take(r.ok)

    ys = property(lambda s: s._vec(lib().ffi_rln_proof_values_get_ys, _take_vec_cfr))
    nullifiers = property(lambda s: s._vec(lib().ffi_rln_proof_values_get_nullifiers, _take_vec_cfr))
    selector_used = property(lambda s: s._vec(lib().ffi_rln_proof_values_get_selector_used, _take_vec_bool))

    def to_bytes_le(self):
        v = lib().ffi_rln_proof_values_to_bytes_le(C.byref(self._h))
        b = C.string_at(v.ptr, v.len)
        lib().ffi_vec_u8_free(v)
        return b

    def to_bytes_be(self):
        v = lib().ffi_rln_proof_values_to_bytes_be(C.byref(self._h))
        b = C.string_at(v.ptr, v.len)
        lib().ffi_vec_u8_free(v)
        return b


class RLNProof:
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ffi_rln_proof_free(self._h)
            self._h = None

    @property
    def values(self):
        return RLNProofValues(C.c_void_p(lib().ffi_rln_proof_get_values(C.byref(self._h))))

    def to_bytes_le(self):
        return _take_bytes(lib().ffi_rln_proof_to_bytes_le(C.byref(self._h)))

    def to_bytes_be(self):
        return _take_bytes(lib().ffi_rln_proof_to_bytes_be(C.byref(self._h)))

    @classmethod
    def from_bytes_le(cls, b):
        v, _k = _vec_u8(b)
        return cls(_ok_ptr(lib().ffi_bytes_le_to_rln_proof(C.byref(v))))

    @classmethod
    def from_bytes_be(cls, b):
        v, _k = _vec_u8(b)
        return cls(_ok_ptr(lib().ffi_bytes_be_to_rln_proof(C.byref(v))))


class RLNPartialWitnessInput:
    """RLNPartialWitnessInput::new (protocol/witness.rs:253-270)"""

    def __init__(self, identity_secret, user_message_limit, path_elements, identity_path_index, _handle=None):
        if _handle is not None:
            self._h = _handle
            return
        pe, _k1 = _vec_cfr(path_elements)
        pi, _k2 = _vec_u8(bytes(identity_path_index))
        self._h = _ok_ptr(lib().ffi_rln_partial_witness_input_new(
            C.byref(_cfr(identity_secret)), C.byref(_cfr(user_message_limit)), C.byref(pe), C.byref(pi)))

    def to_bytes_le(self):
        return _take_bytes(lib().ffi_rln_partial_witness_to_bytes_le(C.byref(self._h)))

    def to_bytes_be(self):
        return _take_bytes(lib().ffi_rln_partial_witness_to_bytes_be(C.byref(self._h)))

    @classmethod
    def from_bytes_le(cls, b):
        v, _k = _vec_u8(b)
        return cls(0, 0, [], [], _handle=_ok_ptr(lib().ffi_bytes_le_to_rln_partial_witness(C.byref(v))))

    @classmethod
    def from_bytes_be(cls, b):
        v, _k = _vec_u8(b)
        return cls(0, 0, [], [], _handle=_ok_ptr(lib().ffi_bytes_be_to_rln_partial_witness(C.byref(v))))

    identity_secret = property(
        lambda s: _take_cfr(lib().ffi_rln_partial_witness_input_get_identity_secret(C.byref(s._h))))
    user_message_limit = property(
        lambda s: _take_cfr(lib().ffi_rln_partial_witness_input_get_user_message_limit(C.byref(s._h))))
    path_elements = property(
        lambda s: _take_vec_cfr(lib().ffi_rln_partial_witness_input_get_path_elements(C.byref(s._h))))

    @property
    def identity_path_index(self):
        v = lib().ffi_rln_partial_witness_input_get_identity_path_index(C.byref(self._h))
        b = C.string_at(v.ptr, v.len)
        lib().ffi_vec_u8_free(v)
        return list(b)

    @classmethod
    def from_witness(cls, w: RLNWitnessInput):
        return cls(0, 0, [], [], _handle=C.c_void_p(lib().ffi_rln_witness_to_partial_witness(C.byref(w._h))))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ffi_rln_partial_witness_input_free(self._h)
            self._h = None


class RLNPartialProof:
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ffi_rln_partial_proof_free(self._h)
            self._h = None

    def to_bytes_le(self):
        return _take_bytes(lib().ffi_rln_partial_proof_to_bytes_le(C.byref(self._h)))

    @classmethod
    def from_bytes_le(cls, b):
        v, _k = _vec_u8(b)
        return cls(_ok_ptr(lib().ffi_bytes_le_to_rln_partial_proof(C.byref(v))))


class RLN:
    """rln::public::RLN.  `RLN(tree_depth)` == RLN::new(tree_depth, "") (public.rs:110-128); `tree_config` is what
    ffi_rln_new takes: the PATH of a JSON file with the PmTreeConfig keys (ffi_rln.rs:24-57);
    `RLN.new_with_params(depth, zkey, graph)` == RLN::new_with_params (public.rs:166-196)."""

    def __init__(self, tree_depth=20, tree_config="", _handle=None):
        self._h = _handle if _handle is not None else _ok_ptr(lib().ffi_rln_new(tree_depth, tree_config.encode()))

    @classmethod
    def new_with_params(cls, tree_depth, zkey: bytes, graph: bytes, tree_config=""):
        z, _k1 = _vec_u8(zkey)
        g, _k2 = _vec_u8(graph)
        return cls(_handle=_ok_ptr(lib().ffi_rln_new_with_params(tree_depth, C.byref(z), C.byref(g),
                                                                 tree_config.encode())))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ffi_rln_free(self._h)
            self._h = None

    def prover_info(self):
        """EXT: sizing of the prover behind this object (window schedule, max_batch, table bytes)"""
        from ._native import ProverInfo
        info = ProverInfo()
        if lib().rlnamd_ffi_prover_info(self._h, C.byref(info)) != 0:
            raise RLNError("no prover")
        return info

    # ---- Merkle-tree APIs (public.rs:298-593)
    def memo_stats(self):
        """the member memo of an object built with "auto_partial" (rlnamd_ffi_memo_stats)"""
        out = (C.c_uint64 * 4)()
        if lib().rlnamd_ffi_memo_stats(self._h, out) != 0:
            raise RLNError("rlnamd_ffi_memo_stats failed")
        return dict(zip(("members", "finishes", "from_scratch", "pending"), [int(v) for v in out]))

    def gather_stats(self):
        """concurrent single-proof calls gathered into batches (rlnamd_ffi_gather_stats)"""
        out = (C.c_uint64 * 8)()
        if lib().rlnamd_ffi_gather_stats(self._h, out) != 0:
            raise RLNError("rlnamd_ffi_gather_stats failed")
        return dict(zip(("batches", "calls", "largest", "cap", "waited", "busy_ns", "finish_batches", "finish_calls"),
                        [int(v) for v in out]))

    def tree_depth(self):
        return int(lib().ffi_rln_get_tree_depth(C.byref(self._h)))

    def max_out(self):
        """message slots of the loaded circuit (1 for the single message-id circuit)"""
        return int(lib().ffi_rln_get_max_out(C.byref(self._h)))

    def set_tree(self, tree_depth):
        _ok_bool(lib().ffi_set_tree(C.byref(self._h), tree_depth))

    def set_leaf(self, index, leaf):
        _ok_bool(lib().ffi_set_leaf(C.byref(self._h), index, C.byref(_cfr(leaf))))

    def get_leaf(self, index):
        r = lib().ffi_get_leaf(C.byref(self._h), index)
        return _take_cfr(C.cast(_ok_ptr(r), C.POINTER(CFr)))

    def set_leaves_from(self, index, leaves):
        v, _k = _vec_cfr(leaves)
        _ok_bool(lib().ffi_set_leaves_from(C.byref(self._h), index, C.byref(v)))

    def init_tree_with_leaves(self, leaves):
        v, _k = _vec_cfr(leaves)
        _ok_bool(lib().ffi_init_tree_with_leaves(C.byref(self._h), C.byref(v)))

    def atomic_operation(self, index, leaves, indices):
        v, _k = _vec_cfr(leaves)
        arr = (C.c_size_t * max(len(indices), 1))(*indices)
        iv = VecSize(C.cast(arr, C.POINTER(C.c_size_t)), len(indices), len(indices))
        _ok_bool(lib().ffi_atomic_operation(C.byref(self._h), index, C.byref(v), C.byref(iv)))

    def leaves_set(self):
        return int(lib().ffi_leaves_set(C.byref(self._h)))

    def set_next_leaf(self, leaf):
        _ok_bool(lib().ffi_set_next_leaf(C.byref(self._h), C.byref(_cfr(leaf))))

    def delete_leaf(self, index):
        _ok_bool(lib().ffi_delete_leaf(C.byref(self._h), index))

    def get_root(self):
        return _take_cfr(lib().ffi_get_root(C.byref(self._h)))

    def get_merkle_proof(self, index):
        """-> (path_elements, identity_path_index) (public.rs:550-556)"""
        r = lib().ffi_get_merkle_proof(C.byref(self._h), index)
        h = _ok_ptr(r)
        mp = C.cast(h, C.POINTER(MerkleProof)).contents
        elems = [int.from_bytes(bytes(mp.path_elements.ptr[i].le), "little") for i in range(mp.path_elements.len)]
        bits = [mp.path_index.ptr[i] for i in range(mp.path_index.len)]
        lib().ffi_merkle_proof_free(h)
        return elems, bits

    def set_metadata(self, metadata: bytes):
        v, _k = _vec_u8(metadata)
        _ok_bool(lib().ffi_set_metadata(C.byref(self._h), C.byref(v)))

    def get_metadata(self):
        return _take_bytes(lib().ffi_get_metadata(C.byref(self._h)))

    def flush(self):
        """ffi_flush (ffi_tree.rs): writes a persistent tree's snapshot; nothing to do for a temporary tree"""
        _ok_bool(lib().ffi_flush(C.byref(self._h)))

    def close(self):
        """drop the object now (a persistent tree is flushed, as sled does when the database is dropped)"""
        self.__del__()

    # ---- zkSNARK APIs (public.rs:595-771)
    def generate_rln_proof(self, witness: RLNWitnessInput) -> RLNProof:
        return RLNProof(_ok_ptr(lib().ffi_generate_rln_proof(C.byref(self._h), C.byref(witness._h))))

    def generate_rln_proof_with_rs(self, witness: RLNWitnessInput, r, s) -> RLNProof:
        """EXT: generate_zk_proof_with_rs (protocol/proof.rs:753-777) + proof values"""
        return RLNProof(_ok_ptr(lib().ffi_generate_rln_proof_with_rs(C.byref(self._h), C.byref(witness._h),
                                                                     C.byref(_cfr(r)), C.byref(_cfr(s)))))

    def generate_rln_proofs_batch(self, witnesses, rs=None):
        """EXT: n proofs in one device batch"""
        n = len(witnesses)
        hs = (C.c_void_p * n)(*[w._h.value for w in witnesses])
        outs = (C.c_void_p * n)()
        rsp = None
        if rs is not None:
            flat, _k = _vec_cfr([v for pair in rs for v in pair])
            rsp = flat.ptr
        _ok_bool(lib().ffi_generate_rln_proofs_batch(C.byref(self._h), hs, n, rsp, outs))
        return [RLNProof(C.c_void_p(outs[i])) for i in range(n)]

    def finish_rln_proofs_batch(self, partials, witnesses, rs=None):
        """EXT: n finishes in one call; partials[i] (the same object may repeat) is what witnesses[i] is finished from"""
        n = len(witnesses)
        ps = (C.c_void_p * n)(*[p._h.value for p in partials])
        hs = (C.c_void_p * n)(*[w._h.value for w in witnesses])
        outs = (C.c_void_p * n)()
        rsp = None
        if rs is not None:
            flat, _k = _vec_cfr([v for pair in rs for v in pair])
            rsp = flat.ptr
        _ok_bool(lib().ffi_finish_rln_proofs_batch(C.byref(self._h), ps, hs, n, rsp, outs))
        return [RLNProof(C.c_void_p(outs[i])) for i in range(n)]

    def generate_rln_proof_with_witness(self, calculated_witness, witness: RLNWitnessInput) -> RLNProof:
        """public.rs:643-658: calculated_witness = the full witness as ints (or decimal strings)"""
        strs = [str(v).encode() for v in calculated_witness]
        arr = (VecU8 * max(len(strs), 1))()
        keep = []
        for i, b in enumerate(strs):
            buf = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b or b"\0")
            keep.append(buf)
            arr[i] = VecU8(C.cast(buf, C.POINTER(C.c_uint8)), len(b), len(b))
        vs = VecString(C.cast(arr, C.POINTER(VecU8)), len(strs), len(strs))
        return RLNProof(_ok_ptr(lib().ffi_generate_rln_proof_with_witness(C.byref(self._h), C.byref(vs),
                                                                          C.byref(witness._h))))

    def generate_partial_zk_proof(self, partial_witness: RLNPartialWitnessInput) -> RLNPartialProof:
        """public.rs:651-658"""
        return RLNPartialProof(_ok_ptr(lib().ffi_generate_partial_zk_proof(C.byref(self._h),
                                                                           C.byref(partial_witness._h))))

    def finish_rln_proof(self, partial: RLNPartialProof, witness: RLNWitnessInput) -> RLNProof:
        """public.rs:674-683"""
        return RLNProof(_ok_ptr(lib().ffi_finish_rln_proof(C.byref(self._h), C.byref(partial._h),
                                                           C.byref(witness._h))))

    def finish_rln_proof_with_rs(self, partial: RLNPartialProof, witness: RLNWitnessInput, r, s) -> RLNProof:
        return RLNProof(_ok_ptr(lib().ffi_finish_rln_proof_with_rs(C.byref(self._h), C.byref(partial._h),
                                                                   C.byref(witness._h), C.byref(_cfr(r)),
                                                                   C.byref(_cfr(s)))))

    def verify_rln_proof(self, proof: RLNProof, x) -> bool:
        return _ok_bool(lib().ffi_verify_rln_proof(C.byref(self._h), C.byref(proof._h), C.byref(_cfr(x))))

    def verify_with_roots(self, proof: RLNProof, x, roots) -> bool:
        v, _k = _vec_cfr(roots)
        return _ok_bool(lib().ffi_verify_with_roots(C.byref(self._h), C.byref(proof._h), C.byref(v),
                                                    C.byref(_cfr(x))))


def compute_id_secret(share1, share2) -> int:
    """protocol/slashing.rs:12-36: shares are (x, y) pairs"""
    r = lib().ffi_compute_id_secret(C.byref(_cfr(share1[0])), C.byref(_cfr(share1[1])), C.byref(_cfr(share2[0])),
                                    C.byref(_cfr(share2[1])))
    return _take_cfr(C.cast(_ok_ptr(r), C.POINTER(CFr)))


def recover_id_secret(values_1: RLNProofValues, values_2: RLNProofValues) -> int:
    """protocol/slashing.rs:43-100"""
    r = lib().ffi_recover_id_secret(C.byref(values_1._h), C.byref(values_2._h))
    return _take_cfr(C.cast(_ok_ptr(r), C.POINTER(CFr)))


def seeded_keygen(seed: bytes):
    """protocol/keygen.rs:50-65 -> (identity_secret, id_commitment)"""
    v, _k = _vec_u8(seed)
    return tuple(_take_vec_cfr(lib().ffi_seeded_key_gen(C.byref(v))))


def extended_keygen():
    """protocol/keygen.rs:31-45 -> (trapdoor, nullifier, identity_secret, id_commitment)"""
    return tuple(_take_vec_cfr(lib().ffi_extended_key_gen()))


def extended_seeded_keygen(seed: bytes):
    """protocol/keygen.rs:72-94"""
    v, _k = _vec_u8(seed)
    return tuple(_take_vec_cfr(lib().ffi_seeded_extended_key_gen(C.byref(v))))
