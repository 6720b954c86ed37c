"""Extension surface (include/rln_amd.h): device-resident batch prover and Poseidon tree."""
import ctypes as C
import os

from ._native import ProverInfo, RLNError, check, lib

_HERE = os.path.dirname(os.path.abspath(__file__))
STAGES = 8


def resource_paths(depth=20, multi=False):
    d = os.path.join(_HERE, "resources", "tree_depth_%d%s" % (depth, "_multi_max_out_4" if multi else ""))
    return os.path.join(d, "rln_final.arkzkey"), os.path.join(d, "graph.bin")


_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def _b(x: int, modulus: int = _R) -> bytes:
    """canonical 32-byte LE form; values outside [0, modulus) are refused (the kernels would silently see a
    different residue from the one the host-side checks compared)"""
    x = int(x)
    if not 0 <= x < modulus:
        raise RLNError("Non-canonical field element: value is not in [0, %s-1]" % ("r" if modulus == _R else "q"))
    return x.to_bytes(32, "little")


def _check_verify_shapes(proofs, public_inputs):
    """the C entry points index proofs + 128 i and values + 32 nv i: refuse anything that is not exactly that shape"""
    if len(proofs) != len(public_inputs):
        raise RLNError("verify_many: %d proofs but %d public-input rows" % (len(proofs), len(public_inputs)))
    nv = len(public_inputs[0])
    for i, (p, row) in enumerate(zip(proofs, public_inputs)):
        if len(p) != 128:
            raise RLNError("verify_many: proof %d is %d bytes, expected 128" % (i, len(p)))
        if len(row) != nv:
            raise RLNError("verify_many: public-input row %d has %d entries, expected %d" % (i, len(row), nv))
    return nv


def _unpack_results(proofs, values, errs):
    out = []
    for i in range(len(errs)):
        v = values[160 * i:160 * (i + 1)]
        vals = [int.from_bytes(v[32 * k:32 * k + 32], "little") for k in range(5)]
        out.append(dict(proof=proofs[128 * i:128 * (i + 1)],
                        values=dict(y=vals[0], root=vals[1], nullifier=vals[2], x=vals[3], external_nullifier=vals[4]),
                        public_inputs=vals, error=int(errs[i])))
    return out


class BatchProver:
    """n x generate_zk_proof_with_rs (/root/reference/rln/src/protocol/proof.rs:753-777) +
    proof_values_from_witness (protocol/witness.rs:759-804) in one device batch."""

    def __init__(self, zkey: bytes = None, graph: bytes = None, max_batch=1024, window_bits=0, depth=20, multi=False):
        if zkey is None or graph is None:
            zp, gp = resource_paths(depth, multi)
            zkey, graph = open(zp, "rb").read(), open(gp, "rb").read()
        self._h = C.c_void_p()
        check(lib().rlnamd_prover_new(zkey, len(zkey), graph, len(graph), max_batch, window_bits, C.byref(self._h)))
        info = ProverInfo()
        check(lib().rlnamd_prover_get_info(self._h, C.byref(info)))
        self.info = info
        self.inputs_size = int(info.inputs_size)
        self.slots = {}
        for name in ("identitySecret", "userMessageLimit", "messageId", "pathElements", "identityPathIndex", "x",
                     "externalNullifier", "selectorUsed"):
            off, ln = C.c_uint32(), C.c_uint32()
            if lib().rlnamd_prover_input_slot(self._h, name.encode(), C.byref(off), C.byref(ln)) == 0:
                self.slots[name] = (off.value, ln.value)
        self.num_public = int(lib().rlnamd_prover_num_public(self._h))

    def close(self):
        if self._h:
            lib().rlnamd_prover_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pack_inputs(self, witnesses):
        """witnesses: dicts with identity_secret, user_message_limit, message_id, path_elements,
        identity_path_index, x, external_nullifier (ints).  -> bytes (n * inputs_size * 32), the
        witness-graph inputs buffer of iden3calc.rs:122-181 (slot 0 = 1)."""
        out = bytearray(len(witnesses) * self.inputs_size * 32)
        names = {"identitySecret": "identity_secret", "userMessageLimit": "user_message_limit",
                 "messageId": "message_id", "pathElements": "path_elements",
                 "identityPathIndex": "identity_path_index", "x": "x", "externalNullifier": "external_nullifier",
                 "selectorUsed": "selector_used"}
        for i, w in enumerate(witnesses):
            base = i * self.inputs_size * 32
            out[base] = 1
            for sig, key in names.items():
                if sig not in self.slots:
                    continue
                off, ln = self.slots[sig]
                v = w[key]
                vals = list(v) if isinstance(v, (list, tuple)) else [v]
                if len(vals) != ln:
                    raise RLNError("invalid input length for %s: expected %d, got %d" % (sig, ln, len(vals)))
                for k, x in enumerate(vals):
                    out[base + (off + k) * 32: base + (off + k + 1) * 32] = _b(x)
        return bytes(out)

    def pack_named_inputs(self, named):
        """named: list of {graph signal name: [ints]} (any circuit): the populate_inputs of iden3calc.rs:122-146"""
        out = bytearray(len(named) * self.inputs_size * 32)
        for i, w in enumerate(named):
            base = i * self.inputs_size * 32
            out[base] = 1
            for sig, vals in w.items():
                if sig not in self.slots:
                    raise RLNError("MissingInput: " + sig)
                off, ln = self.slots[sig]
                if len(vals) != ln:
                    raise RLNError("invalid input length for %s: expected %d, got %d" % (sig, ln, len(vals)))
                for k, x in enumerate(vals):
                    out[base + (off + k) * 32: base + (off + k + 1) * 32] = _b(x)
        return bytes(out)

    def upload(self, inputs: bytes, rs):
        n = len(inputs) // (self.inputs_size * 32)
        rsb = b"".join(_b(r) + _b(s) for r, s in rs)
        assert len(rsb) == 64 * n
        check(lib().rlnamd_prover_upload(self._h, n, inputs, rsb))
        return n

    def run(self, n):
        check(lib().rlnamd_prover_run(self._h, n))

    def run_async(self, n):
        """enqueue only; consecutive batches pipeline on the device (see Prover::run_async)"""
        check(lib().rlnamd_prover_run_async(self._h, n))

    def sync(self):
        check(lib().rlnamd_prover_sync(self._h))

    def download(self, n):
        proofs = C.create_string_buffer(128 * n)
        coords = C.create_string_buffer(256 * n)
        values = C.create_string_buffer(160 * n)
        errs = (C.c_uint32 * n)()
        check(lib().rlnamd_prover_download(self._h, n, proofs, coords, values, errs))
        out = []
        for i in range(n):
            c = coords.raw[256 * i:256 * (i + 1)]
            co = [int.from_bytes(c[32 * k:32 * k + 32], "little") for k in range(8)]
            v = values.raw[160 * i:160 * (i + 1)]
            vals = [int.from_bytes(v[32 * k:32 * k + 32], "little") for k in range(5)]
            out.append(dict(proof=proofs.raw[128 * i:128 * (i + 1)],
                            a=(co[0], co[1]), b=((co[2], co[3]), (co[4], co[5])), c=(co[6], co[7]),
                            values=dict(y=vals[0], root=vals[1], nullifier=vals[2], x=vals[3],
                                        external_nullifier=vals[4]),
                            public_inputs=vals, error=int(errs[i])))
        return out

    def prove(self, witnesses, rs):
        n = self.upload(self.pack_inputs(witnesses), rs)
        self.run(n)
        return self.download(n)

    # ---- streamed batches (rlnamd_prover_submit / _collect): fresh inputs per batch, no pipeline drain
    @staticmethod
    def pack_rs(rs):
        return b"".join(_b(r) + _b(s) for r, s in rs)

    def n_slots(self):
        """workspace slots = batches that can be in flight between submit and collect"""
        return int(lib().rlnamd_prover_slots(self._h))

    def submit(self, inputs: bytes, rsb: bytes, mode=0, partials=None):
        """inputs = pack_inputs(...), rsb = pack_rs(...); returns (ticket, n).  Enqueue only."""
        n = len(inputs) // (self.inputs_size * 32)
        if len(inputs) != n * self.inputs_size * 32 or len(rsb) != 64 * n:
            raise RLNError("submit: inputs / rs sizes do not match")
        pp = None
        if partials is not None:
            pp = b"".join(partials)
            if len(pp) != 320 * n:
                raise RLNError("submit: one 320-byte partial proof per proof expected")
        t = C.c_uint64()
        check(lib().rlnamd_prover_submit(self._h, n, inputs, rsb, mode, pp, C.byref(t)))
        return int(t.value), n

    def hints_for(self, inputs: bytes):
        """the hints of the proofs packed in `inputs`, one proof at a time on this thread (rlnamd_prover_hints_for): a list
        of ctypes arrays, or None when the circuit has no segments form"""
        hw = int(lib().rlnamd_prover_hint_words(self._h))
        if not hw:
            return None
        isz = self.inputs_size * 32
        out = []
        for i in range(len(inputs) // isz):
            h = (C.c_uint32 * hw)()
            check(lib().rlnamd_prover_hints_for(self._h, inputs[i * isz:(i + 1) * isz], h))
            out.append(h)
        return out

    def submit_hinted(self, inputs: bytes, rsb: bytes, hints):
        """submit() for full proofs whose hints (hints_for) are at hand: nothing is hashed inside the call"""
        n = len(inputs) // (self.inputs_size * 32)
        if len(inputs) != n * self.inputs_size * 32 or len(rsb) != 64 * n or len(hints) != n:
            raise RLNError("submit_hinted: inputs / rs / hints sizes do not match")
        hw = len(hints[0])
        flat = (C.c_uint32 * (hw * n))()
        for i, h in enumerate(hints):
            flat[i * hw:(i + 1) * hw] = h[:]
        t = C.c_uint64()
        check(lib().rlnamd_prover_submit_hinted(self._h, n, inputs, rsb, flat, C.byref(t)))
        return int(t.value), n

    def describe(self):
        """the switches this prover was built with (ProverTuning)"""
        buf = C.create_string_buffer(1024)
        check(lib().rlnamd_prover_describe(self._h, buf, 1024))
        return buf.value.decode()

    def wipe(self):
        """overwrite the resident inputs and the last run's witness values (the streamed calls do it by themselves)"""
        check(lib().rlnamd_prover_wipe(self._h))

    def collect_raw(self, ticket, n):
        """waits for that batch only -> (proofs bytes n*128, values bytes n*160, errors list)"""
        proofs = C.create_string_buffer(128 * n)
        values = C.create_string_buffer(160 * n)
        errs = (C.c_uint32 * n)()
        check(lib().rlnamd_prover_collect(self._h, ticket, n, proofs, None, values, errs, None))
        return proofs.raw, values.raw, list(errs)

    def collect(self, ticket, n):
        proofs, values, errs = self.collect_raw(ticket, n)
        return _unpack_results(proofs, values, errs)

    def collect_partial(self, ticket, n):
        buf = C.create_string_buffer(320 * n)
        check(lib().rlnamd_prover_collect(self._h, ticket, n, None, None, None, None, buf))
        return [buf.raw[320 * i:320 * (i + 1)] for i in range(n)]

    def collect_partial_cached(self, ticket, n):
        """collect of a mode-1 batch that keeps the known stored values on the device -> ([partial320], [handle], errors);
        handle 0 = not cached (such a proof finishes through the full interpreter)"""
        buf = C.create_string_buffer(320 * n)
        hs = (C.c_uint64 * n)()
        errs = (C.c_uint32 * n)()
        check(lib().rlnamd_prover_collect_partial_cached(self._h, ticket, n, buf, hs, errs))
        return [buf.raw[320 * i:320 * (i + 1)] for i in range(n)], [int(h) for h in hs], list(errs)

    def submit_finish(self, inputs: bytes, rsb: bytes, partials, handles):
        """finish_zk_proof_with_rs with the partial runs' cache handles: the cone of the witness graph only when every
        handle is live (rlnamd_prover_submit_finish); returns (ticket, n)"""
        n = len(inputs) // (self.inputs_size * 32)
        pp = b"".join(partials)
        if len(inputs) != n * self.inputs_size * 32 or len(rsb) != 64 * n or len(pp) != 320 * n or len(handles) != n:
            raise RLNError("submit_finish: sizes do not match")
        hs = (C.c_uint64 * n)(*handles)
        t = C.c_uint64()
        check(lib().rlnamd_prover_submit_finish(self._h, n, inputs, rsb, pp, hs, C.byref(t)))
        return int(t.value), n

    def release_partial(self, handles):
        hs = (C.c_uint64 * len(handles))(*handles)
        check(lib().rlnamd_prover_release_partial(self._h, hs, len(handles)))

    def hint_stats(self):
        """the witness graph as segments behind hints (rlnamd_prover_hint_stats)"""
        out = (C.c_uint64 * 7)()
        check(lib().rlnamd_prover_hint_stats(self._h, out))
        return dict(zip(("segments", "hints", "longest_segment_steps", "full_steps", "hinted_batches", "fallbacks", "chains_remembered"),
                        [int(v) for v in out]))

    def device_shared(self):
        """bit 0: another prover of this process on the device; bit 1: a prover of another process"""
        who = C.c_int()
        check(lib().rlnamd_prover_device_shared(self._h, C.byref(who)))
        return int(who.value)

    def partial_cache_info(self):
        out = (C.c_uint64 * 8)()
        check(lib().rlnamd_prover_partial_cache_info(self._h, out))
        return dict(zip(("capacity", "in_use", "entry_bytes", "residue_in_free_entries", "cone_batches", "cone_nodes",
                         "cone_steps", "full_steps"), [int(v) for v in out]))

    def prove_stream_raw(self, inputs: bytes, rsb: bytes):
        """any n through rlnamd_prover_prove_stream (chunks of `capacity`, all slots in flight)"""
        n = len(inputs) // (self.inputs_size * 32)
        if len(inputs) != n * self.inputs_size * 32 or len(rsb) != 64 * n:
            raise RLNError("prove_stream: inputs / rs sizes do not match")
        proofs = C.create_string_buffer(128 * n)
        values = C.create_string_buffer(160 * n)
        errs = (C.c_uint32 * max(n, 1))()
        check(lib().rlnamd_prover_prove_stream(self._h, n, inputs, rsb, proofs, values, errs))
        return proofs.raw, values.raw, list(errs)[:n]

    def prove_stream(self, witnesses, rs):
        return _unpack_results(*self.prove_stream_raw(self.pack_inputs(witnesses), self.pack_rs(rs)))

    # ---- partial proofs (protocol/proof.rs:783-849)
    def known_mask(self):
        buf = C.create_string_buffer(int(self.info.num_signals))
        check(lib().rlnamd_prover_known_mask(self._h, buf))
        return list(buf.raw)

    def prove_partial(self, partial_witnesses):
        """partial_witnesses: dicts with identity_secret, user_message_limit, path_elements,
        identity_path_index.  -> list of 320-byte partial proofs (pi_a | rho | pi_b | pi_c, affine LE)."""
        full = [dict(w, message_id=0, x=0, external_nullifier=0) for w in partial_witnesses]
        n = self.upload(self.pack_inputs(full), [(0, 0)] * len(full))
        check(lib().rlnamd_prover_run_mode(self._h, n, 1))
        buf = C.create_string_buffer(320 * n)
        check(lib().rlnamd_prover_download_partial(self._h, n, buf))
        return [buf.raw[320 * i:320 * (i + 1)] for i in range(n)]

    def upload_partial(self, partials):
        check(lib().rlnamd_prover_upload_partial(self._h, len(partials), b"".join(partials)))

    def prove_with_witness(self, witnesses, rs, calculated):
        """generate_zk_proof_with_witness for a batch: `calculated` = per proof the full witness (ints)"""
        n = self.upload(self.pack_inputs(witnesses), rs)
        blob = b"".join(int(v).to_bytes(32, "little") for w in calculated for v in w)
        check(lib().rlnamd_prover_upload_witness(self._h, n, blob))
        check(lib().rlnamd_prover_run_mode(self._h, n, 0))
        return self.download(n)

    def finish(self, witnesses, rs, partials):
        """finish_zk_proof_with_rs for a batch: only the message-dependent rows + h + blinding are walked"""
        n = self.upload(self.pack_inputs(witnesses), rs)
        self.upload_partial(partials)
        check(lib().rlnamd_prover_run_mode(self._h, n, 2))
        return self.download(n)

    def run_async_mode(self, n, mode):
        check(lib().rlnamd_prover_run_async_mode(self._h, n, mode))

    def download_public(self, n):
        """public signals w[1..] of the first n proofs of the last run, from the witness (circuit-generic)"""
        buf = C.create_string_buffer(32 * self.num_public * n)
        check(lib().rlnamd_prover_download_public(self._h, n, buf))
        k = self.num_public
        return [[int.from_bytes(buf.raw[32 * (i * k + j):32 * (i * k + j + 1)], "little") for j in range(k)]
                for i in range(n)]

    def verify_public(self, proof: bytes, public_inputs):
        ok = C.c_int()
        check(lib().rlnamd_verify_public(self._h, proof, b"".join(_b(v) for v in public_inputs), len(public_inputs),
                                         C.byref(ok)))
        return bool(ok.value)

    def stage_ms(self):
        ms = (C.c_float * STAGES)()
        check(lib().rlnamd_prover_stage_ms(self._h, ms))
        return {lib().rlnamd_prover_stage_name(i).decode(): float(ms[i]) for i in range(STAGES)}

    def walk_clock_mhz(self):
        """mean shader clock under the G1 / G2 table walks since the previous call (drains the pipeline)"""
        mhz = (C.c_double * 2)()
        check(lib().rlnamd_prover_walk_clock_mhz(self._h, mhz))
        return {"g1_walk": float(mhz[0]), "g2_walk": float(mhz[1])}

    def fetch_witness(self, index):
        n = int(self.info.num_signals)
        buf = C.create_string_buffer(32 * n)
        check(lib().rlnamd_prover_fetch_witness(self._h, index, buf))
        return [int.from_bytes(buf.raw[32 * i:32 * i + 32], "little") for i in range(n)]

    def fetch_h(self, index):
        n = int(self.info.domain_size)
        buf = C.create_string_buffer(32 * n)
        check(lib().rlnamd_prover_fetch_h(self._h, index, buf))
        return [int.from_bytes(buf.raw[32 * i:32 * i + 32], "little") for i in range(n)]

    def init_ms(self):
        """where the constructor's time went (rlnamd_prover_init_ms)"""
        ms = (C.c_float * 4)()
        check(lib().rlnamd_prover_init_ms(self._h, ms))
        return dict(zip(("parse", "table_alloc", "table_build", "rest"), (round(float(v), 1) for v in ms)))

    def residue(self):
        """tap of the wipes (rlnamd_prover_residue): non-zero 16-byte words left in the last batch's slot"""
        out = (C.c_uint64 * 6)()
        check(lib().rlnamd_prover_residue(self._h, out))
        return dict(zip(("digits_g1", "digits_g2", "abc", "partial_g1", "partial_g2", "inputs"), (int(v) for v in out)))

    def verify_many(self, proofs, public_inputs, threads=0):
        """n independent verifications on host threads (rlnamd_verify_many); returns a list of bools"""
        n = len(proofs)
        if n == 0:
            return []
        nv = _check_verify_shapes(proofs, public_inputs)
        ok = C.create_string_buffer(n)
        check(lib().rlnamd_verify_many(self._h, n, b"".join(proofs), b"".join(_b(v) for pi in public_inputs for v in pi),
                                       nv, threads, ok))
        return [bool(x) for x in ok.raw]

    def verify(self, proof: bytes, public_inputs):
        """verify_zk_proof (protocol/proof.rs:856-894); public_inputs = [y, root, nullifier, x, ext]."""
        ok = C.c_int()
        check(lib().rlnamd_verify(self._h, proof, b"".join(_b(v) for v in public_inputs), C.byref(ok)))
        return bool(ok.value)


def verify_many_with_zkey(zkey: bytes, proofs, public_inputs, threads=0):
    """n independent Groth16 verifications on host threads, straight from arkzkey bytes (no GPU): proofs = list of
    128-byte compressed proofs, public_inputs = list of equally long lists of ints.  Returns a list of bools."""
    n = len(proofs)
    if n == 0:
        return []
    nv = _check_verify_shapes(proofs, public_inputs)
    ok = C.create_string_buffer(n)
    check(lib().rlnamd_verify_many_with_zkey(zkey, len(zkey), n, b"".join(proofs),
                                             b"".join(_b(v) for pi in public_inputs for v in pi), nv, threads, ok))
    return [bool(x) for x in ok.raw]


class ProverPool:
    """rlnamd_pool: one prover replica + one host thread per device of this process; a job of n proofs is cut into
    contiguous index shards (BASELINE config 4: 8 x 8 192) and every replica streams its shard."""

    def __init__(self, devices=None, zkey: bytes = None, graph: bytes = None, max_batch=1024, window_bits=0, depth=20,
                 multi=False):
        if zkey is None or graph is None:
            zp, gp = resource_paths(depth, multi)
            zkey, graph = open(zp, "rb").read(), open(gp, "rb").read()
        self._h = C.c_void_p()
        devs = list(devices) if devices else []
        arr = (C.c_int * max(len(devs), 1))(*devs)
        check(lib().rlnamd_pool_new(zkey, len(zkey), graph, len(graph), max_batch, window_bits, arr if devs else None,
                                    len(devs), C.byref(self._h)))
        self.info = ProverInfo()
        check(lib().rlnamd_pool_get_info(self._h, C.byref(self.info)))
        self.inputs_size = int(self.info.inputs_size)
        self.size = int(lib().rlnamd_pool_size(self._h))
        self.devices = [int(lib().rlnamd_pool_device(self._h, i)) for i in range(self.size)]

    def close(self):
        if self._h:
            lib().rlnamd_pool_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prove_raw(self, inputs: bytes, rsb: bytes):
        n = len(inputs) // (self.inputs_size * 32)
        if len(inputs) != n * self.inputs_size * 32 or len(rsb) != 64 * n:
            raise RLNError("pool: inputs / rs sizes do not match")
        proofs = C.create_string_buffer(128 * n)
        values = C.create_string_buffer(160 * n)
        errs = (C.c_uint32 * max(n, 1))()
        check(lib().rlnamd_pool_prove(self._h, n, inputs, rsb, proofs, values, errs))
        return proofs.raw, values.raw, list(errs)[:n]

    def last_ms(self):
        ms = (C.c_float * self.size)()
        check(lib().rlnamd_pool_last_ms(self._h, ms))
        return [float(x) for x in ms]

    def set_dynamic(self, on=True):
        """chunk-granular dynamic assignment (a cursor shared by the replicas) instead of contiguous shards"""
        check(lib().rlnamd_pool_set_dynamic(self._h, 1 if on else 0))

    def last_proofs(self):
        k = (C.c_size_t * self.size)()
        check(lib().rlnamd_pool_last_proofs(self._h, k))
        return [int(x) for x in k]

    def inject_fault(self, replica, after_chunks=0):
        check(lib().rlnamd_pool_inject_fault(self._h, replica, after_chunks))

    def set_failover(self, rounds=1):
        """rounds > 0: the chunks of a failing replica are proved again by the others (the replica is quarantined until
        revive()); 0: a failing replica fails the job"""
        check(lib().rlnamd_pool_set_failover(self._h, rounds))

    def health(self):
        """[(quarantined, failed dispatches)] per replica"""
        q = (C.c_int * self.size)()
        f = (C.c_size_t * self.size)()
        check(lib().rlnamd_pool_health(self._h, q, f))
        return [(bool(a), int(b)) for a, b in zip(q, f)]

    def revive(self, replica):
        check(lib().rlnamd_pool_revive(self._h, replica))

    def set_probation(self, jobs):
        """jobs > 0: a quarantined replica sits out that many jobs and is then handed work again by itself"""
        check(lib().rlnamd_pool_set_probation(self._h, jobs))

    def verify_many(self, proofs, public_inputs, threads=0):
        n = len(proofs)
        if n == 0:
            return []
        nv = _check_verify_shapes(proofs, public_inputs)
        ok = C.create_string_buffer(n)
        check(lib().rlnamd_pool_verify_many(self._h, n, b"".join(proofs),
                                            b"".join(_b(v) for pi in public_inputs for v in pi), nv, threads, ok))
        return [bool(x) for x in ok.raw]


class Comm:
    """rlnamd_comm: an RCCL communicator created through the C ABI (no torch type crosses it)."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(lib().rlnamd_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def init_rank(cls, uid: bytes, nranks: int, rank: int):
        h = C.c_void_p()
        check(lib().rlnamd_comm_init_rank(uid, nranks, rank, C.byref(h)))
        return cls(h)

    def ranks(self):
        return int(lib().rlnamd_comm_ranks(self._h))

    def close(self):
        if self._h:
            lib().rlnamd_comm_free(self._h)
            self._h = C.c_void_p()


class PoseidonTree:
    """HBM-resident FullMerkleTree (/root/reference/utils/src/merkle_tree/full_merkle_tree.rs)."""

    def __init__(self, depth):
        self._h = C.c_void_p()
        self.depth = depth
        check(lib().rlnamd_tree_new(depth, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().rlnamd_tree_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_range(self, start, leaves):
        buf = b"".join(_b(v) for v in leaves)
        check(lib().rlnamd_tree_set_range(self._h, start, buf, len(leaves)))

    def set(self, index, leaf):
        self.set_range(index, [leaf])

    def set_leaves(self, updates):
        """[(index, leaf)] in any order, later entries win: ONE bottom-up pass over the union of the dirty paths"""
        k = len(updates)
        idx = (C.c_uint64 * max(k, 1))(*[int(i) for i, _ in updates])
        check(lib().rlnamd_tree_set_leaves(self._h, idx, b"".join(_b(v) for _, v in updates), k))

    def root(self):
        out = C.create_string_buffer(32)
        check(lib().rlnamd_tree_root(self._h, out))
        return int.from_bytes(out.raw, "little")

    def get(self, index):
        out = C.create_string_buffer(32)
        check(lib().rlnamd_tree_get_leaf(self._h, index, out))
        return int.from_bytes(out.raw, "little")

    def proof(self, index):
        e = C.create_string_buffer(32 * self.depth)
        b = C.create_string_buffer(max(self.depth, 1))
        check(lib().rlnamd_tree_proof(self._h, index, e, b))
        return ([int.from_bytes(e.raw[32 * i:32 * i + 32], "little") for i in range(self.depth)],
                list(b.raw[:self.depth]))

    def proofs(self, first, count):
        e = C.create_string_buffer(32 * self.depth * count)
        b = C.create_string_buffer(max(self.depth * count, 1))
        check(lib().rlnamd_tree_proofs(self._h, first, count, e, b))
        out = []
        for p in range(count):
            o = p * self.depth
            out.append(([int.from_bytes(e.raw[32 * (o + i):32 * (o + i + 1)], "little") for i in range(self.depth)],
                        list(b.raw[o:o + self.depth])))
        return out

    def fill_sequential(self, start, n, first_value):
        check(lib().rlnamd_tree_fill_sequential(self._h, start, n, first_value))

    def bench(self, n_leaves, first_value=1, verify=True):
        ms = (C.c_float * 2)()
        bad = C.c_size_t()
        check(lib().rlnamd_tree_bench(self._h, n_leaves, first_value, 1 if verify else 0, ms, C.byref(bad)))
        return dict(build_ms=float(ms[0]), proofs_ms=float(ms[1]), bad=int(bad.value))


class MsmG1:
    """Variable-base MSM (VariableBaseMSM::msm_bigint, ark-ec 0.5.0; BASELINE config 5) on G1; MsmG2 below is the same
    object on the twist: points are (x, y) ints for G1, ((x.c0, x.c1), (y.c0, y.c1)) for G2, None = infinity."""
    GROUP = 1

    def __init__(self, capacity):
        self._h = C.c_void_p()
        check((lib().rlnamd_msm_new if self.GROUP == 1 else lib().rlnamd_msm_new_g2)(capacity, C.byref(self._h)))
        self.ws_bytes = int(lib().rlnamd_msm_window_sums_bytes_of(self._h))
        self.pt_bytes = int(lib().rlnamd_msm_point_bytes(self._h))

    def close(self):
        if self._h:
            lib().rlnamd_msm_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _pt_bytes(self, p):
        if p is None:
            return bytes(self.pt_bytes)
        if self.GROUP == 1:
            return _b(p[0], _Q) + _b(p[1], _Q)
        return _b(p[0][0], _Q) + _b(p[0][1], _Q) + _b(p[1][0], _Q) + _b(p[1][1], _Q)

    def _pt_from(self, raw):
        v = [int.from_bytes(raw[32 * k:32 * k + 32], "little") for k in range(self.pt_bytes // 32)]
        if not any(v):
            return None
        return (v[0], v[1]) if self.GROUP == 1 else ((v[0], v[1]), (v[2], v[3]))

    def set(self, points, scalars):
        """points: list of affine points (see the class docstring) or None for infinity"""
        pb = b"".join(self._pt_bytes(p) for p in points)
        check(lib().rlnamd_msm_set(self._h, pb, b"".join(_b(s) for s in scalars), len(points)))

    EQUAL_SCALARS, FOUR_POINTS = 1, 2

    def generate(self, seed, first_index, n, mode=0):
        """config-5 workload in HBM (P_i = k_i G, s_i from the SplitMix64 stream); the expected sum is not ours to
        state -- tests and bench.py take it from the oracle (oracle.c.binding.msm_expected)"""
        check(lib().rlnamd_msm_generate_mode(self._h, seed, first_index, n, mode))

    def fetch(self, first, count):
        """-> [(point or None, scalar)] of the loaded / generated workload"""
        p, s = C.create_string_buffer(self.pt_bytes * count), C.create_string_buffer(32 * count)
        check(lib().rlnamd_msm_fetch(self._h, first, count, p, s))
        return [(self._pt_from(p.raw[self.pt_bytes * i:self.pt_bytes * (i + 1)]),
                 int.from_bytes(s.raw[32 * i:32 * i + 32], "little")) for i in range(count)]

    def run_windows(self):
        """-> (window-sum blob to all-gather, stage ms dict)"""
        buf = C.create_string_buffer(self.ws_bytes)
        ms = (C.c_float * 3)()
        check(lib().rlnamd_msm_run(self._h, buf, ms))
        return buf.raw, dict(sort_ms=float(ms[0]), bucket_acc_ms=float(ms[1]), bucket_reduce_ms=float(ms[2]))

    def combine(self, blobs):
        out = C.create_string_buffer(self.pt_bytes)
        check(lib().rlnamd_msm_combine(self._h, b"".join(blobs), len(blobs), out))
        return self._pt_from(out.raw)

    def run_sharded(self, comm: "Comm"):
        """config 5 on this rank of an RCCL communicator (collective): -> (point or None, stage ms dict)"""
        out = C.create_string_buffer(self.pt_bytes)
        ms = (C.c_float * 4)()
        check(lib().rlnamd_msm_run_sharded(self._h, comm._h, out, ms))
        return self._pt_from(out.raw), dict(sort_ms=float(ms[0]), buckets_ms=float(ms[1]),
                                            all_gather_ms=float(ms[2]), combine_ms=float(ms[3]))

    def msm(self, points, scalars):
        self.set(points, scalars)
        blob, _ = self.run_windows()
        return self.combine([blob])


class MsmG2(MsmG1):
    """the same Pippenger on G2 (rlnamd_msm_new_g2): 128-byte points, 4 KiB of window sums per rank"""
    GROUP = 2
