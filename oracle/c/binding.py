"""ORACLE (test infrastructure): ctypes binding of oracle/c/liboracle.so, the C restatement of the reference's
CPU proving path.  Used by tests/ (parity at batch sizes the Python oracle cannot reach), smoke() and
bench.py's cpu_baseline leg -- never by the product."""
import ctypes as C
import os
import subprocess

import hashlib

_HERE = os.path.dirname(os.path.abspath(__file__))


def _cpu_tag():
    """-march=native objects must not travel between machines: one library per CPU model"""
    try:
        model = next(l for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        model = "unknown"
    return hashlib.sha1(model.encode()).hexdigest()[:10]


_SO = os.path.join(_HERE, "liboracle_%s.so" % _cpu_tag())
_RES = os.path.join(_HERE, "..", "..", "zerokit_amd", "resources")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "rln_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(src) > os.path.getmtime(_SO):
            subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
                                   "-pthread", src, "-o", _SO])
        L = C.CDLL(_SO)
        L.oracle_load.restype = C.c_void_p
        L.oracle_load.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        for f in ("oracle_num_inputs", "oracle_num_signals", "oracle_domain"):
            getattr(L, f).restype = C.c_size_t
            getattr(L, f).argtypes = [C.c_void_p]
        L.oracle_poseidon.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p]
        L.oracle_prove.restype = C.c_int
        L.oracle_prove.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p] + [C.c_char_p] * 5
        L.oracle_prove_many.restype = C.c_double
        L.oracle_prove_many.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_char_p,
                                        C.POINTER(C.c_int)]
        L.oracle_known_mask.restype = None
        L.oracle_known_mask.argtypes = [C.c_void_p, C.c_char_p]
        L.oracle_prove_partial.restype = C.c_int
        L.oracle_prove_partial.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
        L.oracle_finish.restype = C.c_int
        L.oracle_finish.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
        L.oracle_finish_many.restype = C.c_double
        L.oracle_finish_many.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t, C.c_int, C.c_char_p,
                                         C.POINTER(C.c_int)]
        L.oracle_selftest_mul.restype = C.c_int
        L.oracle_selftest_mul.argtypes = [C.c_uint64, C.c_size_t]
        L.oracle_mul_kind.restype = C.c_char_p
        L.oracle_mul_kind.argtypes = []
        L.oracle_num_public.restype = C.c_size_t
        L.oracle_num_public.argtypes = [C.c_void_p]
        L.oracle_input_slot.restype = C.c_int
        L.oracle_input_slot.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.oracle_public_values.restype = None
        L.oracle_public_values.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.oracle_tree_root.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.c_char_p]
        L.oracle_msm_expected.restype = None
        L.oracle_msm_expected.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_char_p]
        L.oracle_msm_workload_item.restype = None
        L.oracle_msm_workload_item.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_char_p, C.c_char_p]
        L.oracle_msm_expected_g2.restype = None
        L.oracle_msm_expected_g2.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_char_p]
        L.oracle_msm_workload_item_g2.restype = None
        L.oracle_msm_workload_item_g2.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_char_p, C.c_char_p]
        L.oracle_msm_pippenger.restype = C.c_double
        L.oracle_msm_pippenger.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_int, C.c_int, C.c_char_p,
                                           C.POINTER(C.c_double)]
        L.oracle_tree_new.restype = C.c_void_p
        L.oracle_tree_new.argtypes = [C.c_int]
        L.oracle_tree_free.argtypes = [C.c_void_p]
        L.oracle_tree_set_range.restype = C.c_int
        L.oracle_tree_set_range.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
        L.oracle_tree_get_root.argtypes = [C.c_void_p, C.c_char_p]
        L.oracle_tree_proof.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_char_p]
        L.oracle_tree_bench.restype = None
        L.oracle_tree_bench.argtypes = [C.c_int, C.c_size_t, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(C.c_double), C.c_char_p, C.c_char_p]
        _lib = L
    return _lib


def _b(x):
    return int(x).to_bytes(32, "little")


class Circuit:
    def __init__(self, depth=20, multi=False):
        d = os.path.join(_RES, "tree_depth_%d%s" % (depth, "_multi_max_out_4" if multi else ""))
        z = open(os.path.join(d, "rln_final.arkzkey"), "rb").read()
        g = open(os.path.join(d, "graph.bin"), "rb").read()
        self.h = lib().oracle_load(z, len(z), g, len(g))
        if not self.h:
            raise RuntimeError("oracle_load failed")
        self.n_inputs = lib().oracle_num_inputs(self.h)
        self.n_signals = lib().oracle_num_signals(self.h)
        self.domain = lib().oracle_domain(self.h)
        self.n_public = lib().oracle_num_public(self.h)
        self.multi = multi
        self.slots = {}
        for name in ("identitySecret", "userMessageLimit", "messageId", "pathElements", "identityPathIndex", "x",
                     "externalNullifier", "selectorUsed"):
            off, ln = C.c_uint32(), C.c_uint32()
            if lib().oracle_input_slot(self.h, name.encode(), C.byref(off), C.byref(ln)) == 0:
                self.slots[name] = (off.value, ln.value)

    # slots of the shipped single-message graphs (graph.bin metadata; asserted in tests/test_oracle_c.py)
    SLOTS = dict(x=1, external_nullifier=2, identity_secret=3, user_message_limit=4, message_id=5, path_elements=6)

    def pack(self, w):
        depth = len(w["path_elements"])
        buf = bytearray(self.n_inputs * 32)
        buf[0] = 1

        def put(slot, v):
            buf[slot * 32:(slot + 1) * 32] = _b(v)
        put(1, w["x"])
        put(2, w["external_nullifier"])
        put(3, w["identity_secret"])
        put(4, w["user_message_limit"])
        put(5, w["message_id"])
        for i, e in enumerate(w["path_elements"]):
            put(6 + i, e)
        for i, e in enumerate(w["identity_path_index"]):
            put(6 + depth + i, e)
        return bytes(buf)

    def pack_named(self, named):
        """named: {graph signal name: [ints]} (witness.rs:832-881) -> the graph's inputs buffer, slot 0 = 1"""
        buf = bytearray(self.n_inputs * 32)
        buf[0] = 1
        for name, vals in named.items():
            off, ln = self.slots[name]
            assert len(vals) == ln, (name, len(vals), ln)
            for k, v in enumerate(vals):
                buf[(off + k) * 32:(off + k + 1) * 32] = _b(v)
        return bytes(buf)

    def public_values(self, packed):
        out = C.create_string_buffer(32 * self.n_public)
        lib().oracle_public_values(self.h, packed, out)
        return [int.from_bytes(out.raw[32 * k:32 * k + 32], "little") for k in range(self.n_public)]

    def prove_packed(self, packed, r, s, want_witness=False):
        """one proof from a packed inputs buffer (any shipped circuit) -> dict(proof, public_inputs[, witness])"""
        proof = C.create_string_buffer(128)
        wit = C.create_string_buffer(32 * self.n_signals) if want_witness else None
        rc = lib().oracle_prove(self.h, packed, _b(r) + _b(s), proof, None, None, wit, None)
        if rc:
            raise RuntimeError("oracle_prove rc=%d" % rc)
        out = dict(proof=proof.raw, public_inputs=self.public_values(packed))
        if wit is not None:
            out["witness"] = [int.from_bytes(wit.raw[32 * i:32 * i + 32], "little") for i in range(self.n_signals)]
        return out

    def prove_many_packed(self, inputs, rsb, threads=None):
        """n proofs from packed buffers on host threads -> (seconds, [proof128], [public inputs])"""
        n = len(rsb) // 64
        assert len(inputs) == n * self.n_inputs * 32
        threads = threads or os.cpu_count() or 1
        proofs = C.create_string_buffer(128 * n)
        values = C.create_string_buffer(32 * self.n_public * n)
        rc = C.c_int(0)
        secs = lib().oracle_prove_many(self.h, inputs, rsb, n, threads, proofs, values, C.byref(rc))
        if rc.value:
            raise RuntimeError("oracle_prove_many rc=%d" % rc.value)
        np_ = self.n_public
        pub = [[int.from_bytes(values.raw[32 * (np_ * i + k):32 * (np_ * i + k + 1)], "little") for k in range(np_)]
               for i in range(n)]
        return secs, [proofs.raw[128 * i:128 * (i + 1)] for i in range(n)], pub

    def known_mask(self):
        """evaluate_partial's knownness per witness signal (entry 0 = the constant 1): PartialProof::mask with a leading True"""
        buf = C.create_string_buffer(self.n_signals)
        lib().oracle_known_mask(self.h, buf)
        return [bool(b) for b in buf.raw]

    def prove_partial_packed(self, packed):
        """generate_partial_zk_proof from a packed inputs buffer (unknown slots are not read) -> 320 bytes
        (pi_a | rho | pi_b | pi_c, affine canonical LE: the layout of rlnamd_prover_download_partial)"""
        out = C.create_string_buffer(320)
        rc = lib().oracle_prove_partial(self.h, packed, out, None)
        if rc:
            raise RuntimeError("oracle_prove_partial rc=%d" % rc)
        return out.raw

    def finish_packed(self, packed, r, s, partial320):
        """finish_zk_proof_with_rs -> the 128-byte compressed proof"""
        proof = C.create_string_buffer(128)
        rc = lib().oracle_finish(self.h, packed, _b(r) + _b(s), partial320, proof)
        if rc:
            raise RuntimeError("oracle_finish rc=%d" % rc)
        return proof.raw

    def finish_many_packed(self, inputs, rsb, partials, threads=None):
        """n finishes on host threads -> (seconds, [proof128])"""
        n = len(rsb) // 64
        assert len(inputs) == n * self.n_inputs * 32 and len(partials) == 320 * n
        threads = threads or os.cpu_count() or 1
        proofs = C.create_string_buffer(128 * n)
        rc = C.c_int(0)
        secs = lib().oracle_finish_many(self.h, inputs, rsb, partials, n, threads, proofs, C.byref(rc))
        if rc.value:
            raise RuntimeError("oracle_finish_many rc=%d" % rc.value)
        return secs, [proofs.raw[128 * i:128 * (i + 1)] for i in range(n)]

    def prove(self, w, r, s, want_witness=False, want_h=False):
        proof = C.create_string_buffer(128)
        coords = C.create_string_buffer(256)
        values = C.create_string_buffer(160)
        wit = C.create_string_buffer(32 * self.n_signals) if want_witness else None
        hb = C.create_string_buffer(32 * self.domain) if want_h else None
        rc = lib().oracle_prove(self.h, self.pack(w), _b(r) + _b(s), proof, coords, values, wit, hb)
        if rc:
            raise RuntimeError("oracle_prove rc=%d" % rc)
        out = dict(proof=proof.raw,
                   coords=[int.from_bytes(coords.raw[32 * i:32 * i + 32], "little") for i in range(8)],
                   public_inputs=[int.from_bytes(values.raw[32 * i:32 * i + 32], "little") for i in range(5)])
        if wit is not None:
            out["witness"] = [int.from_bytes(wit.raw[32 * i:32 * i + 32], "little") for i in range(self.n_signals)]
        if hb is not None:
            out["h"] = [int.from_bytes(hb.raw[32 * i:32 * i + 32], "little") for i in range(self.domain)]
        return out

    def prove_many(self, ws, rs, threads=None):
        """-> (seconds, [proof128], [public_inputs])"""
        n = len(ws)
        threads = threads or os.cpu_count() or 1
        inputs = b"".join(self.pack(w) for w in ws)
        rsb = b"".join(_b(r) + _b(s) for r, s in rs)
        proofs = C.create_string_buffer(128 * n)
        values = C.create_string_buffer(160 * n)
        rc = C.c_int(0)
        secs = lib().oracle_prove_many(self.h, inputs, rsb, n, threads, proofs, values, C.byref(rc))
        if rc.value:
            raise RuntimeError("oracle_prove_many rc=%d" % rc.value)
        pub = [[int.from_bytes(values.raw[160 * i + 32 * k:160 * i + 32 * k + 32], "little") for k in range(5)]
               for i in range(n)]
        return secs, [proofs.raw[128 * i:128 * (i + 1)] for i in range(n)], pub


def poseidon_batch(rows):
    arity = len(rows[0])
    buf = b"".join(_b(v) for r in rows for v in r)
    out = C.create_string_buffer(32 * len(rows))
    lib().oracle_poseidon(buf, len(rows), arity, out)
    return [int.from_bytes(out.raw[32 * i:32 * i + 32], "little") for i in range(len(rows))]


def tree_root(depth, leaves):
    out = C.create_string_buffer(32)
    lib().oracle_tree_root(depth, b"".join(_b(v) for v in leaves), len(leaves), out)
    return int.from_bytes(out.raw, "little")


def _xy(buf):
    x, y = int.from_bytes(buf[:32], "little"), int.from_bytes(buf[32:64], "little")
    return None if x == 0 and y == 0 else (x, y)


# mode bits of the config-5 workload (rln_oracle.c "config 5"): the product's generator takes the same mask
MSM_EQUAL_SCALARS = 1
MSM_FOUR_POINTS = 2


def msm_expected(seed, first, n, mode=0, threads=None):
    """(sum k_i s_i mod r) G over the index range of the config-5 workload: the closed form the device MSM is
    checked against at any size (2^24: ~1 s on 8 threads)"""
    out = C.create_string_buffer(64)
    lib().oracle_msm_expected(seed, first, n, mode, threads or usable_cores(), out)
    return _xy(out.raw)


def msm_workload_item(seed, i, mode=0):
    """-> (point i of the workload as (x, y) or None, scalar i)"""
    p, s = C.create_string_buffer(64), C.create_string_buffer(32)
    lib().oracle_msm_workload_item(seed, i, mode, p, s)
    return _xy(p.raw), int.from_bytes(s.raw, "little")


def _g2(buf):
    v = [int.from_bytes(buf[32 * k:32 * k + 32], "little") for k in range(4)]
    return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))


def msm_expected_g2(seed, first, n, mode=0, threads=None):
    """(sum k_i s_i mod r) G2 for the config-5 workload on the twist -> ((x.c0, x.c1), (y.c0, y.c1)) or None"""
    out = C.create_string_buffer(128)
    lib().oracle_msm_expected_g2(seed, first, n, mode, threads or usable_cores(), out)
    return _g2(out.raw)


def msm_workload_item_g2(seed, i, mode=0):
    p, s = C.create_string_buffer(128), C.create_string_buffer(32)
    lib().oracle_msm_workload_item_g2(seed, i, mode, p, s)
    return _g2(p.raw), int.from_bytes(s.raw, "little")


def msm_pippenger(seed, first, n, mode=0, threads=None):
    """msm_bigint's windowed Pippenger over the materialised workload -> (point, seconds of the MSM, seconds of the
    untimed generation)"""
    out, gen = C.create_string_buffer(64), C.c_double(0)
    secs = lib().oracle_msm_pippenger(seed, first, n, mode, threads or usable_cores(), out, C.byref(gen))
    return _xy(out.raw), secs, gen.value


class Tree:
    """FullMerkleTree restated in C (full_merkle_tree.rs): set_range / root / proof"""

    def __init__(self, depth):
        self.depth = depth
        self.h = lib().oracle_tree_new(depth)

    def close(self):
        if self.h:
            lib().oracle_tree_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_range(self, start, leaves, threads=1):
        if lib().oracle_tree_set_range(self.h, start, b"".join(_b(v) for v in leaves), len(leaves), threads):
            raise RuntimeError("oracle tree: range does not fit")

    def set(self, index, leaf):
        self.set_range(index, [leaf])

    def root(self):
        out = C.create_string_buffer(32)
        lib().oracle_tree_get_root(self.h, out)
        return int.from_bytes(out.raw, "little")

    def proof(self, index):
        e, b = C.create_string_buffer(32 * self.depth), C.create_string_buffer(max(self.depth, 1))
        lib().oracle_tree_proof(self.h, index, e, b)
        return ([int.from_bytes(e.raw[32 * i:32 * i + 32], "little") for i in range(self.depth)],
                list(b.raw[:self.depth]))


def tree_bench(depth, n, first_value=1, threads=None, singles=200, scattered=1000):
    """CPU legs of config 3 and of the tree-mutation calls (see oracle_tree_bench)"""
    out = (C.c_double * 4)()
    root, root_after = C.create_string_buffer(32), C.create_string_buffer(32)
    threads = threads or usable_cores()
    lib().oracle_tree_bench(depth, n, first_value, threads, singles, scattered, out, root, root_after)
    return dict(build_s=out[0], single_update_root_s=out[1], scattered_updates_root_s=out[2], paths_s=out[3],
                root=int.from_bytes(root.raw, "little"), root_after_scattered=int.from_bytes(root_after.raw, "little"),
                threads=threads, singles=singles, scattered=scattered)


def scattered_updates(n, count, seed=0x5CA7, tag=0x5CA7000000000000):
    """the (index, leaf) stream oracle_tree_bench applies in its scattered leg"""
    M = (1 << 64) - 1

    def sm(j):
        z = (seed + (j + 1) * 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)
    return [(sm(k) % n, tag + k) for k in range(count)]


def usable_cores():
    """host threads this process may really use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def time_baseline(ws, rs, target_seconds=12.0):
    """bench.py cpu_baseline: one proof per host thread on every usable core, bounded sample."""
    c = Circuit(20)
    cores = usable_cores()
    t1, _, _ = c.prove_many(ws[:1], rs[:1], threads=1)      # single-thread latency
    n = min(len(ws), max(cores, int(target_seconds * cores / max(t1, 1e-3))))
    secs, _, _ = c.prove_many(ws[:n], rs[:n], threads=cores)
    return {"value": round(n / secs, 3), "unit": "proofs/s", "cores": cores, "kind": "port",
            "single_thread_ms_per_proof": round(t1 * 1e3, 2),
            "field_product": lib().oracle_mul_kind().decode(),
            "sample": "%d proofs of the same config-2 witnesses, one proof per thread on %d threads "
                      "(oracle/c: arkworks-equivalent CPU path restated in C -- 4x64-bit Montgomery with the mulx / adcx / adox "
                      "product ark-ff-asm emits where the CPU has ADX, msm_bigint with ark-ec 0.5.0's signed digits; round 5's "
                      "port, portable product and unsigned digits, took 338 ms per proof on this kind of host)"
                      % (n, cores)}
