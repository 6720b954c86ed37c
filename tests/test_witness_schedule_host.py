"""The host-side scheduler of the lanes = nodes witness interpreter (zerokit_amd/csrc/witness_sched.cpp) without a GPU:
tests/host/witsched.cpp runs the emitted micro-op program -- dependency steps, LDS slot assignment from liveness, product /
addition fusion, reductions, row replication -- with the product's own host field arithmetic and graph operations
(/root/reference/rln/src/circuit/iden3calc/graph.rs:72-143, 246-272, 314-466), and the witness must hash to the committed
golden digests on all three shipped circuits, in lane form and in row form."""
import ctypes
import hashlib
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zerokit_amd", "csrc")
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


@pytest.fixture(scope="module")
def WS():
    so = os.path.join(ROOT, "tests", "host", "libwitsched.so")
    srcs = [os.path.join(ROOT, "tests", "host", "witsched.cpp"), os.path.join(CSRC, "witness_sched.cpp"),
            os.path.join(CSRC, "zkey.cpp")]
    deps = srcs + [os.path.join(CSRC, h) for h in ("witness_sched.h", "witness_ops.h", "zkey.h", "field.h", "curve.h")]
    if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I",
                               "/opt/rocm/include", "-I", CSRC] + srcs + ["-o", so])
    lib = ctypes.CDLL(so)
    lib.witsched_error.restype = ctypes.c_char_p
    return lib


def _run(lib, graph_bytes, named_inputs, rows):
    from oracle.pyref import wtns_graph
    g = wtns_graph.parse(graph_bytes)
    size = g.inputs_size()
    buf = bytearray(size * 32)
    buf[0] = 1
    for name, vals in named_inputs.items():
        off, ln = g.input_mapping[name]
        assert ln == len(vals)
        for k, v in enumerate(vals):
            buf[(off + k) * 32:(off + k + 1) * 32] = (int(v) % R).to_bytes(32, "little")
    out = ctypes.create_string_buffer(32 * len(g.signals))
    stats = (ctypes.c_uint32 * 8)()
    rc = lib.witsched_run(graph_bytes, len(graph_bytes), bytes(buf), size, rows, out, stats)
    assert rc == 0, lib.witsched_error().decode()
    return hashlib.sha256(out.raw).hexdigest(), list(stats)


def _graph(sub):
    return open(os.path.join(ROOT, "zerokit_amd", "resources", sub, "graph.bin"), "rb").read()


@pytest.mark.parametrize("rows", [0, 1])
def test_schedule_reproduces_the_golden_witness_depth20(WS, rows):
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))["cases"]
    gb = _graph("tree_depth_20")
    seen = None
    for c in cases:
        w = c["witness"]
        named = {"identitySecret": [w["identity_secret"]], "userMessageLimit": [w["user_message_limit"]],
                 "messageId": [w["message_id"]], "pathElements": w["path_elements"],
                 "identityPathIndex": w["identity_path_index"], "x": [w["x"]],
                 "externalNullifier": [w["external_nullifier"]]}
        digest, st = _run(WS, gb, named, rows)
        assert digest == c["witness_sha256"], c["name"]
        assert st[7] == 0
        seen = st
    steps, nrow, nfma, nsqr, nadd, nmisc, peak, _ = seen
    # the shipped circuit has 13 972 products at a multiplication depth of 5 736 as the circuit compiler leaves it
    # (7 300 steps with the sums in source order, ~6 100 with the sums re-associated by arrival time); with the linear
    # forms on the critical path re-expressed (witness_sched.cpp: hoist_linear_forms -- a Poseidon partial round as three
    # dependent products instead of four) the depth is ~4 300 and the schedule ~4 700 (lane form) / ~4 800 (row form:
    # four product slots per step, 18 600 micro-ops)
    print("schedule rows=%d: steps %d (row %d fma %d sqr %d add %d misc %d) peak slots %d"
          % (rows, steps, nrow, nfma, nsqr, nadd, nmisc, peak))
    assert 4300 <= steps < 5100 and peak < 600
    if rows:
        assert nrow > 4300 and nfma == 0 and nsqr == 0
    else:
        assert nrow == 0 and nfma + nsqr > 4300


def test_the_schedule_without_the_linear_form_pass_is_the_older_one(WS, monkeypatch):
    """RLNAMD_WL_HOIST=0 keeps the graph as the compiler left it (sums still re-associated): the same witness, ~6 100 steps"""
    monkeypatch.setenv("RLNAMD_WL_HOIST", "0")
    c = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))["cases"][0]
    w = c["witness"]
    named = {"identitySecret": [w["identity_secret"]], "userMessageLimit": [w["user_message_limit"]],
             "messageId": [w["message_id"]], "pathElements": w["path_elements"],
             "identityPathIndex": w["identity_path_index"], "x": [w["x"]],
             "externalNullifier": [w["external_nullifier"]]}
    digest, st = _run(WS, _graph("tree_depth_20"), named, 1)
    assert digest == c["witness_sha256"] and st[7] == 0
    assert 5736 <= st[0] < 6600


@pytest.mark.parametrize("rows", [0, 1])
def test_schedule_reproduces_the_golden_witness_other_circuits(WS, rows):
    for c in json.load(open(os.path.join(ROOT, "tests", "golden", "rln_other_circuits.json")))["cases"]:
        sub = "tree_depth_%d%s" % (c["depth"], "_multi_max_out_4" if c["multi"] else "")
        digest, st = _run(WS, _graph(sub), c["inputs"], rows)
        assert digest == c["witness_sha256"], c["name"]
        assert st[7] == 0


def test_out_of_range_input_sets_the_error_flag(WS):
    """an input >= r is an error of the graph evaluation (graph.rs:42-45: u256_to_fr fails), not a silent reduction"""
    from oracle.pyref import wtns_graph
    gb = _graph("tree_depth_10")
    g = wtns_graph.parse(gb)
    size = g.inputs_size()
    buf = bytearray(size * 32)
    buf[0] = 1
    off, _ = g.input_mapping["x"]
    buf[off * 32:(off + 1) * 32] = R.to_bytes(32, "little")     # x = r: not canonical
    out = ctypes.create_string_buffer(32 * len(g.signals))
    stats = (ctypes.c_uint32 * 8)()
    assert WS.witsched_run(gb, len(gb), bytes(buf), size, 1, out, stats) == 0
    assert stats[7] == 1


def _run_cone(lib, graph_bytes, named_inputs, rows):
    from oracle.pyref import wtns_graph
    g = wtns_graph.parse(graph_bytes)
    size = g.inputs_size()
    buf = bytearray(size * 32)
    buf[0] = 1
    for name, vals in named_inputs.items():
        off, ln = g.input_mapping[name]
        assert ln == len(vals)
        for k, v in enumerate(vals):
            buf[(off + k) * 32:(off + k + 1) * 32] = (int(v) % R).to_bytes(32, "little")
    out = ctypes.create_string_buffer(32 * len(g.signals))
    stats = (ctypes.c_uint32 * 12)()
    rc = lib.witsched_run_cone(graph_bytes, len(graph_bytes), bytes(buf), size, rows, out, stats)
    assert rc == 0, lib.witsched_error().decode()
    return hashlib.sha256(out.raw).hexdigest(), list(stats)


@pytest.mark.parametrize("rows", [0, 1])
def test_the_unknown_cone_on_top_of_a_partial_run_gives_the_golden_witness(WS, rows):
    """Round 6, finish without re-walking the known cone (witness_sched.h: wl_cone).  The FULL program over the PARTIAL
    witness (message id, x, external nullifier zeroed: what generate_partial_zk_proof evaluates, graph.rs:274-312) leaves
    the stored rows; every row of an unknown node is then overwritten with junk; the CONE program -- the nodes
    evaluate_partial leaves None plus the 13 known ones they read -- runs over the full inputs on top of those rows.  The
    witness read off the rows must be the golden one, on all three shipped circuits; and the cone is a twelfth of the
    graph's depth (the 20-level Merkle chain is known)."""
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))["cases"]
    gb = _graph("tree_depth_20")
    for c in cases[:3]:
        w = c["witness"]
        named = {"identitySecret": [w["identity_secret"]], "userMessageLimit": [w["user_message_limit"]],
                 "messageId": [w["message_id"]], "pathElements": w["path_elements"],
                 "identityPathIndex": w["identity_path_index"], "x": [w["x"]],
                 "externalNullifier": [w["external_nullifier"]]}
        digest, st = _run_cone(WS, gb, named, rows)
        assert digest == c["witness_sha256"], c["name"]
        assert st[7] == 0
    # cone nodes, unknown nodes, recomputed known nodes, known stored rows
    assert st[9] == 1934 and st[10] == 13 and st[8] > st[9] and st[11] > 5000
    full_steps = _run(WS, gb, named, rows)[1][0]
    assert st[0] * 8 < full_steps, (st[0], full_steps)
    for c in json.load(open(os.path.join(ROOT, "tests", "golden", "rln_other_circuits.json")))["cases"]:
        sub = "tree_depth_%d%s" % (c["depth"], "_multi_max_out_4" if c["multi"] else "")
        digest, st = _run_cone(WS, _graph(sub), c["inputs"], rows)
        assert digest == c["witness_sha256"], c["name"]
        assert st[7] == 0


def _rln_hints(named, depth):
    """the values between the hashes of the single-message circuit (protocol/witness.rs:759-828): identity commitment, rate
    commitment, the running hash after Merkle levels 1 .. depth - 1, a1 = Poseidon(secret, external nullifier, message id)"""
    from oracle.pyref.poseidon import poseidon
    secret, limit = int(named["identitySecret"][0]) % R, int(named["userMessageLimit"][0]) % R
    idc = poseidon([secret])
    node = poseidon([idc, limit])
    hints = [idc, node]
    for lvl in range(depth):
        e, b = int(named["pathElements"][lvl]) % R, int(named["identityPathIndex"][lvl])
        node = poseidon([e, node]) if b else poseidon([node, e])
        if lvl < depth - 1:
            hints.append(node)
    hints.append(poseidon([secret, int(named["externalNullifier"][0]) % R, int(named["messageId"][0]) % R]))
    return hints


def _run_segments(lib, graph_bytes, named_inputs, hints, rows):
    from oracle.pyref import wtns_graph
    g = wtns_graph.parse(graph_bytes)
    size = g.inputs_size()
    buf = bytearray(size * 32)
    buf[0] = 1
    for name, vals in named_inputs.items():
        off, ln = g.input_mapping[name]
        for k, v in enumerate(vals):
            buf[(off + k) * 32:(off + k + 1) * 32] = (int(v) % R).to_bytes(32, "little")
    hb = b"".join(int(h).to_bytes(32, "little") for h in hints)
    out = ctypes.create_string_buffer(32 * len(g.signals))
    stats = (ctypes.c_uint32 * 12)()
    rc = lib.witsched_run_segments(graph_bytes, len(graph_bytes), bytes(buf), size, hb, len(hints), rows, out, stats)
    assert rc == 0, lib.witsched_error().decode()
    return hashlib.sha256(out.raw).hexdigest(), list(stats)


@pytest.mark.parametrize("rows", [0, 1])
def test_segments_behind_hints_give_the_golden_witness(WS, rows):
    """Round 6: the graph cut at the values between its 22 chained hashes (witness_sched.h: wl_segments) -- every segment a
    program of its own over rows prefilled with junk, the hints from the Python oracle's Poseidon.  The witness read off the
    rows is the golden one, every cut node's computed value equals its hint, and the longest segment is a twentieth of
    the whole graph's program (one hash deep instead of twenty-two)."""
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))["cases"]
    gb = _graph("tree_depth_20")
    for c in cases[:3]:
        w = c["witness"]
        named = {"identitySecret": [w["identity_secret"]], "userMessageLimit": [w["user_message_limit"]],
                 "messageId": [w["message_id"]], "pathElements": w["path_elements"],
                 "identityPathIndex": w["identity_path_index"], "x": [w["x"]],
                 "externalNullifier": [w["external_nullifier"]]}
        digest, st = _run_segments(WS, gb, named, _rln_hints(named, 20), rows)
        assert digest == c["witness_sha256"], c["name"]
        assert st[7] == 0 and st[11] == 0
    full_steps = _run(WS, gb, named, rows)[1][0]
    assert st[8] >= 22 and st[0] * 12 < full_steps, (st, full_steps)
    # a wrong hint is seen: the cut node's own value differs from it
    bad = _rln_hints(named, 20)
    bad[7] = (bad[7] + 1) % R
    with pytest.raises(AssertionError, match="matches no node"):
        _run_segments(WS, gb, named, bad, rows)
    # the depth-10 circuit
    for c in json.load(open(os.path.join(ROOT, "tests", "golden", "rln_other_circuits.json")))["cases"]:
        if c["multi"]:
            continue
        digest, st = _run_segments(WS, _graph("tree_depth_%d" % c["depth"]), c["inputs"], _rln_hints(c["inputs"], c["depth"]), rows)
        assert digest == c["witness_sha256"] and st[7] == 0 and st[11] == 0
