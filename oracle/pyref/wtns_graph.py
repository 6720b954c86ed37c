"""ORACLE (test infrastructure): circom-witnesscalc graph reader + interpreter.

Follows /root/reference/rln/src/circuit/iden3calc/storage.rs:265-302 (file format: magic
`wtns.graph.001`, u64 node count, varint-delimited protobuf nodes, metadata), iden3calc/proto.rs:7-117
(message schema), iden3calc/graph.rs:72-143,180-224,246-272,314-466 (op semantics and evaluation) and
iden3calc.rs:20-60,106-181 (input buffer: slot 0 = 1, named signals at their offsets).
The graph carries its own oracle: w[1..6] must equal [y, root, nullifier, x, ext] from the Poseidon
formulae of protocol/witness.rs:759-828 (tests/test_oracle_kats.py).
"""
from .bn254 import R

MAGIC = b"wtns.graph.001"

# proto.rs:88-110
DUO = ["Mul", "Div", "Add", "Sub", "Pow", "Idiv", "Mod", "Eq", "Neq", "Lt", "Gt", "Leq", "Geq",
       "Land", "Lor", "Shl", "Shr", "Bor", "Band", "Bxor"]
UNO = ["Neg", "Id"]
TRES = ["TernCond"]
HALF_M = 10944121435919637611123202872628637544274182200208017171849102093287904247808  # graph.rs:410-411


def _varint(b, o):
    v = 0
    s = 0
    while True:
        c = b[o]
        o += 1
        v |= (c & 0x7F) << s
        if not c & 0x80:
            return v, o
        s += 7


def _fields(b):
    """decode one protobuf message into {tag: [values]} (varint or length-delimited only)."""
    out = {}
    o = 0
    while o < len(b):
        key, o = _varint(b, o)
        tag, wt = key >> 3, key & 7
        if wt == 0:
            v, o = _varint(b, o)
        elif wt == 2:
            ln, o = _varint(b, o)
            v = b[o:o + ln]
            o += ln
        else:
            raise ValueError("unsupported wire type %d" % wt)
        out.setdefault(tag, []).append(v)
    return out


class Graph:
    def __init__(self, nodes, signals, input_mapping):
        self.nodes = nodes              # tuples: ("Input", i) | ("Const", v) | ("Uno", op, a) | ("Duo", op, a, b) | ("Tres", op, a, b, c)
        self.signals = signals          # witness_signals
        self.input_mapping = input_mapping  # name -> (offset, len)
        self.tree_depth = input_mapping.get("pathElements", (0, 0))[1]       # circuit/mod.rs:163-179
        self.max_out = input_mapping["messageId"][1] if "messageId" in input_mapping else 1

    def inputs_size(self):
        """iden3calc.rs:106-120"""
        start, mx = False, 0
        for n in self.nodes:
            if n[0] == "Input":
                mx = max(mx, n[1])
                start = True
            elif start:
                break
        return mx + 1


def parse(data: bytes) -> Graph:
    if not data:
        raise ValueError("empty graph")
    if data[:len(MAGIC)] != MAGIC:
        raise ValueError("Invalid magic")
    o = len(MAGIC)
    n = int.from_bytes(data[o:o + 8], "little")
    o += 8
    nodes = []
    for _ in range(n):
        ln, o = _varint(data, o)
        msg = _fields(data[o:o + ln])
        o += ln
        if 1 in msg:
            f = _fields(msg[1][0])
            nodes.append(("Input", f.get(1, [0])[0]))
        elif 2 in msg:
            f = _fields(msg[2][0])
            big = _fields(f[1][0]) if 1 in f else {}
            v = int.from_bytes(big.get(1, [b""])[0], "little") % R   # from_le_bytes_mod_order, storage.rs:45-47
            nodes.append(("Const", v))
        elif 3 in msg:
            f = _fields(msg[3][0])
            nodes.append(("Uno", UNO[f.get(1, [0])[0]], f.get(2, [0])[0]))
        elif 4 in msg:
            f = _fields(msg[4][0])
            nodes.append(("Duo", DUO[f.get(1, [0])[0]], f.get(2, [0])[0], f.get(3, [0])[0]))
        elif 5 in msg:
            f = _fields(msg[5][0])
            nodes.append(("Tres", TRES[f.get(1, [0])[0]], f.get(2, [0])[0], f.get(3, [0])[0], f.get(4, [0])[0]))
        else:
            raise ValueError("Proto::Node must have a node field")
    ln, o = _varint(data, o)
    md = _fields(data[o:o + ln])
    signals = []
    for v in md.get(1, []):
        if isinstance(v, (bytes, bytearray)):  # packed repeated uint32
            p = 0
            while p < len(v):
                x, p = _varint(v, p)
                signals.append(x)
        else:
            signals.append(v)
    mapping = {}
    for ent in md.get(2, []):
        e = _fields(ent)
        name = e[1][0].decode()
        d = _fields(e[2][0]) if 2 in e else {}
        mapping[name] = (d.get(1, [0])[0], d.get(2, [0])[0])
    return Graph(nodes, signals, mapping)


def _signed_cmp(a, b, op):
    """graph.rs:413-466: values above M/2 are negative"""
    an, bn = a > HALF_M, b > HALF_M
    if an == bn:
        return int(op(a, b))
    if op.__name__ in ("ge", "gt"):
        return 0 if an else 1
    return 1 if an else 0


def eval_duo(op, a, b):
    import operator as _o
    if op == "Mul":
        return a * b % R
    if op == "Add":
        return (a + b) % R
    if op == "Sub":
        return (a - b) % R
    if op == "Div":
        return 0 if b == 0 else a * pow(b, -1, R) % R
    if op == "Pow":
        return pow(a, b, R)
    if op == "Idiv":
        return 0 if b == 0 else a // b
    if op == "Mod":
        return 0 if b == 0 else a % b
    if op == "Eq":
        return int(a == b)
    if op == "Neq":
        return int(a != b)
    if op == "Lt":
        return _signed_cmp(a, b, _o.lt)
    if op == "Gt":
        return _signed_cmp(a, b, _o.gt)
    if op == "Leq":
        return _signed_cmp(a, b, _o.le)
    if op == "Geq":
        return _signed_cmp(a, b, _o.ge)
    if op == "Land":
        return int(a != 0 and b != 0)
    if op == "Lor":
        return int(a != 0 or b != 0)
    if op == "Shl":  # graph.rs:314-326
        if b == 0:
            return a
        if b >= 254:
            return 0
        v = (a << b) & ((1 << 256) - 1)
        if v >= R:
            raise ValueError("Failed to compute left shift")
        return v
    if op == "Shr":  # graph.rs:328-363
        if b == 0:
            return a
        if b >= 254:
            return 0
        return a >> (b & 0xFF)
    if op in ("Bor", "Band", "Bxor"):  # graph.rs:365-408 (`d > MODULUS` then one subtraction)
        d = {"Bor": a | b, "Band": a & b, "Bxor": a ^ b}[op]
        if d > R:
            d -= R
        if d >= R:
            raise ValueError("Failed to compute bitwise op")
        return d
    raise ValueError(op)


def evaluate(g: Graph, inputs):
    """graph.rs:246-272"""
    vals = []
    for n in g.nodes:
        k = n[0]
        if k == "Const":
            v = n[1]
        elif k == "Input":
            v = inputs[n[1]]
            if v >= R:
                raise ValueError("Failed to convert U256 to Fr")
        elif k == "Duo":
            v = eval_duo(n[1], vals[n[2]], vals[n[3]])
        elif k == "Uno":
            if n[1] != "Neg":
                raise ValueError("uno operator Id not implemented for Montgomery")
            v = (-vals[n[2]]) % R
        else:
            v = vals[n[3]] if vals[n[2]] != 0 else vals[n[4]]
        vals.append(v)
    return [vals[i] for i in g.signals]


def calc_witness(g: Graph, named_inputs: dict):
    """iden3calc.rs:20-60: named_inputs maps signal name -> list of ints."""
    buf = [0] * g.inputs_size()
    buf[0] = 1
    for name, vs in named_inputs.items():
        if name not in g.input_mapping:
            raise KeyError("missing input " + name)
        off, ln = g.input_mapping[name]
        if ln != len(vs):
            raise ValueError("invalid input length for %s: expected %d got %d" % (name, ln, len(vs)))
        for i, v in enumerate(vs):
            buf[off + i] = v
    return evaluate(g, buf)
