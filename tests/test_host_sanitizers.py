"""AddressSanitizer + UBSan build of the product's host code that parses bytes it does not control (SURVEY section 5:
sanitizers belong to the CPU suite; the GPU pool refuses them).  tests/host/sanitize_main.cpp compiles zkey.cpp (arkzkey
and graph parsers, point decompression), pairing.h (the verifier), tree_config.h (the config_path JSON) and
witness_sched.cpp (the interpreter's scheduler) with g++ -fsanitize=address,undefined and runs them on the shipped
resources, on a golden proof and on ~2 500 truncated / mutated inputs: every malformed input must end in an error,
and the sanitizers must stay silent.  Round 6: ffi_wire.h -- the (de)serialisers of the C ABI (rln proofs, proof values,
witnesses, partial witnesses, partial proofs; V1 LE / BE and the V3 forms) -- on golden records and > 10 000 truncations,
bit flips, hostile length prefixes and trailing bytes; every refusal must carry one of the reference's error texts."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zerokit_amd", "csrc")


def test_host_parsers_verifier_and_scheduler_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_main")
    srcs = [os.path.join(ROOT, "tests", "host", "sanitize_main.cpp"), os.path.join(CSRC, "zkey.cpp"),
            os.path.join(CSRC, "witness_sched.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
                           "-I", "/opt/rocm/include", "-I", CSRC] + srcs + ["-o", exe])
    vec = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_vectors.json")))
    case = vec["cases"][0]
    blob = bytes.fromhex(case["proof_compressed"]) + b"".join(int(x).to_bytes(32, "little") for x in case["public_inputs"])
    pf = tmp_path / "proof.bin"
    pf.write_bytes(blob)
    # round 6: the wire parsers (ffi_wire.h) on the pyref-generated V1 LE record of the same case and on the partial points
    # of tests/golden/rln_h20_partial.json
    rec = tmp_path / "rln_proof_le.bin"
    rec.write_bytes(bytes.fromhex(case["rln_proof_le"]))
    part = json.load(open(os.path.join(ROOT, "tests", "golden", "rln_h20_partial.json")))["cases"][0]
    pp = tmp_path / "partial320.bin"
    pp.write_bytes(bytes.fromhex(part["partial320"]))
    res = os.path.join(ROOT, "zerokit_amd", "resources", "tree_depth_20")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, os.path.join(res, "rln_final.arkzkey"), os.path.join(res, "graph.bin"), str(pf), str(rec), str(pp)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr
    assert "0 failures" in r.stdout
    import re
    m = re.search(r"wire parsers: (\d+) inputs, (\d+) parsed, (\d+) refused with a reference error text, 0 failures", r.stdout)
    assert m and int(m.group(1)) >= 10000 and int(m.group(3)) > 5000, r.stdout[-600:]


def test_gather_queue_under_thread_sanitizer(tmp_path):
    """zerokit_amd/csrc/gather.h -- the queue behind ffi_generate_rln_proof / ffi_finish_rln_proof that gathers the calls
    of several threads into batches -- under ThreadSanitizer (tests/host/gather_tsan.cpp): 1 ... 24 threads calling in a
    loop, with and without the leader's wait for recent callers; every call gets its own result, one batch at a time, no
    batch above the cap, a run that throws marks its whole batch, a lone caller never waits; no data race reported."""
    exe = str(tmp_path / "gather_tsan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-I", CSRC,
                           os.path.join(ROOT, "tests", "host", "gather_tsan.cpp"), "-o", exe, "-lpthread"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ThreadSanitizer" not in r.stderr and "0 failures" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
