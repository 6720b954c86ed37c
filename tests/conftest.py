import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


@pytest.fixture(scope="session")
def circuit20():
    """(zkey, graph) of the shipped depth-20 circuit, parsed by the Python oracle."""
    from oracle.pyref import rln
    return rln.load_circuit(20)


_CONFIG2_ORACLE = {}


def oracle_config2(first, n):
    """oracle/c's proofs and public values of config-2 witnesses [first, first + n) -> (ws, rs, [proof128], [public inputs]);
    computed once per session (two -m gpu files judge the same 1 024 witnesses: 1 024 CPU proofs each time before)"""
    key = (first, n)
    if key not in _CONFIG2_ORACLE:
        from oracle.c import binding as ob
        from zerokit_amd import workload
        ws, rs = workload.config2_range(first, n)
        _, proofs, pub = ob.Circuit(20).prove_many(ws, rs)
        _CONFIG2_ORACLE[key] = (ws, rs, proofs, pub)
    return _CONFIG2_ORACLE[key]
