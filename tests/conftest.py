import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """build the product library, the oracle and the host harness once per session (cheap if up to date)"""
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session")
def kats():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "poseidon_kats.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def derived():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "derived_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def jubjub_kat():
    """value-level known answer for the Jubjub group law from the upstream curve crate's own test (see the file's `source`)"""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "jubjub_upstream_kat.json")) as f:
        d = json.load(f)
    return {"f1": int(d["f1"]), "f2": int(d["f2"]), "g": tuple(int(v) for v in d["g"]), "f1f2g": tuple(int(v) for v in d["f1f2g"])}
