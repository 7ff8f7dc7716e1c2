import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def testtext1():
    with open(os.path.join(GOLDEN, "testtext1.txt"), encoding="utf-8", newline="") as f:
        return f.read()


@pytest.fixture(scope="session")
def testheaders1():
    with open(os.path.join(GOLDEN, "testheaders1.txt"), encoding="utf-8", newline="") as f:
        return f.read()


@pytest.fixture(scope="session")
def known():
    with open(os.path.join(GOLDEN, "known_answers.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.lib()
    oracle.install_bmp_tables()
    return oracle
