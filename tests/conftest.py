import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle (oracle/liboracle.so), built on demand"""
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(autouse=True)
def _oracle_switches_do_not_leak():
    """or_set_wide_degree is a process-wide switch of the checker (oracle/gcn_oracle.c): whatever a test did with it — and
    however it ended — the next test starts from the reference's int product."""
    yield
    mod = sys.modules.get("oracle.pyoracle")
    lib = getattr(mod, "_LIB", None) if mod else None
    if lib is not None:
        lib.or_set_wide_degree(0)


@pytest.fixture(scope="session")
def ref():
    """the reference's own objects (oracle/_ref/libref.so): built in the build container, used wherever the built file is present"""
    from oracle.pyoracle import Ref, build
    if not Ref.available() and os.path.isdir("/root/reference/src"):
        build()
    if not Ref.available():
        pytest.skip("oracle/_ref not built (reference tree absent on this box)")
    return Ref()


@pytest.fixture(scope="session")
def experiments():
    """the measured-slower variants (packed dH1 rows, in-launch segment sum, backward pipeline, W in LDS ...) are compiled
    only by `make EXPERIMENTS=1` (include/gcnhip.h, gcnhip_experiments); their bit-identity tests skip on the default build"""
    from cuda_gcn_amd import _lib
    if not _lib.gcnhip().gcnhip_experiments():
        pytest.skip("libgcnhip.so built without GCNHIP_EXPERIMENTS (make EXPERIMENTS=1)")
    return True
