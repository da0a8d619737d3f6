import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mh():
    """The package (its directory name has a hyphen, hence importlib)."""
    return importlib.import_module("multi-h_amd")


@pytest.fixture(scope="session")
def synth(mh):
    return mh.synth


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def engine_lib(mh):
    """Builds the engine if needed (hipcc cross-compiles without a GPU) and dlopens it."""
    if not os.path.exists(mh.LIB_PATH):
        build = importlib.import_module("multi-h_amd.build")
        build.build_all()
    return mh.load_library()


@pytest.fixture()
def engine(mh, engine_lib):
    """A live engine on cuda:0.  Only for @pytest.mark.gpu tests: fails loudly without a GPU."""
    e = mh.Engine(device=0, thr_fund_mat=2.6, thr_hom=2.2, locality=0.005, lam=0.5, min_inliers=20)
    yield e
    e.close()
