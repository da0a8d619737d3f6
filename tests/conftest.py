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


class _ParityScenes:
    """multi-h_amd.synth for the PARITY suite: make_scene defaults to the generator as it stood until round 4
    (legacy_r04=True) — planes drawn independently of each other, 40-90 % of a plane's correspondences within the
    truncation threshold of another plane's homography (tools/plane_trace.py).  Useless as ground truth for result
    quality, but exactly what the solver tests were written around: ambiguous data terms make the alpha-expansion's hard
    instances (several cycles, cores of thousands of sites, arcs-in-memory rows), and every parity assertion compares the
    engine with the oracle on the SAME scene.  Tests of result quality (planes recovered, ARI) take the current
    generator explicitly: mh.synth.make_scene(...)."""

    def __init__(self, synth):
        self._synth = synth

    def make_scene(self, *a, **kw):
        kw.setdefault("legacy_r04", True)
        return self._synth.make_scene(*a, **kw)

    def __getattr__(self, name):
        return getattr(self._synth, name)


@pytest.fixture(scope="session")
def synth(mh):
    return _ParityScenes(mh.synth)


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
