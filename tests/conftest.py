import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLDEN, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle binding (test infrastructure)."""
    from oracle import oracle_py

    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def data_dir():
    return DATA


@pytest.fixture(scope="session")
def config4_gfa(tmp_path_factory):
    """BASELINE config #4 graph ("all HLA-zoo loci merged"): disjoint union of the HLA-zoo loci after readsim.toposort_gfa (the
    stand-in for `odgi sort`); 19 of the 20 -- readsim.HLA_CONFIG4 names the one left out and why"""
    import __graft_entry__ as ge

    out = str(tmp_path_factory.mktemp("cfg4") / "hla19.gfa")
    assert ge.load_package().readsim.config4_graph(DATA, out) == (23980, 33177, 282231)
    return out


@pytest.fixture(scope="session")
def config5_small_gfa(tmp_path_factory):
    """BASELINE config #5 generator at 60 kbp (the oracle's size); the full 1 Mbp graph is built in test_gpu_fullsize"""
    import __graft_entry__ as ge

    out = str(tmp_path_factory.mktemp("cfg5") / "syn60k.gfa")
    ge.load_package().readsim.synth_pangenome(out, 60000, seed=78)
    return out
