import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def wa():
    import wrfchem_arc_amd
    return wrfchem_arc_amd


@pytest.fixture(scope="session")
def sw_table_file(tmp_path_factory, wa):
    """Synthetic RRTMG_SW_DATA (fp64 records), seed 20260001 -- the seed the golden fixtures were made with."""
    p = tmp_path_factory.mktemp("tables") / "RRTMG_SW_DATA"
    wa.synth.write_sw_data(str(p), seed=20260001, real_bytes=8)
    return str(p)


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def sw_oracle(oracle_mod, sw_table_file):
    return oracle_mod.SwOracle(sw_table_file)


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if needed) and load the product library."""
    import __graft_entry__ as ge
    ge.build_product()
    from importlib import import_module
    eng = import_module("wrfchem-arc-interactions_amd.engine")
    eng.load_library()
    return eng
