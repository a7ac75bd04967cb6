"""pytest configuration: import paths, the `gpu` marker, shared fixtures.

`-m "not gpu"` tests run on CPU (oracle vs goldens, host logic, ABI symbols);
`-m gpu` tests are the parity tests proper and call through the C-ABI on cuda:0.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "bpl-next_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import dc_oracle

    return dc_oracle


@pytest.fixture
def dummy_data(oracle):
    """The reference's tests/conftest.py:7-29 fixture (BASELINE config 1)."""
    return oracle.dummy_data_recipe()


@pytest.fixture
def timed_dummy_data(oracle):
    """The reference's tests/conftest.py:32-62 fixture."""
    return oracle.timed_dummy_data_recipe()


@pytest.fixture(scope="session")
def hip_ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from bpl._ffi import HipContext

    ctx = HipContext(0)
    yield ctx
    ctx.close()
