import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the suite (the driver runs `pytest -x`: the first failure hides everything behind it, so what pins the contract comes
# first and what exercises opt-in schedules in several processes comes last): kernels against their fp32/fp64 references ->
# the path against the reference's golden vectors and the oracle -> fine-tuned-model modes -> training step / optimizer ->
# the full-size property tests -> boundary (entry point, checkpoints, launcher) -> multi-process schedules.  Within a file the
# order of definition is kept.
_ORDER = ["test_abi_cpu", "test_oracle_golden", "test_ft_oracle_golden", "test_bench_launcher_cpu", "test_pool_cpu", "test_dp_gloo",
          "test_kernels_gpu", "test_parity_gpu", "test_ft_gpu", "test_train_gpu", "test_fullsize_gpu", "test_fullsize_large_gpu",
          "test_boundary_gpu", "test_dp_gpu"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(name) if name in _ORDER else len(_ORDER)
    items.sort(key=rank)                      # stable: definition order inside a file survives


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
