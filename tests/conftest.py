"""Test configuration.

Markers: ``gpu`` -- needs a real MI355X (run with ``-m gpu`` on the GPU box); everything else runs on CPU.
``oracle`` (tests only!) is the CPU checker; ``piquant`` is the product package under ``pi-quant_amd/``.
"""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "pi-quant_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); deselected by -m 'not gpu'")
    config.addinivalue_line("markers", "slow: takes minutes (eight bench.py ranks sharing the GPU); still part of -m gpu")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle

    oracle.build(with_ref=None)   # compiles liboracle.so; _ref only where /root/reference exists
    return oracle


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# Order of GPU test files under `-x`: what pins results against the oracle first (BASELINE configs, goldens and the dtype matrix, the reference's
# own suites, the epilogue, the C clients), then the Python surface and the sharded paths, and the scheduling / property tests (grid-barrier
# hand-overs, processes sharing the GPU) last.  Round 4's driver run stopped at a scheduling-dependent counter assertion in a file that
# sorted in front of the whole parity suite; nothing of that kind may stand in front of an oracle comparison again.
_GPU_FILE_ORDER = ["test_gpu_00_baseline_configs.py", "test_gpu_01_default_is_reference_exact.py", "test_gpu_parity.py", "test_gpu_reference_suites.py", "test_gpu_epilogue.py", "test_c_client.py",
                   "test_gpu_torch_api.py", "test_gpu_distributed.py", "test_gpu_barrier.py"]


def _file_rank(item):
    name = Path(str(item.fspath)).name
    return _GPU_FILE_ORDER.index(name) if name in _GPU_FILE_ORDER else len(_GPU_FILE_ORDER) - 1   # unknown files: in front of the barrier tests


def pytest_collection_modifyitems(config, items):
    items.sort(key=_file_rank)   # stable: the order inside a file is kept
    # Fail loudly rather than silently skip: a gpu test collected on a box without a GPU is an error.
    if os.environ.get("PIQUANT_ALLOW_GPU_SKIP") == "1":
        import torch

        if not torch.cuda.is_available():
            skip = pytest.mark.skip(reason="no GPU and PIQUANT_ALLOW_GPU_SKIP=1")
            for item in items:
                if "gpu" in item.keywords:
                    item.add_marker(skip)
