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


def pytest_collection_modifyitems(config, items):
    # Fail loudly rather than silently skip: a gpu test collected on a box without a GPU is an error.
    if os.environ.get("PIQUANT_ALLOW_GPU_SKIP") == "1":
        import torch

        if not torch.cuda.is_available():
            skip = pytest.mark.skip(reason="no GPU and PIQUANT_ALLOW_GPU_SKIP=1")
            for item in items:
                if "gpu" in item.keywords:
                    item.add_marker(skip)
