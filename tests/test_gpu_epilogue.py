"""GPU: the DEVICE epilogue (pq::quant_params_epilogue: what the scan's finishing block and the fused kernel run) on more than 10^6
random and adversarial (min, max) pairs -- bit for bit the product's host epilogue (which tests/test_epilogue_exact.py pins to the
exact-rational model of reference src/piquant.cpp:245-258 on the same pairs), plus the model itself on a sample.  VERDICT r01 item 6."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

from epilogue_cases import exact_epilogue, pairs

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
BITS = {8: 4, 4: 3, 2: 2}


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    so = tmp_path_factory.mktemp("probe") / "libepilogue_probe.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", f"-I{ROOT / 'pi-quant_amd' / 'csrc'}",
                    str(ROOT / "tests" / "epilogue_probe.hip"), "-o", str(so)], check=True)
    import piquant  # noqa: F401  (torch + libpiquant first: one HIP runtime)

    lib = ctypes.CDLL(str(so))
    lib.epilogue_probe.restype = ctypes.c_int
    lib.epilogue_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def test_device_epilogue_equals_host_epilogue_and_exact_model(probe):
    from piquant._bootstrap import C_LIB

    lo, hi = pairs(seed=2026, n_random=1_100_000)
    n = lo.size
    assert n >= 1_000_000
    dlo, dhi = torch.from_numpy(lo).cuda(), torch.from_numpy(hi).cuda()
    host = C_LIB.piquant_hip_quant_params_from_minmax
    s, z = ctypes.c_float(), ctypes.c_int64()
    rng = np.random.default_rng(1)
    for bits, code in BITS.items():
        dscale = torch.empty(n, dtype=torch.float32, device="cuda")
        dzp = torch.empty(n, dtype=torch.int64, device="cuda")
        assert probe.epilogue_probe(dlo.data_ptr(), dhi.data_ptr(), n, bits, dscale.data_ptr(), dzp.data_ptr(), None) == 0
        torch.cuda.synchronize()
        gs, gz = dscale.cpu().numpy(), dzp.cpu().numpy()
        hs = np.empty(n, dtype=np.float32)
        hz = np.empty(n, dtype=np.int64)
        for i in range(n):
            host(float(lo[i]), float(hi[i]), code, ctypes.byref(s), ctypes.byref(z))
            hs[i], hz[i] = s.value, z.value
        bad = np.flatnonzero((gs.view(np.uint32) != hs.view(np.uint32)) | (gz != hz))
        assert bad.size == 0, (bits, bad.size, [(float(lo[i]), float(hi[i]), float(gs[i]), int(gz[i]), float(hs[i]), int(hz[i])) for i in bad[:5]])
        # and straight against the exact model: the adversarial tail of the list plus a random sample
        for i in np.r_[rng.integers(0, n, 20000), n - 30000:n]:
            ms, mz = exact_epilogue(float(lo[i]), float(hi[i]), bits)
            assert (np.float32(ms).view(np.uint32), mz) == (gs[i].view(np.uint32), int(gz[i])), (bits, float(lo[i]), float(hi[i]), float(gs[i]), int(gz[i]), ms, mz)
