"""CPU: the C-ABI library loads, exports every symbol the headers declare, and its host-only helpers agree
with the oracle.  No kernel is launched here (no GPU on this box)."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "pi-quant_amd" / "piquant" / "libpiquant.so"


def declared_symbols():
    names = []
    for header in ("piquant.h", "piquant_hip.h"):
        text = (ROOT / "include" / header).read_text()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"PIQUANT_EXPORT\s+[\w\s\*]+?\b(piquant_\w+)\s*\(", text)
    return names


def test_library_is_built():
    assert LIB.exists(), "run `make -C pi-quant_amd/csrc` (or __graft_entry__.build())"


def test_every_declared_symbol_is_exported():
    names = declared_symbols()
    # the six reference entry points (reference include/piquant.h:42-85) must be among them
    for ref_name in ("piquant_context_create", "piquant_context_destroy", "piquant_quantize", "piquant_dequantize",
                     "piquant_compute_quant_params_float32", "piquant_compute_quant_params_bfloat16"):
        assert ref_name in names
    lib = ctypes.CDLL(str(LIB))
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert len(names) >= 16


def test_cpu_companion_exports_its_header():
    """include/piquant_cpu.h <-> pi-quant_amd/piquant/libpiquant_cpu.so; and libpiquant.so does NOT link it (it is dlopen-ed on request only)."""
    import subprocess

    text = re.sub(r"/\*.*?\*/", "", (ROOT / "include" / "piquant_cpu.h").read_text(), flags=re.S)
    names = re.findall(r"PIQUANT_CPU_EXPORT\s+[\w\s\*]+?\b(piquant_cpu_\w+)\s*\(", text)
    assert len(names) >= 11
    cpu_lib = ctypes.CDLL(str(LIB.with_name("libpiquant_cpu.so")))
    for name in names:
        assert hasattr(cpu_lib, name), f"{name} declared in include/piquant_cpu.h but not exported"
    needed = subprocess.run(["readelf", "-d", str(LIB)], capture_output=True, text=True).stdout
    assert "libpiquant_cpu" not in needed


def test_enum_values_match_reference_abi():
    text = (ROOT / "include" / "piquant.h").read_text()
    for name, value in (("PIQUANT_NEAREST", 0), ("PIQUANT_STOCHASTIC", 1), ("PIQUANT_REDUCE_OP_SET", 0), ("PIQUANT_REDUCE_OP_ADD", 1),
                        ("PIQUANT_DTYPE_F32", 0), ("PIQUANT_DTYPE_BF16", 1), ("PIQUANT_DTYPE_UINT2", 2), ("PIQUANT_DTYPE_UINT4", 3),
                        ("PIQUANT_DTYPE_UINT8", 4)):
        assert re.search(rf"\b{name}\s*=\s*{value}\b", text), name


def test_python_surface_mirrors_reference():
    import piquant
    import piquant.torch as pt

    assert [m.name for m in piquant.RoundMode] == ["NEAREST", "STOCHASTIC"]
    assert [m.name for m in piquant.ReduceOp] == ["SET", "ADD"]
    assert [(m.name, m.value) for m in piquant.DataType] == [("F32", 0), ("BF16", 1), ("UINT2", 2), ("UINT4", 3), ("UINT8", 4)]
    assert piquant.DataType.UINT4.bit_size == 4 and piquant.DataType.UINT4.is_quantized and piquant.DataType.BF16.is_dequantized
    for fn in ("quantize_ptr", "dequantize_ptr", "compute_quant_params_ptr_float32", "compute_quant_params_ptr_bfloat16", "get"):
        assert hasattr(piquant.Context, fn)
    for fn in ("torch_to_piquant_dtype", "piquant_to_torch_dtype", "compute_quant_params", "quantize", "dequantize"):
        assert callable(getattr(pt, fn))
    import torch

    assert pt.torch_to_piquant_dtype(torch.quint4x2) is piquant.DataType.UINT4
    assert pt.piquant_to_torch_dtype(piquant.DataType.UINT2) is torch.quint2x4
    with pytest.raises(ValueError):
        pt.torch_to_piquant_dtype(torch.int32)
    assert piquant.DataType.UINT2.packed_nbytes(9) == 3 and piquant.DataType.UINT4.packed_nbytes(9) == 5


def test_host_epilogue_matches_oracle(oracle_mod):
    """piquant_hip_quant_params_from_minmax (host, double math) == oracle restatement of piquant.cpp:245-258."""
    import piquant

    O = oracle_mod
    rng = np.random.default_rng(3)
    cases = [(-1.0, 3.0), (42.0, 42.0), (0.0, 1.0), (-1.0, 1.0), (2.0, 6.0), (-6.0, -2.0), (-0.5, 1.5), (0.0, 0.0), (-1e-30, 1e-30), (-3e38, 3e38)]
    cases += [tuple(sorted(rng.normal(size=2).astype(np.float32).tolist())) for _ in range(200)]
    for lo, hi in cases:
        for dt, odt in ((piquant.DataType.UINT8, O.UINT8), (piquant.DataType.UINT4, O.UINT4), (piquant.DataType.UINT2, O.UINT2)):
            got = piquant.quant_params_from_minmax(lo, hi, dt)
            want = O.quant_params_from_minmax(float(np.float32(lo)), float(np.float32(hi)), odt)
            assert got == want, (lo, hi, dt, got, want)


def test_minmax_key_codec_is_order_preserving():
    import piquant

    lib = ctypes.CDLL(str(LIB))
    vals = np.array([-np.inf, -3.4e38, -1.0, -1e-45, -0.0, 0.0, 1e-45, 1.0, 3.4e38, np.inf], dtype=np.float32)

    def key(f):   # same mapping as csrc/device_math.hpp float_to_key
        b = int(np.float32(f).view(np.int32))
        return b if b >= 0 else b ^ 0x7FFFFFFF

    keys = [key(v) for v in vals]
    assert keys == sorted(keys)
    for v in vals:
        lo, hi = piquant.decode_minmax_keys(key(v), key(-v))
        assert np.float32(lo).tobytes() == np.float32(v).tobytes() or (v == 0 and lo == 0)
        assert hi == v
    assert lib.piquant_hip_version is not None


def test_no_gpu_is_a_loud_python_error_not_a_fallback():
    """On a box without a GPU the package imports (the library loads) but creating a context or quantizing raises --
    there is no CPU compute path to fall back to."""
    import torch

    import piquant

    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    with pytest.raises(RuntimeError, match="no CPU path"):
        piquant.Context()
    with pytest.raises(RuntimeError, match="no CPU path"):
        piquant.torch.quantize(torch.zeros(16), scale=1.0, zero_point=0, dtype=torch.quint8)


def test_native_torch_front_end_loads_and_checks_arguments():
    """piquant/_piquant_torch.so (csrc/torch_binding.cpp, built by __graft_entry__.build()) imports next to libpiquant.so and
    converts its arguments; without a GPU the call itself must be refused, not attempted."""
    import torch

    import piquant.torch as pt

    if pt._native is None:
        pytest.skip("native front end not built (python pi-quant_amd/csrc/build_torch_binding.py)")
    with pytest.raises(RuntimeError, match="device tensors"):
        pt._native.quantize(0, torch.zeros(8), 1.0, 0, torch.quint8, 0, None)
    with pytest.raises(RuntimeError, match="device tensors"):
        pt._native.dequantize(0, torch.zeros(8, dtype=torch.uint8), 1.0, 0, torch.float32, 0, None)
    assert set(dir(pt._native)) >= {"quantize", "dequantize"}
