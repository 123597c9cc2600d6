"""The C99 header is valid C and a plain C program (gcc, host buffers, no HIP headers) drives the library exactly as a
user of the reference's C API would.  Compile checks run on CPU; the run itself needs the GPU."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "c_abi_client.c"
LIBDIR = ROOT / "pi-quant_amd" / "piquant"


def _build(tmp_path):
    exe = tmp_path / "c_abi_client"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", f"-I{ROOT / 'include'}", str(SRC), f"-L{LIBDIR}",
                    "-lpiquant", f"-Wl,-rpath,{LIBDIR}", "-o", str(exe)], check=True)
    return exe


def test_headers_are_valid_c99_and_client_links(tmp_path):
    for header in ("piquant.h", "piquant_hip.h"):
        tu = tmp_path / f"use_{header}.c"
        tu.write_text(f'#include "{header}"\nint main(void) {{ return 0; }}\n')
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", f"-I{ROOT / 'include'}", str(tu)], check=True)
    assert _build(tmp_path).exists()


def _build_cpp(tmp_path):
    exe = tmp_path / "cpp_api_client"
    subprocess.run(["g++", "-std=c++20", "-Wall", "-Wextra", "-O2", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "cpp_api_client.cpp"), f"-L{LIBDIR}",
                    "-lpiquant", f"-Wl,-rpath,{LIBDIR}", "-o", str(exe)], check=True)
    return exe


def test_cpp_header_compiles_and_links(tmp_path):
    """include/piquant.hpp: the reference's C++ spellings (piquant::context, spans, *_generic) as inline forwards to the C ABI."""
    assert _build_cpp(tmp_path).exists()


def _fnv1a(b: bytes) -> int:
    h = 1469598103934665603
    for chunk in np.frombuffer(b, dtype=np.uint8).tolist():
        h = ((h ^ chunk) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.gpu
@pytest.mark.parametrize("host_path", ["stage", "auto"])
@pytest.mark.parametrize("client", ["c99", "cpp20"])
def test_c_client_results_match_oracle(tmp_path, oracle_mod, client, host_path):
    """An unchanged caller of the reference (host buffers, blocking calls): staged through the HIP kernels over PCIe (`stage`) and by the
    library's default (`auto`: the companion libpiquant_cpu.so serves host tensors where they live) -- the same bytes both ways."""
    import os

    O = oracle_mod
    n = 100_003
    exe = _build(tmp_path) if client == "c99" else _build_cpp(tmp_path)
    out = subprocess.run([str(exe), str(n)], check=True, capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, PIQUANT_HIP_HOST_PATH=host_path)).stdout.split()
    s = np.uint32(12345)
    x = np.empty(n, dtype=np.float32)
    with np.errstate(over="ignore"):
        for i in range(n):
            s ^= np.uint32(s << np.uint32(13))
            s ^= np.uint32(s >> np.uint32(17))
            s ^= np.uint32(s << np.uint32(5))
            x[i] = np.float32(int(s) >> 8) * np.float32(2.0 / 16777216.0) - np.float32(1.0)
    s8, z8 = O.compute_quant_params(x, O.F32, O.UINT8)
    s4, z4 = O.compute_quant_params(x, O.F32, O.UINT4)
    assert (np.float32(float(out[0])), int(out[1]), np.float32(float(out[2])), int(out[3])) == (np.float32(s8), z8, np.float32(s4), z4)
    q8 = O.quantize(x, O.F32, O.UINT8, s8, z8)
    q4 = O.quantize(x, O.F32, O.UINT4, s4, z4)
    back = O.dequantize(q8, O.UINT8, O.F32, n, s8, z8, O.ADD, out=np.ones(n, np.float32))
    assert int(out[4], 16) == _fnv1a(q8.tobytes())
    assert int(out[5], 16) == _fnv1a(q4.tobytes())
    assert int(out[6], 16) == _fnv1a(back.tobytes())


def _build_hip_client(tmp_path):
    exe = tmp_path / "hip_client"
    subprocess.run(["g++", "-std=c++20", "-Wall", "-Wextra", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT / 'include'}",
                    str(ROOT / "tests" / "hip_client.cpp"), f"-L{LIBDIR}", "-lpiquant", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{LIBDIR}",
                    "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    return exe


def test_hip_client_compiles_and_links(tmp_path):
    """A C++ program with its own HIP runtime (no Python, no PyTorch) against the GPU-side extras of the library."""
    assert _build_hip_client(tmp_path).exists()


@pytest.mark.gpu
def test_hip_client_device_buffers_match_oracle(tmp_path, oracle_mod):
    """Device buffers on the caller's stream from a torch-free process: fused params + quantize, the batched form, dequantize from
    the device record and dequantize_sum, all checked against the oracle."""
    O = oracle_mod
    n = 1_000_003
    exe = _build_hip_client(tmp_path)
    out = subprocess.run([str(exe), str(n)], check=True, capture_output=True, text=True, timeout=300).stdout.split()
    s = np.uint32(12345)
    x = np.empty(n, dtype=np.float32)
    with np.errstate(over="ignore"):
        for i in range(n):
            s ^= np.uint32(s << np.uint32(13))
            s ^= np.uint32(s >> np.uint32(17))
            s ^= np.uint32(s << np.uint32(5))
            x[i] = np.float32(int(s) >> 8) * np.float32(2.0 / 16777216.0) - np.float32(1.0)
    half = int(out[10])
    assert half == (n // 2) & ~3
    parts = [x, x[:half], x[half:2 * half]]
    params = [O.compute_quant_params(p, O.F32, O.UINT8) for p in parts]
    for i, (sc, zp) in enumerate(params):
        assert (np.float32(float(out[2 * i])), int(out[2 * i + 1])) == (np.float32(sc), zp), i
    qs = [O.quantize(p, O.F32, O.UINT8, sc, zp) for p, (sc, zp) in zip(parts, params)]
    for i in range(3):
        assert int(out[6 + i], 16) == _fnv1a(qs[i].tobytes()), i
    back = O.dequantize(qs[0], O.UINT8, O.F32, n, *params[0])
    acc = back[:half].copy()
    acc = O.dequantize(qs[1], O.UINT8, O.F32, half, params[1][0], params[1][1], O.ADD, out=acc)
    acc = O.dequantize(qs[2], O.UINT8, O.F32, half, params[2][0], params[2][1], O.ADD, out=acc)
    back[:half] = acc
    assert int(out[9], 16) == _fnv1a(back.tobytes())


def _build_hip_client_ext(tmp_path):
    exe = tmp_path / "hip_client_ext"
    subprocess.run(["g++", "-std=c++20", "-Wall", "-Wextra", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", f"-I{ROOT / 'include'}",
                    str(ROOT / "tests" / "hip_client_ext.cpp"), f"-L{LIBDIR}", "-lpiquant", "-L/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{LIBDIR}",
                    "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    return exe


def test_hip_client_ext_compiles_and_links(tmp_path):
    assert _build_hip_client_ext(tmp_path).exists()


@pytest.mark.gpu
def test_hip_client_ext_rest_of_the_extension_surface_matches_oracle(tmp_path, oracle_mod):
    """26 additive exports, 4 of them tested at the C level.  This client drives the rest from a torch-free C++ process:
    minmax_keys / decode / params_from_minmax, compute_quant_params_device + quantize_dp, quantize_dequantize, the stochastic controls,
    the blocking-wait modes, the barrier limit + hand-over counter, reduce_quantize_dynamic, dequantize_dp_batch, reference layout, the flag calls with
    a wait that runs out (piquant_hip_peer_timeout), independent calls."""
    O = oracle_mod
    n = 1_000_003
    exe = _build_hip_client_ext(tmp_path)
    out = subprocess.run([str(exe), str(n)], check=True, capture_output=True, text=True, timeout=300).stdout.split()
    s = np.uint32(12345)
    x = np.empty(n, dtype=np.float32)
    with np.errstate(over="ignore"):
        for i in range(n):
            s ^= np.uint32(s << np.uint32(13))
            s ^= np.uint32(s >> np.uint32(17))
            s ^= np.uint32(s << np.uint32(5))
            x[i] = np.float32(int(s) >> 8) * np.float32(2.0 / 16777216.0) - np.float32(1.0)
    sc, zp = np.float32(1.0) / np.float32(127.0), 128
    s4, z4 = O.compute_quant_params(x, O.F32, O.UINT4)
    assert (np.float32(float(out[0])), int(out[1])) == (np.float32(s4), z4)
    assert int(out[2], 16) == _fnv1a(O.quantize(x, O.F32, O.UINT4, s4, z4).tobytes())
    assert int(out[3], 16) == _fnv1a(O.requantize(x, O.F32, O.UINT8, float(sc), zp).tobytes())
    assert int(out[4], 16) == _fnv1a(O.quantize(x, O.F32, O.UINT8, float(sc), zp, O.STOCHASTIC, 0.25).tobytes())
    assert int(out[5], 16) == _fnv1a(O.quantize_per_element(x, O.F32, O.UINT8, float(sc), zp, 0x1234567890abcdef, 77).tobytes())
    q_near = O.quantize(x, O.F32, O.UINT8, float(sc), zp)
    assert int(out[6], 16) == _fnv1a(q_near.tobytes()) and out[7] == "1"       # four wait modes, same bytes
    s8, z8 = O.compute_quant_params(x, O.F32, O.UINT8)
    q8 = O.quantize(x, O.F32, O.UINT8, s8, z8)
    assert int(out[8], 16) == _fnv1a(q8.tobytes()) and out[9] == "1"           # handing every share over changes nothing but the time
    assert int(out[10]) >= 1                                                   # ... and it happened (deterministic: PIQUANT_HIP_BARRIER_HAND_OVER_ALWAYS)
    total = O.dequantize(q8, O.UINT8, O.F32, n, s8, z8, O.ADD, out=x.copy())
    total = O.dequantize(q8, O.UINT8, O.F32, n, s8, z8, O.ADD, out=total)
    ss, zs = O.compute_quant_params(total, O.F32, O.UINT8)
    qs = O.quantize(total, O.F32, O.UINT8, ss, zs)
    assert int(out[11], 16) == _fnv1a(qs.tobytes())
    assert int(out[12], 16) == _fnv1a(O.dequantize(q8, O.UINT8, O.F32, n, s8, z8).tobytes())
    assert int(out[13], 16) == _fnv1a(O.dequantize(qs, O.UINT8, O.F32, n, ss, zs).tobytes())
    assert int(out[14], 16) == _fnv1a(O.quantize(x, O.F32, O.UINT8, float(sc), zp, form=O.FORM_REFERENCE).tobytes())
    wbuf = np.zeros(n + 32, dtype=np.uint8)
    base = (-wbuf.ctypes.data) % 16                                            # hipMalloc'ed output: aligned, no head
    assert int(out[15], 16) == _fnv1a(O.quantize(x, O.F32, O.UINT8, float(sc), zp, form=O.FORM_REFERENCE, threads=3, out=wbuf[base: base + n]).tobytes())
    assert out[16:20] == ["1", "1", "1", "1"]                                  # fusion off, reseeding, assume_device_pointers, reset_stream
    assert out[20:23] == ["1", "1", "1"]                                       # signal_flags / wait_flags on the stream; peer_timeout reports the late rank; independent calls
    assert out[23] == "1"                                                      # host_path_in_effect
    assert (np.float32(float(out[24])), int(out[25]), np.float32(float(out[26])), int(out[27])) == (np.float32(s8), z8, np.float32(ss), zs)
    assert int(out[28]) == 0 and out[29] == "gfx950"


@pytest.mark.gpu
def test_default_host_path_without_the_companion_library_stages(tmp_path, oracle_mod):
    """PIQUANT_HIP_HOST_PATH_AUTO on an installation that has only libpiquant.so (no libpiquant_cpu.so next to it): host buffers are staged
    through the GPU -- same bytes, nothing aborts -- while an explicit request for the companion still fails loudly."""
    import shutil

    O = oracle_mod
    alone = tmp_path / "alone"
    alone.mkdir()
    shutil.copy(LIBDIR / "libpiquant.so", alone / "libpiquant.so")
    code = f"""
import ctypes, sys
import numpy as np
import torch                      # one HIP runtime per process: the one PyTorch bundles
L = ctypes.CDLL({str(alone / "libpiquant.so")!r})
L.piquant_context_create.restype = ctypes.c_void_p
L.piquant_context_create.argtypes = [ctypes.c_size_t]
L.piquant_hip_host_path_in_effect.argtypes = [ctypes.c_void_p]
L.piquant_hip_set_host_path.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.piquant_quantize.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_float, ctypes.c_int64, ctypes.c_int]
ctx = L.piquant_context_create(0)
print("in_effect", L.piquant_hip_host_path_in_effect(ctx))
x = np.random.default_rng(3).uniform(-1, 1, 200_003).astype(np.float32)
q = np.zeros(x.size, dtype=np.uint8)
L.piquant_quantize(ctx, x.ctypes.data, 0, q.ctypes.data, 4, x.size, 0.0078431377, 127, 0)
print("sum", int(q.astype(np.int64).sum()), "first", int(q[0]))
sys.stdout.flush()
if len(sys.argv) > 1:
    L.piquant_hip_set_host_path(ctx, 1)      # asks for the companion: must abort with a message
    print("not reached")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    x = np.random.default_rng(3).uniform(-1, 1, 200_003).astype(np.float32)
    want = O.quantize(x, O.F32, O.UINT8, 0.0078431377, 127)
    assert "in_effect 0" in r.stdout and f"sum {int(want.astype(np.int64).sum())} first {int(want[0])}" in r.stdout, r.stdout
    r = subprocess.run([sys.executable, "-c", code, "insist"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "not reached" not in r.stdout and "libpiquant_cpu.so" in r.stderr, r.stderr[-1000:]
