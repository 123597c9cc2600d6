"""TEST INFRASTRUCTURE ONLY.

`oracle` is the CPU checker for the HIP path: a plain-C restatement of the reference algorithm
(`piquant_oracle.c`, loaded through ctypes) plus, where it has been built, the reference's own kernel
translation units compiled from /root/reference (`_ref/libpiquant_ref.so`).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  The
product (pi-quant_amd/) never does; it fails loudly when its HIP library is missing.
"""
from .oracle import (  # noqa: F401
    ADD,
    BF16,
    F32,
    FORM_REFERENCE,
    FORM_UNIFORM,
    NEAREST,
    SET,
    STOCHASTIC,
    UINT2,
    UINT4,
    UINT8,
    Ref,
    bf16_to_f32,
    build,
    compute_quant_params,
    dequantize,
    element_threshold,
    f32_to_bf16,
    minmax,
    packed_numel,
    partition,
    quant_params_from_minmax,
    quantize,
    quantize_per_element,
    requantize,
    ref_available,
)
