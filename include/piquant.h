/* piquant.h -- C99 ABI of the MI355X-native libpiquant.so.
 *
 * This header declares exactly the six entry points, three enums and one opaque handle that the
 * reference library exports (reference include/piquant.h:21-85, implemented there by src/capi.cpp:15-104),
 * with the same names, argument order, value encodings and calling convention, so that any binding of
 * the reference (its cffi cdef string, python/src/piquant/_bootstrap.py:15-82, or a C caller) binds this
 * library unchanged.  Behind the boundary everything is new: HIP kernels for gfx950 launched by a thin C++
 * host layer (pi-quant_amd/csrc).  Additive, optional entry points live in piquant_hip.h.
 *
 * Pointer contract (extension of the reference, which knows host memory only):
 *   - device pointers (hipMalloc / PyTorch-ROCm tensor.data_ptr()) are processed in place, in HBM;
 *   - pageable host pointers (the reference's only kind) are served where they live: the call is forwarded whole to the companion
 *     libpiquant_cpu.so (the same arithmetic in AVX-512 on the host cores) when that library is present, and staged through device scratch
 *     over PCIe to the HIP kernels otherwise or when asked (piquant_hip.h, piquant_hip_set_host_path).  libpiquant.so itself contains no CPU
 *     arithmetic, and a context still needs a HIP device.
 * numel always counts logical elements; a packed buffer holds ceil(numel * bits / 8) bytes with the
 * lower-indexed element in the lower bits (reference src/kernels/quantize.inl:36-50).
 *
 * Errors: as in the reference (src/piquant.cpp:88-98) a contract violation prints a message to stderr and
 * calls abort(); there are no return codes.
 */
#ifndef PIQUANT_H
#define PIQUANT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIQUANT_EXPORT __attribute__((visibility("default")))

/* Opaque context.  Reference: include/piquant.h:21, src/capi.cpp:15-17. */
typedef struct piquant_context_t piquant_context_t;

/* Reference include/piquant.h:23-26.  NEAREST is round-half-away-from-zero of x/scale. */
typedef enum piquant_round_mode_t {
    PIQUANT_NEAREST = 0,
    PIQUANT_STOCHASTIC = 1
} piquant_round_mode_t;

/* Reference include/piquant.h:28-31.  SET: out[i] = dq(in[i]);  ADD: out[i] += dq(in[i]). */
typedef enum piquant_reduce_op_t {
    PIQUANT_REDUCE_OP_SET = 0,
    PIQUANT_REDUCE_OP_ADD = 1
} piquant_reduce_op_t;

/* Reference include/piquant.h:33-40 (order is part of the ABI; static-asserted in src/capi.cpp:9-13). */
typedef enum piquant_dtype_t {
    PIQUANT_DTYPE_F32 = 0,   /* IEEE-754 binary32                                   */
    PIQUANT_DTYPE_BF16 = 1,  /* bfloat16, 2 bytes                                    */
    PIQUANT_DTYPE_UINT2 = 2, /* 4 values per byte, value k in bits [2k, 2k+1]        */
    PIQUANT_DTYPE_UINT4 = 3, /* 2 values per byte, even index in the low nibble      */
    PIQUANT_DTYPE_UINT8 = 4  /* 1 value per byte                                     */
} piquant_dtype_t;

/* Reference include/piquant.h:42 / src/capi.cpp:19-22.  num_threads sized the reference's CPU thread
 * pool; the GPU grid replaces it, so the value is accepted and ignored.  The context binds to the HIP
 * device that is current at creation. */
PIQUANT_EXPORT piquant_context_t* piquant_context_create(size_t num_threads);

/* Reference include/piquant.h:43 / src/capi.cpp:24-26. */
PIQUANT_EXPORT void piquant_context_destroy(piquant_context_t* ctx);

/* out[i] = clamp(round(in[i] / scale) + zero_point, 0, 2^bits - 1), packed.
 * Reference include/piquant.h:45-55, src/capi.cpp:28-54, src/piquant.cpp:277-308.
 * dtype_in must be F32/BF16 and dtype_out UINT2/4/8, otherwise the call aborts. */
PIQUANT_EXPORT void piquant_quantize(
    piquant_context_t* ctx,
    const void* in,
    piquant_dtype_t dtype_in,
    void* out,
    piquant_dtype_t dtype_out,
    size_t numel,
    float scale,
    int64_t zero_point,
    piquant_round_mode_t mode);

/* out[i] (op)= (in[i] - zero_point) * scale.
 * Reference include/piquant.h:57-67, src/capi.cpp:56-82, src/piquant.cpp:310-340.
 * dtype_in must be UINT2/4/8 and dtype_out F32/BF16, otherwise the call aborts. */
PIQUANT_EXPORT void piquant_dequantize(
    piquant_context_t* ctx,
    const void* in,
    piquant_dtype_t dtype_in,
    void* out,
    piquant_dtype_t dtype_out,
    size_t numel,
    float scale,
    int64_t zero_point,
    piquant_reduce_op_t op);

/* (scale, zero_point) from the min/max of x for the given quantized dtype.
 * Reference include/piquant.h:69-76, src/capi.cpp:84-93, src/piquant.cpp:222-259, 371-375. */
PIQUANT_EXPORT void piquant_compute_quant_params_float32(
    piquant_context_t* ctx,
    const float* x,
    size_t n,
    piquant_dtype_t target_quant_dtype,
    float* out_scale,
    int64_t* out_zero_point);

/* Same for bfloat16 input given as raw 16-bit patterns.
 * Reference include/piquant.h:78-85, src/capi.cpp:95-104, src/piquant.cpp:377-381. */
PIQUANT_EXPORT void piquant_compute_quant_params_bfloat16(
    piquant_context_t* ctx,
    const uint16_t* x,
    size_t n,
    piquant_dtype_t target_quant_dtype,
    float* out_scale,
    int64_t* out_zero_point);

#ifdef __cplusplus
}
#endif
#endif /* PIQUANT_H */
