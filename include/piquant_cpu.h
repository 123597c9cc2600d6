/*
 * piquant_cpu.h -- host-memory companion of the MI355X library: the same quantize / dequantize / min-max arithmetic as the HIP kernels
 * (position-independent: the reference's SIMD-body formula applied to EVERY element), written for AVX-512 hosts
 * with a scalar form for everything else, over a persistent pool that splits a call by the reference's rule (src/piquant.cpp:145-157).
 *
 * Built into its own shared object (pi-quant_amd/piquant/libpiquant_cpu.so).  libpiquant.so never loads it unless a context is told to
 * (piquant_hip_set_host_path(ctx, PIQUANT_HIP_HOST_PATH_CPU) or PIQUANT_HIP_HOST_PATH=cpu), and then only for calls whose buffers are
 * pageable HOST memory -- the reference's calling convention, which the GPU can only serve by crossing PCIe twice.  Device pointers always
 * run the HIP kernels.  It is also bench.py's reproducible CPU baseline (`cpu_baseline.kind = "restatement"`).  Nothing here touches oracle/.
 *
 * Arithmetic restated from the reference's semantics (not its code): quantize src/kernels/kernels_specialized.inl:35-727 and
 * src/kernels/quantize.inl:8-26, dequantize kernels_specialized.inl:729-1416 and dequantize.inl:8-11, min/max :1418-1607,
 * epilogue src/piquant.cpp:245-258.
 */
#ifndef PIQUANT_CPU_H
#define PIQUANT_CPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIQUANT_CPU_EXPORT __attribute__((visibility("default")))

typedef struct piquant_cpu_context_t piquant_cpu_context_t;

/* num_threads = 0: one worker per PHYSICAL core this process may run on (a second hardware thread of a core adds nothing to kernels that wait for
 * DRAM).  The calling thread is worker 0.  quantize / dequantize cut every worker's share -- its partition by the reference's rule,
 * src/piquant.cpp:145-157 -- into 256 KiB chunks, a worker takes its own chunks first and then the others': results are those of any split. */
PIQUANT_CPU_EXPORT piquant_cpu_context_t* piquant_cpu_context_create(size_t num_threads);
PIQUANT_CPU_EXPORT void piquant_cpu_context_destroy(piquant_cpu_context_t* ctx);
PIQUANT_CPU_EXPORT size_t piquant_cpu_num_threads(const piquant_cpu_context_t* ctx);
/* Number of workers the NEXT calls use (1 .. num_threads; the pool keeps all of them). */
PIQUANT_CPU_EXPORT void piquant_cpu_set_active_threads(piquant_cpu_context_t* ctx, size_t threads);
/* cpus[t] = logical CPU worker t is pinned to (worker 0, the caller, only while a call runs); n = 0 removes the pinning. */
PIQUANT_CPU_EXPORT void piquant_cpu_set_affinity(piquant_cpu_context_t* ctx, const int* cpus, size_t n);
/* 1 when the AVX-512 kernels run on this host, 0 when every call takes the scalar forms. */
PIQUANT_CPU_EXPORT int piquant_cpu_has_avx512(void);
/* Process-wide switch between the AVX-512 kernels and the scalar forms (enable != 0 has no effect on a host without AVX-512); returns what
 * is in force.  The two give identical bytes -- the test suite runs both. */
PIQUANT_CPU_EXPORT int piquant_cpu_use_avx512(int enable);

/* dtype / mode / op codes are those of piquant.h (F32 0, BF16 1, UINT2 2, UINT4 3, UINT8 4; NEAREST 0, STOCHASTIC 1; SET 0, ADD 1).
 * `threshold` is the call's stochastic threshold in [0, 1) (the reference draws one per call, src/piquant.cpp:194-201; the caller draws
 * here).  Contract violations print to stderr and abort(), like the reference (src/piquant.cpp:88-98). */
PIQUANT_CPU_EXPORT void piquant_cpu_quantize(piquant_cpu_context_t* ctx, const void* in, int dtype_in, void* out, int dtype_out, size_t numel, float scale,
                                             int64_t zero_point, int round_mode, float threshold);
PIQUANT_CPU_EXPORT void piquant_cpu_dequantize(piquant_cpu_context_t* ctx, const void* in, int dtype_in, void* out, int dtype_out, size_t numel, float scale,
                                               int64_t zero_point, int reduce_op);
/* Reference-layout mode for host buffers (include/piquant_hip.h, piquant_hip_set_reference_layout): the same calls, and then the scalar heads and
 * tails of the partitions of a reference context with `threads` pool threads (src/piquant.cpp:145-157) are rewritten with the reference's scalar
 * formulas (std::round in the nearest fast paths' heads and tails; (q - zp) * scale and a second rounding for ADD in the bf16 tails; the uint2 ->
 * fp32 ADD tail that stores): every output byte equals what the reference's AVX-512 context of that many threads writes.  Position-dependent by
 * design; the plain calls above are not. */
PIQUANT_CPU_EXPORT void piquant_cpu_quantize_reference_layout(piquant_cpu_context_t* ctx, const void* in, int dtype_in, void* out, int dtype_out, size_t numel,
                                                              float scale, int64_t zero_point, int round_mode, float threshold, size_t threads);
PIQUANT_CPU_EXPORT void piquant_cpu_dequantize_reference_layout(piquant_cpu_context_t* ctx, const void* in, int dtype_in, void* out, int dtype_out, size_t numel,
                                                                float scale, int64_t zero_point, int reduce_op, size_t threads);
/* min / max of x as fp32, NaNs ignored, identities +-FLT_MAX. */
PIQUANT_CPU_EXPORT void piquant_cpu_minmax(piquant_cpu_context_t* ctx, const void* x, int dtype, size_t numel, float* out_min, float* out_max);
PIQUANT_CPU_EXPORT void piquant_cpu_compute_quant_params(piquant_cpu_context_t* ctx, const void* x, int dtype, size_t numel, int target_quant_dtype,
                                                         float* out_scale, int64_t* out_zero_point);
/* dst = src, every worker copying the part of the tensor it will later process: first-touch page placement for NUMA hosts. */
PIQUANT_CPU_EXPORT void piquant_cpu_partition_copy(piquant_cpu_context_t* ctx, const void* src, void* dst, int dtype, size_t numel);

#ifdef __cplusplus
}
#endif
#endif /* PIQUANT_CPU_H */
