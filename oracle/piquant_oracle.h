/* TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of pi-quant's quantize / dequantize /
 * compute_quant_params path.  Nothing in the product (pi-quant_amd/, include/) may include, link or call
 * this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference root).
 * Parity status: PINNED against the reference's own kernel units compiled from /root/reference
 * (oracle/_ref, see oracle/Makefile and tests/test_oracle_vs_ref.py) for quantize, dequantize and
 * min/max; the (min,max) -> (scale, zero_point) epilogue lives in src/piquant.cpp, which is
 * unbuildable here (needs the un-vendored threadpool submodule) -- that one function is pinned only
 * by the known answers recorded in SURVEY.md §8 a11 / tests/golden, i.e. "parity unpinned" by
 * reference execution in this container.
 */
#ifndef PIQUANT_ORACLE_H
#define PIQUANT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dtype codes: include/piquant.h:33-40 */
enum { ORC_F32 = 0, ORC_BF16 = 1, ORC_UINT2 = 2, ORC_UINT4 = 3, ORC_UINT8 = 4 };
/* rounding: include/piquant.h:23-26 ; store op: include/piquant.h:28-31 */
enum { ORC_NEAREST = 0, ORC_STOCHASTIC = 1 };
enum { ORC_SET = 0, ORC_ADD = 1 };

/* Which per-element formula is applied where.
 * ORC_FORM_REFERENCE: position-dependent, exactly what the reference's AVX-512 units do on ONE
 *   contiguous range (scalar head until `out` is 16-B aligned, SIMD body, scalar tail) --
 *   bit-identical to oracle/_ref (isa avx512f, threads 1).
 * ORC_FORM_UNIFORM: the SIMD-body formula on every element (position- and partition-independent).
 *   This is what the HIP kernels implement; it differs from FORM_REFERENCE only on the documented
 *   corner inputs (DESIGN.md "Where the reference disagrees with itself").
 */
enum { ORC_FORM_REFERENCE = 0, ORC_FORM_UNIFORM = 1 };

/* ceil(n / (8/bits)) for sub-byte types: src/piquant_internal.hpp:41-44 */
int64_t orc_packed_numel(int64_t numel, int dtype);
int orc_bit_size(int dtype);

/* bf16 helpers: include/piquant.hpp:86-95 */
uint16_t orc_f32_to_bf16(float x);
float orc_bf16_to_f32(uint16_t b);

/* One contiguous range, as one reference pool thread sees it (src/kernels/kernels.inl:151-173).
 * rnd_threshold is the per-call stochastic threshold of src/piquant.cpp:197-201 made explicit. */
void orc_quantize(const void* in, int dt_in, void* out, int dt_out, int64_t numel, float scale,
                  int64_t zero_point, int round_mode, float rnd_threshold, int form);
void orc_dequantize(const void* in, int dt_in, void* out, int dt_out, int64_t numel, float scale,
                    int64_t zero_point, int reduce_op, int form);

/* Fused quantize -> dequantize ("requant"), float type in == float type out, never materialising the quantized
 * tensor: src/kernels/kernels.inl:30-52 (requant_generic) over the generic scalar steps quantize.inl:8-26 and
 * dequantize.inl:8-11; C++ API include/piquant.hpp:276-285.  One formula for every element (no SIMD fast path). */
void orc_requantize(const void* in, int dt_inout, void* out, int quant_dtype, int64_t numel, float scale,
                    int64_t zero_point, int round_mode, float rnd_threshold, int reduce_op);

/* The static range split of src/piquant.cpp:139-157; returns 0 when the thread gets nothing. */
int orc_partition(int64_t numel, int64_t ti, int64_t tc, int packed_bits, int64_t* begin, int64_t* len);
/* Whole call as a context with `threads` pool threads would run it (src/piquant.cpp:159-169, 203-210). */
void orc_quantize_threads(const void* in, int dt_in, void* out, int dt_out, int64_t numel, float scale,
                          int64_t zero_point, int round_mode, float rnd_threshold, int form, int threads);
void orc_dequantize_threads(const void* in, int dt_in, void* out, int dt_out, int64_t numel, float scale,
                            int64_t zero_point, int reduce_op, int form, int threads);

/* src/kernels/kernels_specialized.inl:1418-1516 / 1518-1607 (NaN-free inputs) */
void orc_minmax_f32(const float* x, int64_t n, float out_min_max[2]);
void orc_minmax_bf16(const uint16_t* x, int64_t n, float out_min_max[2]);
/* src/piquant.cpp:213-259 */
void orc_quant_params_from_minmax(double r_min, double r_max, int quant_dtype, float* scale, int64_t* zero_point);
void orc_compute_quant_params_f32(const float* x, int64_t n, int quant_dtype, float* scale, int64_t* zero_point);
void orc_compute_quant_params_bf16(const uint16_t* x, int64_t n, int quant_dtype, float* scale, int64_t* zero_point);

/* Per-element counter RNG of the product's opt-in "per-element" stochastic mode (an extension; the
 * reference has one threshold per call).  Restated here so the GPU stream can be checked. */
float orc_element_threshold(uint64_t seed, uint64_t element_index);
void orc_quantize_per_element(const void* in, int dt_in, void* out, int dt_out, int64_t numel, float scale,
                              int64_t zero_point, uint64_t seed, uint64_t index_base);

#ifdef __cplusplus
}
#endif
#endif
