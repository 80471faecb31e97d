// szn_common.h -- shared device/host helpers for libszn_hip.so (gfx950 / CDNA4 only).
//
// Conventions used by every kernel in this library:
//   * activations are NHWC ("pixel-major"): [B][H][W][C], C contiguous; the element type T is
//     float (parity path) or bf16 (throughput path); accumulation is always fp32.
//   * conv weights are OHWI: [Cout][KH][KW][Cin] (== torch channels_last of an (O,I,KH,KW) tensor).
//   * every launch goes to the caller's hipStream_t, no host synchronisation inside.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include "../../include/szn.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---- error plumbing -------------------------------------------------------------------------
void szn_set_error(const char* fmt, ...);
#define SZN_FAIL(code, ...) do { szn_set_error(__VA_ARGS__); return (code); } while (0)
void szn_note_kernel(const char* name);          /* thread-local: the kernel the dispatcher picked (szn_last_kernel) */
void szn_note_colsum_rows(int rows);             /* thread-local: partial rows the last call wrote into its colsum slab */
int szn_noted_colsum_rows(void);                /* (internal: read back inside the SAME entry point, published through its out-parameter) */
float szn_noted_work_fraction(void);
static inline void szn_publish_result(const szn_conv_desc_t* d) {
    if (d && d->result) { d->result->colsum_rows = szn_noted_colsum_rows(); d->result->work_fraction = szn_noted_work_fraction(); }
}
void szn_note_work_fraction(float f);
int szn_knob(const char* name, int dflt);        /* environment knob, registered in szn_elementwise.hip's table (aborts on an unlisted name) */
int szn_knob_live(const char* name, int dflt);   /* the same, re-read on every call (the tests flip it inside one process) */            /* thread-local: fraction of the dense tiles the last conv call executed (constant-border hint) */
#define SZN_CHECK_LAUNCH(name) do { hipError_t e__ = hipGetLastError(); szn_note_kernel(name); \
    if (e__ != hipSuccess) SZN_FAIL(SZN_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); } while (0)

// ---- Adam, one element (torch.optim.Adam's scalar chain; no contraction: -ffp-contract=off).  Shared by adam_kernel
//      (szn_elementwise.hip) and the epilogue of conv_wgrad_wide<T, true> (szn_conv2d_wgrad_adam): the same instructions on the same
//      values, so the fused and the separate update agree bit for bit. ----
__device__ __forceinline__ float adam_elem(float& pi, float gi, float& mi, float& vi, float b1, float b2, float eps, float wd,
                                           float step_size, float inv_bc2_sqrt, float gscale) {
    gi = gi * gscale;
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    return pi;
}
// host side of the same step: step_size = lr / (1 - beta1^t), inv_bc2_sqrt = 1 / sqrt(1 - beta2^t)
inline void szn_adam_scalars(float lr, float beta1, float beta2, int step, float* step_size, float* inv_bc2_sqrt) {
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    *step_size = (float)((double)lr / bc1);
    *inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
}

// ---- class sets ---------------------------------------------------------------------------
// class sets travel as kernel arguments (4 words = SZN_MAX_CLASSES bits).  Words are picked by selects, not by a dynamic index
// into the by-value struct (which would put it in scratch).
struct ClassBits {
    uint64_t w[4];
};
__device__ __forceinline__ uint64_t class_word(const ClassBits& s, int i) {
    return i == 0 ? s.w[0] : (i == 1 ? s.w[1] : (i == 2 ? s.w[2] : s.w[3]));
}
__device__ __forceinline__ bool in_set(const ClassBits& s, long k) {
    return k >= 0 && k < SZN_MAX_CLASSES && ((class_word(s, (int)(k >> 6)) >> (k & 63)) & 1ull);
}
inline ClassBits class_bits(const szn_class_set* s) {
    ClassBits b{};
    if (s) for (int i = 0; i < 4; ++i) b.w[i] = s->w[i];
    return b;
}
inline ClassBits class_bits64(uint64_t w0) {
    ClassBits b{};
    b.w[0] = w0;
    return b;
}
inline bool class_bits_any(const ClassBits& b) { return (b.w[0] | b.w[1] | b.w[2] | b.w[3]) != 0; }
inline bool class_bits_fit(const ClassBits& b, int K) {          // no member >= K
    for (int k = K < 0 ? 0 : K; k < SZN_MAX_CLASSES; ++k)
        if ((b.w[k >> 6] >> (k & 63)) & 1ull) return false;
    return true;
}

// ---- element type traits --------------------------------------------------------------------
struct bf16_raw { uint16_t v; };

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
    return __uint_as_float(((uint32_t)b) << 16);
}
// round-to-nearest-even, NaN stays NaN (same rule as torch's float -> bfloat16 cast): gfx950 converts in hardware
// (v_cvt_pk_bf16_f32: one instruction per PAIR instead of ~7 VALU operations per element -- the conversions were a third of the
// VALU work of the 16-bit epilogues)
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }

// IEEE half as a 16-bit storage type (SZN_F16: BASELINE configs[4] "fp16 activations"); converted by the hardware's
// round-to-nearest-even v_cvt_f16_f32 / v_cvt_f32_f16
struct f16_raw { uint16_t v; };
__device__ __forceinline__ float f16_bits_to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

template <typename T> struct elem;
template <> struct elem<float> {
    static constexpr int kPer16B = 4;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct elem<bf16_raw> {
    static constexpr int kPer16B = 8;
    __device__ static __forceinline__ float ld(const bf16_raw* p) { return bf16_bits_to_f32(p->v); }
    __device__ static __forceinline__ void st(bf16_raw* p, float v) { p->v = f32_to_bf16_bits(v); }
};

template <> struct elem<f16_raw> {
    static constexpr int kPer16B = 8;
    __device__ static __forceinline__ float ld(const f16_raw* p) { return f16_bits_to_f32(p->v); }
    __device__ static __forceinline__ void st(f16_raw* p, float v) { p->v = f32_to_f16_bits(v); }
};

// 16-bit kernels are templated on the storage type (bf16_raw | f16_raw): conversions and the MFMA opcode are the only
// differences, the LDS-DMA / fragment plumbing moves raw 16-bit patterns
// (primary templates = bf16, so that code shared with the float instantiation still compiles; f16_raw is specialised)
template <typename T> __device__ __forceinline__ uint16_t to_bits16(float f) { return f32_to_bf16_bits(f); }
template <> __device__ __forceinline__ uint16_t to_bits16<f16_raw>(float f) { return f32_to_f16_bits(f); }
template <typename T> __device__ __forceinline__ float from_bits16(uint16_t b) { return bf16_bits_to_f32(b); }
template <> __device__ __forceinline__ float from_bits16<f16_raw>(uint16_t b) { return f16_bits_to_f32(b); }
typedef __attribute__((ext_vector_type(2))) __bf16 szn_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 szn_f16x2_t;
typedef __attribute__((ext_vector_type(2))) float szn_f32x2_t;
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {       // v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((szn_f32x2_t){lo, hi}, szn_bf16x2_t));
}
template <> __device__ __forceinline__ uint32_t pack2<f16_raw>(float lo, float hi) {        // v_cvt_pk_f16_f32 (RNE)
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((szn_f32x2_t){lo, hi}, szn_f16x2_t));
}
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) uint32_t szn_u32x4_t;
// D += A(16 x 32) * B(32 x 16) with 8 consecutive k per lane in one 128-bit register group
template <typename T> __device__ __forceinline__ f32x4_t mfma16(szn_u32x4_t a, szn_u32x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mfma16<f16_raw>(szn_u32x4_t a, szn_u32x4_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
static inline bool szn_is16(int dtype) { return dtype == SZN_BF16 || dtype == SZN_F16; }
static inline size_t szn_esize(int dtype) { return dtype == SZN_F32 ? 4 : 2; }

// ---- wave helpers (wave = 64 lanes) ---------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int szn_div_up(long a, long b) { return (int)((a + b - 1) / b); }

// Ablation switches (SZN_*_ABLATE) make kernels skip work and return WRONG results; they exist for the cycle-accounting
// experiments of profiles/r01_ablations.txt and are compiled in only by `make ABLATE=1` (-DSZN_ABLATE_BUILD).
static inline int szn_ablate_env(const char* name) {
#ifdef SZN_ABLATE_BUILD
    const char* e = getenv(name);
    return e ? atoi(e) : 0;
#else
    (void)name;
    return 0;
#endif
}
