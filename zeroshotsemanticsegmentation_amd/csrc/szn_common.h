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
#include "../../include/szn.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---- error plumbing -------------------------------------------------------------------------
void szn_set_error(const char* fmt, ...);
#define SZN_FAIL(code, ...) do { szn_set_error(__VA_ARGS__); return (code); } while (0)
void szn_note_kernel(const char* name);          /* thread-local: the kernel the dispatcher picked (szn_last_kernel) */
#define SZN_CHECK_LAUNCH(name) do { hipError_t e__ = hipGetLastError(); szn_note_kernel(name); \
    if (e__ != hipSuccess) SZN_FAIL(SZN_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); } while (0)

// ---- element type traits --------------------------------------------------------------------
struct bf16_raw { uint16_t v; };

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
    return __uint_as_float(((uint32_t)b) << 16);
}
// round-to-nearest-even, NaN preserved (same rule as torch's float -> bfloat16 cast)
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

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
