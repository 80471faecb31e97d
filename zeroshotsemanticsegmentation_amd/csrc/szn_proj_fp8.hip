// szn_proj_fp8.hip -- the pixel-projection GEMM (score_fr || seenmask_score: models.py:93,97,145,149) with fp8 operands
// (OCP e4m3, gfx950 `v_mfma_f32_16x16x32_fp8_fp8`, fp32 accumulation) and per-tensor scales -- BASELINE configs[4]
// ("fp8 MFMA projection GEMM").  Inference / forward only; the backward pass keeps the 16-bit operands.
//
//   sx = amax|x| / 448, sw = amax|w| / 448          (448 = largest e4m3 value; amax == 0 -> scale 1)
//   xq = e4m3(x * (448 / amax|x|)), wq likewise       (round to nearest even, the hardware conversion)
//   out[m][n] = (sum_k xq[m][k] * wq[n][k]) * (sx * sw) + bias[n]      (products exact in fp32, fp32 accumulation)
//
// Shape of the work (M = B*h*w <= a few thousand rows, K = 4096, N <= 320): a few GFLOP, latency-bound; operands are read
// straight from global memory as MFMA fragments (8 consecutive k bytes per lane), no LDS staging -- the whole quantised
// activation matrix is ~10 MB and stays in L2.
#include "szn_common.h"

namespace {

__device__ __forceinline__ float ld_any(const void* p, long i, int dtype) {
    if (dtype == SZN_F32) return ((const float*)p)[i];
    if (dtype == SZN_BF16) return bf16_bits_to_f32(((const uint16_t*)p)[i]);
    return (float)(((const _Float16*)p)[i]);
}

__global__ __launch_bounds__(256) void fp8_amax_kernel(const void* __restrict__ x, long n, int dtype, float* __restrict__ amax) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(ld_any(x, i, dtype)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax((int*)amax, __float_as_int(m));      // non-negative floats order like their bit patterns
}

// four elements per thread -> one packed 32-bit store
__global__ __launch_bounds__(256) void fp8_quant_kernel(const void* __restrict__ x, long n, int dtype, const float* __restrict__ amax,
                                                        uint32_t* __restrict__ q) {
    const float a = *amax;
    const float inv = (a > 0.f) ? 448.f / a : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i * 4 < n; i += (long)gridDim.x * 256) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (i * 4 + e < n) ? ld_any(x, i * 4 + e, dtype) * inv : 0.f;
        int pk = 0;
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], pk, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
        q[i] = (uint32_t)pk;
    }
}

// block = 4 waves; wave w -> pixels [m0 + 16 w, +16) x couts [n0, n0 + 64): four 16x16 accumulators.  A operand = weight
// fragment (lane (r16, g): 8 k-bytes g*8.. of cout r16), B = pixel fragment; D lane (r16, g) element e = (cout g*4+e, pixel r16)
__global__ __launch_bounds__(256) void fp8_gemm_kernel(const uint8_t* __restrict__ xq, const uint8_t* __restrict__ wq,
                                                       const float* __restrict__ amax, const float* __restrict__ bias,
                                                       float* __restrict__ out, long M, int K, int N, int ldo) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const long m = (long)blockIdx.x * 64 + 16 * w + r16;
    const int n0 = blockIdx.y * 64;
    const bool mok = m < M;
    const uint8_t* xp = xq + (mok ? m : 0) * (long)K + g * 8;
    const uint8_t* wp[4];
    bool nok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + i * 16 + r16;
        nok[i] = n < N;
        wp[i] = wq + (long)(nok[i] ? n : 0) * K + g * 8;
    }
    f32x4_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 128) {
        long xb[4], wb[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xb[u] = mok ? *(const long*)(xp + k0 + u * 32) : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) wb[u][i] = nok[i] ? *(const long*)(wp[i] + k0 + u * 32) : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wb[u][i], xb[u], acc[i], 0, 0, 0);
    }
    const float ax = amax[0], aw = amax[1];
    const float s = ((ax > 0.f) ? ax / 448.f : 1.f) * ((aw > 0.f) ? aw / 448.f : 1.f);
    // the pixel index of a D element is r16 (the B-operand row this lane supplied), the cout index g*4 + e
    const long mo = (long)blockIdx.x * 64 + 16 * w + r16;
    if (mo < M) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + i * 16 + g * 4 + e;
                if (n < N) out[mo * ldo + n] = acc[i][e] * s + (bias ? bias[n] : 0.f);
            }
    }
}

inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

extern "C" size_t szn_proj_fp8_workspace_bytes(long M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    return 256 + al256((size_t)M * K) + al256((size_t)N * K);
}

extern "C" int szn_proj_fp8_fwd(int x_dtype, int w_dtype, long M, int K, int N, int ldo, const void* x, const void* w,
                                const float* bias, float* out, void* workspace, szn_stream_t stream) {
    if (!x || !w || !out || !workspace || M <= 0 || N <= 0 || ldo < N) SZN_FAIL(SZN_ERR_ARG, "proj_fp8_fwd: bad argument");
    if (K <= 0 || (K % 128) != 0) SZN_FAIL(SZN_ERR_UNSUPPORTED, "proj_fp8_fwd: K = %d must be a multiple of 128", K);
    if (x_dtype < SZN_F32 || x_dtype > SZN_F16 || w_dtype < SZN_F32 || w_dtype > SZN_F16) SZN_FAIL(SZN_ERR_ARG, "proj_fp8_fwd: bad dtype");
    if (((uintptr_t)workspace) & 15) SZN_FAIL(SZN_ERR_ARG, "proj_fp8_fwd: workspace must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    float* amax = (float*)workspace;
    uint8_t* xq = (uint8_t*)workspace + 256;
    uint8_t* wq = xq + al256((size_t)M * K);
    if (hipMemsetAsync(amax, 0, 2 * sizeof(float), st) != hipSuccess) SZN_FAIL(SZN_ERR_LAUNCH, "proj_fp8_fwd: memset failed");
    const long nx = M * (long)K, nw = (long)N * K;
    hipLaunchKernelGGL(fp8_amax_kernel, dim3((unsigned)min((nx + 255) / 256, 2048L)), dim3(256), 0, st, x, nx, x_dtype, amax);
    hipLaunchKernelGGL(fp8_amax_kernel, dim3((unsigned)min((nw + 255) / 256, 2048L)), dim3(256), 0, st, w, nw, w_dtype, amax + 1);
    SZN_CHECK_LAUNCH("fp8_amax_kernel");
    hipLaunchKernelGGL(fp8_quant_kernel, dim3((unsigned)min((nx / 4 + 255) / 256 + 1, 4096L)), dim3(256), 0, st, x, nx, x_dtype,
                       (const float*)amax, (uint32_t*)xq);
    hipLaunchKernelGGL(fp8_quant_kernel, dim3((unsigned)min((nw / 4 + 255) / 256 + 1, 4096L)), dim3(256), 0, st, w, nw, w_dtype,
                       (const float*)(amax + 1), (uint32_t*)wq);
    SZN_CHECK_LAUNCH("fp8_quant_kernel");
    hipLaunchKernelGGL(fp8_gemm_kernel, dim3((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64)), dim3(256), 0, st,
                       (const uint8_t*)xq, (const uint8_t*)wq, (const float*)amax, bias, out, M, K, N, ldo);
    SZN_CHECK_LAUNCH("fp8_gemm_kernel");
    return SZN_OK;
}
