// szn_proj_fp8.hip -- the pixel-projection GEMM (score_fr || seenmask_score: models.py:93,97,145,149) with fp8 operands
// (OCP e4m3, gfx950 `v_mfma_f32_16x16x32_fp8_fp8`, fp32 accumulation) and per-tensor scales -- BASELINE configs[4]
// ("fp8 MFMA projection GEMM").  Forward (szn_proj_fp8_fwd) and, optionally, the two backward GEMMs of the layer
// (szn_proj_fp8_dgrad / szn_proj_fp8_wgrad: the gradient in OCP e5m2 -- range over precision -- against e4m3 weights /
// activations, `v_mfma_f32_16x16x32_fp8_bf8`).
//
//   sx = amax|x| / 448, sw = amax|w| / 448          (448 = largest e4m3 value; amax == 0 -> scale 1)
//   xq = e4m3(x * (448 / amax|x|)), wq likewise       (round to nearest even, the hardware conversion)
//   out[m][n] = (sum_k xq[m][k] * wq[n][k]) * (sx * sw) + bias[n]      (products exact in fp32, fp32 accumulation)
//
// Shape of the work (M = B*h*w <= a few thousand rows, K = 4096, N <= 320): a few GFLOP, latency-bound; operands are read
// straight from global memory as MFMA fragments (8 consecutive k bytes per lane), no LDS staging -- the whole quantised
// activation matrix is ~10 MB and stays in L2.
#include "szn_common.h"
#include <algorithm>

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

// ---- backward: generalised pieces ---------------------------------------------------------------------------------------------
// quantise a strided 2-D operand src[R][C] (row stride ld) into an 8-bit matrix whose CONTRACTION index is contiguous:
//   TR = false: q[r][c], row stride ldq (columns C .. ldq-1 zero)          -- contraction over c
//   TR = true : q[c][r], row stride ldq (rows r >= R zero up to ldq)        -- contraction over r
// BF8 = OCP e5m2 (max 57344) instead of e4m3 (max 448).  One thread = four consecutive bytes of q.
template <bool BF8, bool TR>
__global__ __launch_bounds__(256) void fp8_quant2d_kernel(const void* __restrict__ src, int dtype, long R, int C, long ld,
                                                          const float* __restrict__ amax, uint32_t* __restrict__ q, long ldq) {
    const float a = *amax;
    const float fmax = BF8 ? 57344.f : 448.f;
    const float inv = (a > 0.f) ? fmax / a : 1.f;
    const long rows_q = TR ? C : R;
    const long quads = ldq / 4;
    const long total = rows_q * quads;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float v[4];
        if (TR) {       // consecutive threads -> consecutive c (coalesced reads of src rows), four r per thread
            const long c = i % rows_q, r4 = (i / rows_q) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (r4 + e < R) ? ld_any(src, (r4 + e) * ld + c, dtype) * inv : 0.f;
            int pk = 0;
            if (BF8) { pk = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], pk, false); pk = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], pk, true); }
            else { pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], pk, false); pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true); }
            q[(c * ldq + r4) / 4] = (uint32_t)pk;
        } else {
            const long r = i / quads, c4 = (i % quads) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (c4 + e < C) ? ld_any(src, r * ld + c4 + e, dtype) * inv : 0.f;
            int pk = 0;
            if (BF8) { pk = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], pk, false); pk = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], pk, true); }
            else { pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], pk, false); pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true); }
            q[(r * ldq + c4) / 4] = (uint32_t)pk;
        }
    }
}

__global__ __launch_bounds__(256) void fp8_amax2d_kernel(const void* __restrict__ x, long R, int C, long ld, int dtype,
                                                         float* __restrict__ amax) {
    float m = 0.f;
    const long n = R * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        m = fmaxf(m, fabsf(ld_any(x, (i / C) * ld + (i % C), dtype)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax((int*)amax, __float_as_int(m));
}

// out[r][c] = epi( (sum_k rq[r][k] * cq[c][k]) * s ), k < Kc (a multiple of 32): rq = e5m2 rows (the gradient), cq = e4m3.
// Same wave layout as fp8_gemm_kernel (wave -> 16 rows x 64 columns).  epi: optional gate (gate[r][c] > 0 ? v : 0), optional
// per-(image, column) factor, output fp32 / bf16 / fp16.
__global__ __launch_bounds__(256) void fp8_gemm_bwd_kernel(const uint8_t* __restrict__ rq, const uint8_t* __restrict__ cq,
                                                           const float* __restrict__ amax, long R, int Kc, int Cn,
                                                           const void* __restrict__ gate, int gate_dtype, long ldgate,
                                                           const float* __restrict__ cscale, int rows_per_image,
                                                           void* __restrict__ out, int out_dtype, long ldo) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const long r = (long)blockIdx.x * 64 + 16 * w + r16;
    const int c0 = blockIdx.y * 64;
    const bool rok = r < R;
    const uint8_t* rp = rq + (rok ? r : 0) * (long)Kc + g * 8;
    const uint8_t* cp[4];
    bool cok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + i * 16 + r16;
        cok[i] = c < Cn;
        cp[i] = cq + (long)(cok[i] ? c : 0) * Kc + g * 8;
    }
    f32x4_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < Kc; k0 += 32) {
        const long rb = rok ? *(const long*)(rp + k0) : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long cb = cok[i] ? *(const long*)(cp[i] + k0) : 0;
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(cb, rb, acc[i], 0, 0, 0);     // A = e4m3 columns, B = e5m2 rows
        }
    }
    const float ar = amax[0], ac = amax[1];
    const float s = ((ar > 0.f) ? ar / 57344.f : 1.f) * ((ac > 0.f) ? ac / 448.f : 1.f);
    const long ro = (long)blockIdx.x * 64 + 16 * w + r16;          // D element e of lane (r16, g): row r16, column g*4 + e
    if (ro < R) {
        const long img = cscale ? ro / rows_per_image : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = c0 + i * 16 + g * 4 + e;
                if (c >= Cn) continue;
                float v = acc[i][e] * s;
                if (gate && !(ld_any(gate, ro * ldgate + c, gate_dtype) > 0.f)) v = 0.f;
                if (cscale) v *= cscale[img * Cn + c];
                if (out_dtype == SZN_F32) ((float*)out)[ro * ldo + c] = v;
                else if (out_dtype == SZN_BF16) ((uint16_t*)out)[ro * ldo + c] = f32_to_bf16_bits(v);
                else ((uint16_t*)out)[ro * ldo + c] = f32_to_f16_bits(v);
            }
    }
}

inline long pad32(long v) { return (v + 31) / 32 * 32; }

}  // namespace

extern "C" size_t szn_proj_fp8_bwd_workspace_bytes(long M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    // dgrad: g [M][pad32 N] + wT [K][pad32 N];  wgrad: gT [N][pad32 M] + xT [K][pad32 M]
    const size_t dg = al256((size_t)M * pad32(N)) + al256((size_t)K * pad32(N));
    const size_t wg = al256((size_t)N * pad32(M)) + al256((size_t)K * pad32(M));
    return 256 + (dg > wg ? dg : wg);
}

static int fp8_quant2d(bool bf8, bool tr, const void* src, int dtype, long R, int C, long ld, const float* amax, void* q, long ldq,
                       hipStream_t st) {
    const long total = (tr ? (long)C : R) * (ldq / 4);
    const dim3 grid((unsigned)std::min<long>((total + 255) / 256, 8192L));
    if (bf8 && tr) hipLaunchKernelGGL((fp8_quant2d_kernel<true, true>), grid, dim3(256), 0, st, src, dtype, R, C, ld, amax, (uint32_t*)q, ldq);
    else if (bf8) hipLaunchKernelGGL((fp8_quant2d_kernel<true, false>), grid, dim3(256), 0, st, src, dtype, R, C, ld, amax, (uint32_t*)q, ldq);
    else if (tr) hipLaunchKernelGGL((fp8_quant2d_kernel<false, true>), grid, dim3(256), 0, st, src, dtype, R, C, ld, amax, (uint32_t*)q, ldq);
    else hipLaunchKernelGGL((fp8_quant2d_kernel<false, false>), grid, dim3(256), 0, st, src, dtype, R, C, ld, amax, (uint32_t*)q, ldq);
    SZN_CHECK_LAUNCH("fp8_quant2d_kernel");
    return SZN_OK;
}

extern "C" int szn_proj_fp8_dgrad(int g_dtype, int w_dtype, long M, int K, int N, int ldg, const void* g, const void* w,
                                  const void* gate, int gate_dtype, int ldgate, const float* chan_scale, int rows_per_image,
                                  int out_dtype, void* dx, int ldx, void* workspace, szn_stream_t stream) {
    if (!g || !w || !dx || !workspace || M <= 0 || N <= 0 || K <= 0 || ldg < N || ldx < K) SZN_FAIL(SZN_ERR_ARG, "proj_fp8_dgrad: bad argument");
    if (g_dtype < SZN_F32 || g_dtype > SZN_F16 || w_dtype < SZN_F32 || w_dtype > SZN_F16 || out_dtype < SZN_F32 || out_dtype > SZN_F16 ||
        (gate && (gate_dtype < SZN_F32 || gate_dtype > SZN_F16 || ldgate < K)) || (chan_scale && rows_per_image < 1))
        SZN_FAIL(SZN_ERR_ARG, "proj_fp8_dgrad: bad dtype / epilogue argument");
    if (((uintptr_t)workspace) & 15) SZN_FAIL(SZN_ERR_ARG, "proj_fp8_dgrad: workspace must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    const long Np = pad32(N);
    float* amax = (float*)workspace;
    uint8_t* gq = (uint8_t*)workspace + 256;                    // [M][Np] e5m2
    uint8_t* wqT = gq + al256((size_t)M * Np);                  // [K][Np] e4m3 (transpose of w [N][K])
    if (hipMemsetAsync(amax, 0, 2 * sizeof(float), st) != hipSuccess) SZN_FAIL(SZN_ERR_LAUNCH, "proj_fp8_dgrad: memset failed");
    hipLaunchKernelGGL(fp8_amax2d_kernel, dim3((unsigned)std::min<long>((M * N + 255) / 256, 2048L)), dim3(256), 0, st, g, M, N, (long)ldg, g_dtype, amax);
    hipLaunchKernelGGL(fp8_amax2d_kernel, dim3((unsigned)std::min<long>(((long)N * K + 255) / 256, 2048L)), dim3(256), 0, st, w, (long)N, K, (long)K, w_dtype, amax + 1);
    SZN_CHECK_LAUNCH("fp8_amax2d_kernel");
    int rc = fp8_quant2d(true, false, g, g_dtype, M, N, ldg, amax, gq, Np, st);
    if (rc) return rc;
    rc = fp8_quant2d(false, true, w, w_dtype, N, K, K, amax + 1, wqT, Np, st);
    if (rc) return rc;
    hipLaunchKernelGGL(fp8_gemm_bwd_kernel, dim3((unsigned)((M + 63) / 64), (unsigned)((K + 63) / 64)), dim3(256), 0, st, (const uint8_t*)gq,
                       (const uint8_t*)wqT, (const float*)amax, M, (int)Np, K, gate, gate_dtype, (long)ldgate, chan_scale, rows_per_image,
                       dx, out_dtype, (long)ldx);
    SZN_CHECK_LAUNCH("fp8_gemm_bwd_kernel");
    return SZN_OK;
}

extern "C" int szn_proj_fp8_wgrad(int g_dtype, int x_dtype, long M, int K, int N, int ldg, int ldx, const void* g, const void* x,
                                  float* dw, void* workspace, szn_stream_t stream) {
    if (!g || !x || !dw || !workspace || M <= 0 || N <= 0 || K <= 0 || ldg < N || ldx < K) SZN_FAIL(SZN_ERR_ARG, "proj_fp8_wgrad: bad argument");
    if (g_dtype < SZN_F32 || g_dtype > SZN_F16 || x_dtype < SZN_F32 || x_dtype > SZN_F16) SZN_FAIL(SZN_ERR_ARG, "proj_fp8_wgrad: bad dtype");
    if (((uintptr_t)workspace) & 15) SZN_FAIL(SZN_ERR_ARG, "proj_fp8_wgrad: workspace must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    const long Mp = pad32(M);
    float* amax = (float*)workspace;
    uint8_t* gqT = (uint8_t*)workspace + 256;                   // [N][Mp] e5m2
    uint8_t* xqT = gqT + al256((size_t)N * Mp);                 // [K][Mp] e4m3
    if (hipMemsetAsync(amax, 0, 2 * sizeof(float), st) != hipSuccess) SZN_FAIL(SZN_ERR_LAUNCH, "proj_fp8_wgrad: memset failed");
    hipLaunchKernelGGL(fp8_amax2d_kernel, dim3((unsigned)std::min<long>((M * N + 255) / 256, 2048L)), dim3(256), 0, st, g, M, N, (long)ldg, g_dtype, amax);
    hipLaunchKernelGGL(fp8_amax2d_kernel, dim3((unsigned)std::min<long>((M * K + 255) / 256, 2048L)), dim3(256), 0, st, x, M, K, (long)ldx, x_dtype, amax + 1);
    SZN_CHECK_LAUNCH("fp8_amax2d_kernel");
    int rc = fp8_quant2d(true, true, g, g_dtype, M, N, ldg, amax, gqT, Mp, st);
    if (rc) return rc;
    rc = fp8_quant2d(false, true, x, x_dtype, M, K, ldx, amax + 1, xqT, Mp, st);
    if (rc) return rc;
    hipLaunchKernelGGL(fp8_gemm_bwd_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((K + 63) / 64)), dim3(256), 0, st, (const uint8_t*)gqT,
                       (const uint8_t*)xqT, (const float*)amax, (long)N, (int)Mp, K, nullptr, 0, 0L, nullptr, 1, dw, SZN_F32, (long)K);
    SZN_CHECK_LAUNCH("fp8_gemm_bwd_kernel");
    return SZN_OK;
}

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
