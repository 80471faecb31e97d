// szn_band.hip -- row / column remapping of an NHWC map (round 5): the constant band of the pad-100 network inside the conv3 block.
//
// models.py:43 pads conv1_1 by 100, so at 1/4 resolution (conv3_1 .. conv3_3, 178 x 178 for a 512 x 512 image) the rows and columns
// between the tensor edge and the image's reach hold ONE value per channel (models._cb_* track the intervals).  A 3x3 convolution maps
// equal neighbourhoods to equal outputs, so most of those rows / columns need not exist: engine code removes an even number of them per
// side (szn_band_remap with one source per output element = crop), runs conv3_1 .. conv3_3 + pool3 on the smaller map (25 % fewer
// pixels at 512 x 512) and puts the pooled rows back by copying a representative one (one source per output again).  Backward: the
// gradient of the copies is their SUM (several sources per output element), the gradient of the removed input rows is zero (no source).
// All four are this one kernel:
//     out[b][y][x][c] = sum over sy in [ys(y), ys(y) + yn(y)), sx in [xs(x), xs(x) + xn(x)) of in[b][sy][sx][c]        (fp32, ascending order)
// with per-axis tables {start, count} on the device.  A 16-B channel group per thread; single-source elements are moved without conversion.
#include "szn_common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void band_remap_kernel(const T* __restrict__ in, T* __restrict__ out, const int* __restrict__ ytab,
                                                         const int* __restrict__ xtab, int B, int Hi, int Wi, int Ho, int Wo, int C) {
    constexpr int CH = 16 / sizeof(T);
    const int cg = C / CH;
    const long total = (long)B * Ho * Wo * cg;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cg);
        long p = i / cg;
        const int x = (int)(p % Wo); p /= Wo;
        const int y = (int)(p % Ho);
        const int b = (int)(p / Ho);
        const int ys = ytab[2 * y], yn = ytab[2 * y + 1], xs = xtab[2 * x], xn = xtab[2 * x + 1];
        szn_u32x4_t o = {0u, 0u, 0u, 0u};
        if (yn == 1 && xn == 1) {
            o = *(const szn_u32x4_t*)(in + (((long)b * Hi + ys) * Wi + xs) * C + c * CH);
        } else if (yn > 0 && xn > 0) {
            float acc[CH];
#pragma unroll
            for (int e = 0; e < CH; ++e) acc[e] = 0.f;
            for (int sy = ys; sy < ys + yn; ++sy)
                for (int sx = xs; sx < xs + xn; ++sx) {
                    const T* q = in + (((long)b * Hi + sy) * Wi + sx) * C + c * CH;
#pragma unroll
                    for (int e = 0; e < CH; ++e) acc[e] += elem<T>::ld(q + e);
                }
            T r[CH];
#pragma unroll
            for (int e = 0; e < CH; ++e) elem<T>::st(r + e, acc[e]);
            o = *(const szn_u32x4_t*)r;
        }
        *(szn_u32x4_t*)(out + (((long)b * Ho + y) * Wo + x) * C + c * CH) = o;
    }
}

// In-place form of the summing rows / columns of a transposed map: every run {start, count} of `runs` is summed (fp32, ascending) into its FIRST row
// (axis 0) or column (axis 1).  The few runs of a band map with count > 1 -- a representative row / column and the copies that stood in for it --
// touch ~10 % of the tensor; afterwards the map is a plain one-source gather, which szn_maxpool2x2_ceil_bwd_code_gather applies while it reads
// (rows first, then columns: the same values as the two szn_band_remap passes, bit for bit).
template <typename T>
__global__ __launch_bounds__(256) void band_fold_kernel(T* __restrict__ d, const int* __restrict__ runs, int nruns, int axis, int B, int H, int W, int C) {
    constexpr int CH = 16 / sizeof(T);
    const int cg = C / CH, other = axis == 0 ? W : H;
    const long total = (long)B * nruns * other * cg;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cg);
        long p = i / cg;
        const int o = (int)(p % other); p /= other;
        const int r = (int)(p % nruns);
        const int b = (int)(p / nruns);
        const int s0 = runs[2 * r], cnt = runs[2 * r + 1];
        T* q = axis == 0 ? d + (((long)b * H + s0) * W + o) * C + c * CH : d + (((long)b * H + o) * W + s0) * C + c * CH;
        const long stride = axis == 0 ? (long)W * C : (long)C;
        float acc[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[e] = 0.f;
        for (int k = 0; k < cnt; ++k)
#pragma unroll
            for (int e = 0; e < CH; ++e) acc[e] += elem<T>::ld(q + k * stride + e);
        T rr[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) elem<T>::st(rr + e, acc[e]);
        *(szn_u32x4_t*)q = *(const szn_u32x4_t*)rr;
    }
}

}  // namespace

extern "C" int szn_band_fold(int dtype, int B, int H, int W, int C, void* d, int axis, const int* runs, int n_runs, szn_stream_t stream) {
    if (!d || !runs || B <= 0 || H <= 0 || W <= 0 || C <= 0 || n_runs <= 0 || (axis != 0 && axis != 1)) SZN_FAIL(SZN_ERR_ARG, "band_fold: bad argument");
    const int ch = szn_is16(dtype) ? 8 : 4;
    if ((dtype != SZN_F32 && !szn_is16(dtype)) || (C % ch) || ((uintptr_t)d & 15))
        SZN_FAIL(SZN_ERR_UNSUPPORTED, "band_fold: dense NHWC map, 16-B aligned, C a multiple of %d", ch);
    const long total = (long)B * n_runs * (axis == 0 ? W : H) * (C / ch);
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SZN_BF16) hipLaunchKernelGGL(band_fold_kernel<bf16_raw>, dim3((unsigned)blocks), dim3(256), 0, st, (bf16_raw*)d, runs, n_runs, axis, B, H, W, C);
    else if (dtype == SZN_F16) hipLaunchKernelGGL(band_fold_kernel<f16_raw>, dim3((unsigned)blocks), dim3(256), 0, st, (f16_raw*)d, runs, n_runs, axis, B, H, W, C);
    else hipLaunchKernelGGL(band_fold_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (float*)d, runs, n_runs, axis, B, H, W, C);
    SZN_CHECK_LAUNCH("band_fold_kernel");
    return SZN_OK;
}

extern "C" int szn_band_remap(int dtype, int B, int Hi, int Wi, int Ho, int Wo, int C, const void* in, void* out, const int* ytab,
                              const int* xtab, szn_stream_t stream) {
    if (!in || !out || !ytab || !xtab || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0)
        SZN_FAIL(SZN_ERR_ARG, "band_remap: bad argument");
    const int ch = szn_is16(dtype) ? 8 : 4;
    if ((dtype != SZN_F32 && !szn_is16(dtype)) || (C % ch) || (((uintptr_t)in | (uintptr_t)out) & 15))
        SZN_FAIL(SZN_ERR_UNSUPPORTED, "band_remap: dense NHWC maps, 16-B aligned, C a multiple of %d", ch);
    const long total = (long)B * Ho * Wo * (C / ch);
    long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SZN_BF16)
        hipLaunchKernelGGL(band_remap_kernel<bf16_raw>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_raw*)in, (bf16_raw*)out, ytab, xtab, B,
                           Hi, Wi, Ho, Wo, C);
    else if (dtype == SZN_F16)
        hipLaunchKernelGGL(band_remap_kernel<f16_raw>, dim3((unsigned)blocks), dim3(256), 0, st, (const f16_raw*)in, (f16_raw*)out, ytab, xtab, B,
                           Hi, Wi, Ho, Wo, C);
    else
        hipLaunchKernelGGL(band_remap_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)in, (float*)out, ytab, xtab, B, Hi, Wi,
                           Ho, Wo, C);
    SZN_CHECK_LAUNCH("band_remap_kernel");
    return SZN_OK;
}
