// szn_conv1_1_wgrad.hip -- weight gradient of conv1_1 (3 -> 64 channels, 3x3, pad P) for bf16 gradients, without an
// im2col image.
//
//   dw[co][t] = sum_pixels dout[p][co] * xcol[p][t],   t = (kh*3 + kw)*3 + ci (27 taps*channels, padded to 32)
// The im2col route writes and re-reads a [pixels][32] bf16 image (258 MB at B = 8) and reads all of dout (516 MB).  With
// pad = 100 only 52 % of the output pixels see the image at all: every other pixel has xcol = 0 and contributes
// nothing, so neither its dout row nor its taps are touched here.
//   * a K step is a segment of 32 consecutive output pixels of one row; only segments whose 3x3 windows overlap the
//     image are enumerated; a WAVE owns segments (no block barriers): its dout rows (32 x 128 B) stream into a private
//     3-deep LDS ring by LDS-DMA, A fragments (dout^T) come out with ds_read_b64_tr_b16;
//   * the B operand (xcol^T, 2 fragments of 16 taps): the 3 x 3 rows x 34 columns of the f32 NCHW image a segment's windows cover
//     are fetched with 5 coalesced buffer loads per lane (out-of-range offsets = the zero padding) one segment ahead, parked in a
//     wave-private LDS patch, and the 16 taps x pixels a lane supplies are read from there and rounded to bf16 like the im2col
//     path did.  (Round 2 gathered them straight from memory: 16 four-byte loads per lane whose 64 lanes hit ~20 different
//     cache lines -- the CU's address path, not HBM, bounded the kernel at 3 TB/s; STAGED = false keeps that form.);
//   * 8 MFMA 16x16x32 per segment into 8 accumulator fragments per wave; at the end the 4 waves of a block add up in
//     LDS and write one fp32 slab [64][32] per block; conv1_1_wgrad_reduce sums the slabs in a fixed order
//     (deterministic) into dw[64][27].
#include "szn_common.h"
#include "szn_cb.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((address_space(3))) void* ldsptr_t;

namespace {

struct C11Args {
    const char* dout; const float* x; float* ws;
    unsigned dout_bytes, x_bytes;
    int B, H, W, pad, Ho, Wo;
    int oh_lo, nrows, seg_lo, nsegx;       // touching rows [oh_lo, oh_lo + nrows), segments [seg_lo, seg_lo + nsegx) of 32 px
    long nseg;                             // B * nrows * nsegx
    BandCut cut;                           // dout is the cropped map [B][Hc][Wc][64] (szn_conv1_1_wgrad_c); empty cut = the full map
    int Hc, Wc;
};

constexpr unsigned kOOB1 = 0x80000000u;
constexpr int RING1 = 3;

constexpr int PS1 = 36;                      // floats per row of the image patch (34 used)
constexpr int PATCH1 = 9 * PS1 * 4;          // bytes per wave: rows (ci, kh)

template <typename T, bool STAGED>      // bf16_raw | f16_raw: rounding of the gathered image taps + the MFMA opcode
__global__ __launch_bounds__(256) void conv1_1_wgrad_kernel(C11Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) bf16x4_t* lp_t;
    __shared__ __attribute__((aligned(16))) char smem[4 * RING1 * 4096 + 4 * PATCH1];      // per wave: 3 stages of 32 px x 128 B | patch
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r16 = lane & 15;
    char* ring = smem + w * RING1 * 4096;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.dout, 0, (int)a.dout_bytes, 0x00020000);
    const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);

    // dout DMA: instruction j (0..3) = rows 8 j .. 8 j + 7; lane -> row 8 j + (lane >> 3), slot lane & 7, source chunk
    // slot ^ (((row >> 1) & 3) << 1) (transpose-read swizzle, see szn_conv_wgrad_taps.hip)
    const int rsub = lane >> 3;
    const unsigned chunkoff = (unsigned)(((lane & 7) ^ (((rsub >> 1) & 3) << 1)) << 4);
    // transpose reads: this lane supplies pixel row kk (and kk + 16) and 8 B = 4 couts
    const int kk = g * 4 + (r16 >> 2);
    const int sw = ((kk >> 1) & 3) << 1;
    int offA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) offA[i] = kk * 128 + (((i * 2) ^ sw) << 4) + (r16 & 3) * 8;

    // B operand taps of this lane: fragment f covers t = r16 + 16 f; element e of the MFMA K index 8 g + e is pixel
    // pix(e) = (e < 4 ? 4 g + e : 16 + 4 g + e - 4) -- the pixel order the transpose reads give the A operand
    int tkh[2], tkw[2]; unsigned tci[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int t = r16 + 16 * f;
        tkh[f] = t < 27 ? t / 9 : -100000;                         // pad taps never hit the image
        tkw[f] = (t / 3) % 3;
        tci[f] = (unsigned)(t % 3);
    }
    const unsigned plane = (unsigned)(a.H * a.W);

    auto seg_coords = [&](long s, int& b, int& oh, int& ow0) {
        const int sx = (int)(s % a.nsegx);
        const long r = s / a.nsegx;
        oh = a.oh_lo + (int)(r % a.nrows);
        b = (int)(r / a.nrows);
        ow0 = (a.seg_lo + sx) * 32;
    };
    auto issue_dma = [&](long s, int stage) {
        int b, oh, ow0;
        seg_coords(s, b, oh, ow0);
        char* sb = ring + stage * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ow = ow0 + 8 * j + rsub;
            const int yc = band_map(oh, a.cut.ya, a.cut.ye, a.cut.ya2, a.cut.ye2);
            const int xc = ow < a.Wo ? band_map(ow, a.cut.xa, a.cut.xe, a.cut.xa2, a.cut.xe2) : -1;     // removed / outside: zeros
            const unsigned v = (xc >= 0 && yc >= 0) ? (unsigned)(((b * a.Hc + yc) * a.Wc + xc) * 128) + chunkoff : kOOB1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(sb + j * 1024), 16, v, 0, 0, 0);
        }
    };
    auto gather = [&](long s, float (&xv)[2][8]) {
        int b, oh, ow0;
        seg_coords(s, b, oh, ow0);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int ih = oh + tkh[f] - a.pad;
            const bool rok = (unsigned)ih < (unsigned)a.H;
            const unsigned rowoff = ((unsigned)(b * 3) + tci[f]) * plane + (unsigned)(ih * a.W);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int pix = e < 4 ? 4 * g + e : 16 + 4 * g + e - 4;
                const int iw = ow0 + pix + tkw[f] - a.pad;
                const unsigned off = (rok && (unsigned)iw < (unsigned)a.W) ? (rowoff + (unsigned)iw) * 4u : kOOB1;
                xv[f][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsX, off, 0, 0));
            }
        }
    };

    // staged form: element idx = lane + 64 k of the patch is (row = idx / 34 = ci * 3 + kh, column c = idx % 34) <-> image pixel
    // (oh + kh - pad, ow0 + c - pad) of channel ci
    float* const patch = (float*)(smem + 4 * RING1 * 4096 + w * PATCH1);
    auto stage_load = [&](long s, float (&xr)[5]) {
        int b, oh, ow0;
        seg_coords(s, b, oh, ow0);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int idx = lane + 64 * k;
            const int row = (idx * 1928) >> 16;                        // idx / 34 for idx < 320
            const int c = idx - row * 34;
            const int ci = (row * 11) >> 5, kh = row - ci * 3;         // row / 3 for row < 10
            const int ih = oh + kh - a.pad, iw = ow0 + c - a.pad;
            const bool ok = idx < 306 && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            const unsigned off = ok ? (((unsigned)(b * 3 + ci)) * plane + (unsigned)(ih * a.W + iw)) * 4u : kOOB1;
            xr[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsX, off, 0, 0));
        }
    };
    auto stage_store = [&](const float (&xr)[5]) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int idx = lane + 64 * k;
            const int row = (idx * 1928) >> 16;
            if (idx < 306) patch[row * PS1 + (idx - row * 34)] = xr[k];
        }
    };
    int poff[2];                                                        // word offset of this lane's first tap pixel per fragment
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int t = r16 + 16 * f;
        poff[f] = t < 27 ? ((int)tci[f] * 3 + t / 9) * PS1 + tkw[f] + 4 * g : -1;
    }
    auto patch_read = [&](float (&xv)[2][8]) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const float* pp = patch + (poff[f] < 0 ? 0 : poff[f]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = pp[e < 4 ? e : 16 + e - 4];
                xv[f][e] = poff[f] < 0 ? 0.f : v;
            }
        }
    };

    f32x4_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[i][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const long nwaves = (long)gridDim.x * 4;
    const long s0 = (long)blockIdx.x * 4 + w;
    if constexpr (STAGED) {
        float xr[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (s0 < a.nseg) { issue_dma(s0, 0); stage_load(s0, xr); }
        if (s0 + nwaves < a.nseg) issue_dma(s0 + nwaves, 1);
        int stage = 0;
        for (long s = s0; s < a.nseg; s += nwaves) {
            // park this segment's image values (forces the wait for their loads; the DMA issued behind them stays in flight)
            stage_store(xr);
            float xn[2][8];
            patch_read(xn);
            u32x4_t xf[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                xf[f].x = pack2<T>(xn[f][0], xn[f][1]);
                xf[f].y = pack2<T>(xn[f][2], xn[f][3]);
                xf[f].z = pack2<T>(xn[f][4], xn[f][5]);
                xf[f].w = pack2<T>(xn[f][6], xn[f][7]);
            }
            const bool more1 = s + nwaves < a.nseg, more2 = s + 2 * nwaves < a.nseg;
            if (more1) stage_load(s + nwaves, xr);
            if (more2) issue_dma(s + 2 * nwaves, stage >= 1 ? stage - 1 : 2);
            // dout rows of THIS segment: newer in flight = this iteration's image loads (5) + DMA (4) and last iteration's DMA (4)
            if (more2) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
            else if (more1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const char* sb = ring + stage * 4096;
            u32x4_t df[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp_t)(sb + offA[i]));
                const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp_t)(sb + 2048 + offA[i]));
                const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
                df[i] = u32x4_t{l2.x, l2.y, h2.x, h2.y};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int f = 0; f < 2; ++f)
                    acc[i][f] = mfma16<T>(df[i], xf[f], acc[i][f]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // stage drained before it is refilled
            stage = stage == 2 ? 0 : stage + 1;
        }
    } else {
    // software pipeline per wave: DMA two segments ahead, gather one segment ahead
    float xn[2][8];
    if (s0 < a.nseg) { issue_dma(s0, 0); gather(s0, xn); }
    if (s0 + nwaves < a.nseg) issue_dma(s0 + nwaves, 1);
    int stage = 0;
    for (long s = s0; s < a.nseg; s += nwaves) {
        // pack the gathered taps of this segment (forces the wait for its loads), then start the next gather
        u32x4_t xf[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            xf[f].x = pack2<T>(xn[f][0], xn[f][1]);
            xf[f].y = pack2<T>(xn[f][2], xn[f][3]);
            xf[f].z = pack2<T>(xn[f][4], xn[f][5]);
            xf[f].w = pack2<T>(xn[f][6], xn[f][7]);
        }
        const bool more1 = s + nwaves < a.nseg, more2 = s + 2 * nwaves < a.nseg;
        // gather first, DMA second: the compiler's wait for the gathered values at the top of the next iteration then
        // leaves the DMA in flight
        if (more1) gather(s + nwaves, xn);
        if (more2) issue_dma(s + 2 * nwaves, stage >= 1 ? stage - 1 : 2);      // (stage + 2) % 3: drained last iteration
        // dout rows of THIS segment: newer in flight = this iteration's DMA (4) + gather (16) and last iteration's DMA (4)
        if (more2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (more1) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const char* sb = ring + stage * 4096;
        u32x4_t df[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp_t)(sb + offA[i]));
            const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp_t)(sb + 2048 + offA[i]));
            const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
            df[i] = u32x4_t{l2.x, l2.y, h2.x, h2.y};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int f = 0; f < 2; ++f)
                acc[i][f] = mfma16<T>(df[i], xf[f], acc[i][f]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // stage drained before it is refilled
        stage = stage == 2 ? 0 : stage + 1;
    }
    }

    // ---- block partial: D[co = 16 i + 4 g + e][t = 16 f + r16], waves added in a fixed order ----
    __syncthreads();
    float* red = (float*)smem;                                       // [4 waves][64][32] = 32 KiB <= 48 KiB ring
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(w * 64 + 16 * i + 4 * g + e) * 32 + 16 * f + r16] = acc[i][f][e];
    __syncthreads();
    float* slab = a.ws + (size_t)blockIdx.x * 2048;
    for (int idx = tid; idx < 2048; idx += 256)
        slab[idx] = (red[idx] + red[2048 + idx]) + (red[4096 + idx] + red[6144 + idx]);
#endif
}

// dw[co][t] (+)= sum over the block slabs: 8 outputs per block, 32 lanes each striding over the slabs, then a fixed-order
// shuffle tree (deterministic)
__global__ __launch_bounds__(256) void conv1_1_wgrad_reduce(const float* __restrict__ ws, float* __restrict__ dw, int nblocks,
                                                            int accumulate) {
    const int i = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    const bool ok = i < 64 * 27;
    const int co = ok ? i / 27 : 0, t = ok ? i - co * 27 : 0;
    float s = 0.f;
    if (ok)
        for (int b = l; b < nblocks; b += 32) s += ws[(size_t)b * 2048 + co * 32 + t];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) s += __shfl_xor(s, d, 32);
    if (ok && l == 0) dw[i] = accumulate ? dw[i] + s : s;
}

}  // namespace

// the rows / 32-pixel column segments of dout whose 3x3 windows overlap the image; false: the fused kernel does not apply
static bool c11_geometry(int dtype, int B, int H, int W, int pad, int& oh_lo, int& oh_hi, int& seg_lo, int& seg_hi) {
    if (!szn_is16(dtype) || B <= 0 || H <= 0 || W <= 0 || pad < 0) return false;
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    const size_t dout_bytes = (size_t)B * Ho * Wo * 128, x_bytes = (size_t)B * 3 * H * W * 4;
    if (Ho <= 0 || Wo <= 0 || dout_bytes >= 0x7fff0000ul || x_bytes >= 0x7fff0000ul) return false;
    // output pixel (oh, ow) sees the image iff oh - pad + kh in [0, H) for some kh in 0..2, same for columns
    oh_lo = pad - 2; if (oh_lo < 0) oh_lo = 0;
    oh_hi = pad + H; if (oh_hi > Ho) oh_hi = Ho;                      // exclusive
    int ow_lo = pad - 2; if (ow_lo < 0) ow_lo = 0;
    int ow_hi = pad + W; if (ow_hi > Wo) ow_hi = Wo;
    if (oh_hi <= oh_lo || ow_hi <= ow_lo) return false;
    seg_lo = ow_lo / 32; seg_hi = (ow_hi + 31) / 32;
    return true;
}

// the part of dout szn_conv1_1_wgrad (db == NULL) reads: rect = rows [r0, r1) x columns [c0, c1); 1 = a proper sub-rectangle
extern "C" int szn_conv1_1_wgrad_reads(int dtype, int B, int H, int W, int pad, int rect[4]) {
    if (!rect) return 0;
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    rect[0] = 0; rect[1] = Ho; rect[2] = 0; rect[3] = Wo;
    int oh_lo, oh_hi, seg_lo, seg_hi;
    if (!c11_geometry(dtype, B, H, W, pad, oh_lo, oh_hi, seg_lo, seg_hi)) return 0;
    rect[0] = oh_lo; rect[1] = oh_hi; rect[2] = seg_lo * 32; rect[3] = seg_hi * 32 < Wo ? seg_hi * 32 : Wo;
    return (rect[0] > 0 || rect[1] < Ho || rect[2] > 0 || rect[3] < Wo) ? 1 : 0;
}

// bf16 path of szn_conv1_1_wgrad (szn_elementwise.hip).  workspace: >= nblocks * 8 KiB.  Returns 1 if not applicable.
int szn_conv1_1_wgrad_fused_try(int dtype, int B, int H, int W, int pad, const float* x, const void* dout, float* dw, int accumulate,
                                void* workspace, size_t workspace_bytes, szn_stream_t stream, const int* cutv) {
    const int Ho = H + 2 * pad - 2, Wo = W + 2 * pad - 2;
    int oh_lo, oh_hi, seg_lo, seg_hi;
    if (!c11_geometry(dtype, B, H, W, pad, oh_lo, oh_hi, seg_lo, seg_hi)) return 1;
    C11Args a;
    a.cut = BandCut{0, 0, Ho, Ho, 0, 0, Wo, Wo};
    if (cutv) a.cut = BandCut{cutv[0], cutv[1], cutv[2], cutv[3], cutv[4], cutv[5], cutv[6], cutv[7]};
    a.Hc = Ho - (a.cut.ye - a.cut.ya) - (a.cut.ye2 - a.cut.ya2); a.Wc = Wo - (a.cut.xe - a.cut.xa) - (a.cut.xe2 - a.cut.xa2);
    const size_t dout_bytes = (size_t)B * a.Hc * a.Wc * 128, x_bytes = (size_t)B * 3 * H * W * 4;
    a.dout = (const char*)dout; a.x = x; a.ws = (float*)workspace;
    a.dout_bytes = (unsigned)dout_bytes; a.x_bytes = (unsigned)x_bytes;
    a.B = B; a.H = H; a.W = W; a.pad = pad; a.Ho = Ho; a.Wo = Wo;
    a.oh_lo = oh_lo; a.nrows = oh_hi - oh_lo;
    a.seg_lo = seg_lo; a.nsegx = seg_hi - seg_lo;
    a.nseg = (long)B * a.nrows * a.nsegx;
    long blocks = (a.nseg + 4 * 16 - 1) / (4 * 16);                   // >= 16 segments per wave
    if (blocks > 768) blocks = 768;
    if (blocks < 1) blocks = 1;
    if (workspace_bytes < (size_t)blocks * 2048 * sizeof(float)) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SZN_F16) hipLaunchKernelGGL((conv1_1_wgrad_kernel<f16_raw, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv1_1_wgrad_kernel<bf16_raw, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    SZN_CHECK_LAUNCH("conv1_1_wgrad_kernel");
    hipLaunchKernelGGL(conv1_1_wgrad_reduce, dim3(64 * 27 / 8), dim3(256), 0, st, (const float*)workspace, dw,
                       (int)blocks, accumulate);
    SZN_CHECK_LAUNCH("conv1_1_wgrad_reduce");
    return SZN_OK;
}
