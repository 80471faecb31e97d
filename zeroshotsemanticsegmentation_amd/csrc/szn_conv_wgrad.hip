// szn_conv_wgrad.hip -- second-generation weight-gradient kernel for gfx950 and the public szn_conv2d_wgrad
// dispatcher.   dw[co][kh][kw][ci] += sum_pixels dout[p][co] * in[p + (kh,kw) - pad][ci]
//
// conv_wgrad_v2: per filter tap a GEMM D[co][ci] = A^T B with A = dout [pixel][co], B = shifted in [pixel][ci]; the
// contraction index (pixels) is the slow one in both operands, so tiles are staged pixel-major and fragments are
// fetched with ds_read_b64_tr_b16 (bf16, hardware transpose) or ds_read_b32 (f32).
//   * block = 256 threads = 4 waves (2 x 2), tile (32 FA) couts x (32 FB) cins, FA, FB in {2, 4}: 64-channel layers
//     (conv1_2, conv2_1, conv1_1's im2col form) no longer pad to 128;
//   * operands go to LDS with buffer_load ... lds; out-of-image taps / tile edges / split ends are out-of-range
//     offsets (zeros); the 16-B chunk index of every row is XOR-swizzled on the SOURCE side so that the transpose
//     reads (4 rows x 32 B per 16-lane group) and the f32 reads are bank-conflict free without row padding;
//   * pixel coordinates advance incrementally (no integer division in the K loop);
//   * 3-stage LDS ring (<= 48 KiB, 3 blocks per CU), counted vmcnt, one raw s_barrier per K step;
//   * split over pixels; partial tiles are added to the fp32 OHWI gradient with global atomics.
#include "szn_common.h"
#include <algorithm>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((address_space(3))) void* ldsptr_t;

int szn_conv2d_wgrad_v1(const szn_conv_desc_t* d, const void* in, const void* dout, float* dw, int accumulate,
                        szn_stream_t stream);

int szn_conv_wgrad_taps_try(const szn_conv_desc_t* d, const void* in, const void* dout, float* dw, int accumulate,
                            int min_tiles_per_block, szn_stream_t stream);

int szn_conv_wgrad_wide_try(const szn_conv_desc_t* d, const void* in, const void* dout, float* dw, int accumulate,
                            int min_tiles, szn_stream_t stream, const szn_adam_args_t* opt = nullptr);

namespace {

struct Wg2Args {
    const char* dout; const char* in; float* dw;
    unsigned dout_bytes, in_bytes;
    int B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, pad;
    int ldi, ldd;
    int M;
    int kspan, nsplit, cotiles, citiles;
    int plain_store;           // single split and no accumulation: write the tile instead of atomically adding it
    float* slab;               // pixel splits > 1 with a workspace: split s stores its partial into slab[s][Co*KH*KW*Ci] (plain
                               // stores; wgrad_slab_reduce adds the splits in a fixed order: deterministic, no atomics / memset)
    int ablate;                // debug (env SZN_WG_ABLATE=1): skip the output epilogue (wrong results)
};

constexpr unsigned kOOBw = 0x80000000u;

template <typename T, int FA, int FB>
__global__ __launch_bounds__(256, 3) void conv_wgrad_v2(Wg2Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int ES = sizeof(T);
    constexpr int CH = 16 / ES;                    // elements per 16-B chunk
    constexpr int KP = (ES == 2) ? 32 : 16;        // pixels per K step
    constexpr int CO_T = 32 * FA, CI_T = 32 * FB;  // tile channels
    constexpr int RBA = CO_T * ES, RBB = CI_T * ES;            // LDS row bytes
    constexpr int CPRA = RBA / 16, CPRB = RBB / 16;            // 16-B chunks per row
    constexpr int RPIA = 1024 / RBA, RPIB = 1024 / RBB;        // rows per LDS-DMA instruction
    constexpr int NA = FA / 2, NBI = FB / 2;                   // LDS-DMA instructions per wave per step
    constexpr int LPC = NA + NBI;
    constexpr int STAGE = KP * (RBA + RBB);
    __shared__ __attribute__((aligned(16))) char smem[3 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;             // wave -> (16 FA) couts x (16 FB) cins
    const int g = lane >> 4, r16 = lane & 15;

    int bid = blockIdx.x;
    const int cit = bid % a.citiles; bid /= a.citiles;
    const int cot = bid % a.cotiles; bid /= a.cotiles;
    const int ntap = a.KH * a.KW;
    const int tap = bid % ntap, split = bid / ntap;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int co0 = cot * CO_T, ci0 = cit * CI_T;
    const int mbeg = split * a.kspan;
    const int mend = min(a.M, mbeg + a.kspan);
    if (mbeg >= mend) return;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.dout, 0, (int)a.dout_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);

    // row swizzle (XOR on the 16-B chunk index), see header comment
    auto swz = [](int row, int rowbytes) -> int {
        if (ES == 2) return rowbytes == 256 ? ((row & 7) << 1) : (((row >> 1) & 3) << 1);
        return (row & 1) << 2;
    };

    // ---- LDS-DMA slots of this thread ----
    int mA[NA]; unsigned chA[NA];       // pixel index and channel byte offset (or OOB) of the dout slots
    int mB[NBI], ohB[NBI], owB[NBI], bB[NBI]; unsigned chB[NBI];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (w * NA + i) * RPIA + lane / CPRA, slot = lane % CPRA;
        const int chunk = slot ^ swz(row, RBA);
        const int co = co0 + chunk * CH;
        mA[i] = mbeg + row;
        chA[i] = (co < a.Co && co + CH <= a.ldd) ? (unsigned)(co * ES) : kOOBw;
    }
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
        const int row = (w * NBI + i) * RPIB + lane / CPRB, slot = lane % CPRB;
        const int chunk = slot ^ swz(row, RBB);
        const int ci = ci0 + chunk * CH;
        const int m = mbeg + row;
        mB[i] = m;
        const int b = m / (a.Ho * a.Wo), r = m - b * (a.Ho * a.Wo);
        bB[i] = b; ohB[i] = r / a.Wo; owB[i] = r - ohB[i] * a.Wo;
        chB[i] = (ci < a.Ci) ? (unsigned)(ci * ES) : kOOBw;
    }
    auto issue = [&](int stage) {
        char* sb = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const unsigned v = (mA[i] < mend && chA[i] != kOOBw) ? (unsigned)mA[i] * (unsigned)(a.ldd * ES) + chA[i] : kOOBw;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(sb + (w * NA + i) * 1024), 16, v, 0, 0, 0);
            mA[i] += KP;
        }
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int ih = ohB[i] + kh - a.pad, iw = owB[i] + kw - a.pad;
            const bool ok = mB[i] < mend && chB[i] != kOOBw && (unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi;
            const unsigned v = ok ? (unsigned)((bB[i] * a.Hi + ih) * a.Wi + iw) * (unsigned)(a.ldi * ES) + chB[i] : kOOBw;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)(sb + KP * RBA + (w * NBI + i) * 1024), 16, v, 0, 0, 0);
            mB[i] += KP; owB[i] += KP;
            while (owB[i] >= a.Wo) { owB[i] -= a.Wo; if (++ohB[i] >= a.Ho) { ohB[i] = 0; ++bB[i]; } }
        }
    };

    f32x4_t acc[FA][FB];
#pragma unroll
    for (int i = 0; i < FA; ++i)
#pragma unroll
        for (int j = 0; j < FB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- per-lane read offsets (bytes within a stage) ----
    int offA[FA], offB[FB];
    if constexpr (ES == 2) {
        const int kk = g * 4 + (r16 >> 2);                 // row supplied by this lane (first read; second is +16 rows)
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            const int cb = wm * FA * 16 + i * 16;
            offA[i] = kk * RBA + ((((cb >> 3)) ^ swz(kk, RBA)) << 4) + (r16 & 3) * 8;
        }
#pragma unroll
        for (int j = 0; j < FB; ++j) {
            const int cb = wn * FB * 16 + j * 16;
            offB[j] = KP * RBA + kk * RBB + ((((cb >> 3)) ^ swz(kk, RBB)) << 4) + (r16 & 3) * 8;
        }
    } else {
#pragma unroll
        for (int i = 0; i < FA; ++i) {
            const int c = wm * FA * 16 + i * 16 + r16;
            offA[i] = g * RBA + (((c >> 2) ^ swz(g, RBA)) << 4) + (c & 3) * 4;
        }
#pragma unroll
        for (int j = 0; j < FB; ++j) {
            const int c = wn * FB * 16 + j * 16 + r16;
            offB[j] = KP * RBA + g * RBB + (((c >> 2) ^ swz(g, RBB)) << 4) + (c & 3) * 4;
        }
    }

    const int nK = (mend - mbeg + KP - 1) / KP;
    issue(0);
    if (nK > 1) issue(1);
    int stage = 0;
    for (int kc = 0; kc < nK; ++kc) {
        if (kc + 1 < nK) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kc + 2 < nK) issue(stage >= 1 ? stage - 1 : 2);
        const char* sb = smem + stage * STAGE;
        if constexpr (ES == 2) {
            // transpose reads as inline asm with an explicit wait: behind the `buffer_load ... lds` of issue() the compiler puts
            // s_waitcnt vmcnt(0) in front of a builtin LDS read (the DMA might alias it) -- the step then waited for the stage it
            // had just issued (profiles/r02_ablations.txt section 12)
            auto rd_tr = [&](int addr, int off) -> u32x2_t {
                u32x2_t v;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
                return v;
            };
            const int sbo = (int)(uintptr_t)(__attribute__((address_space(3))) char*)smem + stage * STAGE;
            u32x2_t dl[FA], dh[FA], xl[FB], xh[FB];
#pragma unroll
            for (int i = 0; i < FA; ++i) { dl[i] = rd_tr(sbo + offA[i], 0); dh[i] = rd_tr(sbo + offA[i], 16 * RBA); }
#pragma unroll
            for (int j = 0; j < FB; ++j) { xl[j] = rd_tr(sbo + offB[j], 0); xh[j] = rd_tr(sbo + offB[j], 16 * RBB); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (volatile asm statements keep their order: every fragment register is redefined BEHIND the wait, so no use can move above it)
#pragma unroll
            for (int i = 0; i < FA; ++i) asm volatile("" : "+v"(dl[i]), "+v"(dh[i]));
#pragma unroll
            for (int j = 0; j < FB; ++j) asm volatile("" : "+v"(xl[j]), "+v"(xh[j]));
            u32x4_t df[FA], xf[FB];
#pragma unroll
            for (int i = 0; i < FA; ++i) df[i] = u32x4_t{dl[i].x, dl[i].y, dh[i].x, dh[i].y};
#pragma unroll
            for (int j = 0; j < FB; ++j) xf[j] = u32x4_t{xl[j].x, xl[j].y, xh[j].x, xh[j].y};
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int j = 0; j < FB; ++j)
                    acc[i][j] = mfma16<T>(df[i], xf[j], acc[i][j]);
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {       // rows 4s + g; the swizzle depends on (row & 1) == (g & 1) only
                float df[FA], xf[FB];
#pragma unroll
                for (int i = 0; i < FA; ++i) df[i] = *(const float*)(sb + offA[i] + 4 * s * RBA);
#pragma unroll
                for (int j = 0; j < FB; ++j) xf[j] = *(const float*)(sb + offB[j] + 4 * s * RBB);
#pragma unroll
                for (int i = 0; i < FA; ++i)
#pragma unroll
                    for (int j = 0; j < FB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(df[i], xf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (++stage == 3) stage = 0;
    }

    if (a.ablate) {
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < FB; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (keep == 123.456f) a.dw[0] = keep;
        return;
    }
    // ---- epilogue: D[co][ci] (lane: rows co = g*4+e, column ci = r16) staged through LDS in two co halves so that
    // every atomic / store instruction covers whole rows of the OHWI gradient (64 consecutive cins = 256 B per wave
    // instead of 4 x 64 B): the atomic tail was up to half of the kernel time on the 512-channel layers.
    constexpr int HR = CO_T / 2;                       // rows per half (wm selects the half)
    constexpr int PT = CI_T + 4;                       // tile pitch in floats
    static_assert(HR * PT * 4 <= 3 * STAGE, "staging tile must fit in the operand ring");
    float* tile = (float*)smem;
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                               // ring / previous half fully consumed
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < FA; ++i)
#pragma unroll
                for (int j = 0; j < FB; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        tile[(i * 16 + g * 4 + e) * PT + wn * FB * 16 + j * 16 + r16] = acc[i][j][e];
        }
        __syncthreads();
        for (int idx = tid; idx < HR * CI_T; idx += 256) {
            const int r = idx / CI_T, c = idx - r * CI_T;
            const int co = co0 + half * HR + r, ci = ci0 + c;
            if (co < a.Co && ci < a.Ci) {
                const long off = ((long)(co * a.KH + kh) * a.KW + kw) * a.Ci + ci;
                const float v = tile[r * PT + c];
                if (a.slab) a.slab[(long)split * ((long)a.Co * a.KH * a.KW * a.Ci) + off] = v;
                else if (a.plain_store) a.dw[off] = v;
                else atomicAdd(a.dw + off, v);
            }
        }
    }
#endif
}

// dw[i] (+)= sum_s slab[s][i].  A block owns 256 / SL groups of four consecutive elements; the SL "split lanes" of a group each add a
// contiguous share of the splits with four independent running sums and lane 0 combines the SL partial sums in ascending order --
// a fixed association for a given (nsplit, SL), so two runs agree bit for bit.  (Round 5: until then ONE thread added all the splits of its
// group one after the other -- conv1_1's fp32 weight gradient, 2,048 elements x 3,072 splits, took 1.03 ms in two blocks of dependent
// loads; profiles/r05_ablations.txt 9.)  Optionally also delivers the 16-bit wire image (szn_conv_desc_t.dw_lp).
template <int SL>
__global__ __launch_bounds__(256) void wgrad_slab_reduce(const float* __restrict__ slab, float* __restrict__ dw, long nw, int nsplit,
                                                         int accumulate) {
    __shared__ f32x4_t part[256];
    constexpr int GPB = 256 / SL;                     // groups per block
    const int gl = threadIdx.x % GPB, sl = threadIdx.x / GPB;
    const long n4 = nw >> 2;
    const int sp0 = (int)((long)nsplit * sl / SL), sp1 = (int)((long)nsplit * (sl + 1) / SL);
    for (long base = (long)blockIdx.x * GPB; base < n4; base += (long)gridDim.x * GPB) {
        const long i = base + gl;
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
        if (i < n4) {
            const f32x4_t* p = (const f32x4_t*)slab + i;
            int k = sp0;
            for (; k + 4 <= sp1; k += 4) {
                s0 += p[(long)k * n4]; s1 += p[(long)(k + 1) * n4]; s2 += p[(long)(k + 2) * n4]; s3 += p[(long)(k + 3) * n4];
            }
            for (; k < sp1; ++k) s0 += p[(long)k * n4];
        }
        part[threadIdx.x] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (sl == 0 && i < n4) {
            f32x4_t s = accumulate ? ((const f32x4_t*)dw)[i] : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < SL; ++r) s += part[r * GPB + gl];
            ((f32x4_t*)dw)[i] = s;
        }
        __syncthreads();
    }
    // (nw % 4 != 0 never reaches this kernel: the launcher requires whole 16-B groups for the slab form)
}

template <typename T, int FA, int FB>
void launch_wg2(const Wg2Args& a, long blocks, hipStream_t st) {
    hipLaunchKernelGGL((conv_wgrad_v2<T, FA, FB>), dim3((unsigned)blocks), dim3(256), 0, st, a);
}

}  // namespace

// Weight gradient + Adam in one launch (fc6 / fc7: the layers that take conv_wgrad_wide, 89 % of the weights).  The separate optimizer
// pass streams 30 B per weight (gradient, master, two moments in; master, moments, 16-bit image out) after the backward pass; here the
// gradient tile is still in LDS when the update is applied, so 4 B per weight are never written and never read back, and the other 26
// move while the other CUs are in their K loops.  Same arithmetic (adam_elem) on the same gradient values: bit-identical.
// fewest 256 x 256 tiles for which conv_wgrad_wide (and with it the fused Adam form) takes a layer
static int wgw_min_tiles() {
    static const int wide_min = szn_knob("SZN_WGW_MINTILES", 96);
    return wide_min;
}

extern "C" int szn_conv2d_wgrad_adam_supported(const szn_conv_desc_t* d) {
    if (!d || !szn_is16(d->dtype) || d->Co < 256 || d->Ci < 256 || (d->ldi & 7) || (d->ldo & 7) || (d->Ci & 7)) return 0;
    if (d->KH == 3 && d->KW == 3 && d->workspace) return 0;                  // the all-taps kernel takes these
    const long cot = szn_div_up(d->Co, 256), cit = szn_div_up(d->Ci, 256);
    const long tiles = cot * cit * d->KH * d->KW;
    if (tiles < wgw_min_tiles() || (long)d->B * d->Ho * d->Wo >= (1L << 22)) return 0;
    if (cot * 256 * cit * 256 > (long)d->Co * d->Ci * 5 / 4) return 0;
    return 1;
}

extern "C" int szn_conv2d_wgrad_adam(const szn_conv_desc_t* d, const void* in, const void* dout, float* dw,
                                     const szn_adam_args_t* opt, szn_stream_t stream) {
    if (!d || !opt) SZN_FAIL(SZN_ERR_ARG, "conv2d_wgrad_adam: null descriptor / optimizer arguments");
    const size_t in_bytes = (size_t)d->B * d->Hi * d->Wi * d->ldi * 2;
    const size_t dout_bytes = (size_t)d->B * d->Ho * d->Wo * d->ldo * 2;
    const bool ok = szn_is16(d->dtype) && in && dout && d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Ci > 0 && d->Co > 0 && d->KH > 0 &&
                    d->KW > 0 && d->pad >= 0 && d->Ho == d->Hi + 2 * d->pad - d->KH + 1 && d->Wo == d->Wi + 2 * d->pad - d->KW + 1 &&
                    d->Ho > 0 && d->Wo > 0 && in_bytes < 0x7fff0000ul && dout_bytes < 0x7fff0000ul &&
                    !(((uintptr_t)in | (uintptr_t)dout) & 15);
    if (!ok) SZN_FAIL(SZN_ERR_ARG, "conv2d_wgrad_adam: bad descriptor / operands");
    if (!opt->param || !opt->exp_avg || !opt->exp_avg_sq || opt->step < 1)
        SZN_FAIL(SZN_ERR_ARG, "conv2d_wgrad_adam: param / exp_avg / exp_avg_sq / step >= 1 are required");
    if (((uintptr_t)opt->param | (uintptr_t)opt->exp_avg | (uintptr_t)opt->exp_avg_sq | (uintptr_t)dw) & 15 || ((uintptr_t)opt->w_lp & 7))
        SZN_FAIL(SZN_ERR_ARG, "conv2d_wgrad_adam: master / moments / gradient must be 16-B aligned, the weight image 8-B");
    if (opt->w_lp && opt->w_lp_dtype != d->dtype) SZN_FAIL(SZN_ERR_ARG, "conv2d_wgrad_adam: the weight image must have the compute dtype");
    if (!szn_conv2d_wgrad_adam_supported(d))
        SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv2d_wgrad_adam: this layer does not take conv_wgrad_wide (use szn_conv2d_wgrad + szn_adam_step)");
    szn_note_colsum_rows(0);
    szn_note_work_fraction(1.f);
    const int rc = szn_conv_wgrad_wide_try(d, in, dout, dw, 0, wgw_min_tiles(), stream, opt);
    szn_publish_result(d);
    if (rc > 0) SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv2d_wgrad_adam: conv_wgrad_wide declined the layer");
    return rc;
}

static int wgrad_dispatch(const szn_conv_desc_t* d, const void* in, const void* dout, float* dw, int accumulate, szn_stream_t stream,
                          int* lp_native);

extern "C" int szn_conv2d_wgrad(const szn_conv_desc_t* d, const void* in, const void* dout, float* dw, int accumulate,
                                szn_stream_t stream) {
    if (!d) SZN_FAIL(SZN_ERR_ARG, "conv2d_wgrad: null descriptor");
    if (d->dw_lp && (accumulate || !szn_is16(d->dw_lp_dtype) || ((uintptr_t)d->dw_lp & 7) || !dw))
        SZN_FAIL(SZN_ERR_ARG, "conv2d_wgrad: dw_lp needs accumulate == 0, a 16-bit dw_lp_dtype, an 8-B aligned image and dw as fp32 scratch");
    int lp_native = 0;
    szn_note_colsum_rows(0);
    const int rc = wgrad_dispatch(d, in, dout, dw, accumulate, stream, &lp_native);
    szn_publish_result(d);
    if (rc != SZN_OK || !d->dw_lp || lp_native) return rc;
    // the kernel family that took the layer finishes in fp32 (head / skip layers, fp32 layers: a few MB): one conversion pass
    return szn_cast(SZN_F32, d->dw_lp_dtype, (long)d->Co * d->KH * d->KW * d->Ci, dw, d->dw_lp, stream);
}

static int wgrad_dispatch(const szn_conv_desc_t* d, const void* in, const void* dout, float* dw, int accumulate, szn_stream_t stream,
                          int* lp_native) {
    szn_note_work_fraction(1.f);
    const size_t es = szn_esize(d->dtype);
    const int ch = (int)(16 / es);
    const size_t in_bytes = (size_t)d->B * d->Hi * d->Wi * d->ldi * es;
    const size_t dout_bytes = (size_t)d->B * d->Ho * d->Wo * d->ldo * es;
    const bool ok = (szn_is16(d->dtype) || d->dtype == SZN_F32) && in && dout && dw && d->B > 0 && d->Hi > 0 && d->Wi > 0 &&
                    d->Ci > 0 && d->Co > 0 && d->KH > 0 && d->KW > 0 && d->pad >= 0 &&
                    d->Ho == d->Hi + 2 * d->pad - d->KH + 1 && d->Wo == d->Wi + 2 * d->pad - d->KW + 1 && d->Ho > 0 && d->Wo > 0 &&
                    (d->Ci % ch) == 0 && (d->ldi % ch) == 0 && (d->ldo % ch) == 0 && in_bytes < 0x7fff0000ul &&
                    dout_bytes < 0x7fff0000ul && (long)d->B * d->Ho * d->Wo < (1L << 31) &&
                    !(((uintptr_t)in | (uintptr_t)dout) & 15);
    if (!ok) return szn_conv2d_wgrad_v1(d, in, dout, dw, accumulate, stream);     // validates and reports / handles >= 2 GiB
    hipStream_t st = (hipStream_t)stream;
    // 3x3 bf16 layers with a workspace: all nine taps from one staged patch, deterministic slab reduction
    // (szn_conv_wgrad_taps.hip); SZN_WGT_MINTILES = fewest 16x16 tiles per block for which it is used
    if (d->KH == 3 && d->KW == 3 && d->workspace) {
        static const int taps_min = szn_knob("SZN_WGT_MINTILES", 8);
        const int rc = szn_conv_wgrad_taps_try(d, in, dout, dw, accumulate, taps_min, stream);
        if (rc <= 0) { *lp_native = 1; return rc; }
    }
    // many channels, few pixels (fc6, fc7): 256 x 256 tiles, one pixel split (szn_conv_wgrad_wide.hip)
    {
        const int rc = szn_conv_wgrad_wide_try(d, in, dout, dw, accumulate, wgw_min_tiles(), stream);
        if (rc <= 0) { *lp_native = 1; return rc; }
    }
    const long nw = (long)d->Co * d->KH * d->KW * d->Ci;
    Wg2Args a;
    a.dout = (const char*)dout; a.in = (const char*)in; a.dw = dw;
    a.dout_bytes = (unsigned)dout_bytes; a.in_bytes = (unsigned)in_bytes;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co;
    a.KH = d->KH; a.KW = d->KW; a.pad = d->pad; a.ldi = d->ldi; a.ldd = d->ldo;
    a.M = d->B * d->Ho * d->Wo;
    const int FA = d->Co <= 64 ? 2 : 4, FB = d->Ci <= 64 ? 2 : 4;
    a.cotiles = szn_div_up(d->Co, 32 * FA); a.citiles = szn_div_up(d->Ci, 32 * FB);
    const long tiles = (long)a.cotiles * a.citiles * d->KH * d->KW;
    // ~3 blocks per CU x 256 CUs x a few waves of blocks; each split covers a multiple of 64 pixels, at least 1024
    const int wg_target = 3072; /* (was SZN_WG_BLOCKS) */
    long want = (wg_target + tiles - 1) / tiles;
    if (want < 1) want = 1;
    long span = (a.M + want - 1) / want;
    if (span < 1024) span = 1024;
    span = (span + 63) / 64 * 64;
    if (tiles >= 512) span = (a.M + 63) / 64 * 64;       // enough tiles to fill the chip: one split, plain stores (fc6, fc7)
    a.nsplit = szn_div_up(a.M, span);
    // a workspace turns the split reduction into fixed-order slabs (deterministic): fewer splits if it is too small for all
    a.slab = nullptr;
    if (a.nsplit > 1 && d->workspace && !((uintptr_t)d->workspace & 15) && (nw & 3) == 0 && !((uintptr_t)dw & 15)) {
        const long fit = (long)(d->workspace_bytes / ((size_t)nw * sizeof(float)));
        if (fit >= 2) {
            if (fit < a.nsplit) { span = ((a.M + fit - 1) / fit + 63) / 64 * 64; a.nsplit = szn_div_up(a.M, span); }
            if (a.nsplit > 1) a.slab = (float*)d->workspace;
        }
    }
    a.kspan = (int)span;
    const long blocks = tiles * a.nsplit;
    if (blocks >= (1L << 31)) SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv2d_wgrad: grid too large");
    a.plain_store = (a.nsplit == 1 && !accumulate) ? 1 : 0;      // fc6: 411 MB written once instead of memset + atomics
    static int wg_abl = -1;
    if (wg_abl < 0) { wg_abl = szn_ablate_env("SZN_WG_ABLATE"); }
    a.ablate = wg_abl;
    if (!accumulate && !a.plain_store && !a.slab) {
        hipError_t e = hipMemsetAsync(dw, 0, nw * sizeof(float), st);
        if (e != hipSuccess) SZN_FAIL(SZN_ERR_LAUNCH, "conv2d_wgrad memset: %s", hipGetErrorString(e));
    }
    if (d->dtype == SZN_BF16) {
        if (FA == 4 && FB == 4) launch_wg2<bf16_raw, 4, 4>(a, blocks, st);
        else if (FA == 4) launch_wg2<bf16_raw, 4, 2>(a, blocks, st);
        else if (FB == 4) launch_wg2<bf16_raw, 2, 4>(a, blocks, st);
        else launch_wg2<bf16_raw, 2, 2>(a, blocks, st);
    } else if (d->dtype == SZN_F16) {
        if (FA == 4 && FB == 4) launch_wg2<f16_raw, 4, 4>(a, blocks, st);
        else if (FA == 4) launch_wg2<f16_raw, 4, 2>(a, blocks, st);
        else if (FB == 4) launch_wg2<f16_raw, 2, 4>(a, blocks, st);
        else launch_wg2<f16_raw, 2, 2>(a, blocks, st);
    } else {
        if (FA == 4 && FB == 4) launch_wg2<float, 4, 4>(a, blocks, st);
        else if (FA == 4) launch_wg2<float, 4, 2>(a, blocks, st);
        else if (FB == 4) launch_wg2<float, 2, 4>(a, blocks, st);
        else launch_wg2<float, 2, 2>(a, blocks, st);
    }
    SZN_CHECK_LAUNCH("conv_wgrad_v2");
    if (a.slab) {
        const long n4 = nw / 4;                        // (the slab form requires nw % 4 == 0, see above)
        // many splits of a small gradient: sixteen split lanes per group; else four
        if (a.nsplit >= 64 && n4 * 16 <= 4096L * 256)
            hipLaunchKernelGGL(wgrad_slab_reduce<16>, dim3((unsigned)std::min<long>((n4 + 15) / 16, 8192L)), dim3(256), 0, st,
                               (const float*)a.slab, dw, nw, a.nsplit, accumulate);
        else
            hipLaunchKernelGGL(wgrad_slab_reduce<4>, dim3((unsigned)std::min<long>((n4 + 63) / 64, 8192L)), dim3(256), 0, st,
                               (const float*)a.slab, dw, nw, a.nsplit, accumulate);
        SZN_CHECK_LAUNCH("wgrad_slab_reduce");
    }
    return SZN_OK;
}
