// szn_conv_igemm.hip -- second-generation forward / dgrad implicit-GEMM convolution for gfx950 and the
// public szn_conv2d_fwd dispatcher.
//
// conv_igemm_v2: block = 512 threads = 8 waves (4 along pixels x 2 along couts), tile 256 pixels x BN couts
// (BN = 128, or 64 for the 64-channel layers), K advanced in 128-byte chunks of one filter tap.
//   * operands go HBM/L2 -> LDS directly with buffer_load_dwordx4 ... lds (no VGPR round trip); padding /
//     out-of-image taps / tile edges are out-of-range buffer offsets, which the hardware returns as zeros;
//   * the per-chunk part of every address (channel offset, tap offset of the weights) rides in the scalar
//     soffset operand, so a K chunk costs no vector address arithmetic; per-lane offsets change only when the
//     filter tap changes;
//   * the LDS image is [row][128 B] with the 16-B chunk index XOR-swizzled by (row & 7): LDS-DMA writes are
//     lane-linear, so the swizzle is applied to the per-lane SOURCE address (lane -> chunk = pos ^ (row & 7));
//   * 3-stage LDS ring, loads run two chunks ahead with counted s_waitcnt vmcnt(N) and one raw s_barrier per chunk;
//   * epilogue staged through LDS so that every store covers whole output rows in 16-B pieces (bias / ReLU / gate /
//     dropout factor / optional column sums = bias gradient of the producer layer);
//   * optional deterministic split-K (long K and a tile count that quantises badly against the CU count: fc6 forward /
//     dgrad, the projection head): every split writes an fp32 slab of the caller's workspace, splitk_epilogue adds the
//     slabs in a fixed order and applies the epilogue.
// The dispatcher tries the specialised kernels first (szn_conv_regw.hip: 64/128-channel 3x3 layers; szn_conv_wide.hip:
// >= 256 couts) and falls back to the register-staged kernel of szn_conv.hip for tensors >= 2 GiB / odd strides.
#include "szn_common.h"
#include "szn_epilogue.h"
#include "szn_wide_args.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((address_space(3))) void* ldsptr_t;

int szn_conv2d_fwd_v1(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias, const void* gate,
                      const float* chan_scale, void* out, szn_stream_t stream);
int szn_proj_stream_try(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias, const void* gate,
                        const float* chan_scale, void* out, int min_tiles, szn_stream_t stream);
int szn_conv_8ph_launch(const void* args, int dtype, int bn, szn_stream_t stream);      // szn_conv_8ph.hip (args = WideArgs)
int szn_conv_wide_try(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias, const void* gate,
                      const float* chan_scale, void* out, unsigned in_bytes, unsigned w_bytes, int min_tiles,
                      float* ws, int nsplit, int chunks_per_split, szn_stream_t stream);
int szn_conv_regw_try(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias, const void* gate,
                     const float* chan_scale, void* out, int min_tiles, szn_stream_t stream);
#include <stdlib.h>

namespace {

template <typename T> struct Mma2;
template <> struct Mma2<bf16_raw> {
    static __device__ __forceinline__ void run(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc,
                                                      0, 0, 0);
    }
};
template <> struct Mma2<f16_raw> {
    static __device__ __forceinline__ void run(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) { acc = mfma16<f16_raw>(a, b, acc); }
};
template <> struct Mma2<float> {
    static __device__ __forceinline__ void run(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

struct Conv2Args {
    const char* in; const char* w; const float* bias; const char* gate; const float* cscale; char* out;
    float* ws;                 // split-K accumulator [M][Co] (nsplit > 1)
    float* colsum;             // optional [Co]: += column sums of the stored tensor (bias gradient of the producer layer)
    float* cslab;              // optional [mtiles][Co]: per-tile partial rows instead of atomics on colsum (fixed-order reduce later)
    unsigned in_bytes, w_bytes;
    int B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, pad;
    int ldi, ldo, ldg, relu, out_f32;
    int M, HoWo, mtiles, ntiles, nsplit, chunks_per_split;
    int nmajor;                // tile order: 1 = pixel tile fastest (weights larger than activations)
    int stagger;               // 1: wave pairs take turns issuing a chunk's LDS-DMA loads (SZN_IGEMM_STAGGER=0: all at once)
    int direct_ep;             // 1: epilogue straight from the accumulator registers (szn_epilogue.h)
    int abl_ep;                // always 0 here (the accounting switch of the wide kernels)
};

__device__ __forceinline__ int xcd_remap2(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

constexpr unsigned kOOB = 0x80000000u;   // any offset >= num_records reads as zero

// ABL (debug ablation, env SZN_ABLATE, results are then WRONG): 1 = no LDS-DMA issue in the loop, 2 = no waits/barrier,
// 3 = fragments read once (no ds_read in the loop), 4 = no MFMA
template <typename T, int WNF, int ABL = 0>   // WNF = 16-cout fragments per wave: 4 -> BN = 128, 2 -> BN = 64
__global__ __launch_bounds__(512, 2) void conv_igemm_v2(Conv2Args a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource type does not exist in the host pass
    constexpr int ES = sizeof(T);
    constexpr int BKE = 128 / ES;                 // elements per 128-B K chunk
    constexpr int BM = 256, BN = 32 * WNF;
    constexpr int STAGE = (BM + BN) * 128;        // bytes per ring stage
    constexpr int NB = BN / 64;                   // weight-tile LDS-DMA instructions per wave per chunk (2 or 1)
    constexpr int LPC = 4 + NB;                   // loads per chunk per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [3][pixels BM x 128 B | weights BN x 128 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;            // wave -> 64 pixels x (BN/2) couts
    const int g = lane >> 4, r16 = lane & 15;

    const int nwg = a.mtiles * a.ntiles;
    const int lid = xcd_remap2(blockIdx.x, nwg);
    // consecutive ids (= one XCD) walk the cout tiles of a pixel tile, or -- when the filter bank is the larger operand
    // (fc6, fc7) -- the pixel tiles of a cout tile, so that the big operand is fetched by one L2 only
    const int nt = a.nmajor ? lid / a.mtiles : lid % a.ntiles, mt = a.nmajor ? lid % a.mtiles : lid / a.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int split = blockIdx.y;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);

    // ---- LDS-DMA assignment: instruction (w, i) fills pixel rows 32w + 8i .. +7; lane -> (row lane>>3, slot lane&7)
    const int chunkA = (lane & 7) ^ (lane >> 3);          // source 16-B chunk that lands in this lane's slot
    unsigned baseA[4];                                     // byte offset of (pixel, tap (0,0), chunkA), wraps mod 2^32
    int ohw[4];                                            // (oh - pad) << 16 | (ow - pad) & 0xffff ; invalid row: 0x7fff7fff
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 32 * w + 8 * i + (lane >> 3);
        if (m < a.M) {
            const int b = m / a.HoWo, r = m - b * a.HoWo;
            const int oh = r / a.Wo, ow = r - oh * a.Wo;
            const int ih0 = oh - a.pad, iw0 = ow - a.pad;
            ohw[i] = (ih0 << 16) | (iw0 & 0xffff);
            const long px = ((long)(b * a.Hi + ih0) * a.Wi + iw0);
            baseA[i] = (unsigned)((px * a.ldi + chunkA * (16 / ES)) * ES);
        } else {
            ohw[i] = 0x7fff7fff;
            baseA[i] = 0;
        }
    }
    unsigned voffB[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + (BN / 8) * w + 8 * i + (lane >> 3);     // BN=128: rows 16w+8i+..; BN=64: rows 8w+..
        voffB[i] = (n < a.Co) ? (unsigned)(((long)n * a.KH * a.KW * a.Ci + chunkA * (16 / ES)) * ES) : kOOB;
    }

    // issue-side iterator (runs two chunks ahead of the compute side)
    const int cpt = a.Ci / BKE;                            // chunks per tap
    const int kbeg = split * a.chunks_per_split;
    const int kend = min(a.KH * a.KW * cpt, kbeg + a.chunks_per_split);
    const int nK = kend - kbeg;
    int itap = kbeg / cpt, ic = kbeg - itap * cpt;         // tap index, chunk within tap
    unsigned voffA[4];
    auto set_tap = [&]() {
        const int kh = itap / a.KW, kw = itap - kh * a.KW;
        const unsigned tapoff = (unsigned)((kh * a.Wi + kw) * a.ldi * ES);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ih = (ohw[i] >> 16) + kh, iw = (int)(short)(ohw[i] & 0xffff) + kw;
            const bool ok = (unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi;
            voffA[i] = ok ? baseA[i] + tapoff : kOOB;
        }
    };
    auto issue = [&](int stage) {
        char* sb = smem + stage * STAGE;
        const int soffA = ic * 128;                                  // channel offset of this chunk (bytes)
        const int soffB = (itap * a.Ci) * ES + ic * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(sb + (32 * w + 8 * i) * 128), 16, voffA[i], soffA, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)(sb + BM * 128 + ((BN / 8) * w + 8 * i) * 128), 16, voffB[i],
                                                     soffB, 0, 0);
        if (++ic == cpt) { ic = 0; ++itap; set_tap(); }
    };

    f32x4_t acc[WNF][4];
#pragma unroll
    for (int i = 0; i < WNF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    set_tap();
    if (nK > 0) issue(0);
    if (nK > 1) issue(1);
    const int offs0 = ((g ^ (r16 & 7)) << 4), offs1 = (((4 + g) ^ (r16 & 7)) << 4);
    int stage = 0;
    if constexpr (ABL == 5) {
        // Software-pipelined loop: the LDS fragment reads of the NEXT half chunk are issued before the MFMAs of the
        // current one (two fragment register sets), also across the chunk boundary, so the LDS pipe and the matrix pipe
        // overlap instead of alternating; three chunks of LDS-DMA in flight; the per-chunk barrier sits between the two
        // MFMA halves, after this wave's reads of the stage that is recycled behind it have retired.
        if (nK > 2) issue(2);
        if (nK > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * LPC) : "memory");
        else if (nK > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        u32x4_t wfA[WNF], pfA[4], wfB[WNF], pfB[4];
        {
            const char* sp = smem + (wm * 64 + r16) * 128;
            const char* sw = smem + BM * 128 + (wn * (BN / 2) + r16) * 128;
#pragma unroll
            for (int i = 0; i < WNF; ++i) wfA[i] = *(const u32x4_t*)(sw + i * 16 * 128 + offs0);
#pragma unroll
            for (int j = 0; j < 4; ++j) pfA[j] = *(const u32x4_t*)(sp + j * 16 * 128 + offs0);
        }
        for (int kc = 0; kc < nK; ++kc) {
            const char* sp = smem + stage * STAGE + (wm * 64 + r16) * 128;
            const char* sw = smem + stage * STAGE + BM * 128 + (wn * (BN / 2) + r16) * 128;
            // ---- half 1: fetch the second-half fragments, multiply the first-half ones ----
#pragma unroll
            for (int i = 0; i < WNF; ++i) wfB[i] = *(const u32x4_t*)(sw + i * 16 * 128 + offs1);
#pragma unroll
            for (int j = 0; j < 4; ++j) pfB[j] = *(const u32x4_t*)(sp + j * 16 * 128 + offs1);
#pragma unroll
            for (int i = 0; i < WNF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma2<T>::run(acc[i][j], wfA[i], pfA[j]);
            // ---- chunk kc+1 visible to everyone; everyone done reading this stage ----
            if (kc + 1 < nK) {
                if (kc + 2 < nK) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPC) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kc + 3 < nK) issue(stage);                          // chunk kc+3 recycles the stage just drained
            const int nstage = (stage == 2) ? 0 : stage + 1;
            // ---- half 2: fetch the next chunk's first-half fragments, multiply the second-half ones ----
            if (kc + 1 < nK) {
                const char* sp2 = smem + nstage * STAGE + (wm * 64 + r16) * 128;
                const char* sw2 = smem + nstage * STAGE + BM * 128 + (wn * (BN / 2) + r16) * 128;
#pragma unroll
                for (int i = 0; i < WNF; ++i) wfA[i] = *(const u32x4_t*)(sw2 + i * 16 * 128 + offs0);
#pragma unroll
                for (int j = 0; j < 4; ++j) pfA[j] = *(const u32x4_t*)(sp2 + j * 16 * 128 + offs0);
            }
#pragma unroll
            for (int i = 0; i < WNF; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma2<T>::run(acc[i][j], wfB[i], pfB[j]);
            stage = nstage;
        }
    } else
    for (int kc = 0; kc < nK; ++kc) {
        // chunk kc has landed once at most the next chunk's loads are still outstanding
        if (ABL != 2) {
            if (kc + 1 < nK) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPC) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // (stage + 2) % 3 was last read in iteration kc - 1.  Wave pair k issues its LDS-DMA loads behind its k-th weight
        // fragment: all eight waves stalled at VMEM issue at once (a CU ingests ~64 B of LDS-DMA per clock) would idle the
        // MFMA pipe for most of a chunk's fill time
        const bool fill = ABL != 1 && kc + 2 < nK;
        const int turn = a.stagger ? (w >> 1) : 0;
        if (fill && turn == 0) issue(stage >= 1 ? stage - 1 : 2);
        const int rstage = (ABL == 3) ? 0 : stage;
        const char* sp = smem + rstage * STAGE + (wm * 64 + r16) * 128;
        const char* sw = smem + rstage * STAGE + BM * 128 + (wn * (BN / 2) + r16) * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int off = s ? offs1 : offs0;
            u32x4_t wf[WNF], pf[4];
#pragma unroll
            for (int i = 0; i < WNF; ++i) wf[i] = *(const u32x4_t*)(sw + i * 16 * 128 + off);
#pragma unroll
            for (int j = 0; j < 4; ++j) pf[j] = *(const u32x4_t*)(sp + j * 16 * 128 + off);
#pragma unroll
            for (int i = 0; i < WNF; ++i) {
                if (s * WNF + i > 0 && s * WNF + i < 4 && fill && turn == s * WNF + i) issue(stage >= 1 ? stage - 1 : 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ABL == 4) { acc[i][j][0] += __uint_as_float(wf[i].x ^ pf[j].x); }
                    else Mma2<T>::run(acc[i][j], wf[i], pf[j]);
                }
            }
        }
        if (++stage == 3) stage = 0;
    }

    // ---- epilogue: lane holds couts nb..nb+3 of pixel m ----
    if (a.nsplit > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + j * 16 + r16;
            if (m >= a.M) continue;
#pragma unroll
            for (int i = 0; i < WNF; ++i) {
                const int nb = n0 + wn * (BN / 2) + i * 16 + g * 4;
                if (nb >= a.Co) continue;
                float* o = a.ws + ((long)split * a.M + m) * a.Co + nb;      // slab of this split: plain stores
                if ((a.Co & 3) == 0) {
                    *(float4*)o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (nb + e < a.Co) o[e] = acc[i][j][e];
                }
            }
        }
        return;
    }
    if (ABL == 6) {                                   // ablation: no output stores
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < WNF; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (keep == 123.456f) ((float*)a.out)[0] = keep;
        return;
    }
    // ---- epilogue from registers (16-bit operands, aligned rows: szn_epilogue.h); the staged one below is the general path ----
    if constexpr (sizeof(T) == 2) {
        if (a.direct_ep) {
            if (a.gate) {
                if (a.cscale) tile_epilogue_direct<T, WNF, true, true>(a, acc, smem, tid, wm, wn, g, r16, m0, n0);
                else tile_epilogue_direct<T, WNF, true, false>(a, acc, smem, tid, wm, wn, g, r16, m0, n0);
            } else {
                if (a.cscale) tile_epilogue_direct<T, WNF, false, true>(a, acc, smem, tid, wm, wn, g, r16, m0, n0);
                else tile_epilogue_direct<T, WNF, false, false>(a, acc, smem, tid, wm, wn, g, r16, m0, n0);
            }
            return;
        }
    }
    // ---- epilogue staged through LDS: the MFMA layout gives each lane 4 couts of 16 different pixels (8-B pieces at a
    // Co-row stride: measured 2x the kernel time on the 710^2 / 355^2 layers); instead the fp32 tile goes to LDS
    // (pitch BN+4 floats: conflict-free float4 writes) and is written out as whole rows, 8 couts (16 B bf16) per lane,
    // with bias / ReLU / gate / dropout applied on the coalesced side (the gate is read in 16-B pieces too).
    constexpr int P = BN + 4;
    float* tile = (float*)smem;
    __syncthreads();                                   // every wave is done with the operand ring
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < WNF; ++i) {
            const int row = wm * 64 + j * 16 + r16, col = wn * (BN / 2) + i * 16 + g * 4;
            *(f32x4_t*)(tile + row * P + col) = acc[i][j];
        }
    __syncthreads();
    const T* __restrict__ gate = (const T*)a.gate;
    constexpr int CPR = BN / 8;                        // 8-cout chunks per tile row
    const bool out32 = a.out_f32 || sizeof(T) == 4;
    const int oes = out32 ? 4 : 2;
    const bool fast_o = (((long)a.ldo * oes) & 15) == 0;
    const bool fast_g = gate && ((((long)a.ldg * ES) & 15) == 0);
    // thread -> fixed 8-cout column group (512 % CPR == 0), rows tid/CPR + k * (512/CPR): bias is loaded once and the
    // fully unrolled row loop lets all gate loads go out before the first store
    const int cc = tid % CPR, row0 = tid / CPR;
    const int n = n0 + cc * 8;
    float cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = 0.f;
    if (n < a.Co) {
        const bool full = n + 8 <= a.Co;
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (a.bias && n + e < a.Co) ? a.bias[n + e] : 0.f;
        constexpr int NIT = 256 * CPR / 512;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int row = row0 + k * (512 / CPR);
            const int m = m0 + row;
            if (m < a.M) {
                const float* tp = tile + row * P + cc * 8;
                float v[8];
                *(f32x4_t*)&v[0] = *(const f32x4_t*)tp;
                *(f32x4_t*)&v[4] = *(const f32x4_t*)(tp + 4);
                float gv[8];
                if (gate) {
                    const T* gp = gate + (long)m * a.ldg + n;
                    if (full && fast_g) {
                        if constexpr (ES == 2) {
                            const u32x4_t q = *(const u32x4_t*)gp;
                            const T* qe = (const T*)&q;
#pragma unroll
                            for (int e = 0; e < 8; ++e) gv[e] = elem<T>::ld(qe + e);
                        } else {
                            *(f32x4_t*)&gv[0] = *(const f32x4_t*)gp;
                            *(f32x4_t*)&gv[4] = *(const f32x4_t*)((const float*)gp + 4);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) gv[e] = (n + e < a.Co) ? elem<T>::ld(gp + e) : 0.f;
                    }
                }
                const long brow = a.cscale ? (long)(m / a.HoWo) * a.Co : 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = v[e] + bv[e];
                    if (a.relu) x = fmaxf(x, 0.f);
                    if (gate) x = (gv[e] > 0.f) ? x : 0.f;
                    if (a.cscale && n + e < a.Co) x *= a.cscale[brow + n + e];
                    v[e] = x;
                    cs[e] += x;
                }
                if (out32) {
                    float* o = (float*)a.out + (long)m * a.ldo + n;
                    if (full && fast_o) {
                        *(f32x4_t*)o = *(const f32x4_t*)&v[0];
                        *(f32x4_t*)(o + 4) = *(const f32x4_t*)&v[4];
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (n + e < a.Co) o[e] = v[e];
                    }
                } else {
                    uint16_t* o = (uint16_t*)a.out + (long)m * a.ldo + n;
                    if (full && fast_o) {
                        u32x4_t pk;
                        pk.x = pack2<T>(v[0], v[1]);
                        pk.y = pack2<T>(v[2], v[3]);
                        pk.z = pack2<T>(v[4], v[5]);
                        pk.w = pack2<T>(v[6], v[7]);
                        *(u32x4_t*)o = pk;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (n + e < a.Co) o[e] = to_bits16<T>(v[e]);
                    }
                }
            }
        }
    }
    if (a.colsum) {                                    // bias gradient of the producer layer: column sums of this tile
        __syncthreads();                               // staging tile no longer needed
        float* red = (float*)smem;                     // [512 / CPR row groups][BN]
        if (n < a.Co) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[row0 * BN + cc * 8 + e] = cs[e];
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.Co) {
            float t = 0.f;
            for (int r = 0; r < 512 / CPR; ++r) t += red[r * BN + tid];
            if (a.cslab) a.cslab[(long)(m0 >> 8) * a.Co + n0 + tid] = t;
            else if (t != 0.f) atomicAdd(a.colsum + n0 + tid, t);
        }
    }
#endif
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_epilogue(const float* __restrict__ ws, Conv2Args a) {
    const long total = (long)a.M * a.Co;
    const T* __restrict__ gate = (const T*)a.gate;
    // four consecutive couts per thread (16-B slab loads) when the row pitches allow it; same additions in the same order
    // (... and the base pointers: bias / chan_scale / an fp32 out are accessed 16 B at a time, a 16-bit out 8 B at a time; a view at
    // an odd offset is legal per szn.h and takes the scalar loop)
    const bool out32 = a.out_f32 || sizeof(T) == 4;
    const bool v4 = !(a.Co & 3) && !(a.ldo & 3) && (!gate || !(a.ldg & 3)) && !((uintptr_t)ws & 15) &&
                    !((uintptr_t)a.bias & 15) && !((uintptr_t)a.cscale & 15) && !((uintptr_t)a.out & (out32 ? 15 : 7));
    if (v4) {
        const int c4n = a.Co >> 2;
        const long tot4 = total >> 2;
        for (long i4 = (long)blockIdx.x * 256 + threadIdx.x; i4 < tot4; i4 += (long)gridDim.x * 256) {
            const int n = (int)(i4 % c4n) * 4;
            const long m = i4 / c4n;
            f32x4_t x = {0.f, 0.f, 0.f, 0.f};
            for (int sp = 0; sp < a.nsplit; ++sp) x += *(const f32x4_t*)(ws + (long)sp * total + i4 * 4);
            if (a.bias) x += *(const f32x4_t*)(a.bias + n);
            if (a.relu) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); x[2] = fmaxf(x[2], 0.f); x[3] = fmaxf(x[3], 0.f); }
            if (gate) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = (elem<T>::ld(gate + m * a.ldg + n + e) > 0.f) ? x[e] : 0.f;
            }
            if (a.cscale) x *= *(const f32x4_t*)(a.cscale + (m / a.HoWo) * a.Co + n);
            if (a.out_f32 || sizeof(T) == 4) *(f32x4_t*)((float*)a.out + m * a.ldo + n) = x;
            else {
                uint2 o;
                o.x = pack2<T>(x[0], x[1]); o.y = pack2<T>(x[2], x[3]);
                *(uint2*)((uint16_t*)a.out + m * a.ldo + n) = o;
            }
        }
        return;
    }
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int n = (int)(idx % a.Co);
        const long m = idx / a.Co;
        float x = 0.f;
        for (int sp = 0; sp < a.nsplit; ++sp) x += ws[(long)sp * total + idx];     // fixed order: bit-reproducible
        if (a.bias) x += a.bias[n];
        if (a.relu) x = fmaxf(x, 0.f);
        if (gate) x = (elem<T>::ld(gate + m * a.ldg + n) > 0.f) ? x : 0.f;
        if (a.cscale) x *= a.cscale[(m / a.HoWo) * a.Co + n];
        if (a.out_f32 || sizeof(T) == 4) ((float*)a.out)[m * a.ldo + n] = x;
        else ((uint16_t*)a.out)[m * a.ldo + n] = to_bits16<T>(x);
    }
}

// The same epilogue with the column sums of the stored tensor (= the bias gradient of the layer that produced the gate): block b owns
// rows [b * rpb, (b + 1) * rpb), a thread owns one group of four couts and every (256 / (Co / 4))-th row of the block, so that its
// running sums stay in registers; the row lanes are combined through LDS in ascending order and the block's partial row goes to row b of
// the caller's slab (szn_colsum_reduce_batch adds the rows in a fixed order) or, without a slab, to colsum by fp32 atomics.  Until round 5
// a dgrad that wanted column sums could not split its K range (few-tile layers of a one-image step) or paid a separate pass over the
// 16-bit din (szn_bias_grad_slab); here the sums come from the fp32 values, like in the unsplit kernels' epilogues.  Same additions
// in the same order as splitk_epilogue for the tensor itself.  Needs Co % 4 == 0, Co / 4 <= 256, 256 % (Co / 4) == 0, the 16-B path.
template <typename T>
__global__ __launch_bounds__(256) void splitk_epilogue_cs(const float* __restrict__ ws, Conv2Args a, int rpb) {
    __shared__ f32x4_t part[256];
    const long total = (long)a.M * a.Co;
    const T* __restrict__ gate = (const T*)a.gate;
    const int c4n = a.Co >> 2, RL = 256 / c4n;
    const int c4 = threadIdx.x % c4n, rl = threadIdx.x / c4n, n = c4 * 4;
    const long m0 = (long)blockIdx.x * rpb, m1 = m0 + rpb < a.M ? m0 + rpb : a.M;
    f32x4_t cs = {0.f, 0.f, 0.f, 0.f};
    // four rows in flight per thread: all their slab / gate loads go out before the first is used (one row at a time, this kernel was a
    // chain of dependent round trips: 41 us for conv3_x's dgrad of a one-image step against ~10 us for the plain epilogue)
    auto finish = [&](long m, f32x4_t x, uint2 gq) {
        if (a.bias) x += *(const f32x4_t*)(a.bias + n);
        if (a.relu) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); x[2] = fmaxf(x[2], 0.f); x[3] = fmaxf(x[3], 0.f); }
        if (gate) {
            if (sizeof(T) == 2) {
                const uint16_t g0 = (uint16_t)(gq.x & 0xffffu), g1 = (uint16_t)(gq.x >> 16), g2 = (uint16_t)(gq.y & 0xffffu), g3 = (uint16_t)(gq.y >> 16);
                x[0] = from_bits16<T>(g0) > 0.f ? x[0] : 0.f; x[1] = from_bits16<T>(g1) > 0.f ? x[1] : 0.f;
                x[2] = from_bits16<T>(g2) > 0.f ? x[2] : 0.f; x[3] = from_bits16<T>(g3) > 0.f ? x[3] : 0.f;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = (elem<T>::ld(gate + m * a.ldg + n + e) > 0.f) ? x[e] : 0.f;
            }
        }
        if (a.cscale) x *= *(const f32x4_t*)(a.cscale + (m / a.HoWo) * a.Co + n);
        cs += x;
        if (a.out_f32 || sizeof(T) == 4) *(f32x4_t*)((float*)a.out + m * a.ldo + n) = x;
        else {
            uint2 o;
            o.x = pack2<T>(x[0], x[1]); o.y = pack2<T>(x[2], x[3]);
            *(uint2*)((uint16_t*)a.out + m * a.ldo + n) = o;
        }
    };
    long m = m0 + rl;
    for (; m + 3L * RL < m1; m += 4L * RL) {
        f32x4_t x[4];
        uint2 gq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i4 = (m + (long)u * RL) * c4n + c4;
            x[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            for (int sp = 0; sp < a.nsplit; ++sp) x[u] += *(const f32x4_t*)(ws + (long)sp * total + i4 * 4);
            gq[u] = uint2{0u, 0u};
            if (gate && sizeof(T) == 2) gq[u] = *(const uint2*)((const uint16_t*)gate + (m + (long)u * RL) * a.ldg + n);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) finish(m + (long)u * RL, x[u], gq[u]);
    }
    for (; m < m1; m += RL) {
        const long i4 = m * c4n + c4;
        f32x4_t x = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < a.nsplit; ++sp) x += *(const f32x4_t*)(ws + (long)sp * total + i4 * 4);
        uint2 gq = {0u, 0u};
        if (gate && sizeof(T) == 2) gq = *(const uint2*)((const uint16_t*)gate + m * a.ldg + n);
        finish(m, x, gq);
    }
    part[threadIdx.x] = cs;
    __syncthreads();
    if (rl == 0) {
        f32x4_t t = part[c4];
        for (int r = 1; r < RL; ++r) t += part[r * c4n + c4];
        if (a.cslab) *(f32x4_t*)(a.cslab + (long)blockIdx.x * a.Co + n) = t;
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (t[e] != 0.f) atomicAdd(a.colsum + n + e, t[e]);
        }
    }
}

template <typename T, int WNF, int ABL>
void launch_abl(const Conv2Args& a, size_t lds, hipStream_t st) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_v2<T, WNF, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((conv_igemm_v2<T, WNF, ABL>), dim3(a.mtiles * a.ntiles, a.nsplit), dim3(512), lds, st, a);
}

template <typename T, int WNF>
int launch_v2(const Conv2Args& a, hipStream_t st) {
    constexpr int BN = 32 * WNF;
    const size_t lds = 3 * (256 + BN) * 128;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_v2<T, WNF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    static int abl = -1;
    if (abl < 0) { abl = szn_ablate_env("SZN_ABLATE"); }
    if (abl && sizeof(T) == 2) {                      // debug ablations of the bf16 kernels (wrong results)
        if (abl == 1) launch_abl<T, WNF, 1>(a, lds, st);
        else if (abl == 2) launch_abl<T, WNF, 2>(a, lds, st);
        else if (abl == 3) launch_abl<T, WNF, 3>(a, lds, st);
        else if (abl == 5) launch_abl<T, WNF, 5>(a, lds, st);
        else if (abl == 6) launch_abl<T, WNF, 6>(a, lds, st);
        else launch_abl<T, WNF, 4>(a, lds, st);
        SZN_CHECK_LAUNCH("conv_igemm_v2(ablation)");
        return SZN_OK;
    }
    hipLaunchKernelGGL((conv_igemm_v2<T, WNF>), dim3(a.mtiles * a.ntiles, a.nsplit), dim3(512), lds, st, a);
    SZN_CHECK_LAUNCH("conv_igemm_v2");
    return SZN_OK;
}

}  // namespace

extern "C" int szn_maxpool2x2_ceil_fwd(int dtype, int B, int Hi, int Wi, int C, const void* in, void* out, szn_stream_t stream);
static int conv2d_fwd_dispatch(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias, const void* gate,
                               const float* chan_scale, void* out, szn_stream_t stream, int* pooled);

static int conv2d_fwd_entry(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias,
                            const void* gate, const float* chan_scale, void* out, szn_stream_t stream);
extern "C" int szn_conv2d_fwd(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias,
                              const void* gate, const float* chan_scale, void* out, szn_stream_t stream) {
    if (!d) SZN_FAIL(SZN_ERR_ARG, "conv2d_fwd: null descriptor");
    szn_note_colsum_rows(0);
    szn_note_work_fraction(1.f);
    const int rc = conv2d_fwd_entry(d, in, w, bias, gate, chan_scale, out, stream);
    szn_publish_result(d);           // (szn_conv_desc_t.result: what the kernel that ran decided -- no state survives the call)
    return rc;
}

static int conv2d_fwd_entry(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias,
                            const void* gate, const float* chan_scale, void* out, szn_stream_t stream) {
    if (d->colsum_slab && ((uintptr_t)d->colsum_slab & 15)) SZN_FAIL(SZN_ERR_ARG, "conv2d: colsum_slab must be 16-B aligned");
    if (d->pool_out && (!d->relu || d->ldo != d->Co || gate || chan_scale))
        SZN_FAIL(SZN_ERR_ARG, "conv2d_fwd: pool_out needs relu, ldo == Co and no gate / chan_scale");
    if ((d->pool_code || d->pool_only) && !d->pool_out) SZN_FAIL(SZN_ERR_ARG, "conv2d_fwd: pool_code / pool_only need pool_out");
    if (d->pool_code && (((uintptr_t)d->pool_code) & 7)) SZN_FAIL(SZN_ERR_ARG, "conv2d_fwd: pool_code must be 8-B aligned");
    int pooled = 0;
    const int rc = conv2d_fwd_dispatch(d, in, w, bias, gate, chan_scale, out, stream, &pooled);
    if (rc || !d->pool_out || pooled) return rc;
    // the kernel that ran has no fused pooling: pool the tensor it wrote (and write the winner codes when asked for)
    return szn_maxpool2x2_ceil_fwd_code((d->out_f32 || d->dtype == SZN_F32) ? SZN_F32 : d->dtype, d->B, d->Ho, d->Wo, d->Co, out,
                                        d->pool_out, d->pool_code, stream);
}

static int conv2d_fwd_dispatch(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias, const void* gate,
                               const float* chan_scale, void* out, szn_stream_t stream, int* pooled) {
    const size_t es = szn_esize(d->dtype);
    const size_t in_bytes = (size_t)d->B * d->Hi * d->Wi * d->ldi * es;
    const size_t w_bytes = (size_t)d->Co * d->KH * d->KW * d->Ci * es;
    const int bke = (int)(128 / es);
    const bool v2_ok = (szn_is16(d->dtype) || d->dtype == SZN_F32) && d->Ci > 0 && (d->Ci % bke) == 0 &&
                       in_bytes < 0x7fff0000ul && w_bytes < 0x7fff0000ul && d->Hi < 32000 && d->Wi < 32000 && d->pad < 16000 &&
                       ((size_t)d->ldi * es) % 16 == 0;
    {   // the pixel projection at full-resolution sizes (>= 256 pixel tiles x one 320-wide cout tile): HBM-streaming kernel
        static const int wide_min0 = szn_knob("SZN_WIDE_MINTILES", 240);
        const int rc = szn_proj_stream_try(d, in, w, bias, gate, chan_scale, out, wide_min0, stream);
        if (rc <= 0) return rc;
    }
    if (!v2_ok) return szn_conv2d_fwd_v1(d, in, w, bias, gate, chan_scale, out, stream);
    // argument validation is shared with the v1 path (same contract)
    if (d->B <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Co <= 0 || d->KH <= 0 || d->KW <= 0 || d->pad < 0 ||
        d->Ho != d->Hi + 2 * d->pad - d->KH + 1 || d->Wo != d->Wi + 2 * d->pad - d->KW + 1 || d->Ho <= 0 || d->Wo <= 0 ||
        d->ldi < d->Ci || d->ldo < d->Co || !in || !w || !out || (((uintptr_t)in | (uintptr_t)w | (uintptr_t)out) & 15) ||
        (long)d->B * d->Ho * d->Wo >= (1L << 31))
        return szn_conv2d_fwd_v1(d, in, w, bias, gate, chan_scale, out, stream);   // reports the precise error
    hipStream_t st = (hipStream_t)stream;
    // 64/128 -> 64/128 channel 3x3 layers (conv1_2, conv2_x forward / dgrad): register-resident filter bank,
    // szn_conv_regw.hip; SZN_REGW_MINTILES = fewest 256-pixel tiles for which it is used
    if (d->KH == 3 && d->KW == 3 && d->Ci <= 128 && d->Co <= 128) {
        static const int regw_min = szn_knob("SZN_REGW_MINTILES", 128);
        const int rc = szn_conv_regw_try(d, in, w, bias, gate, chan_scale, out, regw_min, stream);
        if (rc == 0 && d->pool_out) *pooled = 1;        // conv3x3_regw pools in its epilogue
        if (rc <= 0) return rc;
    }
    Conv2Args a;
    a.in = (const char*)in; a.w = (const char*)w; a.bias = bias; a.gate = (const char*)gate; a.cscale = chan_scale;
    a.out = (char*)out; a.ws = nullptr; a.colsum = d->colsum;
    a.in_bytes = (unsigned)in_bytes; a.w_bytes = (unsigned)w_bytes;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co;
    a.KH = d->KH; a.KW = d->KW; a.pad = d->pad; a.ldi = d->ldi; a.ldo = d->ldo; a.ldg = d->ldg;
    a.relu = d->relu; a.out_f32 = d->out_f32;
    a.M = d->B * d->Ho * d->Wo; a.HoWo = d->Ho * d->Wo;
    { const int stg = 1; /* (was SZN_IGEMM_STAGGER) */ a.stagger = stg; }
    {
        // epilogue from registers (szn_epilogue.h): whole 16-B pieces of 8 couts, so rows and bases have to be 16-B aligned
        static const int de = szn_knob("SZN_IGEMM_DIRECT", 1);
        const size_t oes = d->out_f32 ? 4 : 2;
        const uintptr_t al = (uintptr_t)out | (uintptr_t)gate | (uintptr_t)bias | (uintptr_t)chan_scale;
        a.direct_ep = de && szn_is16(d->dtype) && (d->Co % 8) == 0 && (((size_t)d->ldo * oes) & 15) == 0 && (al & 15) == 0 &&
                      (!gate || (((size_t)d->ldg * 2) & 15) == 0);
        a.abl_ep = 0;
    }
    { const int nm = 0; /* (was SZN_NMAJOR) */ a.nmajor = (nm && w_bytes > in_bytes) ? 1 : 0; }   // measured slower on fc6/fc7: off
    const bool narrow = d->Co <= 64;
    const int BN = narrow ? 64 : 128;
    a.mtiles = szn_div_up(a.M, 256); a.ntiles = szn_div_up(a.Co, BN);
    a.cslab = d->colsum ? d->colsum_slab : nullptr;
    if (a.cslab && d->colsum_slab_rows < a.mtiles) SZN_FAIL(SZN_ERR_ARG, "conv2d: colsum_slab holds %d rows, %d needed", d->colsum_slab_rows, a.mtiles);
    szn_note_colsum_rows(a.cslab ? a.mtiles : 0);
    const int nK = d->KH * d->KW * (d->Ci / bke);
    a.nsplit = 1; a.chunks_per_split = nK;
    const long tiles = (long)a.mtiles * a.ntiles;
    // split-K where K is long and the tile count quantises badly against the CU count (fc6 forward: 320 tiles = 1.25
    // waves; fc6 dgrad: 68 tiles x 3136 chunks): every split writes its own fp32 slab with plain stores, a second
    // kernel sums the slabs in a fixed order + epilogue.  ns minimises a simple time model: MFMA time / wave
    // efficiency + slab traffic.
    bool use_wide = false;                            // split-K on the 256 x 256 tile kernel (szn_conv_wide.hip)
    // column sums under split-K: splitk_epilogue_cs (its conditions are checked here; the slab must hold its row count)
    const bool out32_ = d->out_f32 || d->dtype == SZN_F32;
    const int cs_rpb = 32;
    const long cs_rows = ((long)a.M + cs_rpb - 1) / cs_rpb;
    const int cs_split = 1; /* (was SZN_SPLITK_COLSUM) */
    const bool cs_ok = d->colsum && cs_split && !(d->Co & 3) && (d->Co >> 2) <= 256 && 256 % (d->Co >> 2) == 0 && !(d->ldo & 3) &&
                       (!gate || !(d->ldg & 3)) && !((uintptr_t)d->workspace & 15) && !((uintptr_t)bias & 15) && !((uintptr_t)chan_scale & 15) &&
                       !((uintptr_t)out & (out32_ ? 15 : 7)) && (!d->colsum_slab || d->colsum_slab_rows >= cs_rows) &&
                       (long)a.mtiles * a.ntiles < 128;
    if (d->workspace && nK >= 64 && (!d->colsum || cs_ok)) {
        static int ncu = 0;
        if (!ncu) {
            int dev = 0; hipDeviceProp_t p;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ncu = p.multiProcessorCount;
            if (ncu <= 0) ncu = 256;
        }
        const double flops = 2.0 * a.M * (double)d->Co * d->Ci * d->KH * d->KW;
        const double slab = (double)a.M * d->Co * 8.0;                  // fp32 write + read per split
        // two candidate tilings: 256 x 128 (this file, ~0.95 PF sustained) and 256 x 256 (wide, ~1.15 PF, bf16 / f32 with
        // >= 256 couts and <= 64 wasted columns)
        const long wtiles = (long)a.mtiles * szn_div_up(d->Co, 256);
        const bool wide_ok = d->Co >= 256 && (long)szn_div_up(d->Co, 256) * 256 - d->Co <= 64;
        long best = 1; double best_t = 0; bool best_wide = false;
        for (int cand = 0; cand < (wide_ok ? 2 : 1); ++cand) {
            const long tl = cand ? wtiles : tiles;
            const double rate = cand ? 1.15e15 : 0.95e15;
            for (long ns = 1; ns <= 16 && ns <= nK / 8; ++ns) {
                if (ns > 1 && (size_t)ns * a.M * a.Co * sizeof(float) > d->workspace_bytes) break;
                if (cand && ns == 1) continue;              // unsplit wide tiles are handled below (SZN_WIDE_MINTILES)
                const long blocks = tl * ns;
                const double eff = (double)blocks / (double)((blocks + ncu - 1) / ncu * ncu);
                const double t = flops / (rate * eff) + (ns > 1 ? ns * slab / 3.0e12 + 5e-6 : 0.0);
                if ((cand == 0 && ns == 1) || t < best_t * 0.97) { best = ns; best_t = t; best_wide = cand != 0; }
            }
        }
        const int force_ns = 0; /* (was SZN_SPLITK_NS) */                   // tuning knob: SZN_SPLITK_NS=n forces the split count (0 = model)

        if (force_ns > 0 && force_ns <= nK / 8 && (size_t)force_ns * a.M * a.Co * sizeof(float) <= d->workspace_bytes) best = force_ns;
        if (best > 1) {
            a.chunks_per_split = (int)((nK + best - 1) / best);
            a.nsplit = szn_div_up(nK, a.chunks_per_split);
            a.ws = (float*)d->workspace;
            use_wide = best_wide;
        }
    }
    int rc = 1;
    // >= 256 couts and enough tiles to fill the chip: 256 x 256 tiles (1.5x less LDS fill per FLOP), szn_conv_wide.hip
    if (a.nsplit == 1 || use_wide) {
        static const int wide_min = szn_knob("SZN_WIDE_MINTILES", 240);
        rc = szn_conv_wide_try(d, in, w, bias, gate, chan_scale, out, a.in_bytes, a.w_bytes, use_wide ? 1 : wide_min, a.ws,
                               a.nsplit, a.chunks_per_split, stream);
        if (rc < 0 || (rc == 0 && a.nsplit == 1)) return rc;
    }
    // (conv_igemm_8ph<T, 1, 1> -- the 8-phase schedule on this 256 x 128 tile -- was built in round 4, measured faster alone and slower inside the
    //  step twice (profiles/r04_ablations.txt 10, r05_ablations.txt 4 and 24) and removed in round 6)
    if (rc != 0) {                                    // (rc == 0: the wide / 8-phase kernel ran, or wrote the slabs)
        if (d->dtype == SZN_BF16) rc = narrow ? launch_v2<bf16_raw, 2>(a, st) : launch_v2<bf16_raw, 4>(a, st);
        else if (d->dtype == SZN_F16) rc = narrow ? launch_v2<f16_raw, 2>(a, st) : launch_v2<f16_raw, 4>(a, st);
        else rc = narrow ? launch_v2<float, 2>(a, st) : launch_v2<float, 4>(a, st);
        if (rc) return rc;
    }
    if (a.nsplit > 1 && a.colsum) {
        a.cslab = d->colsum_slab;
        szn_note_colsum_rows(a.cslab ? (int)cs_rows : 0);
        if (d->dtype == SZN_BF16)
            hipLaunchKernelGGL(splitk_epilogue_cs<bf16_raw>, dim3((unsigned)cs_rows), dim3(256), 0, st, (const float*)a.ws, a, cs_rpb);
        else if (d->dtype == SZN_F16)
            hipLaunchKernelGGL(splitk_epilogue_cs<f16_raw>, dim3((unsigned)cs_rows), dim3(256), 0, st, (const float*)a.ws, a, cs_rpb);
        else
            hipLaunchKernelGGL(splitk_epilogue_cs<float>, dim3((unsigned)cs_rows), dim3(256), 0, st, (const float*)a.ws, a, cs_rpb);
        SZN_CHECK_LAUNCH("splitk_epilogue_cs");
        return SZN_OK;
    }
    if (a.nsplit > 1) {
        long blocks = ((long)a.M * a.Co / 4 + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        if (blocks < 1) blocks = 1;
        if (d->dtype == SZN_BF16)
            hipLaunchKernelGGL(splitk_epilogue<bf16_raw>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)a.ws, a);
        else if (d->dtype == SZN_F16)
            hipLaunchKernelGGL(splitk_epilogue<f16_raw>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)a.ws, a);
        else
            hipLaunchKernelGGL(splitk_epilogue<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)a.ws, a);
        SZN_CHECK_LAUNCH("splitk_epilogue");
    }
    return SZN_OK;
}
