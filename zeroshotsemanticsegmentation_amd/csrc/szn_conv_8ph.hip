// szn_conv_8ph.hip -- the 256 x 256-tile forward / dgrad implicit GEMM on the "8-phase" schedule.
//
// Round 4: the CDNA guide's 256^2 8-phase GEMM template, rebuilt as a micro-bench (tools/gemm8, profiles/r04_gemm8_ab*.json), ran the
// conv4_2 GEMM at 1.33-1.40 PF on the box where conv3x3_wide_rows ran the layer at 1.18-1.21 PF (both on relu(randn) / randn operands),
// while STREAMING its 584 MB activation matrix from HBM.  What the template has that conv_igemm_wide / conv3x3_wide_rows (2-stage ring,
// s_waitcnt vmcnt(0) + one s_barrier per K = 64 step, 64 MFMA per wave between barriers, both waves of a SIMD in the same phase) have
// not: (1) the K step cut into 4 phases of 16 MFMA (one 64 x 32 C-quadrant x K = 64) with the phase's fragment reads in front;
// (2) the two wave groups one barrier apart, so that every SIMD has one wave multiplying while its partner reads fragments and
// issues LDS-DMA (worth +9 .. +30 % in the micro-bench); (3) one half-tile of prefetch per phase and a COUNTED s_waitcnt vmcnt(8):
// four half-tiles = 64 KiB per CU always in flight, vmcnt never 0 in the loop.  s_setprio measured nothing and is left out.
//
// This file is that schedule with the address generation of an implicit-GEMM convolution (any KH x KW, stride 1, NHWC, OHWI):
//   * tile 256 pixels x 256 couts, 512 threads = 8 waves as 2 (pixel groups of 128) x 4 (cout groups of 64); MFMA A operand =
//     weights, so lane (g, r16) holds couts 16 i + 4 g .. + 3 of pixel 16 j + r16 (register epilogue of szn_epilogue.h);
//   * LDS 128 KiB = 2 buffers x {P0, P1, W0, W1}: half-tiles of 128 rows x 128 B (K = 64).  P0 / P1 = the lower / upper 64 pixels of
//     each pixel group, W0 / W1 = the lower / upper 32 couts of each cout group -- every wave reads a half-tile in ONE phase;
//   * K tile t = (tap, cin chunk); phases (P0,W0) (P0,W1) (P1,W1) (P1,W0) read 12 / 4 / 8 / 0 ds_read_b128;
//     phase P1 stages W1 of t + 1, P2: P1 of t + 1, P3: P0 of t + 2, P4: W0 of t + 2; the half-tile issued in phase p is waited for
//     in phase p + 4 and read from p + 5 on; a slot is re-staged >= 2 phases after its last read;
//   * pixel rows are gathered per tap: per-lane offsets of the 4 rows a lane stages, re-validated when a stream enters a new tap
//     (padding = out-of-range offset = zeros); weights: one scalar offset per K tile;
//   * fragment reads are inline asm (a compiler-visible LDS read behind an LDS-DMA load gets s_waitcnt vmcnt(0));
//   * split-K ranges (fp32 slabs), bias / ReLU / gate / Dropout2d factor / column-sum epilogue as the other tile kernels.
#include "szn_common.h"
#include "szn_epilogue.h"
#include "szn_wide_args.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((address_space(3))) void* ldsptr_t;

namespace {

constexpr int SLOT = 16384;                           // half-tile: 128 rows x 128 B
constexpr int BUF = 4 * SLOT;                         // P0 | P1 | W0 | W1

template <int OFF> __device__ __forceinline__ void dsr(u32x4_t& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

template <typename T>
__global__ __launch_bounds__(512, 2) void conv_igemm_8ph(WideArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(T) == 2, "16-bit storage only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int g = lane >> 4, r16 = lane & 15;

    const int nwg = a.mtiles * a.ntiles;
    const int lid = xcd_remap_w(blockIdx.x, nwg);
    const int nt = a.nmajor ? lid / a.mtiles : lid % a.ntiles, mt = a.nmajor ? lid % a.mtiles : lid / a.ntiles;
    const int m0 = mt * 256, n0 = nt * 256;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);

    // ---- staging: a half-tile is 16 wave-instructions of 8 rows x 128 B; wave w issues instructions 2 w and 2 w + 1.
    // half-tile image row (2 w + i) 8 + (lane >> 3) of P_h = tile pixel (row >> 6) 128 + 64 h + (row & 63); of W_h = tile cout
    // (row >> 5) 64 + 32 h + (row & 31).  16-B chunk c of an LDS row holds source chunk c ^ (row & 7) (swizzle on the source side).
    const int sc = (lane & 7) ^ (lane >> 3);
    unsigned baseA[2][2], voffA[2][2], voffB[2][2];
    int ohw[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (2 * w + i) * 8 + (lane >> 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + (row >> 6) * 128 + h * 64 + (row & 63);
            if (m < a.M) {
                const int b = m / a.HoWo, r = m - b * a.HoWo;
                const int oh = r / a.Wo, ow = r - oh * a.Wo;
                const int ih0 = oh - a.pad, iw0 = ow - a.pad;
                ohw[h][i] = (ih0 << 16) | (iw0 & 0xffff);
                baseA[h][i] = (unsigned)((((long)(b * a.Hi + ih0) * a.Wi + iw0) * a.ldi + sc * 8) * 2);
            } else {
                ohw[h][i] = 0x7fff7fff;
                baseA[h][i] = 0;
            }
            const int n = n0 + (row >> 5) * 64 + h * 32 + (row & 31);
            voffB[h][i] = n < a.Co ? (unsigned)(((long)n * a.KH * a.KW * a.Ci + sc * 8) * 2) : kOOBx;
        }
    }
    const int cpt = a.Ci >> 6;                                    // cin chunks (K tiles) per tap
    const int kbeg = blockIdx.y * a.chunks_per_split;
    const int kend = min(a.KH * a.KW * cpt, kbeg + a.chunks_per_split);
    // the two pixel streams (P0, P1) walk the K tiles on their own: tile index, chunk within the tap, tap coordinates
    int akt[2], aic[2], akh[2], akw[2];
    auto set_tap = [&](int h) {
        const unsigned tapoff = (unsigned)((akh[h] * a.Wi + akw[h]) * a.ldi * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ih = (ohw[h][i] >> 16) + akh[h], iw = (int)(short)(ohw[h][i] & 0xffff) + akw[h];
            const bool ok = (unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi;
            voffA[h][i] = ok ? baseA[h][i] + tapoff : kOOBx;
        }
    };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int tap = kbeg / cpt;
        akt[h] = kbeg; aic[h] = kbeg - tap * cpt; akh[h] = tap / a.KW; akw[h] = tap - akh[h] * a.KW;
        set_tap(h);
    }
    auto stageA = [&](int h, int buf) {                           // h / buf are compile-time after unrolling
        const unsigned kill = akt[h] < kend ? 0u : kOOBx;          // beyond the K range: zeros into a slot nobody reads (uniform vmcnt)
        char* dst = smem + buf * BUF + h * SLOT + (2 * w) * 1024;
        const int soff = aic[h] * 128;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)dst, 16, voffA[h][0] | kill, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(dst + 1024), 16, voffA[h][1] | kill, soff, 0, 0);
        ++akt[h];
        if (++aic[h] == cpt) {
            aic[h] = 0;
            if (++akw[h] == a.KW) { akw[h] = 0; ++akh[h]; }
            set_tap(h);
        }
    };
    auto stageB = [&](int h, int buf, int kt) {                   // weights [Co][KH][KW][Ci]: K tile kt starts kt * 128 B into a row
        const unsigned kill = kt < kend ? 0u : kOOBx;
        char* dst = smem + buf * BUF + (2 + h) * SLOT + (2 * w) * 1024;
        const int soff = kt * 128;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)dst, 16, voffB[h][0] | kill, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)(dst + 1024), 16, voffB[h][1] | kill, soff, 0, 0);
    };

    // ---- fragment read addresses [buffer][K half]: row base + swizzled chunk (fragments are 16 rows apart: immediates)
    unsigned adA[2][2], adB[2][2];
    {
        const int rowA = wr * 64 + r16, rowB = wc * 32 + r16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            adA[0][s] = (unsigned)(rowA * 128 + (((4 * s + g) ^ (r16 & 7)) << 4));
            adB[0][s] = (unsigned)(2 * SLOT + rowB * 128 + (((4 * s + g) ^ (r16 & 7)) << 4));
            adA[1][s] = adA[0][s] + BUF;
            adB[1][s] = adB[0][s] + BUF;
        }
    }

    f32x4_t acc[4][2][4];                                         // [quadrant][cout fragment i][pixel fragment j]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[q][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    u32x4_t fa[4][2], fb0[2][2], fb1[2][2];                       // pixel sub-tile [j][s]; W0 / W1 sub-tiles [i][s]

    // ---- prologue: P0 W0 W1 P1 of the first tile, P0 W0 of the second (what phases -6 .. -1 of the steady state would have issued)
    stageA(0, 0); stageB(0, 0, kbeg); stageB(1, 0, kbeg); stageA(1, 0); stageA(0, 1); stageB(0, 1, kbeg + 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");              // P0, W0 of the first tile landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                                 // ... everyone's
    if (wr == 1) __builtin_amdgcn_s_barrier();                    // group 1 runs one barrier behind group 0

#define C8_READ_A(KIND, BUFI)                                                                                        \
    {                                                                                                                \
        constexpr int o_ = (KIND) * SLOT;                                                                            \
        dsr<o_ + 0 * 2048>(fa[0][0], adA[BUFI][0]); dsr<o_ + 1 * 2048>(fa[1][0], adA[BUFI][0]);                      \
        dsr<o_ + 2 * 2048>(fa[2][0], adA[BUFI][0]); dsr<o_ + 3 * 2048>(fa[3][0], adA[BUFI][0]);                      \
        dsr<o_ + 0 * 2048>(fa[0][1], adA[BUFI][1]); dsr<o_ + 1 * 2048>(fa[1][1], adA[BUFI][1]);                      \
        dsr<o_ + 2 * 2048>(fa[2][1], adA[BUFI][1]); dsr<o_ + 3 * 2048>(fa[3][1], adA[BUFI][1]);                      \
    }
#define C8_READ_B(FB, KIND, BUFI)                                                                                    \
    {                                                                                                                \
        constexpr int o_ = (KIND) * SLOT;                                                                            \
        dsr<o_ + 0>(FB[0][0], adB[BUFI][0]); dsr<o_ + 2048>(FB[1][0], adB[BUFI][0]);                                 \
        dsr<o_ + 0>(FB[0][1], adB[BUFI][1]); dsr<o_ + 2048>(FB[1][1], adB[BUFI][1]);                                 \
    }
#define C8_SYNC_AND_MMA(Q, FB)                                                                                       \
    {                                                                                                                \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                             \
        __builtin_amdgcn_s_barrier();                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
                     : "+v"(fa[0][0]), "+v"(fa[1][0]), "+v"(fa[2][0]), "+v"(fa[3][0]), "+v"(fa[0][1]), "+v"(fa[1][1]), \
                       "+v"(fa[2][1]), "+v"(fa[3][1]), "+v"(FB[0][0]), "+v"(FB[1][0]), "+v"(FB[0][1]), "+v"(FB[1][1])); \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
                    acc[Q][i][j] = mfma16<T>(FB[i][s], fa[j][s], acc[Q][i][j]);                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                \
    }
// one K tile from buffer BUFI (tile index t): four phases
#define C8_TILE(BUFI, t)                                                                                             \
    {                                                                                                                \
        C8_READ_B(fb0, 0, BUFI) C8_READ_A(0, BUFI) stageB(1, (BUFI) ^ 1, (t) + 1); C8_SYNC_AND_MMA(0, fb0)           \
        C8_READ_B(fb1, 1, BUFI) stageA(1, (BUFI) ^ 1); C8_SYNC_AND_MMA(1, fb1)                                       \
        C8_READ_A(1, BUFI) stageA(0, BUFI); C8_SYNC_AND_MMA(2, fb1)                                                  \
        stageB(0, BUFI, (t) + 2); C8_SYNC_AND_MMA(3, fb0)                                                            \
    }
    int t = kbeg;
    for (; t + 1 < kend; t += 2) {
        C8_TILE(0, t)
        C8_TILE(1, t + 1)
    }
    if (t < kend) C8_TILE(0, t)
#undef C8_TILE
#undef C8_SYNC_AND_MMA
#undef C8_READ_A
#undef C8_READ_B
    if (wr == 0) __builtin_amdgcn_s_barrier();                    // group 0 waits for group 1's last phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the dead prefetches of the tail (they write LDS)

    // ---- epilogue: quadrant q = (P mh, W nh) is the 64-pixel x 32-cout block at (m0 + 128 wr + 64 mh, n0 + 64 wc + 32 nh)
    const bool do_cs = a.colsum != nullptr && !a.ws;
    float* const pw = (float*)smem;                               // [4 pixel blocks (wr, mh)][8 cout blocks (wc, nh)][4 g][8]
    if (do_cs) __syncthreads();                                   // every wave's LDS-DMA has landed before the ring is reused
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mh = q >> 1, nh = (q == 1 || q == 2) ? 1 : 0;   // (P0,W0) (P0,W1) (P1,W1) (P1,W0)
        const int mb = m0 + wr * 128 + mh * 64, nb = n0 + wc * 64 + nh * 32;
        float* pws = do_cs ? pw + (((wr * 2 + mh) * 8 + (wc * 2 + nh)) * 4) * 8 : nullptr;
        if (a.ws) tile_epilogue_raw<2>(a, acc[q], 0, 0, g, r16, mb, nb, (int)blockIdx.y);
        else if (a.gate) {
            if (a.cscale) tile_epilogue_block<T, 2, true, true>(a, acc[q], g, r16, mb, nb, pws);
            else tile_epilogue_block<T, 2, true, false>(a, acc[q], g, r16, mb, nb, pws);
        } else {
            if (a.cscale) tile_epilogue_block<T, 2, false, true>(a, acc[q], g, r16, mb, nb, pws);
            else tile_epilogue_block<T, 2, false, false>(a, acc[q], g, r16, mb, nb, pws);
        }
    }
    if (do_cs) {
        __syncthreads();
        if (tid < 256 && n0 + tid < a.Co) {
            const int cb = tid >> 5, r2 = tid & 31;               // cout block (wc, nh), column within it
            const int gc = (r2 >> 4) | (((r2 >> 3) & 1) << 1), ec = r2 & 7;        // inverse of cl = 16 (g & 1) + 8 (g >> 1)
            float s = 0.f;
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) s += pw[((pb * 8 + cb) * 4 + gc) * 8 + ec];      // fixed order: bit-reproducible
            if (a.cslab) a.cslab[(long)(m0 >> 8) * a.Co + n0 + tid] = s;
            else if (s != 0.f) atomicAdd(a.colsum + n0 + tid, s);
        }
    }
#endif
}

template <typename T>
int launch_8ph(const WideArgs& a, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_8ph<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_igemm_8ph<T>), dim3(a.mtiles * a.ntiles, a.nsplit), dim3(512), 2 * BUF, st, a);
    SZN_CHECK_LAUNCH("conv_igemm_8ph");
    return SZN_OK;
}

}  // namespace

// Called by szn_conv_wide_try with a filled argument block (256-wide cout tiles, 16-bit operands).  Returns 1 when the shape does
// not fit: needs Ci a multiple of 64, 16-B aligned rows for the register epilogue (a.direct_ep) and map sides below 32768.
int szn_conv_8ph_launch(const void* args, int dtype, szn_stream_t stream) {
    const WideArgs& a = *(const WideArgs*)args;
    if (!szn_is16(dtype) || (a.Ci & 63) || (!a.direct_ep && !a.ws) || a.Hi >= 32768 || a.Wi >= 32768 || a.pad >= 16384) return 1;
    if (a.ws && (((uintptr_t)a.ws & 15) || (a.Co & 7))) return 1;
    return dtype == SZN_F16 ? launch_8ph<f16_raw>(a, (hipStream_t)stream) : launch_8ph<bf16_raw>(a, (hipStream_t)stream);
}
