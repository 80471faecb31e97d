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
//   * [MODE 0, the template's form; the shipped default is MODE 2, see the kernel's comment: two phases of 32 MFMA per K tile]
//     K tile t = (tap, cin chunk); phases (P0,W0) (P0,W1) (P1,W1) (P1,W0) read 12 / 4 / 8 / 0 ds_read_b128;
//     phase P1 stages W1 of t + 1, P2: P1 of t + 1, P3: P0 of t + 2, P4: W0 of t + 2; the half-tile issued in phase p is waited for
//     in phase p + 4 and read from p + 5 on; a slot is re-staged >= 2 phases after its last read;
//   * pixel rows are gathered per tap: per-lane offsets of the 4 rows a lane stages, re-validated when a stream enters a new tap
//     (padding = out-of-range offset = zeros); weights: one scalar offset per K tile;
//   * fragment reads are inline asm (a compiler-visible LDS read behind an LDS-DMA load gets s_waitcnt vmcnt(0));
//   * split-K ranges (fp32 slabs), bias / ReLU / gate / Dropout2d factor / column-sum epilogue as the other tile kernels;
//   * <NF0, NF1> = 16-cout fragments of a wave's W0 / W1 sub-tile: <2, 2> is the 256-cout tile, <1, 1> a 128-cout tile (conv5_x; one-image steps);
//   * 1x1 / pad 0 maps are addressed flat, the pixel resource rebased per block (no 2 GiB operand limit).
#include "szn_common.h"
#include "szn_epilogue.h"
#include "szn_wide_args.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((address_space(3))) void* ldsptr_t;

namespace {

constexpr int SLOT = 16384;                           // pixel half-tile: 128 rows x 128 B (W half-tiles: 64 NF rows)

template <int OFF> __device__ __forceinline__ void dsr(u32x4_t& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

template <int N> __device__ __forceinline__ void tie_frags(u32x4_t (&f)[N][2]) {      // uses of f stay behind this point
    static_assert(N >= 1 && N <= 4, "1 .. 4 fragments");
    asm volatile("" : "+v"(f[0][0]), "+v"(f[0][1]));
    if constexpr (N >= 2) asm volatile("" : "+v"(f[1][0]), "+v"(f[1][1]));
    if constexpr (N >= 3) asm volatile("" : "+v"(f[2][0]), "+v"(f[2][1]));
    if constexpr (N >= 4) asm volatile("" : "+v"(f[3][0]), "+v"(f[3][1]));
}

// one cout fragment without a partner (the fifth of a 320-wide tile): lane (g, r16) holds couts nb + 4 g .. + 3 of pixel mb + 16 j + r16
template <typename T, typename Args>
__device__ __forceinline__ void tile_epilogue_single(const Args& a, f32x4_t (&acc)[4], int g, int r16, int mb, int nb) {
    const int n = nb + 4 * g;
    if (n >= a.Co) return;                                        // Co % 4 == 0 (checked by the launcher)
    f32x4_t bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bv = *(const f32x4_t*)(a.bias + n);
    const float lo = a.relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = mb + 16 * j + r16;
        if (m >= a.M || a.abl_ep) continue;
        f32x4_t v = acc[j] + bv;
        v[0] = fmaxf(v[0], lo); v[1] = fmaxf(v[1], lo); v[2] = fmaxf(v[2], lo); v[3] = fmaxf(v[3], lo);
        if (a.out_f32) *(f32x4_t*)((float*)a.out + (long)m * a.ldo + n) = v;
        else {
            uint2 o;
            o.x = pack2<T>(v[0], v[1]); o.y = pack2<T>(v[2], v[3]);
            *(uint2*)((uint16_t*)a.out + (long)m * a.ldo + n) = o;
        }
    }
}

// SPLIT: the second LDS-DMA instruction of a phase's half-tile is issued from the MIDDLE of the phase's MFMA block instead of the read
// segment (an LDS-DMA instruction costs its wave 100-185 cycles beside fragment reads, ~60 among MFMAs, and the read segments are the
// longer ones: profiles/r04_ablations.txt section 6)
// MODE 2 (LONG): two phases of 32 MFMA per K tile instead of four of 16 -- half the barriers.  Phase A = (P0,W0) + (P0,W1): 16 fragment
// reads, phase B = (P1,W1) + (P1,W0): 8; four LDS-DMA loads per phase.  The fragment reads are retired (s_waitcnt lgkmcnt(0)) BEFORE the
// phase's first barrier, so a slot may be re-staged in the next phase: A stages W1, P1 of tile t + 1, B stages P0, W0 of t + 2; a unit
// issued in phase p is waited for in p + 1 (vmcnt(4)) and read in p + 2.
template <typename T, int NF0, int NF1, int MODE, bool KORD = false>
__global__ __launch_bounds__(512, 2) void conv_igemm_8ph(WideArgs a) {
    constexpr bool SPLIT = MODE == 1, LONG = MODE == 2;
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(T) == 2, "16-bit storage only");
    static_assert(NF0 >= NF1 && NF1 >= 1 && NF1 <= 2 && NF0 <= 3, "W0 = 1 .. 3 fragments per wave, W1 = 1 or 2");
    constexpr int NFA = NF0 + NF1, BN = 64 * NFA;                 // couts per wave = 16 NFA, per tile 256 / 320
    constexpr int W0ROWS = 64 * NF0, W1ROWS = 64 * NF1;           // half-tile image rows (4 cout groups x 16 NF)
    constexpr int OFF_W0 = 2 * SLOT, OFF_W1 = OFF_W0 + W0ROWS * 128, BUF = OFF_W1 + W1ROWS * 128;     // 64 / 72 KiB per buffer
    constexpr int VMC = 4 + NF0 + NF1;                            // loads of four consecutive phases: P (2) + P (2) + W0 + W1
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int g = lane >> 4, r16 = lane & 15;

    const int nwg = a.mtiles * a.ntiles;
    const int lid = xcd_remap_w(blockIdx.x, nwg);
    const int nt = a.nmajor ? lid / a.mtiles : lid % a.ntiles, mt = a.nmajor ? lid % a.mtiles : lid / a.ntiles;
    const int m0 = mt * 256, n0 = nt * BN;

    // 1x1 / pad 0 (the pixel projection, fc7): output pixel m reads input pixel m, so the pixel resource is rebased to this block's
    // 256 rows -- no 2 GiB limit on the activation matrix (the full-resolution projection is 2.1 GB)
    const bool flat = a.KH == 1 && a.KW == 1 && a.pad == 0;
    const int rows = min(256, a.M - m0);
    const auto rsA = flat ? __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)m0 * a.ldi * 2), 0, (int)((size_t)rows * a.ldi * 2), 0x00020000)
                          : __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);

    // ---- staging: a half-tile is 16 wave-instructions of 8 rows x 128 B; wave w issues instructions 2 w and 2 w + 1.
    // half-tile image row (2 w + i) 8 + (lane >> 3) of P_h = tile pixel (row >> 6) 128 + 64 h + (row & 63); W_h (NF_h instructions per
    // wave): image row r = (NF_h w + i) 8 + (lane >> 3) = tile cout (r / 16 NF_h) 16 NFA + (h ? 16 NF0 : 0) + r % (16 NF_h).
    // 16-B chunk c of an LDS row holds source chunk c ^ (row & 7) (swizzle on the source side).
    const int sc = (lane & 7) ^ (lane >> 3);
    unsigned baseA[2][2], voffA[2][2], voffB0[NF0], voffB1[NF1];
    int ohw[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (2 * w + i) * 8 + (lane >> 3);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + (row >> 6) * 128 + h * 64 + (row & 63);
            if (flat) {
                ohw[h][i] = m < a.M ? 0 : 0x7fff7fff;
                baseA[h][i] = (unsigned)(((long)(m - m0) * a.ldi + sc * 8) * 2);
            } else if (m < a.M) {
                const int b = m / a.HoWo, r = m - b * a.HoWo;
                const int oh = r / a.Wo, ow = r - oh * a.Wo;
                const int ih0 = oh - a.pad, iw0 = ow - a.pad;
                ohw[h][i] = (ih0 << 16) | (iw0 & 0xffff);
                baseA[h][i] = (unsigned)((((long)(b * a.Hi + ih0) * a.Wi + iw0) * a.ldi + sc * 8) * 2);
            } else {
                ohw[h][i] = 0x7fff7fff;
                baseA[h][i] = 0;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NF0; ++i) {
        const int r = (NF0 * w + i) * 8 + (lane >> 3);
        const int n = n0 + (r / (16 * NF0)) * (16 * NFA) + r % (16 * NF0);
        voffB0[i] = n < a.Co ? (unsigned)(((long)n * a.KH * a.KW * a.Ci + sc * 8) * 2) : kOOBx;
    }
#pragma unroll
    for (int i = 0; i < NF1; ++i) {
        const int r = (NF1 * w + i) * 8 + (lane >> 3);
        const int n = n0 + (r / (16 * NF1)) * (16 * NFA) + 16 * NF0 + r % (16 * NF1);
        voffB1[i] = n < a.Co ? (unsigned)(((long)n * a.KH * a.KW * a.Ci + sc * 8) * 2) : kOOBx;
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
    // KORD: K tiles in cin-chunk-major order (all taps of a 64-channel chunk, then the next chunk) instead of tap-major: the three kh
    // passes over a tile's pixel rows then follow each other within 1 / (Ci / 64) of the tile's time, while the rows are still in the
    // XCD's L2 (tap-major: a third of the tile's time apart, behind 4 MB of other tiles' patches).  Different summation order.
    const int ntap = a.KH * a.KW;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int tap = KORD ? kbeg % ntap : kbeg / cpt;
        akt[h] = kbeg; aic[h] = KORD ? kbeg / ntap : kbeg - tap * cpt; akh[h] = tap / a.KW; akw[h] = tap - akh[h] * a.KW;
        set_tap(h);
    }
    int wch[2], wtp[2];                                             // KORD: (chunk, tap) of the next K tile of the W0 / W1 streams
    wch[0] = wch[1] = kbeg / ntap; wtp[0] = wtp[1] = kbeg % ntap;
    auto stageA = [&](int h, int buf, int part) {                 // h / buf / part are compile-time after unrolling; part 2 = both loads
        const unsigned kill = akt[h] < kend ? 0u : kOOBx;          // beyond the K range: zeros into a slot nobody reads (uniform vmcnt)
        char* dst = smem + buf * BUF + h * SLOT + (2 * w) * 1024;
        const int soff = aic[h] * 128;
        if (part != 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)dst, 16, voffA[h][0] | kill, soff, 0, 0);
        if (part == 0) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(dst + 1024), 16, voffA[h][1] | kill, soff, 0, 0);
        ++akt[h];
        if (KORD) {
            if (++akw[h] == a.KW) { akw[h] = 0; if (++akh[h] == a.KH) { akh[h] = 0; ++aic[h]; } }
            set_tap(h);
        } else if (++aic[h] == cpt) {
            aic[h] = 0;
            if (++akw[h] == a.KW) { akw[h] = 0; ++akh[h]; }
            set_tap(h);
        }
    };
    auto stageB = [&](int h, int buf, int kt, int part) {         // weights [Co][KH][KW][Ci]: K tile kt starts kt * 128 B into a row
        const unsigned kill = kt < kend ? 0u : kOOBx;
        const int soff = KORD ? (wtp[h] * cpt + wch[h]) * 128 : kt * 128;
        if (KORD && part != 0) { if (++wtp[h] == ntap) { wtp[h] = 0; ++wch[h]; } }       // (the stream's last piece of this K tile)
        if (h == 0) {
            char* dst = smem + buf * BUF + OFF_W0 + (NF0 * w) * 1024;
#pragma unroll
            for (int i = 0; i < NF0; ++i)
                if (part == 2 || (part == 0) == (i == 0))
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)(dst + i * 1024), 16, voffB0[i] | kill, soff, 0, 0);
        } else {
            char* dst = smem + buf * BUF + OFF_W1 + (NF1 * w) * 1024;
#pragma unroll
            for (int i = 0; i < NF1; ++i)
                if (part == 2 || (part == 0) == (i == 0))
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)(dst + i * 1024), 16, voffB1[i] | kill, soff, 0, 0);
        }
    };

    // ---- fragment read addresses [buffer][K half]: row base + swizzled chunk (fragments are 16 rows apart: immediates)
    unsigned adA[2][2], adB0[2][2], adB1[2][2];
    {
        const int rowA = wr * 64 + r16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int sw = (((4 * s + g) ^ (r16 & 7)) << 4);
            adA[0][s] = (unsigned)(rowA * 128 + sw);
            adB0[0][s] = (unsigned)(OFF_W0 + (wc * 16 * NF0 + r16) * 128 + sw);
            adB1[0][s] = (unsigned)(OFF_W1 + (wc * 16 * NF1 + r16) * 128 + sw);
            adA[1][s] = adA[0][s] + BUF;
            adB0[1][s] = adB0[0][s] + BUF;
            adB1[1][s] = adB1[0][s] + BUF;
        }
    }

    // quadrants (P0,W0) (P0,W1) (P1,W1) (P1,W0): [cout fragment i][pixel fragment j]
    f32x4_t acc00[NF0][4], acc01[NF1][4], acc11[NF1][4], acc10[NF0][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < NF0; ++i) acc00[i][j] = acc10[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NF1; ++i) acc01[i][j] = acc11[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    u32x4_t fa[4][2], fb0[NF0][2], fb1[NF1][2];                   // pixel sub-tile [j][s]; W0 / W1 sub-tiles [i][s]

    // ---- prologue: P0 W0 W1 P1 of the first tile, P0 W0 of the second (what phases -6 .. -1 of the steady state would have issued)
    stageA(0, 0, 2); stageB(0, 0, kbeg, 2); stageB(1, 0, kbeg, 2); stageA(1, 0, 2); stageA(0, 1, 2); stageB(0, 1, kbeg + 1, 2);
    // short phases: P0, W0 of the first tile landed (this wave's pieces); long phases: the whole first tile
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LONG ? 2 + NF0 : VMC) : "memory");
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
#define C8_READ_B0(BUFI)                                                                                             \
    {                                                                                                                \
        dsr<0>(fb0[0][0], adB0[BUFI][0]);                                                                            \
        if constexpr (NF0 >= 2) dsr<2048>(fb0[NF0 >= 2 ? 1 : 0][0], adB0[BUFI][0]);                                  \
        if constexpr (NF0 == 3) dsr<4096>(fb0[NF0 - 1][0], adB0[BUFI][0]);                                           \
        dsr<0>(fb0[0][1], adB0[BUFI][1]);                                                                            \
        if constexpr (NF0 >= 2) dsr<2048>(fb0[NF0 >= 2 ? 1 : 0][1], adB0[BUFI][1]);                                  \
        if constexpr (NF0 == 3) dsr<4096>(fb0[NF0 - 1][1], adB0[BUFI][1]);                                           \
    }
#define C8_READ_B1(BUFI)                                                                                             \
    {                                                                                                                \
        dsr<0>(fb1[0][0], adB1[BUFI][0]);                                                                            \
        if constexpr (NF1 >= 2) dsr<2048>(fb1[NF1 >= 2 ? 1 : 0][0], adB1[BUFI][0]);                                  \
        dsr<0>(fb1[0][1], adB1[BUFI][1]);                                                                            \
        if constexpr (NF1 >= 2) dsr<2048>(fb1[NF1 >= 2 ? 1 : 0][1], adB1[BUFI][1]);                                  \
    }
#define C8_SYNC_AND_MMA(ACC, FB, NFB, MID)                                                                           \
    {                                                                                                                \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPLIT ? VMC - 1 : VMC) : "memory");                                 \
        __builtin_amdgcn_s_barrier();                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
                     : "+v"(fa[0][0]), "+v"(fa[1][0]), "+v"(fa[2][0]), "+v"(fa[3][0]), "+v"(fa[0][1]), "+v"(fa[1][1]), \
                       "+v"(fa[2][1]), "+v"(fa[3][1]));                                                              \
        tie_frags<NFB>(FB);                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        _Pragma("unroll") for (int i = 0; i < NFB; ++i)                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
                ACC[i][j] = mfma16<T>(FB[i][0], fa[j][0], ACC[i][j]);                                                \
        if constexpr (SPLIT) {                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            MID;                                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < NFB; ++i)                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
                ACC[i][j] = mfma16<T>(FB[i][1], fa[j][1], ACC[i][j]);                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                \
    }
// one K tile from buffer BUFI (tile index t): four phases
#define C8_TILE(BUFI, t)                                                                                             \
    {                                                                                                                \
        C8_READ_B0(BUFI) C8_READ_A(0, BUFI) stageB(1, (BUFI) ^ 1, (t) + 1, P0_);                                     \
        C8_SYNC_AND_MMA(acc00, fb0, NF0, stageB(1, (BUFI) ^ 1, (t) + 1, 1))                                          \
        C8_READ_B1(BUFI) stageA(1, (BUFI) ^ 1, P0_); C8_SYNC_AND_MMA(acc01, fb1, NF1, stageA(1, (BUFI) ^ 1, 1))      \
        C8_READ_A(1, BUFI) stageA(0, BUFI, P0_); C8_SYNC_AND_MMA(acc11, fb1, NF1, stageA(0, BUFI, 1))                \
        stageB(0, BUFI, (t) + 2, P0_); C8_SYNC_AND_MMA(acc10, fb0, NF0, stageB(0, BUFI, (t) + 2, 1))                 \
    }
#define C8_MMA32(ACCX, FBX, NFX, ACCY, FBY, NFY, NLOADS)                                                                     \
    {                                                                                                                \
        asm volatile("s_waitcnt vmcnt(%[vm]) lgkmcnt(0)"                                                                 \
                     : "+v"(fa[0][0]), "+v"(fa[1][0]), "+v"(fa[2][0]), "+v"(fa[3][0]), "+v"(fa[0][1]), "+v"(fa[1][1]), \
                       "+v"(fa[2][1]), "+v"(fa[3][1])                                                                \
                     : [vm] "n"(NLOADS) : "memory");                                                                     \
        tie_frags<NF0>(fb0);                                                                                         \
        tie_frags<NF1>(fb1);                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                \
            _Pragma("unroll") for (int i = 0; i < NFX; ++i)                                                          \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
                    ACCX[i][j] = mfma16<T>(FBX[i][s], fa[j][s], ACCX[i][j]);                                         \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                \
            _Pragma("unroll") for (int i = 0; i < NFY; ++i)                                                          \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
                    ACCY[i][j] = mfma16<T>(FBY[i][s], fa[j][s], ACCY[i][j]);                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                \
    }
#define C8_TILE_LONG(BUFI, t)                                                                                        \
    {                                                                                                                \
        C8_READ_B0(BUFI) C8_READ_B1(BUFI) C8_READ_A(0, BUFI)                                                         \
        stageB(1, (BUFI) ^ 1, (t) + 1, 2); stageA(1, (BUFI) ^ 1, 2);                                                 \
        C8_MMA32(acc00, fb0, NF0, acc01, fb1, NF1, 2 + NF1)      /* this phase issued W1 + P1: everything older has landed */                                                                             \
        C8_READ_A(1, BUFI)                                                                                           \
        stageA(0, BUFI, 2); stageB(0, BUFI, (t) + 2, 2);                                                             \
        C8_MMA32(acc11, fb1, NF1, acc10, fb0, NF0, 2 + NF0)      /* this phase issued P0 + W0 */                                                                             \
    }
    constexpr int P0_ = SPLIT ? 0 : 2;                            // what the read segment issues: the first load / both
    int t = kbeg;
    if constexpr (LONG) {
        for (; t + 1 < kend; t += 2) {
            C8_TILE_LONG(0, t)
            C8_TILE_LONG(1, t + 1)
        }
        if (t < kend) C8_TILE_LONG(0, t)
    } else {
        for (; t + 1 < kend; t += 2) {
            C8_TILE(0, t)
            C8_TILE(1, t + 1)
        }
        if (t < kend) C8_TILE(0, t)
    }
#undef C8_TILE_LONG
#undef C8_MMA32
#undef C8_TILE
#undef C8_SYNC_AND_MMA
#undef C8_READ_A
#undef C8_READ_B0
#undef C8_READ_B1
    if (wr == 0) __builtin_amdgcn_s_barrier();                    // group 0 waits for group 1's last phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the dead prefetches of the tail (they write LDS)

    // ---- epilogue: the wave's 128 x 16 NFA block as two pixel halves (P0, P1) of NFA cout fragments: W0's then W1's; fragments go
    // through the register epilogue in pairs (v_permlane16_swap), the fifth of a 320-wide tile alone
    const bool do_cs = a.colsum != nullptr && !a.ws;
    float* const pw = (float*)smem;                               // [4 pixel blocks (wr, mh)][4 wc][NFA / 2 pairs][4 g][8]
    if (do_cs) __syncthreads();                                   // every wave's LDS-DMA has landed before the ring is reused
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
        const int mb = m0 + wr * 128 + mh * 64, nbw = n0 + wc * 16 * NFA;
        f32x4_t(&aw0)[NF0][4] = mh ? acc10 : acc00;
        f32x4_t(&aw1)[NF1][4] = mh ? acc11 : acc01;
#pragma unroll
        for (int p = 0; p < NFA / 2; ++p) {                       // pair p = fragments 2 p, 2 p + 1 of the wave's NFA
            f32x4_t pr[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (NF0 == 1) {                         // one pair: W0[0] W1[0]
                    pr[0][j] = aw0[0][j];
                    pr[1][j] = aw1[0][j];
                } else if constexpr (NF0 == 2) {
                    pr[0][j] = p == 0 ? aw0[0][j] : aw1[0][j];
                    pr[1][j] = p == 0 ? aw0[NF0 - 1][j] : aw1[NF1 - 1][j];
                } else {                                          // fragments: W0[0] W0[1] | W0[2] W1[0] | W1[1]
                    pr[0][j] = p == 0 ? aw0[0][j] : aw0[NF0 - 1][j];
                    pr[1][j] = p == 0 ? aw0[1][j] : aw1[0][j];
                }
            }
            const int nb = nbw + 32 * p;
            float* pws = do_cs ? pw + ((((wr * 2 + mh) * 4 + wc) * (NFA / 2) + p) * 4) * 8 : nullptr;
            if (a.ws) tile_epilogue_raw<2>(a, pr, 0, 0, g, r16, mb, nb, (int)blockIdx.y);
            else if (a.gate) {
                if (a.cscale) tile_epilogue_block<T, 2, true, true>(a, pr, g, r16, mb, nb, pws);
                else tile_epilogue_block<T, 2, true, false>(a, pr, g, r16, mb, nb, pws);
            } else {
                if (a.cscale) tile_epilogue_block<T, 2, false, true>(a, pr, g, r16, mb, nb, pws);
                else tile_epilogue_block<T, 2, false, false>(a, pr, g, r16, mb, nb, pws);
            }
        }
        if constexpr (NFA & 1) tile_epilogue_single<T>(a, aw1[NF1 - 1], g, r16, mb, nbw + 16 * (NFA - 1));
    }
    if (do_cs) {
        __syncthreads();
        if (tid < BN && n0 + tid < a.Co) {
            const int cb = tid >> 5, r2 = tid & 31;               // (wc, pair) = wc NFA / 2 + pair, column within the pair
            const int gc = (r2 >> 4) | (((r2 >> 3) & 1) << 1), ec = r2 & 7;        // inverse of cl = 16 (g & 1) + 8 (g >> 1)
            float s = 0.f;
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) s += pw[((pb * (2 * NFA) + cb) * 4 + gc) * 8 + ec];      // fixed order: bit-reproducible
            if (a.cslab) a.cslab[(long)(m0 >> 8) * a.Co + n0 + tid] = s;
            else if (s != 0.f) atomicAdd(a.colsum + n0 + tid, s);
        }
    }
#endif
}

template <typename T, int NF0, int NF1, int MODE>
int launch_8ph_v(const WideArgs& a, hipStream_t st) {
    constexpr int lds = 2 * (2 * SLOT + 64 * (NF0 + NF1) * 128);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_8ph<T, NF0, NF1, MODE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (MODE == 2) (void)hipFuncSetAttribute((const void*)conv_igemm_8ph<T, NF0, NF1, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    // SZN_8PH_KORD (default 1): cin-chunk-major K order for the K x K layers (MODE 2 only; see the kernel).  The kernel alone is 0-4 %
    // slower that way, the step 0.17 ms FASTER (same box, eight alternating runs: 9.08-9.16 -> 8.93-8.96 ms): its fabric traffic no longer
    // pushes everybody else's operands out of the caches.  0 = tap-major: bit-identical to conv_igemm_wide / conv_igemm_v2.
    static const int kord = szn_knob("SZN_8PH_KORD", 1);
    if (kord && MODE == 2 && a.KH * a.KW > 1 && !(kord == 2 && a.KH != 3)) {       // (2: the 3 x 3 layers only -- A/B of fc6's 7 x 7 forward)
        hipLaunchKernelGGL((conv_igemm_8ph<T, NF0, NF1, 2, true>), dim3(a.mtiles * a.ntiles, a.nsplit), dim3(512), lds, st, a);
        SZN_CHECK_LAUNCH(NF0 + NF1 == 2 ? "conv_igemm_8ph_n128" : "conv_igemm_8ph");
        return SZN_OK;
    }
    hipLaunchKernelGGL((conv_igemm_8ph<T, NF0, NF1, MODE, false>), dim3(a.mtiles * a.ntiles, a.nsplit), dim3(512), lds, st, a);
    SZN_CHECK_LAUNCH(NF0 + NF1 == 2 ? "conv_igemm_8ph_n128" : "conv_igemm_8ph");
    return SZN_OK;
}

template <typename T, int NF0, int NF1>
int launch_8ph(const WideArgs& a, hipStream_t st) {
    // SZN_8PH_MODE: 2 (default) = two phases of 32 MFMA per K tile; 0 = four phases of 16 (the template's form: 2-4 % slower per kernel,
    // profiles/r04_ablations.txt section 8); 1 = four phases with the second LDS-DMA load issued among the MFMAs (6-8 % slower, section 6)
    return launch_8ph_v<T, NF0, NF1, 2>(a, st);
}

}  // namespace

// Called by szn_conv_wide_try / szn_proj_stream_try with a filled argument block (16-bit operands).  bn = 256: every epilogue;
// bn = 320 (the 300-d projection as one cout tile): bias / ReLU only.  Returns 1 when the shape does not fit: needs Ci a multiple of
// 64, 16-B aligned rows for the register epilogue (a.direct_ep) and map sides below 32768.
int szn_conv_8ph_launch(const void* args, int dtype, int bn, szn_stream_t stream) {
    const WideArgs& a = *(const WideArgs*)args;
    if (!szn_is16(dtype) || (a.Ci & 63) || (!a.direct_ep && !a.ws) || a.Hi >= 32768 || a.Wi >= 32768 || a.pad >= 16384) return 1;
    if (a.ws && (((uintptr_t)a.ws & 15) || (a.Co & 7))) return 1;
    // (a 320-cout tile -- the 300-d projection as one cout tile -- does not fit this form: 160 accumulator + 72 fragment registers
    // spill, <3, 2> is not instantiated; a form with the whole weight operand resident and the pixel operand in quarters fits but
    // concentrates the LDS-DMA issue in two phases and measured 12-25 % slower: profiles/r04_ablations.txt)
    if (bn != 256) return 1;           // (the 256 x 128 form <1, 1> was removed in round 6: never faster inside a step)
    return dtype == SZN_F16 ? launch_8ph<f16_raw, 2, 2>(a, (hipStream_t)stream) : launch_8ph<bf16_raw, 2, 2>(a, (hipStream_t)stream);
}
