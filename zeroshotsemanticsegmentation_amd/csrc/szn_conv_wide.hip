// szn_conv_wide.hip -- 256 x 256 (or 256 x 320) tile variant of the forward / dgrad implicit GEMM for layers with >= 256
// couts; the 320-wide tile (10 weight fragments per wave, 160 accumulator VGPRs) makes the 300-d pixel projection one
// cout tile with 6 % padding instead of 3 x 128 (22 %).
//
// rocprof + ablations (profiles/r01_ablations.txt) show conv_igemm_v2 (256 px x 128 couts) pinned by the LDS-DMA fill
// rate of a CU (~44 GB/s): 48 KB of operands per 4.2 MFLOP.  A 256 x 256 tile moves 64 KB per 8.4 MFLOP (1.5x less per
// FLOP).  Cost: 128 accumulator registers per lane and a 64 KB ring stage, so the ring is 2 stages (one chunk of
// prefetch) and the LDS-staged epilogue runs in four 64-pixel passes.
//   * 512 threads = 8 waves (4 x 2): wave (wm, wn) -> 64 pixels x 128 couts (8 weight fragments x 4 pixel fragments);
//   * otherwise identical to conv_igemm_v2: buffer_load ... lds with source-side XOR swizzle, scalar soffset per chunk,
//     OOB offsets for padding, bias / ReLU / gate / dropout / column-sum epilogue on whole output rows.
#include "szn_common.h"
#include "szn_epilogue.h"
#include "szn_wide_args.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((address_space(3))) void* ldsptr_t;

namespace {

// ---- epilogue staged through LDS in four 64-pixel passes ----
template <typename T, int WNF>
__device__ __forceinline__ void wide_epilogue(const WideArgs& a, f32x4_t (&acc)[WNF][4], char* smem, int tid, int wm, int wn,
                                              int g, int r16, int m0, int n0, int split) {
    constexpr int ES = sizeof(T);
    constexpr int BN = 32 * WNF;
    constexpr int P = BN + 4;                          // tile pitch (floats): 64 x 260 x 4 B = 66,560 B per pass
    constexpr int CPR = BN / 8;                        // 8-cout chunks per row (32 / 40)
    constexpr int RG = 512 / CPR;                      // row groups (16 / 12; threads >= RG * CPR idle in the epilogue)
    constexpr int NIT = (64 + RG - 1) / RG;
    float* tile = (float*)smem;
    if (a.abl_ep == 2) return;
    const T* __restrict__ gate = a.abl_ep ? nullptr : (const T*)a.gate;
    const bool out32 = a.out_f32 || sizeof(T) == 4;
    const int oes = out32 ? 4 : 2;
    const bool fast_o = (((long)a.ldo * oes) & 15) == 0;
    const bool fast_g = gate && ((((long)a.ldg * ES) & 15) == 0);
    const int cc = tid % CPR, row0 = tid / CPR;
    const int n = n0 + cc * 8;
    const bool full = n + 8 <= a.Co;
    float bv[8], cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bv[e] = (a.bias && n + e < a.Co) ? a.bias[n + e] : 0.f; cs[e] = 0.f; }
    // ReLU-gate rows (dgrad: 16 B of the forward activation per 8 outputs) are fetched ONE PASS AHEAD: issued in front of the two
    // barriers and the LDS staging of the pass before, so their latency is no longer paid NIT times per pass in the store loop
    // (dgrad ran ~10 % behind the forward pass of the same layer).  16-bit gates with 16-B rows only; SZN_WIDE_GATEPF=0: off.
    // (the 320-wide tile keeps 160 accumulator registers and is never gated: no prefetch registers there)
    constexpr int NG = (ES == 2 && WNF <= 8) ? NIT : 1;
    const bool gpf = ES == 2 && WNF <= 8 && fast_g && full && a.gate_prefetch && !a.ws;
    u32x4_t gcur[NG], gnext[NG];
    auto load_gates = [&](int pass, u32x4_t (&dst)[NG]) {
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int row = row0 + k * RG;
            const int m = m0 + pass * 64 + row;
            const bool ok = row0 < RG && row < 64 && m < a.M;
            dst[k] = ok ? *(const u32x4_t*)(gate + (long)m * a.ldg + n) : u32x4_t{0u, 0u, 0u, 0u};
        }
    };
    if (gpf) load_gates(0, gcur);
    for (int pass = 0; pass < 4; ++pass) {
        if (gpf && pass + 1 < 4) load_gates(pass + 1, gnext);
        __syncthreads();
        if (wm == pass) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < WNF; ++i)
                    *(f32x4_t*)(tile + (j * 16 + r16) * P + wn * (BN / 2) + i * 16 + g * 4) = acc[i][j];
        }
        __syncthreads();
        if (a.ws) {                                        // split-K slab: raw fp32 partial sums, whole rows
            if (n < a.Co && row0 < RG) {
                float* slab = a.ws + (size_t)split * a.M * a.Co;
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int row = row0 + k * RG;
                    const int m = m0 + pass * 64 + row;
                    if (row < 64 && m < a.M) {
                        const float* tp = tile + row * P + cc * 8;
                        float* o = slab + (size_t)m * a.Co + n;
                        if (full && (a.Co & 3) == 0) {
                            *(f32x4_t*)o = *(const f32x4_t*)tp;
                            *(f32x4_t*)(o + 4) = *(const f32x4_t*)(tp + 4);
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) if (n + e < a.Co) o[e] = tp[e];
                        }
                    }
                }
            }
            continue;
        }
        if (n < a.Co && row0 < RG) {
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int row = row0 + k * RG;
                const int m = m0 + pass * 64 + row;
                if (row < 64 && m < a.M) {
                    const float* tp = tile + row * P + cc * 8;
                    float v[8];
                    *(f32x4_t*)&v[0] = *(const f32x4_t*)tp;
                    *(f32x4_t*)&v[4] = *(const f32x4_t*)(tp + 4);
                    float gv[8];
                    if (gpf) {
                        const T* qe = (const T*)&gcur[k < NG ? k : 0];
#pragma unroll
                        for (int e = 0; e < 8; ++e) gv[e] = elem<T>::ld(qe + e);
                    } else if (gate) {
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
                    if (a.abl_ep) { if (v[0] == 1.2345e-33f) ((float*)a.out)[0] = v[1]; }
                    else if (out32) {
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
        if (gpf) {
#pragma unroll
            for (int k = 0; k < NG; ++k) gcur[k] = gnext[k];
        }
    }
    if (a.colsum && !a.ws) {
        __syncthreads();
        float* red = (float*)smem;                     // [RG row groups][BN]
        if (n < a.Co && row0 < RG) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[row0 * BN + cc * 8 + e] = cs[e];
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.Co) {
            float t = 0.f;
            for (int r = 0; r < RG; ++r) t += red[r * BN + tid];
            if (a.cslab) a.cslab[(long)(m0 >> 8) * a.Co + n0 + tid] = t;         // tile row m0 / 256: reduced in a fixed order later
            else if (t != 0.f) atomicAdd(a.colsum + n0 + tid, t);
        }
    }
}


template <typename T, int WNF>
__device__ __forceinline__ void wide_finish(const WideArgs& a, f32x4_t (&acc)[WNF][4], char* smem, int tid, int wm, int wn,
                                            int g, int r16, int m0, int n0, int split) {
    if constexpr (sizeof(T) == 2) {
        if (a.direct_ep) {
            if constexpr (WNF == 10) {             // the 320-wide projection tile (160 accumulator registers): ungated rows only
                if (!a.ws && !a.gate && !a.cscale) {
                    tile_epilogue_direct<T, WNF, false, false>(a, acc, smem, tid, wm, wn, g, r16, m0, n0);
                    return;
                }
            } else if (a.ws) tile_epilogue_raw<WNF>(a, acc, wm, wn, g, r16, m0, n0, split);
            else if (a.gate) {
                if (a.cscale) tile_epilogue_direct<T, WNF, true, true>(a, acc, smem, tid, wm, wn, g, r16, m0, n0);
                else tile_epilogue_direct<T, WNF, true, false>(a, acc, smem, tid, wm, wn, g, r16, m0, n0);
            } else {
                if (a.cscale) tile_epilogue_direct<T, WNF, false, true>(a, acc, smem, tid, wm, wn, g, r16, m0, n0);
                else tile_epilogue_direct<T, WNF, false, false>(a, acc, smem, tid, wm, wn, g, r16, m0, n0);
            }
            if constexpr (WNF != 10) return;
        }
    }
    wide_epilogue<T, WNF>(a, acc, smem, tid, wm, wn, g, r16, m0, n0, split);
}

// ABL (debug, SZN_WIDE_ABLATE, wrong results): 1 = no LDS-DMA in the loop, 2 = no fragment reads / MFMA
template <typename T, int WNF, int ABL = 0>       // WNF = 16-cout fragments per wave: 6 -> BN = 192, 8 -> 256, 10 -> 320
__global__ __launch_bounds__(512, 2) void conv_igemm_wide(WideArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int ES = sizeof(T);
    constexpr int BKE = 128 / ES;
    constexpr int BM = 256, BN = 32 * WNF;
    constexpr int NBW = BN / 64;                  // weight LDS-DMA instructions per wave per chunk (3 / 4 / 5)
    constexpr int STAGE = (BM + BN) * 128;        // 64 / 72 KiB
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2][pixels 256 x 128 B | weights 256 x 128 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int g = lane >> 4, r16 = lane & 15;

    const int nwg = a.mtiles * a.ntiles;
    const int lid = xcd_remap_w(blockIdx.x, nwg);
    const int nt = a.nmajor ? lid / a.mtiles : lid % a.ntiles, mt = a.nmajor ? lid % a.mtiles : lid / a.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);

    const int chunkA = (lane & 7) ^ (lane >> 3);
    unsigned baseA[4], voffA[4], voffB[NBW];
    int ohw[4];
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
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int n = n0 + (NBW * w + i) * 8 + (lane >> 3);
        voffB[i] = (n < a.Co) ? (unsigned)(((long)n * a.KH * a.KW * a.Ci + chunkA * (16 / ES)) * ES) : kOOBx;
    }
    const int cpt = a.Ci / BKE;
    const int split = blockIdx.y;
    const int kbeg = split * a.chunks_per_split;
    const int nK = min(a.KH * a.KW * cpt, kbeg + a.chunks_per_split);          // end of this split's chunk range
    int itap = kbeg / cpt, ic = kbeg - itap * cpt;
    auto set_tap = [&]() {
        const int kh = itap / a.KW, kw = itap - kh * a.KW;
        const unsigned tapoff = (unsigned)((kh * a.Wi + kw) * a.ldi * ES);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ih = (ohw[i] >> 16) + kh, iw = (int)(short)(ohw[i] & 0xffff) + kw;
            const bool ok = (unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi;
            voffA[i] = ok ? baseA[i] + tapoff : kOOBx;
        }
    };
    auto issue = [&](int stage) {
        char* sb = smem + stage * STAGE;
        const int soffA = ic * 128;
        const int soffB = (itap * a.Ci) * ES + ic * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(sb + (32 * w + 8 * i) * 128), 16, voffA[i], soffA, 0, 0);
#pragma unroll
        for (int i = 0; i < NBW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)(sb + BM * 128 + (NBW * w + i) * 1024), 16, voffB[i],
                                                     soffB, 0, 0);
        if (++ic == cpt) { ic = 0; ++itap; set_tap(); }
    };

    f32x4_t acc[WNF][4];
#pragma unroll
    for (int i = 0; i < WNF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    set_tap();
    issue(0);
    const int offs0 = ((g ^ (r16 & 7)) << 4), offs1 = (((4 + g) ^ (r16 & 7)) << 4);
    int stage = 0;
    for (int kc = kbeg; kc < nK; ++kc) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // chunk kc landed (the only one outstanding)
        __builtin_amdgcn_s_barrier();                             // ... for every wave; everyone left the other stage
        // The eight LDS-DMA loads of a wave stall it at VMEM issue (a CU ingests ~64 B of LDS-DMA per clock: 64 KiB = ~1000
        // cycles per chunk, half of the chunk's MFMA time): wave pair k issues behind its k-th pair of weight fragments of
        // the first K half, so the eight waves are never all stalled at once and their SIMD partners keep the MFMA pipe busy.
        const bool fill = ABL != 1 && kc + 1 < nK;
        // turn = the weight-fragment index (of the first K half) behind which this wave issues: stagger 1 -> pairs at
        // 0, 2, 4, 6; stagger 2 -> every wave its own slot
        // (WNF < 8, the 192-wide tile: pairs at 0, 1, 2, 3 -- every slot has to be below WNF)
        const int turn = WNF < 8 ? (a.stagger ? (w >> 1) : 0) : a.stagger == 2 ? w : (a.stagger ? 2 * (w >> 1) : 0);
        if (fill && turn == 0) issue(stage ^ 1);
        const char* sp = smem + stage * STAGE + (wm * 64 + r16) * 128;
        const char* sw = smem + stage * STAGE + BM * 128 + (wn * (BN / 2) + r16) * 128;
        if constexpr (ABL == 2) { if (fill && turn != 0) issue(stage ^ 1); } else
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int off = s ? offs1 : offs0;
            u32x4_t wf[WNF], pf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) pf[j] = *(const u32x4_t*)(sp + j * 16 * 128 + off);
#pragma unroll
            for (int i = 0; i < WNF; ++i) wf[i] = *(const u32x4_t*)(sw + i * 16 * 128 + off);
#pragma unroll
            for (int i = 0; i < WNF; ++i) {
                if (s == 0 && i > 0 && i < 8 && fill && turn == i) issue(stage ^ 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (ES == 2) {
                        acc[i][j] = mfma16<T>(wf[i], pf[j], acc[i][j]);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[i].x), __uint_as_float(pf[j].x), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[i].y), __uint_as_float(pf[j].y), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[i].z), __uint_as_float(pf[j].z), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[i].w), __uint_as_float(pf[j].w), acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        stage ^= 1;
    }

    wide_finish<T, WNF>(a, acc, smem, tid, wm, wn, g, r16, m0, n0, split);
#endif
}

// ---- 3x3 / pad 1 / same-size layers (conv3_x .. conv5_x, forward and dgrad): row-resident pixel operand ------------------------
// For these layers the input pixel of output m under tap (kh, kw) is the FLATTENED pixel m + (kh - 1) Wi + (kw - 1), so the three
// kw taps of one kernel row read the same 258 consecutive input pixels shifted by one LDS row.  The pixel operand is therefore
// staged once per (kh, cin chunk) -- 264 x 128 B -- and used for three K steps; only the weights (256 x 128 B) change per step:
// 43 KiB of LDS-DMA per 8.4 MFLOP step instead of 64 KiB (the fill rate is what bounds conv_igemm_wide, see the header).
// Pixels whose tap falls outside the image (borders; tiles also run across rows and images) read a real neighbour from the
// flattened buffer: their fragment is replaced by zeros with a per-lane select (a lane of the pixel fragment holds ONE pixel).
// K order: (kh, cin chunk, kw).  LDS: pixels 2 x 33 KiB + weights 2 x 32 KiB = 130 KiB.
template <typename T>
__global__ __launch_bounds__(512, 2) void conv3x3_wide_rows(WideArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(T) == 2, "16-bit storage only");
    constexpr int ES = 2, BM = 256, BN = 256, WNF = 8, NBW = 4;
    constexpr int AROWS = 264;                              // 256 pixels + 2 halo rows, padded to whole 8-row DMA groups
    constexpr int ABYTES = AROWS * 128, BBYTES = BN * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // pixels [2][264 x 128 B] | weights [2][256 x 128 B]
    char* const sA = smem;
    char* const sB = smem + 2 * ABYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int g = lane >> 4, r16 = lane & 15;

    const int nwg = a.mtiles * a.ntiles;
    const int lid = xcd_remap_w(blockIdx.x, nwg);
    const int nt = lid % a.ntiles, mt = lid / a.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);

    const int chunkA = (lane & 7) ^ (lane >> 3);
    const long npix = (long)a.B * a.Hi * a.Wi;
    // LDS row r of the pixel buffer <-> flattened input pixel m0 - 1 + (kh - 1) Wi + r.  Wave w fills row groups 4w .. 4w+3,
    // wave 0 also group 32 (rows 256 .. 263, of which 256 and 257 are used)
    unsigned voffA[5], voffB[NBW];
    auto set_kh = [&](int kh) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int row = (i < 4 ? 32 * w + 8 * i : 256) + (lane >> 3);
            const long p = (long)m0 - 1 + (long)(kh - 1) * a.Wi + row;
            const bool ok = p >= 0 && p < npix && row < BM + 2;
            voffA[i] = ok ? (unsigned)((p * a.ldi + chunkA * 8) * ES) : kOOBx;
        }
    };
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
        const int n = n0 + (NBW * w + i) * 8 + (lane >> 3);
        voffB[i] = (n < a.Co) ? (unsigned)(((long)n * 9 * a.Ci + chunkA * 8) * ES) : kOOBx;
    }
    // tap validity of the four pixels this lane supplies (pixel fragment j, row r16): bit kh * 3 + kw
    unsigned vmask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + 16 * j + r16;
        unsigned mk = 0;
        if (m < a.M) {
            const int r = m % a.HoWo;
            const int oh = r / a.Wo, ow = r - oh * a.Wo;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    if ((unsigned)(oh + kh - 1) < (unsigned)a.Hi && (unsigned)(ow + kw - 1) < (unsigned)a.Wi) mk |= 1u << (kh * 3 + kw);
        }
        vmask[j] = mk;
    }

    const int cpt = a.Ci / 64;
    const int nK = 9 * cpt;
    // the step being issued FOR (one ahead of the step being computed)
    int ikh = 0, iic = 0, ikw = 0, igrp = 0, voff_kh = 0;
    auto issue = [&](int step) {          // loads for `step`: its weights, and -- at kw 1 / 2 -- half of the NEXT group's pixels
        const int soffB = ((ikh * 3 + ikw) * a.Ci) * ES + iic * 128;
        char* sb = sB + (step & 1) * BBYTES;
#pragma unroll
        for (int i = 0; i < NBW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)(sb + (NBW * w + i) * 1024), 16, voffB[i], soffB, 0, 0);
        if (ikw > 0) {
            // pixels of the group AFTER the one `step` belongs to.  issue(step) runs while step - 1 is being computed: at kw 1 / 2
            // that is kw 0 / 1 of the same group, so the other pixel buffer (last read by the previous group) is free
            int nkh = ikh, nic = iic + 1;
            if (nic == cpt) { nic = 0; ++nkh; }
            if (nkh < 3) {
                if (nkh != voff_kh) { set_kh(nkh); voff_kh = nkh; }
                char* sa = sA + ((igrp + 1) & 1) * ABYTES;
                const int soffA = nic * 128;
                if (ikw == 1) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(sa + (32 * w + 0) * 128), 16, voffA[0], soffA, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(sa + (32 * w + 8) * 128), 16, voffA[1], soffA, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(sa + (32 * w + 16) * 128), 16, voffA[2], soffA, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(sa + (32 * w + 24) * 128), 16, voffA[3], soffA, 0, 0);
                    if (w == 0)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(sa + 256 * 128), 16, voffA[4], soffA, 0, 0);
                }
            }
        }
        if (++ikw == 3) { ikw = 0; ++igrp; if (++iic == cpt) { iic = 0; ++ikh; } }
    };

    f32x4_t acc[WNF][4];
#pragma unroll
    for (int i = 0; i < WNF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // prologue: the pixels of group 0 (kh = 0, chunk 0) and everything `issue(0)` brings
    set_kh(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(sA + (32 * w + 8 * i) * 128), 16, voffA[i], 0, 0, 0);
    if (w == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(sA + 256 * 128), 16, voffA[4], 0, 0, 0);
    issue(0);

    const int turn = a.stagger == 2 ? w : (a.stagger ? 2 * (w >> 1) : 0);
    int ckw = 0, cgrp = 0, ctap_base = 0, cic = 0;       // the step being computed: tap = ctap_base + ckw
    for (int kc = 0; kc < nK; ++kc) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool fill = kc + 1 < nK;
        if (fill && turn == 0) issue(kc + 1);
        const int rsh = (r16 + ckw) & 7;
        const char* sp = sA + (cgrp & 1) * ABYTES + (wm * 64 + r16 + ckw) * 128;
        const char* sw = sB + (kc & 1) * BBYTES + (wn * (BN / 2) + r16) * 128;
        const int tap = ctap_base + ckw;
        bool keep[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) keep[j] = (vmask[j] >> tap) & 1u;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int offp = (((4 * s + g) ^ rsh) << 4), offw = (((4 * s + g) ^ (r16 & 7)) << 4);
            u32x4_t wf[WNF], pf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4_t v = *(const u32x4_t*)(sp + j * 16 * 128 + offp);
                pf[j].x = keep[j] ? v.x : 0u; pf[j].y = keep[j] ? v.y : 0u; pf[j].z = keep[j] ? v.z : 0u; pf[j].w = keep[j] ? v.w : 0u;
            }
#pragma unroll
            for (int i = 0; i < WNF; ++i) wf[i] = *(const u32x4_t*)(sw + i * 16 * 128 + offw);
#pragma unroll
            for (int i = 0; i < WNF; ++i) {
                if (s == 0 && i > 0 && fill && turn == i) issue(kc + 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<T>(wf[i], pf[j], acc[i][j]);
            }
        }
        if (++ckw == 3) { ckw = 0; ++cgrp; if (++cic == cpt) { cic = 0; ctap_base += 3; } }
    }

    wide_finish<T, WNF>(a, acc, smem, tid, wm, wn, g, r16, m0, n0, 0);
#endif
}

// ---- pixel projection (1x1, K >> N, N <= 320): activation operand streamed from HBM ----------------------------------------------
// score_fr is M x 4096 x 300: every activation row is used by ONE block, once (AI = 300 FLOP per activation byte -- at the HBM
// ridge), while the 2.6 MB filter matrix is re-read by every block from its XCD's L2.  conv_igemm_wide<T, 10> keeps one 72 KiB chunk
// of prefetch in flight per CU (8 MB on the chip): enough for operands that come from L2 / MALL, not for 2 us of loaded HBM latency
// (measured: 1.7 TB/s of activation stream, 0.5 PF).  Here the two operands get their own rings:
//   * activations: 3 units of 256 px x 128 B (K = 64), two units = 64 KiB per CU in flight (16 MB on the chip), whole 128-B lines per
//     request, non-temporal (aux = nt: streamed once, must not push the filter matrix out of L2);
//   * filters: 3 units of 320 x 64 B (K = 32), two in flight, default policy (L2 hits);
//   * one s_barrier per K = 32 half step (40 MFMA per wave between barriers), counted s_waitcnt vmcnt: a wave's loads complete in
//     issue order, and behind the unit a half step needs there are always exactly one filter unit and one activation unit.
//     (Staggering the issue across waves, which pays in the conv kernels, measured nothing here: the loads per half step are few.)
// 64-B filter rows: ds_read_b128 is served 8 lanes (128 B) at a time, so the eight rows r .. r + 7 a lane group reads (same chunk)
// must fall on eight different 16-B slots of a 128-B window: slot = chunk ^ ((row >> 1) & 3) (with (row >> 2) the PMC pass showed
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.40).  Accumulators / epilogue as conv_igemm_wide<T, 10> (bias, 16-bit or fp32 rows).
// Co <= 304 (the 300-d projection): the 20th 16-cout fragment is pure padding.  Its filter rows are not loaded and the waves of the
// upper cout half run 9 fragments; waves w and w + 4 share a SIMD, so (wm, wn) = (w & 3, w >> 2) gives every SIMD 40 + 36 MFMA per half
// step instead of 80.
// STAG (round 4, SZN_PROJ_STAG): the two wave groups (wn = 0 / 1: waves w and w + 4 share a SIMD) run one barrier apart, every half step
// becomes {fragment reads (inline asm) + this half step's LDS-DMA issue + counted vmcnt + lgkmcnt(0); barrier; 40 MFMA; barrier}: one wave of
// every SIMD multiplies while its partner reads and issues (the schedule of szn_conv_8ph.hip).  Reads are retired before the barrier, so a
// slot is re-staged in the half step after its last read (the 3-unit rings stay); a unit issued in half step p is waited for in p + 1 (filters:
// read in p + 2) -- the activation units keep three half steps of flight instead of four.
template <typename T, bool NT, bool NF19, bool STAG>
__global__ __launch_bounds__(512, 2) void proj_gemm_stream(WideArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(T) == 2, "16-bit storage only");
    constexpr int BM = 256, BN = 320, WNF = 10;
    constexpr int AUNIT = BM * 128, WUNIT = BN * 64;           // 32 KiB, 20 KiB
    constexpr int AUXA = NT ? 2 : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // activations [3][256 x 128 B] | filters [3][320 x 64 B]
    char* const sA = smem;
    char* const sW = smem + 3 * AUNIT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w & 3, wn = w >> 2;
    const int g = lane >> 4, r16 = lane & 15;
    const int mt = xcd_remap_w(blockIdx.x, a.mtiles);
    const int m0 = mt * BM;

    // the activation resource covers this block's rows only (base = row m0): no 2 GB limit on the matrix, rows >= M fall outside
    const int rows = min(BM, a.M - m0);
    const size_t abase = a.proj_abl == 1 ? 0 : (size_t)m0 * a.ldi * 2;     // ablation 1: every block streams the rows of block 0 (L2 hits)
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + abase), 0, (int)((size_t)rows * a.ldi * 2), 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);

    // activations: wave w fills pixel rows 32 w .. 32 w + 31, one instruction = 8 rows x 128 B (lane -> row lane >> 3, 16-B chunk
    // (lane & 7) ^ row: the source-side XOR swizzle of the other kernels)
    unsigned voffA[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 32 * w + 8 * i + (lane >> 3);
        voffA[i] = (r < rows) ? (unsigned)(((long)r * a.ldi + ((lane & 7) ^ (lane >> 3)) * 8) * 2) : kOOBx;
    }
    // filters: 20 instructions of 16 rows x 64 B per unit; wave w takes row blocks w, w + 8, w + 16 (the third only for w < 4)
    constexpr int W3 = NF19 ? 3 : 4;                            // waves below W3 issue three filter loads per unit, the others two
    const int nwi = w < W3 ? 3 : 2;
    unsigned voffB[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int row = (w + 8 * i) * 16 + (lane >> 2);
        const int n = row;                                      // one cout tile: n0 = 0
        const int chunk = (lane & 3) ^ ((row >> 1) & 3);
        voffB[i] = (i < nwi && n < a.Co) ? (unsigned)(((long)n * a.Ci + chunk * 8) * 2) : kOOBx;
    }
    const int nU = a.Ci / 64, H = 2 * nU;
    auto issueA = [&](int u) {
        char* sa = sA + (u % 3) * AUNIT;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(sa + (32 * w + 8 * i) * 128), 16, voffA[i], u * 128, 0, AUXA);
    };
    auto issueW = [&](int h) {
        char* sw = sW + (h % 3) * WUNIT;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)(sw + (w + 0) * 1024), 16, voffB[0], h * 64, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)(sw + (w + 8) * 1024), 16, voffB[1], h * 64, 0, 0);
        if (w < W3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)(sw + (w + 16) * 1024), 16, voffB[2], h * 64, 0, 0);
    };

    f32x4_t acc[WNF][4];
#pragma unroll
    for (int i = 0; i < WNF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // issue order A0 W0 A1 W1 | W2 A2 | W3 | W4 A3 | W5 | ...: when half step h needs (A[h/2], W[h]) the loads issued after them are
    // always one filter unit and one activation unit -> vmcnt(4 + nwi)
    issueA(0); issueW(0);
    if (nU > 1) issueA(1);
    if (H > 1) issueW(1);
    const int offw = ((g ^ ((r16 >> 1) & 3)) << 4);
    if constexpr (STAG) {
        const int smem_lds = (int)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        auto rd = [&](int addr) -> u32x4_t {
            u32x4_t v;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
            return v;
        };
        // everything the un-staggered prologue issued stays; its first wait happens in half step 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (A0, W0, A1, W1 of this wave landed: a one-off)
        __builtin_amdgcn_s_barrier();
        if (wn == 1) __builtin_amdgcn_s_barrier();            // group 1 runs one barrier behind group 0
        for (int h = 0; h < H; ++h) {
            const int base = smem_lds + (int)(sA - smem);
            const int ap = base + ((h >> 1) % 3) * AUNIT + (wm * 64 + r16) * 128 + (((4 * (h & 1) + g) ^ (r16 & 7)) << 4);
            const int wp = smem_lds + (int)(sW - smem) + (h % 3) * WUNIT + (wn * 160 + r16) * 64 + offw;
            u32x4_t pf[4], wf[WNF];
#pragma unroll
            for (int j = 0; j < 4; ++j) pf[j] = rd(ap + j * 16 * 128);
#pragma unroll
            for (int i = 0; i < WNF; ++i) wf[i] = rd(wp + i * 16 * 64);
            if (h + 2 < H) issueW(h + 2);                     // into the slot of W[h - 1] (its reads were retired before the last barrier)
            if (!(h & 1) && (h >> 1) + 2 < nU) issueA((h >> 1) + 2);
            // W[h + 1] (issued in half step h - 1) has to have landed: behind it in issue order there are at most one activation unit and
            // this half step's filter unit
            if (h + 4 >= H) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the activation issue stops four half steps before the end)
            else if (w < W3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3]), "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]),
                           "+v"(wf[4]), "+v"(wf[5]), "+v"(wf[6]), "+v"(wf[7]), "+v"(wf[8]), "+v"(wf[9]));
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < WNF - 1; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<T>(wf[i], pf[j], acc[i][j]);
            if (!NF19 || wn == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[WNF - 1][j] = mfma16<T>(wf[WNF - 1], pf[j], acc[WNF - 1][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        if (wn == 0) __builtin_amdgcn_s_barrier();            // group 0 waits for group 1's last half step
    } else
    for (int h = 0; h < H; ++h) {
        if (h + 4 >= H) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (w < W3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // (A[h/2], W[h]) landed for every wave; everyone left half step h - 1
        if (h + 2 < H) issueW(h + 2);                 // into the slot of W[h - 1]
        if (!(h & 1) && (h >> 1) + 2 < nU) issueA((h >> 1) + 2);     // into the slot of A[h/2 - 1]
        const char* sp = sA + ((h >> 1) % 3) * AUNIT + (wm * 64 + r16) * 128 + (((4 * (h & 1) + g) ^ (r16 & 7)) << 4);
        const char* sw = sW + (h % 3) * WUNIT + (wn * 160 + r16) * 64 + offw;
        u32x4_t pf[4], wf[WNF];
#pragma unroll
        for (int j = 0; j < 4; ++j) pf[j] = *(const u32x4_t*)(sp + j * 16 * 128);
#pragma unroll
        for (int i = 0; i < WNF; ++i) wf[i] = *(const u32x4_t*)(sw + i * 16 * 64);
#pragma unroll
        for (int i = 0; i < WNF - 1; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<T>(wf[i], pf[j], acc[i][j]);
        if (!NF19 || wn == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[WNF - 1][j] = mfma16<T>(wf[WNF - 1], pf[j], acc[WNF - 1][j]);
        }
    }
    wide_finish<T, WNF>(a, acc, smem, tid, wm, wn, g, r16, m0, 0, 0);
#endif
}

template <typename T, bool NT, bool NF19>
void launch_proj_variant(const WideArgs& a, size_t lds, hipStream_t st) {
    const int stag = 1; /* (was SZN_PROJ_STAG) */
    if (stag) {
        (void)hipFuncSetAttribute((const void*)proj_gemm_stream<T, NT, NF19, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((proj_gemm_stream<T, NT, NF19, true>), dim3(a.mtiles), dim3(512), lds, st, a);
        return;
    }
    (void)hipFuncSetAttribute((const void*)proj_gemm_stream<T, NT, NF19, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((proj_gemm_stream<T, NT, NF19, false>), dim3(a.mtiles), dim3(512), lds, st, a);
}

template <typename T>
int launch_proj_stream(const WideArgs& a, hipStream_t st) {
    const size_t lds = 3 * 256 * 128 + 3 * 320 * 64;
    const int nt = 1; /* (was SZN_PROJ_NT) */
    const bool nf19 = a.Co <= 304;
    if (nt) { if (nf19) launch_proj_variant<T, true, true>(a, lds, st); else launch_proj_variant<T, true, false>(a, lds, st); }
    else { if (nf19) launch_proj_variant<T, false, true>(a, lds, st); else launch_proj_variant<T, false, false>(a, lds, st); }
    SZN_CHECK_LAUNCH("proj_gemm_stream");
    return SZN_OK;
}

template <typename T>
int launch_wide_rows(const WideArgs& a, hipStream_t st) {
    const size_t lds = 2 * 264 * 128 + 2 * 256 * 128;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv3x3_wide_rows<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv3x3_wide_rows<T>), dim3(a.mtiles * a.ntiles), dim3(512), lds, st, a);
    SZN_CHECK_LAUNCH("conv3x3_wide_rows");
    return SZN_OK;
}

template <typename T, int WNF>
int launch_wide(const WideArgs& a, hipStream_t st) {
    const size_t lds = 2 * (256 + 32 * WNF) * 128;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_wide<T, WNF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    static int abl = -1;
    if (abl < 0) { abl = szn_ablate_env("SZN_WIDE_ABLATE"); }
    if (abl && sizeof(T) == 2 && WNF == 8) {          // debug ablations of the bf16 256 x 256 kernel (wrong results)
        if (abl == 1) {
            (void)hipFuncSetAttribute((const void*)conv_igemm_wide<T, WNF, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((conv_igemm_wide<T, WNF, 1>), dim3(a.mtiles * a.ntiles, a.nsplit), dim3(512), lds, st, a);
        } else {
            (void)hipFuncSetAttribute((const void*)conv_igemm_wide<T, WNF, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((conv_igemm_wide<T, WNF, 2>), dim3(a.mtiles * a.ntiles, a.nsplit), dim3(512), lds, st, a);
        }
        SZN_CHECK_LAUNCH("conv_igemm_wide(ablation)");
        return SZN_OK;
    }
    hipLaunchKernelGGL((conv_igemm_wide<T, WNF>), dim3(a.mtiles * a.ntiles, a.nsplit), dim3(512), lds, st, a);
    SZN_CHECK_LAUNCH("conv_igemm_wide");
    return SZN_OK;
}

}  // namespace

int szn_conv_8ph_launch(const void* args, int dtype, int bn, szn_stream_t stream);      // szn_conv_8ph.hip (args = WideArgs)

// Called by szn_conv2d_fwd (which has validated the descriptor).  Returns 1 when the shape is not a good fit.
int szn_conv_wide_try(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias, const void* gate,
                      const float* chan_scale, void* out, unsigned in_bytes, unsigned w_bytes, int min_tiles,
                      float* ws, int nsplit, int chunks_per_split, szn_stream_t stream) {
    if (d->Co < 256) return 1;
    WideArgs a;
    a.proj_abl = 0;
    { static int ea = -1; if (ea < 0) ea = szn_ablate_env("SZN_WIDE_EPABL"); a.abl_ep = ea; }
    a.ws = nsplit > 1 ? ws : nullptr; a.nsplit = nsplit > 1 ? nsplit : 1;
    { const int stg = 1; /* (was SZN_WIDE_STAGGER) */ a.stagger = stg; }
    { const int gp = 1; /* (was SZN_WIDE_GATEPF) */ a.gate_prefetch = gp; }
    a.chunks_per_split = nsplit > 1 ? chunks_per_split : (1 << 30);
    a.M = d->B * d->Ho * d->Wo;
    // cout tile 256, or 320 (bf16) when that wastes fewer columns: the 300-d projection is one 320-wide tile
    const int waste256 = szn_div_up(d->Co, 256) * 256 - d->Co, waste320 = szn_div_up(d->Co, 320) * 320 - d->Co;
    int bn = (szn_is16(d->dtype) && waste320 < waste256) ? 320 : 256;
    a.mtiles = szn_div_up(a.M, 256); a.ntiles = szn_div_up(d->Co, bn);
    // tile order within an XCD's share of the grid: pixel tile fastest when the filter bank is the larger operand (fc6: 205 MB against a
    // 4 MB map), so that the blocks of one XCD share cout tiles and the bank crosses the fabric once, not once per XCD
    { const int nm = 1; /* (was SZN_WIDE_NMAJOR) */ a.nmajor = (nm && w_bytes > in_bytes) ? 1 : 0; }
    if ((long)a.mtiles * a.ntiles * a.nsplit < min_tiles) {         // too few blocks to fill the chip: keep 256 x 128 ...
        // ... unless 256 x 192 tiles fit ONE round of the chip where the 256 x 128 tiles need two (fc7 at B = 8, 512 x 512:
        // 2,312 x 4096 is 10 x 32 = 320 narrow tiles = 1.25 rounds, but 10 x 22 = 220 tiles of 192 couts: 0.117 -> ~0.09 ms)
        static int ncu = 0;
        if (!ncu) {
            int dev = 0; hipDeviceProp_t p;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ncu = p.multiProcessorCount;
            if (ncu <= 0) ncu = 256;
        }
        const long n192 = (long)a.mtiles * szn_div_up(d->Co, 192), n128 = (long)a.mtiles * szn_div_up(d->Co, 128);
        const long cost192 = (n192 + ncu - 1) / ncu * 192, cost128 = (n128 + ncu - 1) / ncu * 128;
        if (!szn_is16(d->dtype) || a.nsplit != 1 || cost192 * 100 >= cost128 * 90) return 1;
        bn = 192; a.ntiles = szn_div_up(d->Co, 192);
    } else if ((long)a.ntiles * bn - d->Co > 64) return 1;          // would waste > 64 columns of the last tile
    a.in = (const char*)in; a.w = (const char*)w; a.bias = bias; a.gate = (const char*)gate; a.cscale = chan_scale;
    a.out = (char*)out; a.colsum = d->colsum;
    a.cslab = (d->colsum && !a.ws) ? d->colsum_slab : nullptr;
    if (a.cslab && d->colsum_slab_rows < a.mtiles) SZN_FAIL(SZN_ERR_ARG, "conv2d: colsum_slab holds %d rows, %d needed", d->colsum_slab_rows, a.mtiles);
    szn_note_colsum_rows(a.cslab ? a.mtiles : 0);
    a.in_bytes = in_bytes; a.w_bytes = w_bytes;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co;
    a.KH = d->KH; a.KW = d->KW; a.pad = d->pad; a.ldi = d->ldi; a.ldo = d->ldo; a.ldg = d->ldg;
    a.relu = d->relu; a.out_f32 = d->out_f32; a.HoWo = d->Ho * d->Wo;
    {
        // epilogue from registers (wide_epilogue_direct): whole 16-B pieces of 8 couts, so rows and bases have to be 16-B aligned
        static const int de = szn_knob("SZN_WIDE_DIRECT", 1);
        const size_t oes = (d->out_f32 || a.ws) ? 4 : 2;
        const uintptr_t al = (uintptr_t)out | (uintptr_t)gate | (uintptr_t)bias | (uintptr_t)chan_scale | (uintptr_t)a.ws;
        a.direct_ep = de && szn_is16(d->dtype) && (d->Co % 8) == 0 && (((size_t)d->ldo * oes) & 15) == 0 && (al & 15) == 0 &&
                      (!gate || (((size_t)d->ldg * 2) & 15) == 0);
    }
    if (bn == 256 && szn_is16(d->dtype)) {
        // the 8-phase schedule (szn_conv_8ph.hip): every 256-wide 16-bit shape, split-K included; SZN_WIDE_8PH=0: the round 1-3 kernels
        static const int ph8 = szn_knob("SZN_WIDE_8PH", 1);
        if (ph8) {
            const int rc = szn_conv_8ph_launch(&a, d->dtype, 256, stream);
            if (rc <= 0) return rc;
        }
    }
    {
        static const int rows = szn_knob("SZN_WIDE_ROWS", 1);
        if (rows && bn == 256 && szn_is16(d->dtype) && a.nsplit == 1 && d->KH == 3 && d->KW == 3 && d->pad == 1 && d->Hi == d->Ho &&
            d->Wi == d->Wo && (d->Ci % 64) == 0)
            return d->dtype == SZN_F16 ? launch_wide_rows<f16_raw>(a, (hipStream_t)stream)
                                       : launch_wide_rows<bf16_raw>(a, (hipStream_t)stream);
    }
    if (bn == 192)
        return d->dtype == SZN_F16 ? launch_wide<f16_raw, 6>(a, (hipStream_t)stream) : launch_wide<bf16_raw, 6>(a, (hipStream_t)stream);
    if (bn == 320)
        return d->dtype == SZN_F16 ? launch_wide<f16_raw, 10>(a, (hipStream_t)stream) : launch_wide<bf16_raw, 10>(a, (hipStream_t)stream);
    if (d->dtype == SZN_F16) return launch_wide<f16_raw, 8>(a, (hipStream_t)stream);
    return d->dtype == SZN_BF16 ? launch_wide<bf16_raw, 8>(a, (hipStream_t)stream) : launch_wide<float, 8>(a, (hipStream_t)stream);
}

// The streaming projection kernel (proj_gemm_stream): 1x1 / pad 0, 16-bit operands, 256 < Co <= 320 (one cout tile), Ci a multiple
// of 64, no gate / dropout factor / column sums / split-K, at least `min_tiles` pixel tiles.  Called by szn_conv2d_fwd BEFORE the
// 2 GB operand check of the generic kernels: the activation resource is rebased per block.  Returns 1 when the shape does not fit.
int szn_proj_stream_try(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias, const void* gate,
                        const float* chan_scale, void* out, int min_tiles, szn_stream_t stream) {
    const int proj = 1; /* (was SZN_PROJ_STREAM) */
    if (!proj || !szn_is16(d->dtype) || d->KH != 1 || d->KW != 1 || d->pad != 0 || d->Co <= 256 || d->Co > 320 || d->Ci < 256 ||
        (d->Ci % 64) || gate || chan_scale || d->colsum || d->pool_out || d->relu)
        return 1;
    if (d->B <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ho != d->Hi || d->Wo != d->Wi || d->ldi < d->Ci || d->ldo < d->Co || !in || !w ||
        !out || (((uintptr_t)in | (uintptr_t)w | (uintptr_t)out) & 15) || (((size_t)d->ldi * 2) & 15) ||
        (long)d->B * d->Ho * d->Wo >= (1L << 31) || (size_t)256 * d->ldi * 2 >= 0x7fff0000ul)
        return 1;
    WideArgs a;
    a.M = d->B * d->Ho * d->Wo;
    a.mtiles = szn_div_up(a.M, 256); a.ntiles = 1; a.nmajor = 0;
    if (a.mtiles < min_tiles) return 1;
    a.ws = nullptr; a.nsplit = 1; a.chunks_per_split = 1 << 30; a.stagger = 0; a.gate_prefetch = 0;
    {
        static int abl = -1;
        if (abl < 0) abl = szn_ablate_env("SZN_PROJ_ABLATE");
        a.proj_abl = abl; a.abl_ep = 0;
        static const int de = szn_knob("SZN_WIDE_DIRECT", 1);
        const size_t oes = d->out_f32 ? 4 : 2;
        a.direct_ep = de && (d->Co % 8) == 0 && (((size_t)d->ldo * oes) & 15) == 0 && (((uintptr_t)out | (uintptr_t)bias) & 15) == 0;
    }
    a.in = (const char*)in; a.w = (const char*)w; a.bias = bias; a.gate = nullptr; a.cscale = nullptr;
    a.out = (char*)out; a.colsum = nullptr; a.cslab = nullptr;
    a.in_bytes = 0; a.w_bytes = (unsigned)((size_t)d->Co * d->Ci * 2);
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co;
    a.KH = 1; a.KW = 1; a.pad = 0; a.ldi = d->ldi; a.ldo = d->ldo; a.ldg = 0;
    a.relu = 0; a.out_f32 = d->out_f32; a.HoWo = d->Ho * d->Wo;
    return d->dtype == SZN_F16 ? launch_proj_stream<f16_raw>(a, (hipStream_t)stream) : launch_proj_stream<bf16_raw>(a, (hipStream_t)stream);
}
