// szn_epilogue.h -- epilogue of the 256-pixel tile kernels (conv_igemm_v2, conv_igemm_wide, conv3x3_wide_rows, proj_gemm_stream)
// straight from the accumulator registers.  Shared by szn_conv_igemm.hip and szn_conv_wide.hip: both kernels give wave (wm, wn) a
// 64-pixel x (16 WNF)-cout block of the tile as acc[WNF][4] (cout fragment i, pixel fragment j; MFMA A operand = weights, so lane
// (g, r16) holds couts 16 i + 4 g .. + 3 of pixel 16 j + r16).  `Args` is the kernel's argument struct (WideArgs / Conv2Args: same
// field names).
#pragma once
#include "szn_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t ep_u32x4_t;

__device__ __forceinline__ float row16_sum_w(float x) {                    // sum over the 16 lanes of a DPP row
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x141, 0xF, 0xF, true));   // row_half_mirror
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x140, 0xF, 0xF, true));   // row_mirror
    return x;
}

// ---- epilogue straight from the accumulator registers (16-bit operands, Co a multiple of 8, 16-B aligned rows) ----------------
// The LDS-staged epilogue of the tile kernels costs 10-13 us per 256 x 256 tile (SZN_WIDE_EPABL accounting, profiles/r03_ablations.txt section 4:
// conv3_2 forward 0.305 ms, 0.299 without its global stores, 0.255 without the epilogue): four passes in which two of the eight
// waves write 64 KiB into LDS while six wait at the barrier, then a branchy read-back loop.  Here nothing is staged: a lane of the
// MFMA result holds 4 consecutive couts of one pixel for each cout fragment; v_permlane16_swap on a PAIR of fragments (the DPP rows
// g and g ^ 1 exchange one fragment each) leaves 8 consecutive couts of one pixel in every lane -- column 16 (g & 1) + 8 (g >> 1) of
// the 32-cout pair -- so bias / ReLU / gate / Dropout2d factor / column sums run on whole 16-B pieces and every lane stores one
// 16-B piece (two for fp32 rows) per (pair, pixel fragment).  Gate pieces (dgrad: the forward activation) are loaded in the same
// layout, the four of a pair up front.  No barrier unless column sums are wanted (bias gradient of the producer layer: DPP row sums ->
// LDS -> thread c adds the four pixel groups in ascending order: fixed order, bit-reproducible).  Same arithmetic per element as the
// staged epilogue: outputs are bit-identical; the column sums add the same terms in another order.
// One wave block: 64 pixels (rows mb + 16 j + r16) x 32 NP couts (columns nb ..) held as acc[WNF][4].  pws = where this block's row
// sums go ([4 g][NP][8] floats in LDS; nullptr: no column sums).  No barrier inside.  RAGGED: Co may be a multiple of 4 only (the last
// 8-cout piece is then written element by element); otherwise Co % 8 == 0 is the caller's contract.
template <typename T, int WNF, bool GATE, bool SCALE, bool RAGGED = false, typename Args>
__device__ __forceinline__ void tile_epilogue_block(const Args& a, f32x4_t (&acc)[WNF][4], int g, int r16, int mb, int nb, float* pws) {
    static_assert(sizeof(T) == 2 && (WNF % 2) == 0, "16-bit storage, fragment pairs");
    constexpr int NP = WNF / 2;
    const T* __restrict__ gate = (const T*)a.gate;
    const bool out32 = a.out_f32 != 0;
    const bool do_cs = pws != nullptr;
    const float lo = a.relu ? 0.f : -__builtin_inff();        // ReLU as max(x, lo): max(x, -inf) == x
    const int cl = 16 * (g & 1) + 8 * (g >> 1);
    const int nw = nb + cl;                                    // column of this lane's piece in pair 0
    int mrow[4];
    bool okm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mrow[j] = mb + 16 * j + r16;
        okm[j] = mrow[j] < a.M;
    }
    // gfx950 counts loads AND stores in vmcnt, in issue order: a load issued behind the stores of the previous pair would make its
    // s_waitcnt sit out those stores' round trip (16 times per tile).  Bias and gate pieces of pair p + 1 are therefore issued in
    // front of the stores of pair p, and the wait in front of their first use is vmcnt(stores of p).
    float bvn[8];
    auto fetch = [&](int p) {
        const int n = nw + 32 * p;
#pragma unroll
        for (int e = 0; e < 8; ++e) bvn[e] = 0.f;
        if (a.bias && n + 8 <= a.Co) {                         // Co % 8 == 0: the whole piece is inside or outside ...
            *(f32x4_t*)&bvn[0] = *(const f32x4_t*)(a.bias + n);
            *(f32x4_t*)&bvn[4] = *(const f32x4_t*)(a.bias + n + 4);
        } else if (RAGGED && a.bias && n < a.Co) {             // ... a ragged Co (the 300-d projection, RAGGED callers only): its last piece
#pragma unroll
            for (int e = 0; e < 8; ++e) bvn[e] = n + e < a.Co ? a.bias[n + e] : 0.f;
        }
    };
    // gate pieces (dgrad: 16 B of the forward activation per piece): a ring of two pairs, pair p + 2 is requested when pair p is done
    // (all NP pairs up front need 16 NP registers beside the accumulators: spills)
    ep_u32x4_t gq[GATE ? 2 : 1][4];
    auto fetch_gate = [&](int p, int slot) {
        const int n = nw + 32 * p;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            gq[slot][j] = (okm[j] && n < a.Co) ? *(const ep_u32x4_t*)(gate + (long)mrow[j] * a.ldg + n) : ep_u32x4_t{0u, 0u, 0u, 0u};
    };
    if constexpr (GATE) {
        fetch_gate(0, 0);
        if (NP > 1) fetch_gate(1, 1);
    }
    fetch(0);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int n = nw + 32 * p;
        const bool okn = n < a.Co;
        float bv[8], cs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { bv[e] = bvn[e]; cs[e] = 0.f; }
        float sc[SCALE ? 4 : 1][8];
        if constexpr (SCALE) {                                 // Dropout2d factor of (image, channel): fc7 only (few tiles)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sc[j][e] = 1.f;
                if (okm[j] && okn) {
                    const float* sp = a.cscale + (long)(mrow[j] / a.HoWo) * a.Co + n;
                    *(f32x4_t*)&sc[j][0] = *(const f32x4_t*)sp;
                    *(f32x4_t*)&sc[j][4] = *(const f32x4_t*)(sp + 4);
                }
            }
        }
        if (p + 1 < NP) fetch(p + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * p][j][c]), __float_as_uint(acc[2 * p + 1][j][c]),
                                                                false, false);
                v[c] = __uint_as_float(r[0]);
                v[4 + c] = __uint_as_float(r[1]);
            }
            const bool ok = okm[j] && okn;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = fmaxf(v[e] + bv[e], lo);
                if constexpr (GATE) {
                    const uint32_t gw = gq[p & 1][j][e >> 1];
                    const float gv = from_bits16<T>((uint16_t)((e & 1) ? (gw >> 16) : (gw & 0xffffu)));
                    x = (gv > 0.f) ? x : 0.f;
                }
                if constexpr (SCALE) x *= sc[j][e];
                v[e] = x;
            }
            if (do_cs) {
#pragma unroll
                for (int e = 0; e < 8; ++e) cs[e] += ok ? v[e] : 0.f;
            }
            if (RAGGED && ok && !a.abl_ep && n + 8 > a.Co) {   // ragged last piece: element by element
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (n + e < a.Co) {
                        if (out32) ((float*)a.out)[(long)mrow[j] * a.ldo + n + e] = v[e];
                        else ((uint16_t*)a.out)[(long)mrow[j] * a.ldo + n + e] = to_bits16<T>(v[e]);
                    }
                }
            } else if (ok && !a.abl_ep) {
                if (out32) {
                    float* o = (float*)a.out + (long)mrow[j] * a.ldo + n;
                    *(f32x4_t*)o = *(const f32x4_t*)&v[0];
                    *(f32x4_t*)(o + 4) = *(const f32x4_t*)&v[4];
                } else {
                    ep_u32x4_t pk;
                    pk.x = pack2<T>(v[0], v[1]);
                    pk.y = pack2<T>(v[2], v[3]);
                    pk.z = pack2<T>(v[4], v[5]);
                    pk.w = pack2<T>(v[6], v[7]);
                    *(ep_u32x4_t*)((uint16_t*)a.out + (long)mrow[j] * a.ldo + n) = pk;
                }
            }
        }
        if constexpr (GATE) {
            if (p + 2 < NP) fetch_gate(p + 2, p & 1);
        }
        if (do_cs) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = row16_sum_w(cs[e]);
                if (r16 == 0) pws[((g * NP) + p) * 8 + e] = x;
            }
        }
    }
}

// the 4 x 2 wave layout of conv_igemm_v2 / conv_igemm_wide / conv3x3_wide_rows / proj_gemm_stream: wave (wm, wn) owns the block at
// (m0 + 64 wm, n0 + (BN / 2) wn); column sums: the four pixel groups of a column are added in ascending order by one thread
template <typename T, int WNF, bool GATE, bool SCALE, typename Args>
__device__ __forceinline__ void tile_epilogue_direct(const Args& a, f32x4_t (&acc)[WNF][4], char* smem, int tid, int wm, int wn,
                                                     int g, int r16, int m0, int n0) {
    constexpr int BN = 32 * WNF, NP = WNF / 2;
    const bool do_cs = a.colsum != nullptr;
    float* const pw = (float*)smem;                            // [8 waves][4 g][NP][8] row sums (column sums only)
    if (do_cs) __syncthreads();                                // every wave has left the operand ring
    tile_epilogue_block<T, WNF, GATE, SCALE>(a, acc, g, r16, m0 + wm * 64, n0 + wn * (BN / 2),
                                             do_cs ? pw + ((wm * 2 + wn) * 4) * NP * 8 : nullptr);
    if (do_cs) {
        __syncthreads();
        if (tid < BN && n0 + tid < a.Co) {
            const int wnc = tid / (BN / 2), rem = tid % (BN / 2);
            const int pc = rem >> 5, r2 = rem & 31;
            const int gc = (r2 >> 4) | (((r2 >> 3) & 1) << 1), ec = r2 & 7;        // inverse of cl = 16 (g & 1) + 8 (g >> 1)
            float t = 0.f;
#pragma unroll
            for (int wmc = 0; wmc < 4; ++wmc) t += pw[((((wmc * 2 + wnc) * 4 + gc) * NP) + pc) * 8 + ec];
            if (a.cslab) a.cslab[(long)(m0 >> 8) * a.Co + n0 + tid] = t;
            else if (t != 0.f) atomicAdd(a.colsum + n0 + tid, t);
        }
    }
}

// split-K slab: the raw fp32 partial sums of this split, whole 32-B pieces per lane
template <int WNF, typename Args>
__device__ __forceinline__ void tile_epilogue_raw(const Args& a, f32x4_t (&acc)[WNF][4], int wm, int wn, int g, int r16, int m0,
                                                  int n0, int split) {
    constexpr int BN = 32 * WNF, NP = WNF / 2;
    float* const slab = a.ws + (size_t)split * a.M * a.Co;
    const int nw = n0 + wn * (BN / 2) + 16 * (g & 1) + 8 * (g >> 1);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int n = nw + 32 * p;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + wm * 64 + 16 * j + r16;
            float v[8];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * p][j][c]), __float_as_uint(acc[2 * p + 1][j][c]),
                                                                false, false);
                v[c] = __uint_as_float(r[0]);
                v[4 + c] = __uint_as_float(r[1]);
            }
            if (m < a.M && n < a.Co && !a.abl_ep) {
                float* o = slab + (size_t)m * a.Co + n;
                *(f32x4_t*)o = *(const f32x4_t*)&v[0];
                *(f32x4_t*)(o + 4) = *(const f32x4_t*)&v[4];
            }
        }
    }
}

}  // namespace
