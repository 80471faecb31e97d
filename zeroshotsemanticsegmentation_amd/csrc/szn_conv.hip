// szn_conv.hip -- stride-1 convolution forward / dgrad / wgrad as implicit GEMM on MFMA (gfx950).
//
// Replaces nn.Conv2d for conv1_2..conv5_3, fc6, fc7, score_fr||seenmask_score
// (reference models.py:45-97 construction, :117-149 forward; backward via loss.backward(),
// trainer_fcn.py:157).
//
// Forward / dgrad kernel (conv_igemm):
//   GEMM view  out[m][n] = sum_k A[m][k] * W[n][k],  m = (b,oh,ow), n = cout, k = (kh,kw,cin).
//   Block = 256 threads = 4 waves (2x2), tile 128 pixels x 128 couts, K advanced in 128-byte
//   chunks (64 bf16 / 32 f32 of one filter tap).  Both operands are staged through LDS as
//   [row][128 B] with the 16-B chunk index XOR-swizzled by (row & 7) so that ds_read_b128 of an
//   MFMA fragment (16 rows x one chunk) is bank-conflict free.  The MFMA "A" operand is the WEIGHT
//   fragment and "B" the PIXEL fragment, so each lane ends up holding 4 consecutive couts of one
//   pixel: the NHWC epilogue store is 8/16 contiguous bytes per lane.
//   bf16: v_mfma_f32_16x16x32_bf16 (one per fragment pair per 32 k); f32: 4 x v_mfma_f32_16x16x4_f32
//   per 16-B pair (exact fp32 fmaf chain).
// dgrad is the same kernel run on dout with the flipped/transposed weight image (szn_pack_weight_dgrad)
// and pad' = KH-1-pad; its epilogue applies the ReLU gate (forward input > 0) and the dropout factor.
//
// wgrad kernel (conv_wgrad): D[co][ci] = sum_pixels dout[p][co] * in[p+tap][ci] per filter tap; both
// operands are pixel-major so the contraction index is the slow one: tiles are staged as
// [pixel][128 ch] and fragments are fetched with ds_read_b64_tr_b16 (bf16, hardware transpose) or
// plain ds_read_b32 (f32, one element per lane).  Split over pixels; partial tiles are added to the
// fp32 OHWI gradient with global atomics.
#include "szn_common.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

namespace {

// ------------------------------------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_raw> {
    // one 16x16x32 step: lane (r = lane&15, g = lane>>4) holds k = 8*g .. 8*g+7 of row r
    static __device__ __forceinline__ void run(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                      __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct Mma<f16_raw> {
    static __device__ __forceinline__ void run(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) { acc = mfma16<f16_raw>(a, b, acc); }
};
template <> struct Mma<float> {
    // four 16x16x4 steps; step e uses element e of every lane's 16-B chunk (k = 4*g + e)
    static __device__ __forceinline__ void run(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

struct ConvArgs {
    const char* in; const char* w; const float* bias; const char* gate; const float* cscale; char* out;
    int B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, pad;
    int ldi, ldo, ldg, relu, out_f32;
    int M, HoWo, mtiles, ntiles;
};

// bijective XCD-aware remap: consecutive logical tiles land on the same XCD (block b runs on XCD b%8)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

constexpr int TILE = 128;           // tile rows for both operands
constexpr int ROWB = 128;           // bytes of K per LDS row
constexpr int OPB = TILE * ROWB;    // bytes per operand per buffer (16 KiB)

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_igemm(ConvArgs a) {
    constexpr int CH = elem<T>::kPer16B;   // elements per 16-B chunk
    constexpr int BKE = 8 * CH;            // elements per 128-B K chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][pixels 16K | weights 16K]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;          // wave -> 64-pixel x 64-cout quadrant
    const int g = lane >> 4, r16 = lane & 15;

    const int nwg = a.mtiles * a.ntiles;
    const int lid = xcd_remap(blockIdx.x, nwg);
    const int nt = lid % a.ntiles, mt = lid / a.ntiles;
    const int m0 = mt * TILE, n0 = nt * TILE;

    const T* __restrict__ in = (const T*)a.in;
    const T* __restrict__ w = (const T*)a.w;

    // ---- per-thread staging assignment: rows crow+32*i, 16-B chunk cchunk ----
    const int crow = tid >> 3, cchunk = tid & 7;
    long pbase[4];
    int ih0[4], iw0[4];
    long wbase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + crow + 32 * i;
        if (m < a.M) {
            const int b = m / a.HoWo, r = m - b * a.HoWo;
            const int oh = r / a.Wo, ow = r - oh * a.Wo;
            ih0[i] = oh - a.pad; iw0[i] = ow - a.pad;
            pbase[i] = ((long)(b * a.Hi + ih0[i]) * a.Wi + iw0[i]) * a.ldi + cchunk * CH;
        } else {
            ih0[i] = -(1 << 24); iw0[i] = -(1 << 24); pbase[i] = 0;
        }
        const int n = n0 + crow + 32 * i;
        wbase[i] = (n < a.Co) ? (long)n * a.KH * a.KW * a.Ci + cchunk * CH : -1;
    }
    const int swz_st = (cchunk ^ (crow & 7)) << 4;    // (row & 7) == (crow & 7) for all 4 rows

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    u32x4_t ra[4], rb[4];
    int kh = 0, kw = 0, c0 = 0;
    const int nK = a.KH * a.KW * (a.Ci / BKE);

    auto gload = [&]() {
        const long tapoff = ((long)kh * a.Wi + kw) * a.ldi + c0;
        const long wtap = (long)(kh * a.KW + kw) * a.Ci + c0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ih = ih0[i] + kh, iw = iw0[i] + kw;
            const bool ok = (unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi;
            ra[i] = ok ? *(const u32x4_t*)(in + pbase[i] + tapoff) : u32x4_t{0, 0, 0, 0};
            rb[i] = (wbase[i] >= 0) ? *(const u32x4_t*)(w + wbase[i] + wtap) : u32x4_t{0, 0, 0, 0};
        }
        c0 += BKE;
        if (c0 >= a.Ci) { c0 = 0; if (++kw >= a.KW) { kw = 0; ++kh; } }
    };
    auto lstore = [&](int buf) {
        char* sp = smem + buf * (2 * OPB);
        char* sw = sp + OPB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = crow + 32 * i;
            *(u32x4_t*)(sp + row * ROWB + swz_st) = ra[i];
            *(u32x4_t*)(sw + row * ROWB + swz_st) = rb[i];
        }
    };

    gload();
    lstore(0);
    __syncthreads();

    for (int kc = 0; kc < nK; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < nK) gload();
        const char* sp = smem + cur * (2 * OPB) + (wm * 64 + r16) * ROWB;
        const char* sw = smem + cur * (2 * OPB) + OPB + (wn * 64 + r16) * ROWB;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int off = (((s * 4 + g) ^ (r16 & 7)) << 4);
            u32x4_t wf[4], pf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *(const u32x4_t*)(sw + i * 16 * ROWB + off);
#pragma unroll
            for (int j = 0; j < 4; ++j) pf[j] = *(const u32x4_t*)(sp + j * 16 * ROWB + off);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<T>::run(acc[i][j], wf[i], pf[j]);
        }
        if (kc + 1 < nK) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds couts nb..nb+3 (rows of D) of pixel m (column of D) ----
    const T* __restrict__ gate = (const T*)a.gate;
    const bool vec_ok = ((a.ldo & 3) == 0) && (!a.gate || (a.ldg & 3) == 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + r16;
        if (m >= a.M) continue;
        const int b = a.cscale ? (m / a.HoWo) : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nb = n0 + wn * 64 + i * 16 + g * 4;
            if (nb >= a.Co) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = nb + e;
                float x = acc[i][j][e];
                if (n < a.Co) {
                    if (a.bias) x += a.bias[n];
                    if (a.relu) x = fmaxf(x, 0.f);
                    if (gate) x = (elem<T>::ld(gate + (long)m * a.ldg + n) > 0.f) ? x : 0.f;
                    if (a.cscale) x *= a.cscale[(long)b * a.Co + n];
                }
                v[e] = x;
            }
            if (a.out_f32 || sizeof(T) == 4) {
                float* o = (float*)a.out + (long)m * a.ldo + nb;
                if (vec_ok && nb + 3 < a.Co) {
                    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (nb + e < a.Co) o[e] = v[e];
                }
            } else {
                uint16_t* o = (uint16_t*)a.out + (long)m * a.ldo + nb;
                if (vec_ok && nb + 3 < a.Co) {
                    u32x2_t pk;
                    pk.x = pack2<T>(v[0], v[1]);
                    pk.y = pack2<T>(v[2], v[3]);
                    *(u32x2_t*)o = pk;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (nb + e < a.Co) o[e] = to_bits16<T>(v[e]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight repack for dgrad: wT[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci]
template <typename T>
__global__ void pack_dgrad_kernel(const T* __restrict__ w, T* __restrict__ wT, int Co, int KH, int KW, int Ci) {
    // one block per (tap, 32-co tile, 32-ci tile); LDS transpose keeps both sides coalesced
    __shared__ T tile[32][33];
    const int tap = blockIdx.z, kh = tap / KW, kw = tap - kh * KW;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        if (co < Co && ci < Ci) tile[r][tx] = w[((long)(co * KH + kh) * KW + kw) * Ci + ci];
    }
    __syncthreads();
    const int tapT = (KH - 1 - kh) * KW + (KW - 1 - kw);
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (co < Co && ci < Ci) wT[((long)ci * KH * KW + tapT) * Co + co] = tile[tx][r];
    }
}

// 16-bit fast path (Co, Ci multiples of 64): 64 x 64 tiles, 16-B global loads and stores on both sides (full 128-B
// rows), transpose through LDS with 2-B accesses ([64][66] pitch: conflict-free column reads)
__global__ __launch_bounds__(256) void pack_dgrad16_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ wT, int Co,
                                                           int KH, int KW, int Ci) {
    __shared__ uint16_t tile[64][66];
    const int tap = blockIdx.z, kh = tap / KW, kw = tap - kh * KW;
    const int co0 = blockIdx.y * 64, ci0 = blockIdx.x * 64;
    const int c8 = threadIdx.x & 7, r0 = threadIdx.x >> 3;       // 8 x 16-B chunks per 128-B row, 32 rows per pass
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = r0 + 32 * p;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);                    // (rows past a ragged Co: Co is a multiple of 8 there, not of 64)
        if (co0 + r < Co) v = *(const uint4*)(w + ((long)((co0 + r) * KH + kh) * KW + kw) * Ci + ci0 + c8 * 8);
        uint32_t* d32 = (uint32_t*)&tile[r][c8 * 8];             // row pitch 132 B: 4-B aligned
        d32[0] = v.x; d32[1] = v.y; d32[2] = v.z; d32[3] = v.w;
    }
    __syncthreads();
    const int tapT = (KH - 1 - kh) * KW + (KW - 1 - kw);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int ci = r0 + 32 * p;                               // output row = input channel
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = (uint32_t)tile[c8 * 8 + 2 * e][ci] | ((uint32_t)tile[c8 * 8 + 2 * e + 1][ci] << 16);
        if (co0 + c8 * 8 < Co) *(uint4*)(wT + ((long)(ci0 + ci) * KH * KW + tapT) * Co + co0 + c8 * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// several layers in one launch (the per-step refresh of every flipped / transposed image: 15 of the 17 launches are a few blocks
// each and cost their ~4.6 us dispatch floor): block -> (item, ci tile, co tile, tap) through the items' block prefix
constexpr int kPackMax = 24;
struct PackItem { const uint16_t* w; uint16_t* wT; int Co, K, Ci, block0; };
struct PackBatch { PackItem it[kPackMax]; int n; };

__global__ __launch_bounds__(256) void pack_dgrad16_batch_kernel(PackBatch pb) {
    __shared__ uint16_t tile[64][66];
    int i = 0;
    while (i + 1 < pb.n && (int)blockIdx.x >= pb.it[i + 1].block0) ++i;
    const uint16_t* __restrict__ w = pb.it[i].w;
    uint16_t* __restrict__ wT = pb.it[i].wT;
    const int Co = pb.it[i].Co, K = pb.it[i].K, Ci = pb.it[i].Ci;
    int lb = blockIdx.x - pb.it[i].block0;
    const int nci = Ci / 64, nco = Co / 64;
    const int cit = lb % nci; lb /= nci;
    const int cot = lb % nco;
    const int tap = lb / nco, kh = tap / K, kw = tap - kh * K;
    const int co0 = cot * 64, ci0 = cit * 64;
    const int c8 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = r0 + 32 * p;
        const uint4 v = *(const uint4*)(w + ((long)((co0 + r) * K + kh) * K + kw) * Ci + ci0 + c8 * 8);
        uint32_t* d32 = (uint32_t*)&tile[r][c8 * 8];
        d32[0] = v.x; d32[1] = v.y; d32[2] = v.z; d32[3] = v.w;
    }
    __syncthreads();
    const int tapT = (K - 1 - kh) * K + (K - 1 - kw);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int ci = r0 + 32 * p;
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = (uint32_t)tile[c8 * 8 + 2 * e][ci] | ((uint32_t)tile[c8 * 8 + 2 * e + 1][ci] << 16);
        *(uint4*)(wT + ((long)(ci0 + ci) * K * K + tapT) * Co + co0 + c8 * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    const char* dout; const char* in; float* dw;
    int B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, pad;
    int ldi, ldd;
    int M, HoWo;
    int kspan;        // pixels per split
    int nsplit, cotiles, citiles;
};

template <typename T> struct WgTraits;
template <> struct WgTraits<bf16_raw> { static constexpr int KP = 32; static constexpr int TRS = 256 + 32; };
template <> struct WgTraits<f16_raw> { static constexpr int KP = 32; static constexpr int TRS = 256 + 32; };
template <> struct WgTraits<float>    { static constexpr int KP = 16; static constexpr int TRS = 512 + 64; };

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_wgrad(WgradArgs a) {
    constexpr int CH = elem<T>::kPer16B;
    constexpr int KP = WgTraits<T>::KP;         // pixels per K step
    constexpr int TRS = WgTraits<T>::TRS;       // LDS bytes per pixel row (128 ch + pad)
    constexpr int CPR = 128 / CH;               // 16-B chunks per pixel row (16 bf16 / 32 f32)
    constexpr int OPBW = KP * TRS;              // 9216 B per operand per buffer
    __shared__ __attribute__((aligned(16))) char smem[4 * OPBW];   // [2][dout | in]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;    // wave -> 64 co x 64 ci quadrant
    const int g = lane >> 4, r16 = lane & 15;

    int bid = blockIdx.x;
    const int cit = bid % a.citiles; bid /= a.citiles;
    const int cot = bid % a.cotiles; bid /= a.cotiles;
    const int tap = bid % (a.KH * a.KW); const int split = bid / (a.KH * a.KW);
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int co0 = cot * 128, ci0 = cit * 128;
    const int mbeg = split * a.kspan;
    const int mend = min(a.M, mbeg + a.kspan);
    if (mbeg >= mend) return;

    const T* __restrict__ dout = (const T*)a.dout;
    const T* __restrict__ in = (const T*)a.in;

    // staging: chunk q = tid + 256*i, i = 0,1 -> pixel row q / CPR, 16-B chunk q % CPR
    int srow[2], scc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int q = tid + 256 * i; srow[i] = q / CPR; scc[i] = q % CPR; }

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    u32x4_t rd[2], ri[2];
    int mcur = mbeg;
    auto gload = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mcur + srow[i];
            u32x4_t vd = u32x4_t{0, 0, 0, 0}, vi = u32x4_t{0, 0, 0, 0};
            if (m < mend) {
                const int co = co0 + scc[i] * CH;
                if (co + CH <= a.ldd && co < a.Co) vd = *(const u32x4_t*)(dout + (long)m * a.ldd + co);
                const int b = m / a.HoWo, r = m - b * a.HoWo;
                const int oh = r / a.Wo, ow = r - oh * a.Wo;
                const int ih = oh + kh - a.pad, iw = ow + kw - a.pad;
                const int ci = ci0 + scc[i] * CH;
                if ((unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi && ci < a.Ci)
                    vi = *(const u32x4_t*)(in + ((long)(b * a.Hi + ih) * a.Wi + iw) * a.ldi + ci);
            }
            rd[i] = vd; ri[i] = vi;
        }
        mcur += KP;
    };
    auto lstore = [&](int buf) {
        char* sd = smem + buf * (2 * OPBW);
        char* si = sd + OPBW;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(u32x4_t*)(sd + srow[i] * TRS + scc[i] * 16) = rd[i];
            *(u32x4_t*)(si + srow[i] * TRS + scc[i] * 16) = ri[i];
        }
    };

    const int nK = (mend - mbeg + KP - 1) / KP;
    gload();
    lstore(0);
    __syncthreads();
    for (int kc = 0; kc < nK; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < nK) gload();
        const char* sd = smem + cur * (2 * OPBW);
        const char* si = sd + OPBW;
        if constexpr (sizeof(T) == 2) {
            // transpose reads: within a 16-lane group lane q supplies row (q>>2), cols 4*(q&3)..+3 and
            // receives column q of the 4x16 block; k of read h (h=0,1) = 16*h + 4*g + {0..3}
            u32x4_t df[4], xf[4];
            const int rowsel = (g * 4 + (r16 >> 2)) * TRS + (r16 & 3) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cb = (wm * 64 + i * 16) * 2;
                typedef __attribute__((address_space(3))) bf16x4_t* lp_t;
                bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp_t)(sd + rowsel + cb));
                bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp_t)(sd + 16 * TRS + rowsel + cb));
                u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
                df[i] = u32x4_t{l2.x, l2.y, h2.x, h2.y};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cb = (wn * 64 + j * 16) * 2;
                typedef __attribute__((address_space(3))) bf16x4_t* lp_t;
                bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp_t)(si + rowsel + cb));
                bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lp_t)(si + 16 * TRS + rowsel + cb));
                u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
                xf[j] = u32x4_t{l2.x, l2.y, h2.x, h2.y};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<T>::run(acc[i][j], df[i], xf[j]);
        } else {
            // f32: one element per lane per 16x16x4 step: lane (r16, g) reads [k = 4*s + g][ch r16]
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float df[4], xf[4];
                const int ro = (4 * s + g) * TRS + r16 * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) df[i] = *(const float*)(sd + ro + (wm * 64 + i * 16) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[j] = *(const float*)(si + ro + (wn * 64 + j * 16) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(df[i], xf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (kc + 1 < nK) lstore(cur ^ 1);
        __syncthreads();
    }

    // D[co][ci]: lane holds rows co = g*4+e, column ci = r16
    const long tapstride = (long)a.Ci;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = co0 + wm * 64 + i * 16 + g * 4 + e;
            if (co >= a.Co) continue;
            float* row = a.dw + ((long)(co * a.KH + kh) * a.KW + kw) * tapstride;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ci = ci0 + wn * 64 + j * 16 + r16;
                if (ci < a.Ci) atomicAdd(row + ci, acc[i][j][e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// db[n] += sum_m dout[m][n]: 16-B loads, thread = (16-B channel chunk, row lane), LDS reduce, one atomic per
// (block, channel)
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(const T* __restrict__ dout, float* __restrict__ db, long M, int Co,
                                                        int ldd, int rows_per_block, float* __restrict__ slab) {
    constexpr int CH = elem<T>::kPer16B;
    __shared__ float red[256 * CH];
    const int cpr = (Co + CH - 1) / CH;                  // 16-B chunks per row
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    for (int cg = 0; cg < cpr; cg += 256) {              // chunk groups (only Co > 256*CH needs more than one)
        const int ncg = min(256, cpr - cg);
        const int rl = 256 / ncg;                        // row lanes
        const int chunk = threadIdx.x % ncg, lr = threadIdx.x / ncg;
        float acc[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) acc[e] = 0.f;
        if (lr < rl) {
            const T* p = dout + (long)(cg + chunk) * CH;
            for (long r = r0 + lr; r < r1; r += rl) {
                const u32x4_t v = *(const u32x4_t*)(p + r * ldd);
                const T* ve = (const T*)&v;
#pragma unroll
                for (int e = 0; e < CH; ++e) acc[e] += elem<T>::ld(ve + e);
            }
        }
        __syncthreads();
        if (lr < rl) {
#pragma unroll
            for (int e = 0; e < CH; ++e) red[(lr * ncg + chunk) * CH + e] = acc[e];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < ncg * CH; i += 256) {
            float s = 0.f;
            for (int l = 0; l < rl; ++l) s += red[l * ncg * CH + i];
            const int n = cg * CH + i;
            if (n < Co) {
                if (slab) slab[(long)blockIdx.x * Co + n] = s;       // one partial row per block (fixed-order reduce later)
                else if (s != 0.f) atomicAdd(db + n, s);
            }
        }
    }
}

template <typename T>
int launch_conv(const ConvArgs& a, hipStream_t st) {
    const size_t lds = 4 * OPB;   // 64 KiB
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_igemm<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(conv_igemm<T>, dim3(a.mtiles * a.ntiles), dim3(256), lds, st, a);
    SZN_CHECK_LAUNCH("conv_igemm");
    return SZN_OK;
}

int check_desc(const szn_conv_desc_t* d) {
    if (!d) SZN_FAIL(SZN_ERR_ARG, "conv: null descriptor");
    if (d->dtype != SZN_F32 && !szn_is16(d->dtype)) SZN_FAIL(SZN_ERR_ARG, "conv: bad dtype %d", d->dtype);
    if (d->B <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ci <= 0 || d->Co <= 0 || d->KH <= 0 || d->KW <= 0 || d->pad < 0)
        SZN_FAIL(SZN_ERR_ARG, "conv: non-positive dimension");
    if (d->Ho != d->Hi + 2 * d->pad - d->KH + 1 || d->Wo != d->Wi + 2 * d->pad - d->KW + 1 || d->Ho <= 0 || d->Wo <= 0)
        SZN_FAIL(SZN_ERR_ARG, "conv: Ho/Wo (%d,%d) inconsistent with Hi/Wi (%d,%d) pad %d k %dx%d", d->Ho, d->Wo, d->Hi,
                 d->Wi, d->pad, d->KH, d->KW);
    if ((long)d->B * d->Ho * d->Wo >= (1L << 31) || (long)d->B * d->Hi * d->Wi >= (1L << 31))
        SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv: more than 2^31 pixels");
    return SZN_OK;
}

}  // namespace

// first-generation kernel (register-staged, 128x128 tile): fallback of szn_conv2d_fwd (szn_conv_igemm.hip)
int szn_conv2d_fwd_v1(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias, const void* gate,
                      const float* chan_scale, void* out, szn_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    const int bke = szn_is16(d->dtype) ? 64 : 32;
    if (d->Ci % bke) SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv2d_fwd: Ci=%d must be a multiple of %d", d->Ci, bke);
    if (d->ldi < d->Ci || d->ldo < d->Co || (d->ldi % (bke / 8)))
        SZN_FAIL(SZN_ERR_ARG, "conv2d_fwd: bad pixel strides ldi=%d ldo=%d", d->ldi, d->ldo);
    if (!in || !w || !out) SZN_FAIL(SZN_ERR_ARG, "conv2d_fwd: null pointer");
    if (((uintptr_t)in | (uintptr_t)w | (uintptr_t)out) & 15) SZN_FAIL(SZN_ERR_ARG, "conv2d_fwd: pointers must be 16-B aligned");
    ConvArgs a;
    a.in = (const char*)in; a.w = (const char*)w; a.bias = bias; a.gate = (const char*)gate; a.cscale = chan_scale;
    a.out = (char*)out;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co;
    a.KH = d->KH; a.KW = d->KW; a.pad = d->pad; a.ldi = d->ldi; a.ldo = d->ldo; a.ldg = d->ldg;
    a.relu = d->relu; a.out_f32 = d->out_f32;
    a.M = d->B * d->Ho * d->Wo; a.HoWo = d->Ho * d->Wo;
    a.mtiles = szn_div_up(a.M, TILE); a.ntiles = szn_div_up(a.Co, TILE);
    rc = d->dtype == SZN_BF16 ? launch_conv<bf16_raw>(a, (hipStream_t)stream)
       : d->dtype == SZN_F16 ? launch_conv<f16_raw>(a, (hipStream_t)stream) : launch_conv<float>(a, (hipStream_t)stream);
    if (rc || !d->colsum) return rc;
    if (d->out_f32 && szn_is16(d->dtype)) SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv2d_fwd(v1): colsum with out_f32 is unsupported");
    return szn_bias_grad_slab(d->dtype, a.M, d->Co, d->ldo, out, d->colsum, 1, d->colsum_slab, d->colsum_slab_rows, nullptr, stream);   // fallback path: separate pass
}

extern "C" int szn_pack_weight_dgrad(int dtype, int Co, int KH, int KW, int Ci, const void* w, void* wT,
                                     szn_stream_t stream) {
    if (!w || !wT || Co <= 0 || Ci <= 0 || KH <= 0 || KW <= 0) SZN_FAIL(SZN_ERR_ARG, "pack_weight_dgrad: bad argument");
    dim3 grid(szn_div_up(Ci, 32), szn_div_up(Co, 32), KH * KW);
    if (szn_is16(dtype) && (Co & 7) == 0 && (Ci & 63) == 0 && !(((uintptr_t)w | (uintptr_t)wT) & 15)) {
        hipLaunchKernelGGL(pack_dgrad16_kernel, dim3(Ci / 64, (Co + 63) / 64, KH * KW), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)w, (uint16_t*)wT, Co, KH, KW, Ci);
        SZN_CHECK_LAUNCH("pack_dgrad16_kernel");
        return SZN_OK;
    }
    if (szn_is16(dtype))
        hipLaunchKernelGGL(pack_dgrad_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)w,
                           (uint16_t*)wT, Co, KH, KW, Ci);
    else if (dtype == SZN_F32)
        hipLaunchKernelGGL(pack_dgrad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)w, (float*)wT,
                           Co, KH, KW, Ci);
    else
        SZN_FAIL(SZN_ERR_ARG, "pack_weight_dgrad: bad dtype %d", dtype);
    SZN_CHECK_LAUNCH("pack_dgrad_kernel");
    return SZN_OK;
}

extern "C" int szn_pack_weight_dgrad_batch(int dtype, int n, const void* const* w, void* const* wT, const int* Co, const int* K,
                                           const int* Ci, szn_stream_t stream) {
    if (n <= 0 || !w || !wT || !Co || !K || !Ci) SZN_FAIL(SZN_ERR_ARG, "pack_weight_dgrad_batch: bad argument");
    if (!szn_is16(dtype)) SZN_FAIL(SZN_ERR_UNSUPPORTED, "pack_weight_dgrad_batch: 16-bit images only (use szn_pack_weight_dgrad)");
    if (n > kPackMax) SZN_FAIL(SZN_ERR_UNSUPPORTED, "pack_weight_dgrad_batch: at most %d layers per call", kPackMax);
    PackBatch pb;
    long blocks = 0;
    for (int i = 0; i < n; ++i) {
        if (!w[i] || !wT[i] || Co[i] <= 0 || K[i] <= 0 || Ci[i] <= 0) SZN_FAIL(SZN_ERR_ARG, "pack_weight_dgrad_batch: bad item %d", i);
        if ((Co[i] & 63) || (Ci[i] & 63) || (((uintptr_t)w[i] | (uintptr_t)wT[i]) & 15))
            SZN_FAIL(SZN_ERR_UNSUPPORTED, "pack_weight_dgrad_batch: item %d needs Co, Ci multiples of 64 and 16-B aligned images", i);
        pb.it[i].w = (const uint16_t*)w[i]; pb.it[i].wT = (uint16_t*)wT[i];
        pb.it[i].Co = Co[i]; pb.it[i].K = K[i]; pb.it[i].Ci = Ci[i]; pb.it[i].block0 = (int)blocks;
        blocks += (long)(Ci[i] / 64) * (Co[i] / 64) * K[i] * K[i];
        if (blocks > 0x7fffffffL) SZN_FAIL(SZN_ERR_UNSUPPORTED, "pack_weight_dgrad_batch: too many tiles");
    }
    pb.n = n;
    hipLaunchKernelGGL(pack_dgrad16_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pb);
    SZN_CHECK_LAUNCH("pack_dgrad16_batch_kernel");
    return SZN_OK;
}

extern "C" int szn_conv2d_dgrad(const szn_conv_desc_t* d, const void* dout, const void* wT, const void* gate,
                                const float* chan_scale, void* din, szn_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    // swap roles: the "input" of the dgrad conv is dout [B][Ho][Wo][Co], its "output" din [B][Hi][Wi][Ci]
    szn_conv_desc_t s = *d;
    s.Hi = d->Ho; s.Wi = d->Wo; s.Ci = d->Co;
    s.Ho = d->Hi; s.Wo = d->Wi; s.Co = d->Ci;
    s.pad = d->KH - 1 - d->pad;
    if (d->KH != d->KW || s.pad < 0) SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv2d_dgrad: needs a square kernel with pad <= K-1");
    s.ldi = d->ldo; s.ldo = d->ldi;
    s.relu = 0;
    return szn_conv2d_fwd(&s, dout, wT, nullptr, gate, chan_scale, din, stream);
}

// ------------------------------------------------------------------------------------------------
// dgrad of a large-window convolution (fc6: 7x7 valid on a 23x23 map) as GEMM + col2im.  As a convolution over the
// (K-1)-padded dout map it executes 1.83x its algorithmic FLOPs (most taps of the border pixels are padding); as
//   Y[m = (b,oh,ow)][(kh,kw,ci)] = sum_co dout[m][co] * w[co][kh][kw][ci]           (one GEMM, N = KH*KW*Ci)
//   din[b][ih][iw][ci]           = sum_{kh,kw} Y[(b, ih+pad-kh, iw+pad-kw)][(kh,kw,ci)]
// it executes exactly them; every Y element is read once by col2im (16-B loads along ci).
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ Y, T* __restrict__ din, int B, int Hi, int Wi,
                                                     int Ci, int Ho, int Wo, int KH, int KW, int pad, int ldi) {
    const int c4n = Ci >> 2;
    const long total = (long)B * Hi * Wi * c4n;
    const long N = (long)KH * KW * Ci;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % c4n);
        const long p = idx / c4n;
        const int iw = (int)(p % Wi);
        const long t = p / Wi;
        const int ih = (int)(t % Hi), b = (int)(t / Hi);
        // taps whose output pixel exists: kh in [kh0, kh1], kw in [kw0, kw1] (ranges instead of per-tap tests: the loads of a row of taps
        // are independent and issue back to back); two running sums per kernel row, added in a fixed order
        const int kh0 = max(0, ih + pad - (Ho - 1)), kh1 = min(KH - 1, ih + pad);
        const int kw0 = max(0, iw + pad - (Wo - 1)), kw1 = min(KW - 1, iw + pad);
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        for (int kh = kh0; kh <= kh1; ++kh) {
            const int oh = ih + pad - kh;
            const float* yrow = Y + (((long)b * Ho + oh) * Wo + (iw + pad)) * N + (long)kh * KW * Ci + c4 * 4;
            f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
            int kw = kw0;
#pragma unroll 4
            for (; kw + 1 <= kw1; kw += 2) {
                a0 += *(const f32x4_t*)(yrow - (long)kw * N + (long)kw * Ci);
                a1 += *(const f32x4_t*)(yrow - (long)(kw + 1) * N + (long)(kw + 1) * Ci);
            }
            if (kw <= kw1) a0 += *(const f32x4_t*)(yrow - (long)kw * N + (long)kw * Ci);
            acc += a0 + a1;
        }
        T* o = din + p * ldi + c4 * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) elem<T>::st(o + e, acc[e]);
    }
}

static int launch_col2im(const szn_conv_desc_t* d, const float* Y, void* din, szn_stream_t stream);
int szn_conv_wgrad_wide_try(const szn_conv_desc_t* d, const void* in, const void* dout, float* dw, int accumulate,
                            int min_tiles, szn_stream_t stream, const szn_adam_args_t* opt = nullptr);

extern "C" size_t szn_conv2d_dgrad_gemm_workspace_bytes(const szn_conv_desc_t* d) {
    if (!d || d->B <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->KH <= 0 || d->KW <= 0 || d->Ci <= 0) return 0;
    return (size_t)d->B * d->Ho * d->Wo * d->KH * d->KW * d->Ci * sizeof(float);
}

extern "C" int szn_conv2d_dgrad_gemm(const szn_conv_desc_t* d, const void* dout, const void* wG, void* din,
                                     szn_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dout || !wG || !din) SZN_FAIL(SZN_ERR_ARG, "conv2d_dgrad_gemm: null pointer");
    if (d->Ci & 3) SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv2d_dgrad_gemm: Ci must be a multiple of 4");
    const size_t need = szn_conv2d_dgrad_gemm_workspace_bytes(d);
    if (!d->workspace || d->workspace_bytes < need || ((uintptr_t)d->workspace & 15))
        SZN_FAIL(SZN_ERR_ARG, "conv2d_dgrad_gemm: needs a 16-B aligned workspace of %zu bytes", need);
    if (d->B * d->Ho >= 32000 || d->Wo >= 32000) SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv2d_dgrad_gemm: map too large");
    const long N = (long)d->KH * d->KW * d->Ci;
    if (N >= (1L << 31)) SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv2d_dgrad_gemm: KH*KW*Ci too large");
    // Y = dout x wG^T as a 1x1 convolution over the (B*Ho) x Wo pixel grid: "Ci" = Co, "Co" = KH*KW*Ci, f32 output
    szn_conv_desc_t g = {};
    g.dtype = d->dtype; g.B = 1; g.Hi = d->B * d->Ho; g.Wi = d->Wo; g.Ci = d->Co;
    g.Ho = g.Hi; g.Wo = g.Wi; g.Co = (int)N; g.KH = 1; g.KW = 1; g.pad = 0;
    g.ldi = d->ldo; g.ldo = (int)N; g.ldg = 0; g.relu = 0; g.out_f32 = 1;
    g.workspace = nullptr; g.workspace_bytes = 0; g.colsum = nullptr;
    rc = szn_conv2d_fwd(&g, dout, wG, nullptr, nullptr, nullptr, d->workspace, stream);
    if (rc) return rc;
    return launch_col2im(d, (const float*)d->workspace, din, stream);
}

static int launch_col2im(const szn_conv_desc_t* d, const float* Y, void* din, szn_stream_t stream) {
    const long total = (long)d->B * d->Hi * d->Wi * (d->Ci >> 2);
    long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (d->dtype == SZN_BF16)
        hipLaunchKernelGGL(col2im_kernel<bf16_raw>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           Y, (bf16_raw*)din, d->B, d->Hi, d->Wi, d->Ci, d->Ho, d->Wo, d->KH, d->KW,
                           d->pad, d->ldi);
    else if (d->dtype == SZN_F16)
        hipLaunchKernelGGL(col2im_kernel<f16_raw>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           Y, (f16_raw*)din, d->B, d->Hi, d->Wi, d->Ci, d->Ho, d->Wo, d->KH, d->KW,
                           d->pad, d->ldi);
    else
        hipLaunchKernelGGL(col2im_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           Y, (float*)din, d->B, d->Hi, d->Wi, d->Ci, d->Ho, d->Wo, d->KH, d->KW,
                           d->pad, d->ldi);
    SZN_CHECK_LAUNCH("col2im_kernel");
    return SZN_OK;
}

// dout [M][C] (16-bit) -> doutT [C][Mp], Mp = M rounded up to 8 (columns M .. Mp - 1 zero): the K-major A operand of the native dgrad
// GEMM below for pixel counts that are not a multiple of 8 (M = 289 at the reference's batch size of one image).  64 x 64 tiles through
// LDS; C is a multiple of 64.
__global__ __launch_bounds__(256) void transpose_pad16_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int M, int Mp,
                                                              int C) {
    __shared__ uint16_t tile[64][66];
    const int m0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int c8 = threadIdx.x & 7, r0 = threadIdx.x >> 3;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = r0 + 32 * p;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (m0 + r < M) v = *(const uint4*)(src + (long)(m0 + r) * C + c0 + c8 * 8);
        uint32_t* d32 = (uint32_t*)&tile[r][c8 * 8];
        d32[0] = v.x; d32[1] = v.y; d32[2] = v.z; d32[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int c = r0 + 32 * p;
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = (uint32_t)tile[c8 * 8 + 2 * e][c] | ((uint32_t)tile[c8 * 8 + 2 * e + 1][c] << 16);
        if (m0 + c8 * 8 < Mp) *(uint4*)(dst + (long)(c0 + c) * Mp + m0 + c8 * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ---- the same dgrad on the filter bank in its FORWARD layout (round 3) -------------------------------------------------------------
// szn_conv2d_dgrad_gemm needs wG = the plain transpose of the [Co][KH*KW*Ci] forward image: for fc6 a 205 MB read + 205 MB write per
// step (85 of the 115 us of szn_pack_weight_dgrad_batch) just to make co the contiguous index.  conv_wgrad_wide multiplies two
// K-MAJOR operands (D[a][b] = sum_k A[k][a] B[k][b], transposing LDS reads on both): with k = co, A = dout^T [Co][M] and B = the
// forward image [Co][N] it yields Y[m][n] directly -- only the 19 MB dout is transposed.  Same workspace Y, same col2im.
static bool dgrad_native_shape_ok(const szn_conv_desc_t* d) {
    if (!d || !szn_is16(d->dtype) || d->KH != d->KW || d->ldo != d->Co || (d->Ci & 3)) return false;
    const long M = (long)d->B * d->Ho * d->Wo, N = (long)d->KH * d->KW * d->Ci;
    if (M < 256 || N < 256 || (N & 7) || (d->Co & 63) || M >= (1L << 22) || N >= (1L << 31)) return false;
    const long cot = (M + 255) / 256, cit = (N + 255) / 256;
    if (cot * cit < 96) return false;                                                   // conv_wgrad_wide's own admission rule
    // padding waste of the last pixel tile: at most a quarter -- or at most 256 blocks in all (one round of the chip: the launch then
    // takes one K loop whatever the rows of its second pixel tile hold, M = 289 at B = 1, and re-transposing the 205 MB filter bank
    // for the packed form costs as much as that K loop)
    if (cot * 256 * cit * 256 > M * N * 5 / 4 && cot * cit > 256) return false;
    return (size_t)d->Co * N * 2 < 0xffff0000ul && (size_t)d->Co * M * 2 < 0xffff0000ul;
}

extern "C" int szn_conv2d_dgrad_gemm_native_supported(const szn_conv_desc_t* d) { return dgrad_native_shape_ok(d) ? 1 : 0; }

extern "C" size_t szn_conv2d_dgrad_gemm_native_workspace_bytes(const szn_conv_desc_t* d) {
    const size_t y = szn_conv2d_dgrad_gemm_workspace_bytes(d);
    if (!y) return 0;
    const size_t Mp = ((size_t)d->B * d->Ho * d->Wo + 7) / 8 * 8;
    return (y + 255) / 256 * 256 + (size_t)d->Co * Mp * 2;                               // Y | dout^T (rows padded to 8 pixels)
}

extern "C" int szn_conv2d_dgrad_gemm_native(const szn_conv_desc_t* d, const void* dout, const void* w, void* din,
                                            szn_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dout || !w || !din) SZN_FAIL(SZN_ERR_ARG, "conv2d_dgrad_gemm_native: null pointer");
    if (!dgrad_native_shape_ok(d)) SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv2d_dgrad_gemm_native: shape not supported (use szn_conv2d_dgrad_gemm)");
    const size_t need = szn_conv2d_dgrad_gemm_native_workspace_bytes(d);
    if (!d->workspace || d->workspace_bytes < need || ((uintptr_t)d->workspace & 15))
        SZN_FAIL(SZN_ERR_ARG, "conv2d_dgrad_gemm_native: needs a 16-B aligned workspace of %zu bytes", need);
    const int M = d->B * d->Ho * d->Wo;
    const long N = (long)d->KH * d->KW * d->Ci;
    float* Y = (float*)d->workspace;
    void* doutT = (char*)d->workspace + (szn_conv2d_dgrad_gemm_workspace_bytes(d) + 255) / 256 * 256;
    const int Mp = (M + 7) / 8 * 8;
    if (Mp == M) {
        rc = szn_pack_weight_dgrad(d->dtype, M, 1, 1, d->Co, dout, doutT, stream);      // [M][Co] -> [Co][M]
        if (rc) return rc;
    } else {
        if ((uintptr_t)dout & 15) SZN_FAIL(SZN_ERR_ARG, "conv2d_dgrad_gemm_native: dout must be 16-B aligned");
        hipLaunchKernelGGL(transpose_pad16_kernel, dim3(d->Co / 64, (Mp + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dout,
                           (uint16_t*)doutT, M, Mp, d->Co);
        SZN_CHECK_LAUNCH("transpose_pad16_kernel");
    }
    szn_conv_desc_t g = {};
    g.dtype = d->dtype; g.B = 1; g.Hi = 1; g.Wi = d->Co; g.Ci = (int)N; g.Ho = 1; g.Wo = d->Co; g.Co = M;
    g.KH = 1; g.KW = 1; g.pad = 0; g.ldi = (int)N; g.ldo = Mp; g.ldg = 0;
    rc = szn_conv_wgrad_wide_try(&g, w, doutT, Y, 0, 1, stream);
    if (rc > 0) SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv2d_dgrad_gemm_native: conv_wgrad_wide refused the shape");
    if (rc) return rc;
    return launch_col2im(d, Y, din, stream);
}

// first-generation kernel (register-staged, 128x128 tile): fallback of szn_conv2d_wgrad (szn_conv_wgrad.hip)
int szn_conv2d_wgrad_v1(const szn_conv_desc_t* d, const void* in, const void* dout, float* dw, int accumulate,
                        szn_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!in || !dout || !dw) SZN_FAIL(SZN_ERR_ARG, "conv2d_wgrad: null pointer");
    const int ch = szn_is16(d->dtype) ? 8 : 4;
    if ((d->Ci % ch) || (d->ldi % ch) || (d->ldo % ch))
        SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv2d_wgrad: Ci/ldi/ldo must be multiples of %d", ch);
    hipStream_t st = (hipStream_t)stream;
    const long nw = (long)d->Co * d->KH * d->KW * d->Ci;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(dw, 0, nw * sizeof(float), st);
        if (e != hipSuccess) SZN_FAIL(SZN_ERR_LAUNCH, "conv2d_wgrad memset: %s", hipGetErrorString(e));
    }
    WgradArgs a;
    a.dout = (const char*)dout; a.in = (const char*)in; a.dw = dw;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co;
    a.KH = d->KH; a.KW = d->KW; a.pad = d->pad; a.ldi = d->ldi; a.ldd = d->ldo;
    a.M = d->B * d->Ho * d->Wo; a.HoWo = d->Ho * d->Wo;
    a.cotiles = szn_div_up(d->Co, 128); a.citiles = szn_div_up(d->Ci, 128);
    const long tiles = (long)a.cotiles * a.citiles * d->KH * d->KW;
    // aim for ~2048 blocks; each split covers a multiple of 64 pixels, at least 512
    long want = (2048 + tiles - 1) / tiles;
    if (want < 1) want = 1;
    long span = (a.M + want - 1) / want;
    if (span < 512) span = 512;
    span = (span + 63) / 64 * 64;
    a.kspan = (int)span;
    a.nsplit = szn_div_up(a.M, span);
    const long blocks = tiles * a.nsplit;
    if (blocks >= (1L << 31)) SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv2d_wgrad: grid too large");
    if (d->dtype == SZN_BF16)
        hipLaunchKernelGGL(conv_wgrad<bf16_raw>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else if (d->dtype == SZN_F16)
        hipLaunchKernelGGL(conv_wgrad<f16_raw>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(conv_wgrad<float>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    SZN_CHECK_LAUNCH("conv_wgrad");
    return SZN_OK;
}

extern "C" int szn_bias_grad_slab(int dtype, long M, int Co, int ldd, const void* dout, float* db, int accumulate,
                                  float* colsum_slab, int colsum_slab_rows, int* colsum_rows_out, szn_stream_t stream);
extern "C" int szn_bias_grad(int dtype, long M, int Co, int ldd, const void* dout, float* db, int accumulate,
                             szn_stream_t stream) {
    return szn_bias_grad_slab(dtype, M, Co, ldd, dout, db, accumulate, nullptr, 0, nullptr, stream);
}

extern "C" int szn_bias_grad_slab(int dtype, long M, int Co, int ldd, const void* dout, float* db, int accumulate,
                                  float* slab, int slab_rows, int* colsum_rows_out, szn_stream_t stream) {
    if (colsum_rows_out) *colsum_rows_out = 0;
    if (!dout || !db || M <= 0 || Co <= 0 || ldd < Co) SZN_FAIL(SZN_ERR_ARG, "bias_grad: bad argument");
    if (ldd % (szn_is16(dtype) ? 8 : 4) || ((uintptr_t)dout & 15))
        SZN_FAIL(SZN_ERR_UNSUPPORTED, "bias_grad: rows must be 16-B aligned (ldd multiple of %d)", szn_is16(dtype) ? 8 : 4);
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(db, 0, (size_t)Co * sizeof(float), st);
        if (e != hipSuccess) SZN_FAIL(SZN_ERR_LAUNCH, "bias_grad memset: %s", hipGetErrorString(e));
    }
    long rpb = (M + 2047) / 2048;
    if (rpb < 32) rpb = 32;
    const int blocks = szn_div_up(M, rpb);
    if (slab && slab_rows < blocks) SZN_FAIL(SZN_ERR_ARG, "bias_grad: colsum_slab holds %d rows, %d needed", slab_rows, blocks);
    szn_note_colsum_rows(slab ? blocks : 0);
    if (colsum_rows_out) *colsum_rows_out = slab ? blocks : 0;
    if (dtype == SZN_BF16)
        hipLaunchKernelGGL(bias_grad_kernel<bf16_raw>, dim3(blocks), dim3(256), 0, st, (const bf16_raw*)dout, db, M, Co, ldd,
                           (int)rpb, slab);
    else if (dtype == SZN_F16)
        hipLaunchKernelGGL(bias_grad_kernel<f16_raw>, dim3(blocks), dim3(256), 0, st, (const f16_raw*)dout, db, M, Co, ldd,
                           (int)rpb, slab);
    else if (dtype == SZN_F32)
        hipLaunchKernelGGL(bias_grad_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)dout, db, M, Co, ldd,
                           (int)rpb, slab);
    else
        SZN_FAIL(SZN_ERR_ARG, "bias_grad: bad dtype %d", dtype);
    SZN_CHECK_LAUNCH("bias_grad_kernel");
    return SZN_OK;
}

extern "C" int szn_gemm_proj_fwd(int dtype, long M, int K, int N, int ldo, const void* x, const void* w,
                                 const float* bias, float* out_f32, szn_stream_t stream) {
    if (M <= 0 || M >= (1L << 31)) SZN_FAIL(SZN_ERR_ARG, "gemm_proj_fwd: bad M");
    szn_conv_desc_t d = {dtype, 1, 1, (int)M, K, 1, (int)M, N, 1, 1, 0, K, ldo, 0, 0, 1};
    return szn_conv2d_fwd(&d, x, w, bias, nullptr, nullptr, out_f32, stream);
}
extern "C" int szn_gemm_proj_dgrad(int dtype, long M, int K, int N, int ldd, const void* dout, const void* wT,
                                   const void* gate, const float* chan_scale, void* dx, szn_stream_t stream) {
    if (M <= 0 || M >= (1L << 31)) SZN_FAIL(SZN_ERR_ARG, "gemm_proj_dgrad: bad M");
    if (chan_scale) SZN_FAIL(SZN_ERR_UNSUPPORTED, "gemm_proj_dgrad: chan_scale needs image geometry; use szn_conv2d_dgrad");
    szn_conv_desc_t d = {dtype, 1, 1, (int)M, K, 1, (int)M, N, 1, 1, 0, K, ldd, K, 0, 0};
    return szn_conv2d_dgrad(&d, dout, wT, gate, nullptr, dx, stream);
}
extern "C" int szn_gemm_proj_wgrad(int dtype, long M, int K, int N, int ldd, const void* x, const void* dout,
                                   float* dw, int accumulate, szn_stream_t stream) {
    if (M <= 0 || M >= (1L << 31)) SZN_FAIL(SZN_ERR_ARG, "gemm_proj_wgrad: bad M");
    szn_conv_desc_t d = {dtype, 1, 1, (int)M, K, 1, (int)M, N, 1, 1, 0, K, ldd, 0, 0, 0};
    return szn_conv2d_wgrad(&d, x, dout, dw, accumulate, stream);
}
