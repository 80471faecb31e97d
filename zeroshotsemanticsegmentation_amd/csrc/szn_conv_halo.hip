// szn_conv_halo.hip -- 3x3 convolution (forward / dgrad) with the input patch held in LDS across the 9 filter taps.
//
// The implicit-GEMM kernel (szn_conv_igemm.hip) re-fetches its 256-pixel operand tile for every tap: 32 KB (pixels) +
// 16 KB (weights) of L2->LDS traffic per 4.2 MFLOP, and rocprof shows the 3x3 layers pinned at ~60 % L2 channel
// utilisation with waves 40 % of the time in s_waitcnt.  Here a block owns a 16 x 16 OUTPUT tile; per 64-channel
// slice it loads the (16+2) x (16+2) input patch ONCE (41 KB) and runs the 9 taps out of LDS by shifting the
// fragment read address, so only the weights stream per tap: 4.5 + 16 KB per 4.2 MFLOP (2.3x less; 3.2x less
// for the 64-cout layers).
//   * 512 threads = 8 waves (4 x 2): wave (wm, wn) -> output rows 4 wm .. 4 wm + 3 (one 16-pixel MFMA fragment per
//     row) x BN/2 couts;
//   * LDS: 2 patch buffers [324 px][128 B] (double-buffered over channel slices) + 3-stage weight ring [BN][128 B]
//     + a 1 KiB-per-wave dump page; every iteration issues exactly NB weight + 1 patch LDS-DMA per wave (the patch
//     of the next slice is spread over taps 2..7, the other taps issue an out-of-range load into the dump page), so one
//     counted s_waitcnt vmcnt(NB + 1) + one s_barrier per tap;
//   * both images are [row][128 B] with the 16-B chunk index XOR-swizzled by (row & 7), applied on the DMA source
//     side; a fragment reads 16 CONSECUTIVE patch rows, which keeps ds_read_b128 conflict-free for any start row;
//   * padding, image borders and tile edges are out-of-range buffer offsets (zeros).
// Same epilogue (bias, ReLU, gate, dropout factor) and accumulation order per output as conv_igemm_v2 up to the
// order of the K terms (channel slice outer, tap inner instead of tap outer).
#include "szn_common.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((address_space(3))) void* ldsptr_t;

namespace {

struct HaloArgs {
    const char* in; const char* w; const float* bias; const char* gate; const float* cscale; char* out;
    unsigned in_bytes, w_bytes;
    int B, Hi, Wi, Ci, Ho, Wo, Co, pad;
    int ldi, ldo, ldg, relu, out_f32;
    int tiles_x, tiles_y, ntiles;
};

constexpr unsigned kOOBh = 0x80000000u;
constexpr int PW = 18, PROWS = PW * PW;            // patch 18 x 18 pixels
constexpr int PATCHB = 328 * 128;                   // 324 rows rounded up to 41 DMA instructions of 8 rows

template <typename T, int WNF>
__global__ __launch_bounds__(512, 2) void conv3x3_halo(HaloArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int ES = sizeof(T);
    constexpr int BKE = 128 / ES;
    constexpr int BN = 32 * WNF;
    constexpr int NB = BN / 64;                    // weight DMA instructions per wave per tap
    constexpr int LPC = NB + 1;
    constexpr int WSTAGE = BN * 128;
    constexpr int OFF_W = 2 * PATCHB;              // weight ring after the two patch buffers
    constexpr int OFF_DUMP = OFF_W + 3 * WSTAGE;   // 8 x 1 KiB dump pages
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int g = lane >> 4, r16 = lane & 15;

    // tile decode: cout tile fastest (the patch is shared by the cout tiles of one pixel tile through L2)
    int bid = blockIdx.x;
    const int nt = bid % a.ntiles; bid /= a.ntiles;
    const int tx = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty = bid % a.tiles_y; const int b = bid / a.tiles_y;
    const int oh0 = ty * 16, ow0 = tx * 16, n0 = nt * BN;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);

    // ---- patch DMA slots of this thread: instruction p (0..5) of wave w covers patch rows 8 (w + 8 p) .. + 7 ----
    unsigned voffP[6];
    const int chunk_l = lane & 7, rsub = lane >> 3;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const int q = 8 * (w + 8 * p) + rsub;                      // linear patch row
        unsigned v = kOOBh;
        if (q < PROWS) {
            const int pr = q / PW, pc = q - pr * PW;
            const int ih = oh0 - a.pad + pr, iw = ow0 - a.pad + pc;
            if ((unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi) {
                const int chunk = chunk_l ^ (q & 7);
                v = (unsigned)((((long)(b * a.Hi + ih) * a.Wi + iw) * a.ldi + chunk * (16 / ES)) * ES);
            }
        }
        voffP[p] = v;
    }
    unsigned voffB[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = (BN / 8) * w + 8 * i + rsub;
        const int n = n0 + row;
        const int chunk = chunk_l ^ (row & 7);
        voffB[i] = (n < a.Co) ? (unsigned)(((long)n * 9 * a.Ci + chunk * (16 / ES)) * ES) : kOOBh;
    }

    const int nslice = a.Ci / BKE;                 // channel slices
    const int nIter = nslice * 9;
    // issue iteration `it` (slice = it / 9, tap = it % 9): weights of (slice, tap) -> ring stage, + one patch piece of
    // slice + 1 (taps 0..5) or a dump load
    auto issue = [&](int it, int wstage) {
        const int sl = it / 9, tap = it - sl * 9;
        const int soffB = (tap * a.Ci) * ES + sl * 128;
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)(smem + OFF_W + wstage * WSTAGE + ((BN / 8) * w + 8 * i) * 128),
                                                     16, voffB[i], soffB, 0, 0);
        // pieces ride on taps 2..7: the other patch buffer (slice sl - 1) was last READ in iteration 9 sl - 1 and this
        // call happens behind the barrier of iteration >= 9 sl, and the last piece (tap 7) is retired by the counted
        // wait of iteration 9 (sl + 1)
        const int pc = tap - 2;
        const bool real = (pc >= 0) && (pc < 6) && (sl + 1 < nslice) && (w + 8 * pc) < 41;
        if (real) {
            char* dst = smem + ((sl + 1) & 1) * PATCHB + (w + 8 * pc) * 1024;
            // voffP index must be a compile-time constant to stay in registers: unrolled select
            unsigned v = voffP[0];
            if (pc == 1) v = voffP[1]; else if (pc == 2) v = voffP[2]; else if (pc == 3) v = voffP[3];
            else if (pc == 4) v = voffP[4]; else if (pc == 5) v = voffP[5];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)dst, 16, v, (sl + 1) * 128, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(smem + OFF_DUMP + w * 1024), 16, kOOBh, 0, 0, 0);
        }
    };

    f32x4_t acc[WNF][4];
#pragma unroll
    for (int i = 0; i < WNF; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: patch of slice 0 (6 pieces per wave), then the first two weight stages ----
#pragma unroll
    for (int p = 0; p < 6; ++p)
        if (w + 8 * p < 41)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(smem + (w + 8 * p) * 1024), 16, voffP[p], 0, 0, 0);
    issue(0, 0);
    if (nIter > 1) issue(1, 1);
    if (nIter > 2) issue(2, 2);

    // fragment read bases: pixel fragment j = output row 4 wm + j, lane r16 = column; patch row q = (R+kh)*18 + kw + r16
    int qb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) qb[j] = (wm * 4 + j) * PW + r16;
    const int woff0 = ((g ^ (r16 & 7)) << 4), woff1 = (((4 + g) ^ (r16 & 7)) << 4);

    auto mma = [&](const u32x4_t (&wf)[WNF], const u32x4_t (&pf)[4]) {
#pragma unroll
        for (int i = 0; i < WNF; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (ES == 2) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[i]),
                                                                        __builtin_bit_cast(bf16x8_t, pf[j]), acc[i][j], 0, 0, 0);
                } else {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[i].x), __uint_as_float(pf[j].x), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[i].y), __uint_as_float(pf[j].y), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[i].z), __uint_as_float(pf[j].z), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[i].w), __uint_as_float(pf[j].w), acc[i][j], 0, 0, 0);
                }
            }
    };
    // fragments of half `s` (0/1) of iteration `it` whose weights sit in ring stage `wst`
    auto fetch = [&](int it, int wst, int s, u32x4_t (&wf)[WNF], u32x4_t (&pf)[4]) {
        const int sl = it / 9, tap = it - sl * 9;
        const int kh = tap / 3, kw = tap - kh * 3;
        const char* pb = smem + (sl & 1) * PATCHB;
        const char* sw = smem + OFF_W + wst * WSTAGE + (wn * (BN / 2) + r16) * 128;
        const int qs = kh * PW + kw;
#pragma unroll
        for (int i = 0; i < WNF; ++i) wf[i] = *(const u32x4_t*)(sw + i * 16 * 128 + (s ? woff1 : woff0));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = qb[j] + qs;
            pf[j] = *(const u32x4_t*)(pb + q * 128 + (((s * 4 + g) ^ (q & 7)) << 4));
        }
    };

    // software-pipelined loop (see conv_igemm_v2 ABL == 5): fragment reads of the next half run under the MFMAs of the
    // current half, three tap-iterations of LDS-DMA in flight, the barrier sits between the two halves
    if (nIter > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * LPC) : "memory");
    else if (nIter > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    u32x4_t wfA[WNF], pfA[4], wfB[WNF], pfB[4];
    fetch(0, 0, 0, wfA, pfA);
    int stage = 0;
    for (int it = 0; it < nIter; ++it) {
        fetch(it, stage, 1, wfB, pfB);
        mma(wfA, pfA);
        if (it + 1 < nIter) {
            if (it + 2 < nIter) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPC) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + 3 < nIter) issue(it + 3, stage);                  // recycles the weight stage just drained
        const int nstage = (stage == 2) ? 0 : stage + 1;
        if (it + 1 < nIter) fetch(it + 1, nstage, 0, wfA, pfA);
        mma(wfB, pfB);
        stage = nstage;
    }

    // ---- epilogue staged through LDS (same scheme as conv_igemm_v2): fp32 tile -> whole output rows, 16 B per lane ----
    constexpr int P = BN + 4;
    float* tile = (float*)smem;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < WNF; ++i) {
            const int row = (wm * 4 + j) * 16 + r16, col = wn * (BN / 2) + i * 16 + g * 4;
            *(f32x4_t*)(tile + row * P + col) = acc[i][j];
        }
    __syncthreads();
    const T* __restrict__ gate = (const T*)a.gate;
    constexpr int CPR = BN / 8;
    const bool out32 = a.out_f32 || sizeof(T) == 4;
    const int oes = out32 ? 4 : 2;
    const bool fast_o = (((long)a.ldo * oes) & 15) == 0;
    const bool fast_g = gate && ((((long)a.ldg * ES) & 15) == 0);
    const int cc = tid % CPR, row0 = tid / CPR;
    const int n = n0 + cc * 8;
    if (n < a.Co) {
        const bool full = n + 8 <= a.Co;
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (a.bias && n + e < a.Co) ? a.bias[n + e] : 0.f;
        constexpr int NIT = 256 * CPR / 512;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int row = row0 + k * (512 / CPR);
            const int oh = oh0 + (row >> 4), ow = ow0 + (row & 15);
            if (oh < a.Ho && ow < a.Wo) {
                const long m = ((long)b * a.Ho + oh) * a.Wo + ow;
                const float* tp = tile + row * P + cc * 8;
                float v[8];
                *(f32x4_t*)&v[0] = *(const f32x4_t*)tp;
                *(f32x4_t*)&v[4] = *(const f32x4_t*)(tp + 4);
                float gv[8];
                if (gate) {
                    const T* gp = gate + m * a.ldg + n;
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
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float x = v[e] + bv[e];
                    if (a.relu) x = fmaxf(x, 0.f);
                    if (gate) x = (gv[e] > 0.f) ? x : 0.f;
                    if (a.cscale && n + e < a.Co) x *= a.cscale[(long)b * a.Co + n + e];
                    v[e] = x;
                }
                if (out32) {
                    float* o = (float*)a.out + m * a.ldo + n;
                    if (full && fast_o) {
                        *(f32x4_t*)o = *(const f32x4_t*)&v[0];
                        *(f32x4_t*)(o + 4) = *(const f32x4_t*)&v[4];
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (n + e < a.Co) o[e] = v[e];
                    }
                } else {
                    uint16_t* o = (uint16_t*)a.out + m * a.ldo + n;
                    if (full && fast_o) {
                        u32x4_t pk;
                        pk.x = (uint32_t)f32_to_bf16_bits(v[0]) | ((uint32_t)f32_to_bf16_bits(v[1]) << 16);
                        pk.y = (uint32_t)f32_to_bf16_bits(v[2]) | ((uint32_t)f32_to_bf16_bits(v[3]) << 16);
                        pk.z = (uint32_t)f32_to_bf16_bits(v[4]) | ((uint32_t)f32_to_bf16_bits(v[5]) << 16);
                        pk.w = (uint32_t)f32_to_bf16_bits(v[6]) | ((uint32_t)f32_to_bf16_bits(v[7]) << 16);
                        *(u32x4_t*)o = pk;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (n + e < a.Co) o[e] = f32_to_bf16_bits(v[e]);
                    }
                }
            }
        }
    }
#endif
}

template <typename T, int WNF>
int launch_halo(const HaloArgs& a, hipStream_t st) {
    constexpr int BN = 32 * WNF;
    const size_t lds = 2 * PATCHB + 3 * BN * 128 + 8 * 1024;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo<T, WNF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const long blocks = (long)a.B * a.tiles_y * a.tiles_x * a.ntiles;
    hipLaunchKernelGGL((conv3x3_halo<T, WNF>), dim3((unsigned)blocks), dim3(512), lds, st, a);
    SZN_CHECK_LAUNCH("conv3x3_halo");
    return SZN_OK;
}

}  // namespace

// Called by szn_conv2d_fwd for 3x3 layers whose geometry it has already validated. Returns 1 if not applicable.
int szn_conv3x3_halo_try(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias, const void* gate,
                         const float* chan_scale, void* out, szn_stream_t stream) {
    if (d->KH != 3 || d->KW != 3 || d->pad > 2) return 1;
    const size_t es = d->dtype == SZN_BF16 ? 2 : 4;
    const size_t in_bytes = (size_t)d->B * d->Hi * d->Wi * d->ldi * es;
    const size_t w_bytes = (size_t)d->Co * 9 * d->Ci * es;
    if (in_bytes >= 0x7fff0000ul || w_bytes >= 0x7fff0000ul) return 1;
    HaloArgs a;
    a.in = (const char*)in; a.w = (const char*)w; a.bias = bias; a.gate = (const char*)gate; a.cscale = chan_scale;
    a.out = (char*)out;
    a.in_bytes = (unsigned)in_bytes; a.w_bytes = (unsigned)w_bytes;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co; a.pad = d->pad;
    a.ldi = d->ldi; a.ldo = d->ldo; a.ldg = d->ldg; a.relu = d->relu; a.out_f32 = d->out_f32;
    a.tiles_x = szn_div_up(d->Wo, 16); a.tiles_y = szn_div_up(d->Ho, 16);
    const bool narrow = d->Co <= 64;
    a.ntiles = szn_div_up(d->Co, narrow ? 64 : 128);
    if ((long)a.B * a.tiles_y * a.tiles_x * a.ntiles >= (1L << 31)) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == SZN_BF16) return narrow ? launch_halo<bf16_raw, 2>(a, st) : launch_halo<bf16_raw, 4>(a, st);
    return narrow ? launch_halo<float, 2>(a, st) : launch_halo<float, 4>(a, st);
}
