// szn_conv_c64.hip -- 3x3 convolution, 64 -> 64 channels, bf16 (conv1_2 forward and conv1_2 dgrad: the two 710 x 710
// launches of the pad-100 network, 11 % of the forward+dgrad time in conv_igemm_v2).
//
// A 64 x 9 x 64 filter bank is 72 KiB in bf16: split over the 8 waves of a block by (cout half) it is 36 MFMA A
// fragments = 144 VGPRs per lane, so the WEIGHTS LIVE IN REGISTERS for the whole kernel.  The block is persistent
// (one per CU) and walks a contiguous run of 16 x 16 output tiles; the only LDS traffic is the input patch:
//   * LDS = 3 patch buffers [18 x 18 px][128 B] filled by LDS-DMA (raw buffer loads, out-of-range offsets give the
//     zero padding), chunk index XOR-swizzled by (row & 7) on the DMA source side; the patch of tile t + 2 streams in
//     while tile t computes;
//   * wave (wm, wn): output rows 4 wm .. 4 wm + 3 x couts 32 wn .. + 31.  A pixel fragment (one patch row segment at
//     one kw shift) is read once and used by every (kh, j) pair that touches it: 36 ds_read_b128 per 144 MFMA;
//   * epilogue straight from registers: v_permlane16_swap pairs the two cout fragments so that a lane holds 8
//     consecutive couts of one pixel (16-B store, 64-B contiguous per pixel per instruction), + bias / ReLU / gate /
//     column sums (bias gradient of the producer layer, one atomicAdd per cout per block at the end);
//   * one s_barrier per tile.  Per-wave vmcnt order inside a tile: 6 patch pieces (t + 2) ... MFMAs ... vmcnt(6)
//     [=> pieces of t + 1 and the stores of t - 1 retired] ... stores (t).
// Accumulation order per output: tap-major inside each 32-channel half exactly like conv_igemm_v2 up to the order of
// the K terms (fp32 accumulate, so parity with the oracle is within the same tolerance as the other kernels).
#include "szn_common.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((address_space(3))) void* ldsptr_t;

namespace {

struct C64Args {
    const char* in; const char* w; const float* bias; const char* gate; char* out; float* colsum;
    unsigned in_bytes, gate_bytes;
    int B, Hi, Wi, Ho, Wo, pad;
    int ldi, ldo, ldg, relu;
    int tiles_x, tiles_y, ntiles;      // ntiles = B * tiles_y * tiles_x
};

constexpr unsigned kOOBc = 0x80000000u;
constexpr int PWc = 18, PROWSc = PWc * PWc;
constexpr int PATCHBc = 328 * 128;                  // 41 DMA instructions of 8 rows
constexpr int NBUF = 3;
constexpr int OFF_GATEc = NBUF * PATCHBc;           // gate tile: [8 waves][4 j][64 lanes][16 B], filled by LDS-DMA per tile
constexpr int OFF_DUMPc = OFF_GATEc + 8 * 4096;     // 1 KiB dump page (idle DMA slots)
constexpr int OFF_REDc = OFF_DUMPc + 1024;          // bias[64] + final colsum reduction [4 wm][64]
constexpr int OFF_CSc = OFF_REDc + 256 + 1024;      // column-sum accumulators [8 waves][4 g][8 floats]
constexpr int LDS_C64 = OFF_CSc + 1024;

__device__ __forceinline__ float row16_sum(float x) {                      // sum over the 16 lanes of a DPP row
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x141, 0xF, 0xF, true));   // row_half_mirror
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x140, 0xF, 0xF, true));   // row_mirror
    return x;
}

template <bool GATED, bool COLSUM>
__global__ __launch_bounds__(512) void conv3x3_c64(C64Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int g = lane >> 4, r16 = lane & 15;

    // contiguous run of tiles per block; blocks of one XCD (blockIdx % 8) own neighbouring runs so that the halo rows
    // shared by vertically adjacent tiles meet in the same L2
    const int G = gridDim.x;
    const int vb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int first = (int)((long)a.ntiles * vb / G), last = (int)((long)a.ntiles * (vb + 1) / G);
    if (first >= last) return;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc((void*)(GATED ? a.gate : a.in), 0, (int)(GATED ? a.gate_bytes : 0u), 0x00020000);

    // ---- the filter bank of this wave: A fragments W[tap][s][i], lane (g, r16) = cout 32 wn + 16 i + r16, channels
    //      32 s + 8 g .. + 7 (OHWI) ----
    u32x4_t W[9][2][2];
    {
        const char* wp = a.w + ((long)(wn * 32 + r16) * 9 * 64 + g * 8) * 2;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    W[tap][s][i] = *(const u32x4_t*)(wp + ((long)(i * 16) * 9 * 64 + tap * 64 + s * 32) * 2);
    }
    if (tid < 64) ((float*)(smem + OFF_REDc))[tid] = a.bias ? a.bias[tid] : 0.f;

    // ---- patch DMA slots: instruction p (0..5) of wave w covers patch rows 8 (w + 8 p) .. + 7 ----
    const int chunk_l = lane & 7, rsub = lane >> 3;
    const unsigned chunkoff = (unsigned)((chunk_l ^ rsub) << 4);          // (q & 7) == rsub
    const int q0 = 8 * w + rsub;                                          // patch row of slot p: q0 + 64 p
    auto issue = [&](int t, int buf) {
        int bb = t;
        const int tx = bb % a.tiles_x; bb /= a.tiles_x;
        const int ty = bb % a.tiles_y; const int b = bb / a.tiles_y;
        const int ih0 = ty * 16 - a.pad, iw0 = tx * 16 - a.pad;
        char* base = smem + buf * PATCHBc;
        int q0v = q0;
        asm volatile("" : "+v"(q0v));       // opaque per call: keeps the per-slot coordinates out of loop-carried VGPRs
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            unsigned v = kOOBc;
            const int q = q0v + 64 * p;
            if (q < PROWSc) {
                const int pr = (q * 3641) >> 16, pc = q - pr * PWc;           // q / 18 for q < 324
                const int ih = ih0 + pr, iw = iw0 + pc;
                if ((unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi)
                    v = (unsigned)(((b * a.Hi + ih) * a.Wi + iw) * a.ldi * 2) + chunkoff;
            }
            const int piece = w + 8 * p;
            char* dst = (piece < 41) ? base + piece * 1024 : smem + OFF_DUMPc;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)dst, 16, v, 0, 0, 0);
        }
    };

    // column sums: per tile reduced over the lane's 4 pixels and the 16 lanes of its row, then ds_add_f32 into a slot per
    // (wave, g) -- 8 loop-carried VGPRs would not fit next to the 144 of the filter bank
    float* csl = (float*)(smem + OFF_CSc + (w * 4 + g) * 32);
    if constexpr (COLSUM) {
        if (tid < 256) ((float*)(smem + OFF_CSc))[tid] = 0.f;
    }
    const int cstart = wn * 32 + (g & 1) * 16 + (g >> 1) * 8;              // first of this lane's 8 couts after the swap
    const int qb = (wm * 4) * PWc + r16;

    issue(first, 0);
    if (first + 1 < last) {
        issue(first + 1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    int buf = 0;
    for (int t = first; t < last; ++t) {
        const bool more = t + 2 < last;
        int bb = t;
        const int tx = bb % a.tiles_x; bb /= a.tiles_x;
        const int ty = bb % a.tiles_y; const int b = bb / a.tiles_y;
        const int ow = tx * 16 + r16;
        const int oh0 = ty * 16 + wm * 4;
        const unsigned m0 = (unsigned)((b * a.Ho + oh0) * a.Wo + ow);       // < 2^31 pixels (checked by the caller)
        const bool okw = ow < a.Wo;
        if constexpr (GATED) {
            // each lane's 16 gate bytes per pixel row j go to its own LDS slot: an asynchronous, register-free prefetch
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = okw && oh0 + j < a.Ho;
                const unsigned v = ok ? ((m0 + j * a.Wo) * a.ldg + cstart) * 2u : kOOBc;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (ldsptr_t)(smem + OFF_GATEc + w * 4096 + j * 1024), 16, v, 0, 0, 0);
            }
        }
        if (more) issue(t + 2, buf == 0 ? 2 : buf - 1);                    // (buf + 2) % 3

        f32x4_t acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const char* pb = smem + buf * PATCHBc;
        // 18 steps (patch row pr, shift kw), two fragments (channel halves) each; the fragments of step n + 1 are read
        // while the MFMAs of step n run (sched_barrier keeps the compiler from hoisting all 36 reads: 144 VGPRs)
        u32x4_t P[2][2];
        auto ldP = [&](int step, u32x4_t (&dst)[2]) {
            const int pr = step / 3, kw = step - pr * 3;
            const int q = qb + pr * PWc + kw;
            const char* rowp = pb + q * 128;
            const int sw = q & 7;
            dst[0] = *(const u32x4_t*)(rowp + ((g ^ sw) << 4));
            dst[1] = *(const u32x4_t*)(rowp + (((4 + g) ^ sw) << 4));
        };
        ldP(0, P[0]);
#pragma unroll
        for (int step = 0; step < 18; ++step) {
            const int pr = step / 3, kw = step - pr * 3;
            if (step + 1 < 18) ldP(step + 1, P[(step + 1) & 1]);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int j = pr - kh;
                    if (j < 0 || j > 3) continue;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, W[kh * 3 + kw][s][i]),
                                                                            __builtin_bit_cast(bf16x8_t, P[step & 1][s]), acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue from registers ----
        float bv[8];
        *(f32x4_t*)&bv[0] = *(const f32x4_t*)(smem + OFF_REDc + cstart * 4);
        *(f32x4_t*)&bv[4] = *(const f32x4_t*)(smem + OFF_REDc + cstart * 4 + 16);
        // retire the gate pieces of t, the patch pieces of t + 1 and the stores of t - 1
        if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float cs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = okw && oh0 + j < a.Ho;
            float v[8];
            u32x4_t gq;
            if constexpr (GATED) gq = *(const u32x4_t*)(smem + OFF_GATEc + w * 4096 + j * 1024 + lane * 16);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][j][c]), __float_as_uint(acc[1][j][c]), false, false);
                v[c] = __uint_as_float(r[0]);
                v[4 + c] = __uint_as_float(r[1]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = v[e] + bv[e];
                if (a.relu) x = fmaxf(x, 0.f);
                if constexpr (GATED) {
                    const uint32_t gw = gq[e >> 1];
                    const float gv = __uint_as_float((e & 1) ? (gw & 0xffff0000u) : (gw << 16));
                    x = (gv > 0.f) ? x : 0.f;
                }
                v[e] = x;
                if constexpr (COLSUM) cs[e] += ok ? x : 0.f;
            }
            u32x4_t pk;
            pk.x = (uint32_t)f32_to_bf16_bits(v[0]) | ((uint32_t)f32_to_bf16_bits(v[1]) << 16);
            pk.y = (uint32_t)f32_to_bf16_bits(v[2]) | ((uint32_t)f32_to_bf16_bits(v[3]) << 16);
            pk.z = (uint32_t)f32_to_bf16_bits(v[4]) | ((uint32_t)f32_to_bf16_bits(v[5]) << 16);
            pk.w = (uint32_t)f32_to_bf16_bits(v[6]) | ((uint32_t)f32_to_bf16_bits(v[7]) << 16);
            if (ok) *(u32x4_t*)(a.out + ((size_t)(m0 + j * a.Wo) * a.ldo + cstart) * 2) = pk;
        }
        if constexpr (COLSUM) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = row16_sum(cs[e]);
                if (r16 == 0) __hip_atomic_fetch_add(csl + e, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_f32
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        buf = (buf == 2) ? 0 : buf + 1;
    }

    if constexpr (COLSUM) {
        // slot (w, g) holds the sums of couts cstart .. + 7 of that wave: add the 4 wm waves
        float* red = (float*)(smem + OFF_REDc + 256);                      // [wm][64 couts]
        if (r16 == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wm * 64 + cstart + e] = csl[e];
        }
        __syncthreads();
        if (tid < 64) {
            const float s = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
            if (s != 0.f) atomicAdd(a.colsum + tid, s);
        }
    }
#endif
}

template <bool GATED, bool COLSUM>
int launch_c64(const C64Args& a, int grid, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv3x3_c64<GATED, COLSUM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_C64);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv3x3_c64<GATED, COLSUM>), dim3((unsigned)grid), dim3(512), LDS_C64, st, a);
    SZN_CHECK_LAUNCH("conv3x3_c64");
    return SZN_OK;
}

}  // namespace

// Called by szn_conv2d_fwd after it has validated the descriptor. Returns 1 if the layer is not this kernel's shape.
int szn_conv_c64_try(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias, const void* gate,
                     const float* chan_scale, void* out, int min_tiles, szn_stream_t stream) {
    if (d->dtype != SZN_BF16 || d->KH != 3 || d->KW != 3 || d->Ci != 64 || d->Co != 64 || d->pad > 2 || d->out_f32 ||
        chan_scale) return 1;
    if ((d->ldo & 7) || (d->ldi & 7) || (gate && (d->ldg & 7))) return 1;
    const size_t in_bytes = (size_t)d->B * d->Hi * d->Wi * d->ldi * 2;
    if (in_bytes >= 0x7fff0000ul) return 1;
    C64Args a;
    a.in = (const char*)in; a.w = (const char*)w; a.bias = bias; a.gate = (const char*)gate; a.out = (char*)out;
    a.colsum = d->colsum;
    a.in_bytes = (unsigned)in_bytes;
    const size_t gate_bytes = gate ? (size_t)d->B * d->Ho * d->Wo * d->ldg * 2 : 0;
    if (gate_bytes >= 0x7fff0000ul || (size_t)d->B * d->Ho * d->Wo * d->ldo * 2 >= 0xffff0000ul) return 1;
    a.gate_bytes = (unsigned)gate_bytes;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ho = d->Ho; a.Wo = d->Wo; a.pad = d->pad;
    a.ldi = d->ldi; a.ldo = d->ldo; a.ldg = d->ldg; a.relu = d->relu;
    a.tiles_x = szn_div_up(d->Wo, 16); a.tiles_y = szn_div_up(d->Ho, 16);
    const long nt = (long)a.B * a.tiles_y * a.tiles_x;
    if (nt >= (1L << 30) || nt < min_tiles) return 1;
    a.ntiles = (int)nt;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ncu = p.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
        ncu &= ~7;                                   // whole XCD groups
        if (ncu < 8) ncu = 8;
    }
    hipStream_t st = (hipStream_t)stream;
    if (gate) return d->colsum ? launch_c64<true, true>(a, ncu, st) : launch_c64<true, false>(a, ncu, st);
    return d->colsum ? launch_c64<false, true>(a, ncu, st) : launch_c64<false, false>(a, ncu, st);
}
