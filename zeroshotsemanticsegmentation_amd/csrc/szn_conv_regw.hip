// szn_conv_regw.hip -- 3x3 convolution with the filter bank held in REGISTERS (bf16; 64/128 -> 64/128 channels: conv1_2
// and conv2_x forward and dgrad, the 710^2 / 355^2 launches of the pad-100 network).
//
// A wave holds the MFMA A fragments of 32 couts x 64 cins x 9 taps = 144 VGPRs per lane for the whole kernel, so a
// block of 8 waves holds up to 128 x 128 x 9 filters.  The block is persistent (one per CU) and walks a contiguous run
// of output tiles; the only LDS traffic is the input patch, staged ONCE for all nine taps:
//   * wave = (pixel group pg, cout group cog of 32, cin group cig of 64); COG x CIG x PG = 8.  The tile is 4 PG rows x
//     16 columns; a wave computes 4 rows (4 pixel fragments) x 32 couts over its 64 cins: 144 MFMA per tile, pixel
//     fragments reused across the taps that touch them (36 ds_read_b128 per 144 MFMA);
//   * LDS = NBUF patch buffers [(4 PG + 2) x 18 px][64 CIG bf16] filled by LDS-DMA (raw buffer loads, out-of-range
//     offsets give the zero padding), 16-B chunk index XOR-swizzled by the pixel row on the DMA source side; the patch
//     of tile t + NBUF - 1 streams in while tile t computes;
//   * CIG = 2: the two cin halves of a cout group sit in neighbouring waves; after the MFMAs each wave hands the
//     partial sums of two of its four pixel fragments to its partner through LDS and finishes the other two;
//   * epilogue straight from registers: v_permlane16_swap pairs the two cout fragments so that a lane holds 8
//     consecutive couts of one pixel (16-B store), + bias / ReLU / gate (gate tile prefetched by LDS-DMA into a
//     private slot per lane) / column sums (bias gradient of the producer layer: DPP row sums + ds_add_f32, one global
//     atomicAdd per cout per block);
//   * one s_barrier per tile (two when CIG = 2).  Per-wave vmcnt order inside a tile: gate pieces (t), patch pieces
//     (t + NBUF - 1) ... MFMAs ... counted wait [=> gate (t), patch (t + 1), stores (t - 1) retired] ... stores (t).
// Accumulation is fp32; per output the K terms are added tap-major inside 32-channel groups (a different order from
// conv_igemm_v2, same tolerance against the oracle).
#include "szn_common.h"
#include "szn_cb.h"
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((address_space(3))) void* ldsptr_t;

namespace {

struct RwArgs {
    const char* in; const char* w; const float* bias; const char* gate; char* out; float* colsum;
    float* cslab;              // optional [grid][Co]: this block's column sums go to row blockIdx.x (fixed-order reduce later)
    char* pool;                        // optional: MaxPool2d(2,2,ceil) of the (ReLU'd) output, [B][Hp][Wp][Co] dense
    unsigned char* pcode;              // optional (with pool): winner code per pooled element, szn_conv_desc_t.pool_code
    int skip_x;                        // 1: the un-pooled output is not stored (szn_conv_desc_t.pool_only)
    CbGeom cb;                         // constant-border hint (cb.on): ntiles counts the kept tiles only
    // constant-border hint of a GATED launch (dgrad): tiles wholly outside rows [gy0, gy1) x columns [gx0, gx1) read the gate of
    // pixel `gref` (their own gate rows hold the same values), tiles wholly outside [sy0, sy1) x [sx0, sx1) store nothing
    int dg_on, gy0, gy1, gx0, gx1, sy0, sy1, sx0, sx1;
    unsigned gref;
    unsigned in_bytes, gate_bytes;
    unsigned out_bytes, pool_bytes;    // (< 0xffff0000: the stores go through buffer descriptors with 32-bit offsets, masked lanes by an out-of-range one)
    int Hp, Wp;
    int B, Hi, Wi, Ho, Wo, pad;
    int ldi, ldo, ldg, relu;
    int tiles_x, tiles_y, ntiles;      // ntiles = B * tiles_y * tiles_x
    int ablate;                        // SZN_REGW_ABLATE (make ABLATE=1 only; WRONG results): 1 no MFMA phase, 2 no global stores, 4 no patch DMA
    int shift;                         // 1: phase-shifted wave groups (CIG = 1 kernels; SZN_REGW_SHIFT=0 restores the lockstep order)
};
#ifdef SZN_ABLATE_BUILD
#define RW_ABL(bit) (a.ablate & (bit))
#else
#define RW_ABL(bit) 0
#endif

constexpr unsigned kOOBr = 0x80000000u;
constexpr unsigned kOOBst = 0xfffffff0u;                // store offset beyond any output this kernel takes (out_bytes, pool_bytes < 0xffff0000)
constexpr int PWr = 18;

__device__ __forceinline__ float row16_sum(float x) {                      // sum over the 16 lanes of a DPP row
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x141, 0xF, 0xF, true));   // row_half_mirror
    x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), 0x140, 0xF, 0xF, true));   // row_mirror
    return x;
}

template <int COG, int CIG>
struct RwGeom {
    static constexpr int PG = 8 / (COG * CIG);          // pixel groups of 4 output rows
    static constexpr int TR = 4 * PG;                   // tile rows
    static constexpr int PROWS = (TR + 2) * PWr;        // patch pixels
    static constexpr int RB = 128 * CIG;                // bytes per patch pixel
    static constexpr int CPR = RB / 16;                 // 16-B chunks per patch pixel
    static constexpr int RPP = 1024 / RB;               // patch pixels per 1-KiB DMA piece
    static constexpr int NPIECE = (PROWS * RB + 1023) / 1024;
    static constexpr int SLOTS = (NPIECE + 7) / 8;      // patch DMA instructions per wave per tile
    static constexpr int PATCHB = NPIECE * 1024;
    static constexpr int NBUF = CIG == 1 ? 3 : 2;
    static constexpr int CO = 32 * COG, CI = 64 * CIG;
    static constexpr int NJ = 4 / CIG;                  // pixel fragments finished (epilogue) by one wave
    static constexpr int OFF_GATE = NBUF * PATCHB;      // [8 waves][NJ][64 lanes][16 B]
    static constexpr int OFF_XCH = OFF_GATE + 8 * NJ * 1024;            // CIG = 2: [8 waves][2 jj][2 i][64 lanes][16 B]
    static constexpr int OFF_DUMP = OFF_XCH + (CIG == 2 ? 8 * 4096 : 0);
    static constexpr int OFF_BIAS = OFF_DUMP + 1024;
    static constexpr int OFF_RED = OFF_BIAS + CO * 4;
    static constexpr int LDS = OFF_RED + CO * 4;
    static_assert(COG * CIG * PG == 8 && LDS <= 160 * 1024, "wave decomposition / LDS budget");
};

template <typename T, int COG, int CIG, bool GATED, bool COLSUM>      // T = bf16_raw | f16_raw (conversions + MFMA opcode only)
__global__ __launch_bounds__(512) void conv3x3_regw(RwArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef RwGeom<COG, CIG> G_;
    constexpr int PG = G_::PG, TR = G_::TR, PROWS = G_::PROWS, RB = G_::RB, CPR = G_::CPR, RPP = G_::RPP;
    constexpr int NPIECE = G_::NPIECE, SLOTS = G_::SLOTS, PATCHB = G_::PATCHB, NBUF = G_::NBUF, CO = G_::CO, CI = G_::CI;
    constexpr int NJ = G_::NJ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cig = w % CIG, cog = (w / CIG) % COG, pg = w / (CIG * COG);
    const int g = lane >> 4, r16 = lane & 15;

    // contiguous run of tiles per block; blocks of one XCD (blockIdx % 8) own neighbouring runs so that the halo rows
    // shared by vertically adjacent tiles meet in the same L2
    const int G = gridDim.x;
    const int vb = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int first = (int)((long)a.ntiles * vb / G), last = (int)((long)a.ntiles * (vb + 1) / G);
    if (first >= last) {
        if constexpr (COLSUM) { if (a.cslab && tid < CO) a.cslab[(long)blockIdx.x * CO + tid] = 0.f; }
        return;
    }

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc((void*)(GATED ? a.gate : a.in), 0, (int)(GATED ? a.gate_bytes : 0u), 0x00020000);
    // stores: one 32-bit offset per lane (a lane that stores nothing gets an out-of-range one) instead of 64-bit address arithmetic and an
    // exec-mask branch around every store -- ~40 VALU / SALU instructions per tile and wave less in the epilogue (profiles/r05_ablations.txt 17)
    const auto rsO = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (int)a.out_bytes, 0x00020000);
    // (pooling only exists in the un-gated kernels: no descriptor SGPRs for it in the gated ones, which hold rsG)
    const auto rsPo = __builtin_amdgcn_make_buffer_rsrc((void*)((!GATED && a.pool) ? a.pool : a.out), 0, (int)((!GATED && a.pool) ? a.pool_bytes : 0u), 0x00020000);
    const auto rsPc = __builtin_amdgcn_make_buffer_rsrc((void*)((!GATED && a.pcode) ? (char*)a.pcode : a.out), 0, (int)((!GATED && a.pcode) ? a.pool_bytes >> 1 : 0u), 0x00020000);

    // ---- the filter bank of this wave: A fragments W[tap][s][i], lane (g, r16) = cout 32 cog + 16 i + r16, channels
    //      64 cig + 32 s + 8 g .. + 7 (OHWI) ----
    u32x4_t W[9][2][2];
    {
        const char* wp = a.w + ((long)(cog * 32 + r16) * 9 * CI + cig * 64 + g * 8) * 2;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    W[tap][s][i] = *(const u32x4_t*)(wp + ((long)(i * 16) * 9 * CI + tap * CI + s * 32) * 2);
    }
    if (tid < CO) {
        ((float*)(smem + G_::OFF_BIAS))[tid] = a.bias ? a.bias[tid] : 0.f;
        ((float*)(smem + G_::OFF_RED))[tid] = 0.f;
    }

    // ---- patch DMA slots: instruction p of wave w is piece w + 8 p = patch pixels RPP * piece .. + RPP - 1; the source
    //      chunk of a lane's slot is slot ^ (pixel & (CPR - 1)), a per-lane constant for both row widths ----
    const int slot = lane % CPR, psub = lane / CPR;
    const unsigned chunkoff = (unsigned)((CPR == 8 ? (slot ^ psub) : (slot ^ ((4 * (w & 3) + psub) & 15))) << 4);
    const int q0 = RPP * w + psub;                                         // patch pixel of slot p: q0 + 8 RPP p
    auto tile_of = [&](int t, int& b, int& ty, int& tx) {
        if (a.cb.on) {
            b = t / a.cb.per_image;
            cb_decode(a.cb, t - b * a.cb.per_image, ty, tx);
        } else {
            int bb = t;
            tx = bb % a.tiles_x; bb /= a.tiles_x;
            ty = bb % a.tiles_y; b = bb / a.tiles_y;
        }
    };
    // A block's tiles are consecutive in the (kept-)tile numbering: two cursors -- the tile being issued and the tile being computed --
    // are decoded once and then stepped (scalar compares; the divisions of tile_of per tile and per cursor were ~10 % of a
    // constant-border launch)
    auto step = [&](int& b, int& ty, int& tx) {
        ++tx;
        for (int pass = 0; pass < a.tiles_y + 2; ++pass) {     // (whole tile rows may be skippable: the dgrad form keeps the window rows only)
            if (a.cb.on && ty >= a.cb.fy0 && ty < a.cb.fy1 && tx >= a.cb.fx0 && tx < a.cb.fx1) {       // inside the padding-free rectangle
                if (ty < a.cb.wy0 || ty >= a.cb.wy1) tx = a.cb.fx1;
                else if (tx < a.cb.wx0) tx = a.cb.wx0;
                else if (tx >= a.cb.wx1) tx = a.cb.fx1;
            }
            if (tx < a.tiles_x) break;
            tx = 0;                                        // next tile row (its column 0 may be skippable too: second pass)
            if (++ty == a.tiles_y) { ty = 0; ++b; }
        }
    };
    int ib, ity, itx;                                      // the next tile to issue
    tile_of(first, ib, ity, itx);
    int cb_b = ib, cb_ty = ity, cb_tx = itx;               // the tile being computed
    auto issue = [&](int buf) {
        const int b = ib, ty = ity, tx = itx;
        step(ib, ity, itx);
        const int ih0 = ty * TR - a.pad, iw0 = tx * 16 - a.pad;
        char* base = smem + buf * PATCHB;
        int q0v = q0;
        asm volatile("" : "+v"(q0v));       // opaque per call: keeps the per-slot coordinates out of loop-carried VGPRs
#pragma unroll
        for (int p = 0; p < SLOTS; ++p) {
            unsigned v = kOOBr;
            const int q = q0v + 8 * RPP * p;
            if (q < PROWS) {
                const int pr = (q * 3641) >> 16, pc = q - pr * PWr;           // q / 18 for q < 324
                const int ih = ih0 + pr, iw = iw0 + pc;
                if ((unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi)
                    v = (unsigned)(((b * a.Hi + ih) * a.Wi + iw) * a.ldi * 2) + chunkoff;
            }
            const int piece = w + 8 * p;
            char* dst = (piece < NPIECE) ? base + piece * 1024 : smem + G_::OFF_DUMP;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)dst, 16, v, 0, 0, 0);
        }
    };

    const int cstart = cog * 32 + (g & 1) * 16 + (g >> 1) * 8;             // first of this lane's 8 couts after the swap
    const int qb = (pg * 4) * PWr + r16;
    const int j0 = CIG == 2 ? 2 * cig : 0;                                 // first pixel fragment this wave finishes

    issue(0);
    if constexpr (NBUF == 3) {
        if (first + 1 < last) {
            issue(1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(SLOTS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // Phase shift (CIG = 1).  The two waves of a SIMD (w, w + 4) used to run the same phase at the same time -- MFMA phase, then
    // epilogue (permlane / bias / ReLU / pack / stores) -- and the phases simply added up (SZN_REGW_ABLATE accounting of conv1_2
    // forward: 0.42 ms = 0.14 epilogue + barriers, + 0.19 MFMA phase, + 0.05 stores, + 0.09 patch DMA).  Waves 4 .. 7 (group B) now take
    // the per-tile block barrier BETWEEN their MFMA phase and their epilogue, waves 0 .. 3 (group A) behind the epilogue as before:
    //     A:  MFMA(t) epi(t) | MFMA(t+1) epi(t+1) | ...          B:  MFMA(t) | epi(t) MFMA(t+1) | epi(t+1) MFMA(t+2) | ...
    // so between two barriers every SIMD has one wave in each phase.  Same number of barriers per wave, same buffer hand-over rule
    // (after barrier k everyone has finished the MFMAs of tile first + k - 1, whose patch buffer is the one refilled next, and has
    // waited for its own pieces of patch first + k), no extra registers.  CIG = 2 keeps the lockstep order: its mid-tile partner
    // exchange needs a block barrier that both groups reach at the same point.
    const bool grpB = CIG == 1 && a.shift && w >= 4;
    // column sums (bias gradient of the producer layer): each lane keeps running sums of its 8 couts over all the tiles of the
    // block; they are combined once, in a fixed order, behind the tile loop (no LDS atomics: bit-reproducible)
    float cst[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cst[e] = 0.f;
    const float relu_lo = a.relu ? 0.f : -__builtin_inff();
    int buf = 0;
#ifdef SZN_ABLATE_BUILD
    // SZN_REGW_ABLATE bit 8 (tools/probe_regw_cycles.py): clock64 split of the tile loop per wave -- top (gate / patch DMA issue), MFMA phase,
    // counted wait, group B's barrier, epilogue, group A's barrier -- written over the first bytes of `out` (which is garbage then)
    long long pc[6] = {0, 0, 0, 0, 0, 0}, pk0 = 0;
#define RW_PROBE(i) do { if (a.ablate & 8) { const long long now_ = clock64(); pc[i] += now_ - pk0; pk0 = now_; } } while (0)
#else
#define RW_PROBE(i) do { } while (0)
#endif
    for (int t = first; t < last; ++t) {
#ifdef SZN_ABLATE_BUILD
        if (a.ablate & 8) pk0 = clock64();
#endif
        const bool more = t + NBUF - 1 < last;
        const int b = cb_b, ty = cb_ty, tx = cb_tx;
        step(cb_b, cb_ty, cb_tx);
        const int ow = tx * 16 + r16;
        const int oh0 = ty * TR + pg * 4 + j0;                             // output row of this wave's first finished fragment
        const unsigned m0 = (unsigned)((b * a.Ho + oh0) * a.Wo + ow);       // < 2^31 pixels (checked by the caller)
        const bool okw = ow < a.Wo;
        bool gconst = false, nostore = false;
        if constexpr (GATED) {
            if (a.dg_on) {
                const int y0 = ty * TR, x0 = tx * 16;
                gconst = y0 + TR <= a.gy0 || y0 >= a.gy1 || x0 + 16 <= a.gx0 || x0 >= a.gx1;
                nostore = y0 + TR <= a.sy0 || y0 >= a.sy1 || x0 + 16 <= a.sx0 || x0 >= a.sx1;
            }
        }
        if constexpr (GATED) {
            // each lane's 16 gate bytes per finished pixel row go to its own LDS slot: an asynchronous, register-free prefetch
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const bool ok = okw && oh0 + jj < a.Ho;
                const unsigned v = ok ? ((gconst ? a.gref : m0 + jj * a.Wo) * a.ldg + cstart) * 2u : kOOBr;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (ldsptr_t)(smem + G_::OFF_GATE + (w * NJ + jj) * 1024), 16, v, 0, 0, 0);
            }
        }
        if (more && !RW_ABL(4)) issue((buf + NBUF - 1) % NBUF);
        RW_PROBE(0);

        f32x4_t acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const char* pb = smem + buf * PATCHB;
        int qbv = qb;
        asm volatile("" : "+v"(qbv));               // per-tile opaque: the 36 swizzled read addresses are not hoisted into VGPRs
        // 18 steps (patch row pr, shift kw), two fragments (channel halves) each; the fragments of step n + 1 are read
        // while the MFMAs of step n run (sched_barrier keeps the compiler from hoisting all 36 reads: 144 VGPRs)
        u32x4_t P[2][2];
        auto ldP = [&](int step, u32x4_t (&dst)[2]) {
            const int pr = step / 3, kw = step - pr * 3;
            const int q = qbv + pr * PWr + kw;
            const char* rowp = pb + q * RB;
            const int sw = q & (CPR - 1);
            dst[0] = *(const u32x4_t*)(rowp + (((cig * 8 + g) ^ sw) << 4));
            dst[1] = *(const u32x4_t*)(rowp + (((cig * 8 + 4 + g) ^ sw) << 4));
        };
        ldP(0, P[0]);
        if (!RW_ABL(1))
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
                        acc[i][j] = mfma16<T>(W[kh * 3 + kw][s][i], P[step & 1][s], acc[i][j]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }

        if constexpr (CIG == 2) {
            // hand two pixel fragments to the partner wave (w ^ 1, the other cin half), take its sums for the other two
            f32x4_t* xw = (f32x4_t*)(smem + G_::OFF_XCH + w * 4096) + lane;
            const f32x4_t* xr = (const f32x4_t*)(smem + G_::OFF_XCH + (w ^ 1) * 4096) + lane;
            if (cig == 0) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < 2; ++i) xw[(jj * 2 + i) * 64] = acc[i][2 + jj];
            } else {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < 2; ++i) xw[(jj * 2 + i) * 64] = acc[i][jj];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (cig == 0) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][jj] += xr[(jj * 2 + i) * 64];
            } else {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][jj] = acc[i][2 + jj] + xr[(jj * 2 + i) * 64];   // finished set in [0], [1]
            }
        }

        // ---- epilogue from registers: fragments acc[.][0 .. NJ - 1] are output rows oh0 .. oh0 + NJ - 1 ----
        float bv[8];
        *(f32x4_t*)&bv[0] = *(const f32x4_t*)(smem + G_::OFF_BIAS + cstart * 4);
        *(f32x4_t*)&bv[4] = *(const f32x4_t*)(smem + G_::OFF_BIAS + cstart * 4 + 16);
        RW_PROBE(1);
        // retire the gate pieces of t, the patch pieces of t + 1 and the stores of t - 1
        if (NBUF == 3 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(SLOTS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RW_PROBE(2);
        if (grpB) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        RW_PROBE(3);
        float pm[8];                                                       // pooling: the even row of the current row pair
        u32x4_t pkm = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) pm[e] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool ok = okw && oh0 + j < a.Ho;
            float v[8];
            u32x4_t gq;
            if constexpr (GATED) gq = *(const u32x4_t*)(smem + G_::OFF_GATE + (w * NJ + j) * 1024 + lane * 16);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][j][c]), __float_as_uint(acc[1][j][c]), false, false);
                v[c] = __uint_as_float(r[0]);
                v[4 + c] = __uint_as_float(r[1]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = fmaxf(v[e] + bv[e], relu_lo);             // (one v_max_f32; `if (a.relu)` cost a v_cndmask per element)
                if constexpr (GATED) {
                    const uint32_t gw = gq[e >> 1];
                    const float gv = __uint_as_float((e & 1) ? (gw & 0xffff0000u) : (gw << 16));
                    x = (gv > 0.f) ? x : 0.f;
                }
                v[e] = x;
                if constexpr (COLSUM) cst[e] += ok ? x : 0.f;
            }
            u32x4_t pk;
            pk.x = pack2<T>(v[0], v[1]);
            pk.y = pack2<T>(v[2], v[3]);
            pk.z = pack2<T>(v[4], v[5]);
            pk.w = pack2<T>(v[6], v[7]);
            if (!RW_ABL(2) && !a.skip_x && !nostore)             // (wave-uniform conditions)
                __builtin_amdgcn_raw_buffer_store_b128(pk, rsO, ok ? ((m0 + j * a.Wo) * a.ldo + cstart) * 2u : kOOBst, 0, 0);
            if constexpr (!GATED) {
                if (a.pool && a.pcode) {
                    // pooling on the packed 16-bit patterns (post-ReLU values are >= 0: they order like unsigned integers)
                    const u32x4_t pz = ok ? pk : u32x4_t{0u, 0u, 0u, 0u};
                    if ((j & 1) == 0) pkm = pz;
                    else {
                        u32x4_t Mo, Co4;
                        auto pk_max = [](uint32_t x, uint32_t y) { uint32_t r; asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
                        auto pk_min = [](uint32_t x, uint32_t y) { uint32_t r; asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
                        auto pk_sub = [](uint32_t x, uint32_t y) { uint32_t r; asm volatile("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
                        auto pk_mad = [](uint32_t x, uint32_t y, uint32_t z) { uint32_t r; asm volatile("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z)); return r; };
                        const uint32_t one = 0x00010001u, four = 0x00040004u;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t A = pkm[i], Cq = pz[i];
                            const uint32_t Bq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)A, 0xB1, 0xF, 0xF, true);
                            const uint32_t Dq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)Cq, 0xB1, 0xF, 0xF, true);
                            const uint32_t M = pk_max(pk_max(A, Bq), pk_max(Cq, Dq));
                            const uint32_t na = pk_min(pk_sub(M, A), one), nb = pk_min(pk_sub(M, Bq), one), nc = pk_min(pk_sub(M, Cq), one);
                            // first maximum in scan order (a, b, c, d): na (1 + nb (1 + nc)); not positive: 4
                            const uint32_t t = pk_mad(nb, pk_mad(nc, one, one), one);          // 1 + nb (1 + nc)
                            uint32_t code = pk_mad(na, t, 0u);
                            const uint32_t z = pk_min(M, one);
                            code = pk_mad(pk_sub(code, four), z, four);
                            Mo[i] = M;
                            Co4[i] = code;
                        }
                        const int poh = (oh0 + j) >> 1, pw = ow >> 1;
                        if (!RW_ABL(2)) {
                            const bool pst = (r16 & 1) == 0 && okw && poh < a.Hp;
                            const unsigned pe = (unsigned)((b * a.Hp + poh) * a.Wp + pw) * (32 * COG) + cstart;
                            __builtin_amdgcn_raw_buffer_store_b128(Mo, rsPo, pst ? pe * 2u : kOOBst, 0, 0);
                            u32x2_t cb;
                            cb.x = __builtin_amdgcn_perm(Co4[1], Co4[0], 0x06040200u);
                            cb.y = __builtin_amdgcn_perm(Co4[3], Co4[2], 0x06040200u);
                            __builtin_amdgcn_raw_buffer_store_b64(cb, rsPc, pst ? pe : kOOBst, 0, 0);
                        }
                    }
                } else
                // fused MaxPool2d(2,2,ceil): rows (j, j + 1) pair up in this lane (oh0 is even), columns (ow, ow ^ 1) in
                // neighbouring lanes; post-ReLU values are >= 0, so out-of-range window members count as 0
                if (a.pool) {
                    if ((j & 1) == 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) pm[e] = ok ? v[e] : 0.f;
                    } else {
                        float mx[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float m2 = fmaxf(pm[e], ok ? v[e] : 0.f);
                            const float nb = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(m2), 0xB1, 0xF, 0xF, true));
                            mx[e] = fmaxf(m2, nb);                         // quad_perm [1,0,3,2]: the lane of column ow ^ 1
                        }
                        const int poh = (oh0 + j) >> 1, pw = ow >> 1;
                        if ((r16 & 1) == 0 && okw && poh < a.Hp && !RW_ABL(2)) {
                            u32x4_t pq;
                            pq.x = pack2<T>(mx[0], mx[1]);
                            pq.y = pack2<T>(mx[2], mx[3]);
                            pq.z = pack2<T>(mx[4], mx[5]);
                            pq.w = pack2<T>(mx[6], mx[7]);
                            *(u32x4_t*)(a.pool + ((size_t)((b * a.Hp + poh) * a.Wp + pw) * (32 * COG) + cstart) * 2) = pq;
                        }
                    }
                }
            }
        }
        RW_PROBE(4);
        if (!grpB) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        RW_PROBE(5);
        buf = (buf == NBUF - 1) ? 0 : buf + 1;
    }
#ifdef SZN_ABLATE_BUILD
    if ((a.ablate & 8) && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float* pr = (float*)a.out + ((size_t)blockIdx.x * 8 + w) * 8;
        for (int i = 0; i < 6; ++i) pr[i] = (float)pc[i];
        pr[6] = (float)(last - first);
        pr[7] = 1.f;
    }
#endif
    if constexpr (COLSUM) {
        // lane (g, r16 == 0) of wave w holds, after the row sum, the 8 couts cstart .. cstart + 7 of its wave: 8 waves x 4 rows x 8
        // values go to LDS (the patch buffers are free behind the barrier) and thread c adds the waves that own cout c in
        // ascending wave order
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        float* pw = (float*)smem;                           // [8 waves][4 g][8]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = row16_sum(cst[e]);
            if (r16 == 0) pw[(w * 4 + g) * 8 + e] = x;
        }
        __syncthreads();
        if (tid < CO) {
            const int cogc = tid >> 5, rem = tid & 31;
            const int gc = (((rem & 15) >> 3) << 1) | (rem >> 4), ec = rem & 7;     // inverse of cstart = 32 cog + 16 (g & 1) + 8 (g >> 1)
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww)
                if ((ww / CIG) % COG == cogc) s += pw[(ww * 4 + gc) * 8 + ec];
            if (a.cslab) a.cslab[(long)blockIdx.x * CO + tid] = s;
            else if (s != 0.f) atomicAdd(a.colsum + tid, s);
        }
    }
#endif
}

// the skipped tiles: every pixel = the reference pixel (the un-pooled output unless pool_only, the pooled output and its winner codes).
// One block per (image, tile row, group of CB_FILL_G tile columns): it walks the skippable tiles of its group (a block per tile
// was 16 k blocks of a few KB each, a block per whole row too few blocks for the wide layers).
constexpr int CB_FILL_G = 8;
__global__ __launch_bounds__(256) void regw_const_fill_kernel(CbGeom c, int TR, int Co, int Ho, int Wo, int Hp, int Wp, uint16_t* __restrict__ out,
                                                             int ldo, uint16_t* __restrict__ pool, unsigned char* __restrict__ pcode,
                                                             int ref_oh, int ref_ow) {
    const int ngrp = (c.tiles_x + CB_FILL_G - 1) / CB_FILL_G;
    const int grp = blockIdx.x % ngrp, ty = (blockIdx.x / ngrp) % c.tiles_y, b = blockIdx.x / (ngrp * c.tiles_y);
    if (ty < c.fy0 || ty >= c.fy1) return;
    const int g0 = grp * CB_FILL_G, g1 = g0 + CB_FILL_G;
    const int cpp = Co / 8;                                   // 16-B chunks per pixel
    const bool inwin = ty >= c.wy0 && ty < c.wy1;
    // skippable columns of this row: [fx0, fx1) minus the window columns -> pixel columns [x0, x1) in up to two runs
    for (int run = 0; run < 2; ++run) {
        int t0, t1;
        if (!inwin) { if (run) break; t0 = c.fx0; t1 = c.fx1; }
        else if (run == 0) { t0 = c.fx0; t1 = c.wx0; }
        else { t0 = c.wx1; t1 = c.fx1; }
        t0 = t0 > g0 ? t0 : g0; t1 = t1 < g1 ? t1 : g1;
        if (t1 <= t0) continue;
        const int x0 = t0 * 16, nx = (t1 - t0) * 16;
        if (out) {
            const uint4* ref = (const uint4*)(out + ((long)ref_oh * Wo + ref_ow) * ldo);
            for (int i = threadIdx.x; i < TR * nx * cpp; i += 256) {
                const int cc = i % cpp, px = i / cpp;
                const int oh = ty * TR + px / nx, ow = x0 + px % nx;
                *(uint4*)(out + ((long)(b * Ho + oh) * Wo + ow) * ldo + cc * 8) = ref[cc];
            }
        }
        if (pool) {
            const int rp = ref_oh >> 1, rq = ref_ow >> 1;
            const uint4* pref = (const uint4*)(pool + ((long)rp * Wp + rq) * Co);
            const uint2* cref = pcode ? (const uint2*)(pcode + ((long)rp * Wp + rq) * Co) : nullptr;
            const int px0 = x0 >> 1, npx = nx >> 1;
            for (int i = threadIdx.x; i < (TR / 2) * npx * cpp; i += 256) {
                const int cc = i % cpp, px = i / cpp;
                const int ph = ty * (TR / 2) + px / npx, pw = px0 + px % npx;
                const long pe = ((long)(b * Hp + ph) * Wp + pw) * Co + cc * 8;
                *(uint4*)(pool + pe) = pref[cc];
                if (cref) *(uint2*)(pcode + pe) = cref[cc];
            }
        }
    }
}

template <typename T, int COG, int CIG, bool GATED, bool COLSUM>
int launch_regw(const RwArgs& a, int grid, hipStream_t st) {
    constexpr int lds = RwGeom<COG, CIG>::LDS;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv3x3_regw<T, COG, CIG, GATED, COLSUM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv3x3_regw<T, COG, CIG, GATED, COLSUM>), dim3((unsigned)grid), dim3(512), lds, st, a);
    SZN_CHECK_LAUNCH("conv3x3_regw");
    return SZN_OK;
}

template <typename T, int COG, int CIG>
int launch_regw_flags(const RwArgs& a, int grid, hipStream_t st) {
    if (a.gate) return a.colsum ? launch_regw<T, COG, CIG, true, true>(a, grid, st) : launch_regw<T, COG, CIG, true, false>(a, grid, st);
    return a.colsum ? launch_regw<T, COG, CIG, false, true>(a, grid, st) : launch_regw<T, COG, CIG, false, false>(a, grid, st);
}

}  // namespace

// dgrad form of the constant-border hint with cb_on == 2 (d = the swapped descriptor szn_conv2d_dgrad hands to szn_conv2d_fwd: its
// "output" is din): the kept tiles are those that meet the rows x columns the caller reads (cb_const) or the rows x columns in which the
// gate varies (cb_rect); everything else -- whole tile rows above and below, the columns left and right -- is skipped.
static bool regw_dgrad_skip_geom(const szn_conv_desc_t* d, int tr, CbGeom& c) {
    const int ty_n = szn_div_up(d->Ho, tr), tx_n = szn_div_up(d->Wo, 16);
    const int y0 = std::max(std::min(d->cb_rect[0], d->cb_const[0]), 0), y1 = std::min(std::max(d->cb_rect[1], d->cb_const[1]), d->Ho);
    const int x0 = std::max(std::min(d->cb_rect[2], d->cb_const[2]), 0), x1 = std::min(std::max(d->cb_rect[3], d->cb_const[3]), d->Wo);
    if (y1 <= y0 || x1 <= x0 || d->pad != 1) return false;
    c = CbGeom{};
    c.on = 1; c.tiles_y = ty_n; c.tiles_x = tx_n;
    c.fy0 = 0; c.fy1 = ty_n; c.fx0 = 0; c.fx1 = tx_n;
    c.wy0 = y0 / tr; c.wy1 = (y1 + tr - 1) / tr; c.wx0 = x0 / 16; c.wx1 = (x1 + 15) / 16;
    cb_finish(c);
    const long all = (long)ty_n * tx_n;
    return c.per_image > 0 && (all - c.per_image) * 10 >= all;          // worth it from 10 % skipped tiles
}

// Called by szn_conv2d_fwd after it has validated the descriptor. Returns 1 if the layer is not this kernel's shape.
int szn_conv_regw_try(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias, const void* gate,
                      const float* chan_scale, void* out, int min_tiles, szn_stream_t stream) {
    if (!szn_is16(d->dtype) || d->KH != 3 || d->KW != 3 || d->pad > 2 || d->out_f32 || chan_scale) return 1;
    if ((d->Ci != 64 && d->Ci != 128) || (d->Co != 64 && d->Co != 128)) return 1;
    if ((d->ldo & 7) || (d->ldi & 7) || (gate && (d->ldg & 7))) return 1;
    if (d->pool_out && (gate || (size_t)d->B * ((d->Ho + 1) / 2) * ((d->Wo + 1) / 2) * d->Co * 2 >= 0xffff0000ul)) return 1;
    const size_t in_bytes = (size_t)d->B * d->Hi * d->Wi * d->ldi * 2;
    const size_t gate_bytes = gate ? (size_t)d->B * d->Ho * d->Wo * d->ldg * 2 : 0;
    if (in_bytes >= 0x7fff0000ul || gate_bytes >= 0x7fff0000ul || (size_t)d->B * d->Ho * d->Wo * d->ldo * 2 >= 0xffff0000ul)
        return 1;
    const int cog = d->Co / 32, cig = d->Ci / 64;
    const int tr = 4 * (8 / (cog * cig));
    RwArgs a;
    a.in = (const char*)in; a.w = (const char*)w; a.bias = bias; a.gate = (const char*)gate; a.out = (char*)out;
    a.colsum = d->colsum;
    a.cslab = d->colsum ? d->colsum_slab : nullptr;
    a.pool = (char*)d->pool_out; a.Hp = (d->Ho + 1) / 2; a.Wp = (d->Wo + 1) / 2;
    a.pcode = a.pool ? (unsigned char*)d->pool_code : nullptr;
    a.skip_x = (a.pool && d->pool_only) ? 1 : 0;
    a.in_bytes = (unsigned)in_bytes; a.gate_bytes = (unsigned)gate_bytes;
    a.out_bytes = (unsigned)((size_t)d->B * d->Ho * d->Wo * d->ldo * 2);
    a.pool_bytes = a.pool ? (unsigned)((size_t)d->B * a.Hp * a.Wp * d->Co * 2) : 0u;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ho = d->Ho; a.Wo = d->Wo; a.pad = d->pad;
    a.ldi = d->ldi; a.ldo = d->ldo; a.ldg = d->ldg; a.relu = d->relu;
    a.tiles_x = szn_div_up(d->Wo, 16); a.tiles_y = szn_div_up(d->Ho, tr);
    const long nt = (long)a.B * a.tiles_y * a.tiles_x;
    if (nt >= (1L << 30) || nt * tr * 16 < (long)min_tiles * 256) return 1;      // min_tiles counts 256-pixel tiles
    a.ntiles = (int)nt;
    // constant-border hint: run the tiles the image or the zero padding can reach (+ one more tile row, which holds the reference
    // pixel), broadcast the rest
    CbGeom cb = {};
    bool use_cb = false;
    int ref_oh = 0, ref_ow = 0;
    {
        static const int cbe = szn_knob("SZN_CONST_BORDER", 1);
        if (cbe && d->cb_on && !gate && !d->colsum) {
            cb.tiles_y = a.tiles_y; cb.tiles_x = a.tiles_x;
            cb.fy0 = (std::max(d->cb_const[0], 0) + tr - 1) / tr; cb.fy1 = std::min(d->cb_const[1], d->Ho) / tr;
            cb.fx0 = (std::max(d->cb_const[2], 0) + 15) / 16;     cb.fx1 = std::min(d->cb_const[3], d->Wo) / 16;
            cb.wy0 = std::max(d->cb_rect[0], 0) / tr;             cb.wy1 = (std::min(d->cb_rect[1], d->Ho) + tr - 1) / tr;
            cb.wx0 = std::max(d->cb_rect[2], 0) / 16;             cb.wx1 = (std::min(d->cb_rect[3], d->Wo) + 15) / 16;
            if (cb.fy1 > cb.fy0 && cb.fx1 > cb.fx0 && cb.wy0 - 1 >= cb.fy0 && cb.wy0 - 1 < cb.fy1 && cb.wy1 > cb.wy0 && cb.wx1 > cb.wx0 && cb.wx0 >= cb.fx0 && cb.wx0 < cb.fx1) {
                // (the reference tile (wy0 - 1, wx0) has to lie INSIDE the padding-free rectangle: a hint whose window starts at or
                // beyond it would broadcast a pixel that sees the zero padding -- such a hint runs dense)
                const int ry = cb.wy0 - 1, rx = cb.wx0;          // a tile above the window, inside the padding-free rectangle
                cb.wy0 -= 1;                                     // ... its whole row of window tiles is run (no special case in decode())
                cb_finish(cb);
                const long kept = (long)cb.per_image * a.B;
                if (kept * 10 <= nt * 9 && cb.nF > 0 && cb.nW > 0) {                  // worth it from 10 % skipped tiles
                    use_cb = true;
                    cb.on = 1;
                    a.ntiles = (int)kept;
                    ref_oh = ry * tr; ref_ow = rx * 16;
                    szn_note_work_fraction((float)((double)kept / (double)nt));
                }
            }
        }
    }
    a.cb = cb;
    a.dg_on = 0; a.gy0 = a.gy1 = a.gx0 = a.gx1 = a.sy0 = a.sy1 = a.sx0 = a.sx1 = 0; a.gref = 0;
    {
        static const int cbe = szn_knob("SZN_CONST_BORDER", 1);
        // (dgrad form of the hint: see szn_conv_desc_t.cb_on)
        if (cbe && d->cb_on && gate && d->cb_rect[0] >= 1 && d->cb_rect[0] <= d->Ho && d->cb_rect[2] >= 0 && d->cb_rect[2] < d->Wo) {
            a.dg_on = 1;
            a.gy0 = d->cb_rect[0]; a.gy1 = d->cb_rect[1]; a.gx0 = d->cb_rect[2]; a.gx1 = d->cb_rect[3];
            a.sy0 = d->cb_const[0]; a.sy1 = d->cb_const[1]; a.sx0 = d->cb_const[2]; a.sx1 = d->cb_const[3];
            a.gref = (unsigned)((d->cb_rect[0] - 1) * d->Wo + d->cb_rect[2]);
            // cb_on == 2: the tiles that neither store anything nor see a varying gate are not run at all; their share of the column
            // sums is added by szn_conv2d_dgrad_border_finish from region sums of dout (the caller's promise)
            CbGeom sk;
            if (d->cb_on == 2 && d->colsum && regw_dgrad_skip_geom(d, tr, sk)) {
                a.cb = sk;
                a.ntiles = a.B * sk.per_image;
                szn_note_work_fraction((float)sk.per_image / (float)(a.tiles_y * a.tiles_x));
            }
        }
    }
    { static int abl = -1; if (abl < 0) abl = szn_ablate_env("SZN_REGW_ABLATE"); a.ablate = abl; }
    { const int sh = 1; /* (was SZN_REGW_SHIFT) */ a.shift = sh; }
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ncu = p.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
        ncu &= ~7;                                   // whole XCD groups
        if (ncu < 8) ncu = 8;
    }
    // reserved_cus: leave that many CUs (whole XCD groups of 8 blocks) to another queue -- the RCCL all-reduce that runs under the
    // backward pass; a persistent block per CU cannot share its CU's LDS with an RCCL workgroup, which would otherwise wait for a
    // whole kernel (or make one block of this kernel wait for a whole collective)
    const int ncu_all = ncu;
    int grid_cus = ncu_all;
    if (d->reserved_cus > 0) { grid_cus = (ncu_all - d->reserved_cus) & ~7; if (grid_cus < 8) grid_cus = 8; }
    hipStream_t st = (hipStream_t)stream;
    if (a.cslab && d->colsum_slab_rows < grid_cus) SZN_FAIL(SZN_ERR_ARG, "conv2d: colsum_slab holds %d rows, %d needed", d->colsum_slab_rows, grid_cus);
    szn_note_colsum_rows(a.cslab ? grid_cus : 0);
    int rc;
    if (d->dtype == SZN_F16) {
        if (cog == 2 && cig == 1) rc = launch_regw_flags<f16_raw, 2, 1>(a, grid_cus, st);
        else if (cog == 4 && cig == 1) rc = launch_regw_flags<f16_raw, 4, 1>(a, grid_cus, st);
        else if (cog == 2 && cig == 2) rc = launch_regw_flags<f16_raw, 2, 2>(a, grid_cus, st);
        else rc = launch_regw_flags<f16_raw, 4, 2>(a, grid_cus, st);
    } else if (cog == 2 && cig == 1) rc = launch_regw_flags<bf16_raw, 2, 1>(a, grid_cus, st);
    else if (cog == 4 && cig == 1) rc = launch_regw_flags<bf16_raw, 4, 1>(a, grid_cus, st);
    else if (cog == 2 && cig == 2) rc = launch_regw_flags<bf16_raw, 2, 2>(a, grid_cus, st);
    else rc = launch_regw_flags<bf16_raw, 4, 2>(a, grid_cus, st);
    if (rc || !use_cb) return rc;
    hipLaunchKernelGGL(regw_const_fill_kernel, dim3((unsigned)(a.B * a.tiles_y * ((a.tiles_x + CB_FILL_G - 1) / CB_FILL_G))), dim3(256), 0, st, cb, tr, d->Co, d->Ho, d->Wo, a.Hp, a.Wp,
                       a.skip_x ? nullptr : (uint16_t*)a.out, d->ldo, (uint16_t*)a.pool, a.pcode, ref_oh, ref_ow);
    SZN_CHECK_LAUNCH("conv3x3_regw");
    return SZN_OK;
}

// ---- the skipped tiles' share of the column sums (cb_on == 2) -------------------------------------------------------------------------
// din = conv3x3(dout, wT), pad 1, gated.  Over the skipped region R = map \ Wpx the gate is one value per channel, so
//   sum_{p in R} din[p][ci] = g[ci] * sum_{tap, co} wT[ci][tap][co] * D_tap[co],    D_tap[co] = sum_{p in R} dout[p + delta_tap][co],
// and D_tap differs from S0 = sum_{q in R} dout[q] (which the producer of dout delivers: szn_maxpool2x2_ceil_bwd_code_cb) only by
// one-pixel strips along the edges of the map and of Wpx:  D_tap = S0 + Delta(map, delta) - Delta(Wpx, delta),  Delta(rect, delta) =
// (sum of dout over the shifted rect, clipped to the map) - (sum over the rect).  border_strips_kernel sums the strips -- per rect four
// row segments (rows y0 - 1, y0, y1 - 1, y1 over [x0, x1)), four column segments (columns x0 - 1, x0, x1 - 1, x1 over [y0, y1)) and the
// sixteen pixels where they cross -- and border_finish_kernel assembles the nine D_tap and applies the filter bank.
namespace {
struct StripArgs {
    const char* dout; int B, H, W, Co, ldd, is_f16;
    int rect[2][4];            // {y0, y1, x0, x1} of the map and of Wpx
    float* prim;               // [2][24][B][Co]
};
// block (item, rect, image): eight independent 16-B loads in flight per thread (one at a time made these two small kernels cost more
// than the tiles they replace: 0.07 ms slower per step instead of 0.15 faster)
__global__ __launch_bounds__(256) void border_strips_kernel(StripArgs a) {
    __shared__ float red[256][9];
    const int item = blockIdx.x, r = blockIdx.y, b = blockIdx.z;
    const int y0 = a.rect[r][0], y1 = a.rect[r][1], x0 = a.rect[r][2], x1 = a.rect[r][3];
    const int ys[4] = {y0 - 1, y0, y1 - 1, y1}, xs[4] = {x0 - 1, x0, x1 - 1, x1};
    // items 0..3: row ys[item] over [x0, x1); 4..7: column xs[item - 4] over [y0, y1); 8..23: pixel (ys[(item - 8) / 4], xs[(item - 8) % 4])
    int py, px, n, dy, dx;
    if (item < 4) { py = ys[item]; px = x0; n = x1 - x0; dy = 0; dx = 1; }
    else if (item < 8) { py = y0; px = xs[item - 4]; n = y1 - y0; dy = 1; dx = 0; }
    else { py = ys[(item - 8) >> 2]; px = xs[(item - 8) & 3]; n = 1; dy = 0; dx = 0; }
    const int chunks = a.Co >> 3, ppi = 256 / chunks;
    const int ch = threadIdx.x % chunks, pl = threadIdx.x / chunks;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto ld = [&](int i) -> uint4 {
        const int y = py + dy * i, x = px + dx * i;
        if (i >= n || y < 0 || y >= a.H || x < 0 || x >= a.W) return make_uint4(0u, 0u, 0u, 0u);            // outside the map: zeros
        return *(const uint4*)(a.dout + ((size_t)((b * a.H + y) * a.W + x) * a.ldd + ch * 8) * 2);
    };
    auto add = [&](const uint4& v) {
        const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint16_t lo = (uint16_t)(wv[e] & 0xffffu), hi = (uint16_t)(wv[e] >> 16);
            s[2 * e] += a.is_f16 ? f16_bits_to_f32(lo) : bf16_bits_to_f32(lo);
            s[2 * e + 1] += a.is_f16 ? f16_bits_to_f32(hi) : bf16_bits_to_f32(hi);
        }
    };
    if (pl < ppi)
        for (int q = pl; q < n; q += 8 * ppi) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld(q + u * ppi);
#pragma unroll
            for (int u = 0; u < 8; ++u) add(v[u]);
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
    __syncthreads();
    for (int co = threadIdx.x; co < a.Co; co += 256) {
        const int c8 = co >> 3, e = co & 7;
        float t = 0.f;
        for (int p = 0; p < ppi; ++p) t += red[p * chunks + c8][e];
        a.prim[(((size_t)r * 24 + item) * a.B + b) * a.Co + co] = t;
    }
}

struct FinishArgs {
    const float* prim; const float* s0; const char* wT; const char* gate; float* colsum;
    int B, Co, Ci, is_f16; unsigned gref_bytes;
};
// Delta(rect, (dy, dx)) from the 24 primitives of a rect (P[item] = that primitive of channel co, summed over the images)
__device__ __forceinline__ float border_delta(const float* P, int dy, int dx) {
    auto row = [&](int i) { return P[i]; };                                // rows y0 - 1, y0, y1 - 1, y1
    auto col = [&](int i) { return P[4 + i]; };                            // columns x0 - 1, x0, x1 - 1, x1
    auto pix = [&](int yi, int xi) { return P[8 + yi * 4 + xi]; };
    float d = 0.f;
    if (dy == 1) d += row(3) - row(1);
    else if (dy == -1) d += row(0) - row(2);
    // x shift of the rows [y0 + dy, y1 + dy): a column segment over [y0, y1) corrected by its two end pixels
    auto colshift = [&](int xi) {
        float c = col(xi);
        if (dy == 1) c += pix(3, xi) - pix(1, xi);
        else if (dy == -1) c += pix(0, xi) - pix(2, xi);
        return c;
    };
    if (dx == 1) d += colshift(3) - colshift(1);
    else if (dx == -1) d += colshift(0) - colshift(2);
    return d;
}
__global__ __launch_bounds__(256) void border_finish_kernel(FinishArgs a) {
    __shared__ float D[9][128];
    __shared__ float Pl[48][128];
    // the 48 primitives per channel, summed over the images in order: 256 / Co thread groups share the (rect, item) pairs and keep eight
    // loads in flight (one thread per channel walking all 48 x B values took 100 us)
    {
        const int co = threadIdx.x % a.Co, grp = threadIdx.x / a.Co, ngrp = 256 / a.Co;
        for (int pr = grp; pr < 48; pr += ngrp) {
            const float* src = a.prim + (size_t)pr * a.B * a.Co + co;
            float t = 0.f;
            int b = 0;
            for (; b + 8 <= a.B; b += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(b + u) * a.Co];
#pragma unroll
                for (int u = 0; u < 8; ++u) t += v[u];
            }
            for (; b < a.B; ++b) t += src[(size_t)b * a.Co];
            Pl[pr][co] = t;
        }
    }
    __syncthreads();
    for (int co = threadIdx.x; co < a.Co; co += 256) {
        float P[2][24];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int it = 0; it < 24; ++it) P[r][it] = Pl[r * 24 + it][co];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                D[kh * 3 + kw][co] = a.s0[co] + border_delta(P[0], kh - 1, kw - 1) - border_delta(P[1], kh - 1, kw - 1);
    }
    __syncthreads();
    // thread = (ci, quarter of the couts): 9 x Co / 4 products from 16-B loads of the filter bank, then the four quarters by DPP-free
    // shuffles in a fixed order
    const int qtr = threadIdx.x & 3, cq = a.Co >> 2;
    for (int ci = threadIdx.x >> 2; ci < a.Ci; ci += 64) {
        float acc = 0.f;
        for (int tap = 0; tap < 9; ++tap) {
            const uint16_t* wr = (const uint16_t*)a.wT + ((size_t)ci * 9 + tap) * a.Co + qtr * cq;
            for (int c8 = 0; c8 < cq; c8 += 8) {
                const uint4 v = *(const uint4*)(wr + c8);
                const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t lo = (uint16_t)(wv[e] & 0xffffu), hi = (uint16_t)(wv[e] >> 16);
                    acc = fmaf(a.is_f16 ? f16_bits_to_f32(lo) : bf16_bits_to_f32(lo), D[tap][qtr * cq + c8 + 2 * e], acc);
                    acc = fmaf(a.is_f16 ? f16_bits_to_f32(hi) : bf16_bits_to_f32(hi), D[tap][qtr * cq + c8 + 2 * e + 1], acc);
                }
            }
        }
        const float a1 = __shfl_xor(acc, 1, 64);
        const float s01 = (qtr & 1) ? a1 + acc : acc + a1;                 // (quarter 0 + quarter 1), (2 + 3): same order in both lanes
        const float s23 = __shfl_xor(s01, 2, 64);
        const float tot = (qtr & 2) ? s23 + s01 : s01 + s23;
        if (qtr == 0) {
            const uint16_t gb = *(const uint16_t*)(a.gate + a.gref_bytes + (size_t)ci * 2);
            const float gv = a.is_f16 ? f16_bits_to_f32(gb) : bf16_bits_to_f32(gb);
            if (gv > 0.f) a.colsum[ci] += tot;                              // the ReLU gate of the skipped region, one value per channel
        }
    }
}
}  // namespace

static bool border_desc_ok(const szn_conv_desc_t* d) {
    return d && szn_is16(d->dtype) && d->KH == 3 && d->KW == 3 && d->pad == 1 && d->Hi == d->Ho && d->Wi == d->Wo && d->cb_on == 2 &&
           (d->Ci == 64 || d->Ci == 128) && (d->Co == 64 || d->Co == 128) && !(d->ldo & 7) && !(d->ldi & 7) && !(d->ldg & 7) &&
           d->cb_rect[0] >= 1 && d->cb_rect[0] <= d->Hi && d->cb_rect[2] >= 0 && d->cb_rect[2] < d->Wi;
}

// region[8] = pixels {0, Hi, 0, Wi, wy0, wy1, wx0, wx1}: a szn_conv2d_dgrad call with this descriptor (cb_on == 2, gate, colsum) runs only
// the tiles inside rows [wy0, wy1) x columns [wx0, wx1); the sum of dout over the rest of the map is what szn_conv2d_dgrad_border_finish
// wants as skip_sum.  Returns 1, or 0 when the call would run every tile (then no finish call either: result->work_fraction == 1).
extern "C" int szn_conv2d_dgrad_border_region(const szn_conv_desc_t* d, int region[8]) {
    if (!border_desc_ok(d) || !region) return 0;
    static const int cbe = szn_knob("SZN_CONST_BORDER", 1);
    static const int dgb = szn_knob("SZN_DGRAD_BORDER", 1);
    if (!cbe || !dgb) return 0;
    szn_conv_desc_t s = *d;                                                // the swap of szn_conv2d_dgrad
    s.Hi = d->Ho; s.Wi = d->Wo; s.Ci = d->Co; s.Ho = d->Hi; s.Wo = d->Wi; s.Co = d->Ci; s.pad = 1;
    const int cog = s.Co / 32, cig = s.Ci / 64, tr = 4 * (8 / (cog * cig));
    CbGeom c;
    if (!regw_dgrad_skip_geom(&s, tr, c)) return 0;
    region[0] = 0; region[1] = d->Hi; region[2] = 0; region[3] = d->Wi;
    region[4] = c.wy0 * tr; region[5] = std::min(c.wy1 * tr, d->Hi); region[6] = c.wx0 * 16; region[7] = std::min(c.wx1 * 16, d->Wi);
    return 1;
}

extern "C" int szn_conv2d_dgrad_border_finish(const szn_conv_desc_t* d, const void* dout, const void* wT, const void* gate,
                                              const float* skip_sum, float* colsum, void* workspace, szn_stream_t stream) {
    int region[8];
    if (!dout || !wT || !gate || !skip_sum || !colsum || !workspace || ((uintptr_t)workspace & 15))
        SZN_FAIL(SZN_ERR_ARG, "conv2d_dgrad_border_finish: null / unaligned pointer");
    if (!szn_conv2d_dgrad_border_region(d, region)) SZN_FAIL(SZN_ERR_UNSUPPORTED, "conv2d_dgrad_border_finish: this call skips nothing");
    StripArgs sa;
    sa.dout = (const char*)dout; sa.B = d->B; sa.H = d->Ho; sa.W = d->Wo; sa.Co = d->Co; sa.ldd = d->ldo; sa.is_f16 = d->dtype == SZN_F16;
    sa.rect[0][0] = 0; sa.rect[0][1] = d->Ho; sa.rect[0][2] = 0; sa.rect[0][3] = d->Wo;
    sa.rect[1][0] = region[4]; sa.rect[1][1] = region[5]; sa.rect[1][2] = region[6]; sa.rect[1][3] = region[7];
    sa.prim = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(border_strips_kernel, dim3(24, 2, (unsigned)d->B), dim3(256), 0, st, sa);
    SZN_CHECK_LAUNCH("border_strips_kernel");
    FinishArgs fa;
    fa.prim = sa.prim; fa.s0 = skip_sum; fa.wT = (const char*)wT; fa.gate = (const char*)gate; fa.colsum = colsum;
    fa.B = d->B; fa.Co = d->Co; fa.Ci = d->Ci; fa.is_f16 = sa.is_f16;
    fa.gref_bytes = (unsigned)(((size_t)(d->cb_rect[0] - 1) * d->Wi + d->cb_rect[2]) * d->ldg * 2);
    hipLaunchKernelGGL(border_finish_kernel, dim3(1), dim3(256), 0, st, fa);
    SZN_CHECK_LAUNCH("border_finish_kernel");
    return SZN_OK;
}
