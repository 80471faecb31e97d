// szn_conv_wgrad_taps.hip -- weight gradient of the 3x3 layers, all nine taps from ONE staged input patch (bf16).
//
// conv_wgrad_v2 runs one GEMM per filter tap and re-stages both operands for each of them (256 B of LDS fill per MFMA);
// on the 64/128-channel layers at 710^2 / 355^2 it sits at ~250 TFLOP/s.  Here a persistent block owns a
// (64 couts) x (64 cins) x (9 taps) slice of dW -- 144 accumulator fragments = 18 per wave, kept in registers for the
// whole kernel -- and walks a run of 16 x 16 output tiles.  Per tile it stages the dout tile [256 px][64 co] and the
// input patch [18 x 18 px][64 ci] ONCE (73 KiB per 1152 MFMA = 64 B per MFMA) with LDS-DMA (out-of-range offsets give
// the zero padding / ragged edges), double-buffered against the MFMAs of the previous tile.
//   * the contraction index is the pixel: fragments come out of the pixel-major images with ds_read_b64_tr_b16; a K step
//     is 32 pixels = 2 tile rows.  Tap (kh, kw) of K step p reads patch rows 2p + kh, 2p + kh + 1 at column shift kw: the
//     4 x 3 (row, shift) reads of a step serve all nine taps, and two of the four rows carry over to the next step,
//     so a wave issues 4 (dout) + 6 (patch) ds_read_b64_tr per 18 MFMA;
//   * wave (h, c): couts 32 h .. + 31 (2 fragments) x cins 16 c .. + 15 x 9 taps;
//   * both images use the row swizzle chunk ^= ((row >> 1) & 3) << 1 (source side of the DMA): any 8 consecutive rows
//     are conflict-free for the transpose reads, whatever the start row (the patch reads start anywhere);
//   * every block writes its fp32 partial [64][9][64] into its own slab of the caller's workspace with coalesced
//     plain stores; wgrad_taps_reduce adds the slabs in a fixed order (deterministic, no atomics, no memset).
#include "szn_common.h"
#include "szn_cb.h"
#include <stdlib.h>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((address_space(3))) void* ldsptr_t;

namespace {

struct WtArgs {
    const char* dout; const char* in; float* dw; float* ws;
    uint16_t* dw_lp; int lp_f16;       // szn_conv_desc_t.dw_lp: the reduce kernel delivers the gradient as a 16-bit image instead
    unsigned dout_bytes, in_bytes;
    int B, Hi, Wi, Ci, Ho, Wo, Co, pad;
    int ldi, ldd;
    int tiles_x, tiles_y, ntiles;      // ntiles = B * tiles_y * tiles_x
    int cotiles, citiles, nsplit;
    int accumulate;
    int xcd_mode;              // block -> (split, combo) mapping, see the kernel
    int ablate;                // debug (env SZN_WGT_ABLATE, wrong results): 1 = no LDS-DMA in the loop, 2 = no reads / MFMA
    int fill_mode;             // who issues the fill of the next tile when: 0 = wave pair p at K step p (p < 4), 1 = wave group g (waves 4 g .. 4 g + 3) at K step g
    // constant-border hint (szn_conv_desc_t.cb_on for szn_conv2d_wgrad): the tiles whose whole 18 x 18 input patch holds ONE value per
    // channel are not run (cb numbers the others; ntiles = B * cb.per_image).  Their share of dW is a rank-one term, the same for
    // all nine taps:  dW[co][tap][ci] += (sum of dout[px][co] over their pixels) * x_const[ci]  -- csum [Co] comes from
    // wgrad_cb_colsum / wgrad_cb_colsum_reduce, x_const is read from the input at pixel cref; wgrad_taps_reduce adds the product.
    CbGeom cb;
    float* crow;               // [cb_rows][Co]: per-block sums over the skipped tiles (wgrad_cb_colsum)
    float* csum;               // [Co]
    unsigned cref;             // byte offset of the reference pixel in `in`
    int is_f16;
};

constexpr unsigned kOOBt = 0x80000000u;
constexpr int PWt = 18, PROWSt = PWt * PWt;
constexpr int DOUTB = 256 * 128;
constexpr int PATCHBt = 328 * 128;
constexpr int STAGEt = DOUTB + PATCHBt;
constexpr int OFF_DUMPt = 2 * STAGEt;
constexpr int LDS_WT = OFF_DUMPt + 1024;
constexpr int SLAB = 64 * 9 * 64;                   // floats per block partial

template <typename T>      // bf16_raw | f16_raw: only the MFMA opcode differs (operands move as raw 16-bit patterns)
__global__ __launch_bounds__(512) void conv_wgrad_taps(WtArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) bf16x4_t* lp_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = w >> 2, c = w & 3;
    const int g = lane >> 4, r16 = lane & 15;

    // block -> (pixel split, cout tile, cin slice).  The cotiles x citiles combos of one pixel split read the SAME dout tiles
    // (citiles times) and input patches (cotiles times); a.xcd_mode places them on as few XCDs as possible (block b runs on
    // XCD b % 8 -- observed placement, used for speed only), so that the repeats are hits in that XCD's L2 instead of one
    // fabric fetch per XCD: 1 = the split count is a multiple of 8 (split = b % nsplit lives on XCD split % 8),
    // 2 = the split count divides 8 (a split owns 8 / nsplit XCDs), 0 = launch order (combos of a split are neighbours).
    const int ncombo_k = a.cotiles * a.citiles;
    int split, combo;
    if (a.xcd_mode == 1) { split = blockIdx.x % a.nsplit; combo = blockIdx.x / a.nsplit; }
    else if (a.xcd_mode == 2) {
        const int xps = 8 / a.nsplit, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        split = xcd / xps; combo = (xcd % xps) * (ncombo_k / xps) + idx;
    } else { combo = blockIdx.x % ncombo_k; split = blockIdx.x / ncombo_k; }
    const int cit = combo % a.citiles, cot = combo / a.citiles;
    const int slab_id = split * ncombo_k + combo;                  // what wgrad_taps_reduce expects
    const int first = (int)((long)a.ntiles * split / a.nsplit), last = (int)((long)a.ntiles * (split + 1) / a.nsplit);

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.dout, 0, (int)a.dout_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);

    // ---- LDS-DMA slots: 8 rows of 128 B per instruction; lane -> (row = 8 piece + rsub, 16-B slot) ----
    const int rsub = lane >> 3;
    const unsigned chunkoff = (unsigned)(((lane & 7) ^ (((rsub >> 1) & 3) << 1)) << 4);      // (row >> 1) & 3 == (rsub >> 1) & 3
    const unsigned coA = (unsigned)(cot * 128) + chunkoff, coB = (unsigned)(cit * 128) + chunkoff;
    const int q0 = 8 * w + rsub;
    // Offsets of a tile = scalar tile base + a per-lane relative part unless out of range (cheap VALU right behind the
    // barrier; spreading the ten loads over the K steps costs VGPRs the 18 accumulator fragments do not leave).
    unsigned vA[4], vB[6];
    auto prepare = [&](int t) {
        int tx, ty, b;
        if (a.cb.on) {
            b = t / a.cb.per_image;
            cb_decode(a.cb, t - b * a.cb.per_image, ty, tx);
        } else {
            int bb = t;
            tx = bb % a.tiles_x; bb /= a.tiles_x;
            ty = bb % a.tiles_y; b = bb / a.tiles_y;
        }
        int q0v = q0;
        asm volatile("" : "+v"(q0v));               // per-call opaque: no per-slot constants in loop-carried VGPRs
        const unsigned baseA = (unsigned)(((b * a.Ho + ty * 16) * a.Wo + tx * 16) * a.ldd * 2) + coA;
        const int nrow = a.Ho - ty * 16, ncol = a.Wo - tx * 16;     // valid rows / columns of this tile
        const int r0 = q0v >> 4, c0 = q0v & 15;                     // tile pixel of slot p: row r0 + 4 p, column c0
        const bool colok = c0 < ncol;
        const unsigned rel0 = (unsigned)((r0 * a.Wo + c0) * a.ldd * 2), relstep = (unsigned)(4 * a.Wo * a.ldd * 2);
#pragma unroll
        for (int p = 0; p < 4; ++p) vA[p] = (colok && (r0 + 4 * p) < nrow) ? baseA + rel0 + relstep * p : kOOBt;
        const int ih0 = ty * 16 - a.pad, iw0 = tx * 16 - a.pad;
        const unsigned baseB = (unsigned)(((b * a.Hi + ih0) * a.Wi + iw0) * a.ldi * 2) + coB;   // may wrap below zero:
#pragma unroll                                                                                  // only used in range
        for (int p = 0; p < 6; ++p) {
            const int q = q0v + 64 * p;
            const int pr = (q * 3641) >> 16, pc = q - pr * PWt;       // q / 18 for q < 384
            const bool ok = q < PROWSt && (unsigned)(ih0 + pr) < (unsigned)a.Hi && (unsigned)(iw0 + pc) < (unsigned)a.Wi;
            vB[p] = ok ? baseB + (unsigned)((pr * a.Wi + pc) * a.ldi * 2) : kOOBt;
        }
    };
    auto fireA = [&](int p, int stage) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(smem + stage * STAGEt + (w + 8 * p) * 1024), 16, vA[p], 0, 0, 0);
    };
    auto fireB = [&](int p, int stage) {
        const int piece = w + 8 * p;
        char* dst = (piece < 41) ? smem + stage * STAGEt + DOUTB + piece * 1024 : smem + OFF_DUMPt;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)dst, 16, vB[p], 0, 0, 0);
    };

    f32x4_t acc[2][9];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[i][k] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // per-lane transpose-read coordinates: this lane supplies pixel kk (of 16) and 8 B = 4 channels
    const int kk = g * 4 + (r16 >> 2);
    const int sub = (r16 & 3) * 8;
    // dout image rows are 32 p + kk (+ 16): (row >> 1) & 3 == (kk >> 1) & 3 for both
    int offA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) offA[i] = kk * 128 + (((h * 4 + i * 2) ^ (((kk >> 1) & 3) << 1)) << 4) + sub;
    // patch reads: per-lane bases for (parity of s, (s >> 1) & 3), s = 18 R + kw (see the K loop)
    int Xb[2][4];
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int m = 0; m < 4; ++m)
            Xb[par][m] = kk * 128 + sub + (((c * 2) ^ (((((kk + par) >> 1) + m) & 3) << 1)) << 4);

    if (first < last) {
        prepare(first);
#pragma unroll
        for (int p = 0; p < 4; ++p) fireA(p, 0);
#pragma unroll
        for (int p = 0; p < 6; ++p) fireB(p, 0);
    }
    const int smem_lds = (int)(uintptr_t)(__attribute__((address_space(3))) char*)smem;      // LDS byte address of the dynamic segment
    int stage = 0;
    long long tw = 0, ti = 0, tc = 0, c0 = 0, c1 = 0, c2 = 0;   // SZN_WGT_ABLATE=9: cycles in wait+barrier / issue / compute
    // Fragment addresses = per-lane base + compile-time offset (the ds_read offset field), and reads issued ONE K step ahead.
    // A patch read of row R at column shift kw touches flattened patch pixel q = s + kk with s = 18 R + kw known at compile
    // time; its row swizzle ((q >> 1) & 3) is ((kk + (s & 1)) >> 1) + (s >> 1) mod 4, so eight per-lane bases Xb[s & 1][(s >> 1) & 3]
    // cover every (R, kw) and the address costs no VALU work (the compiler had hoisted ~60 per-site offsets into VGPRs and
    // then had no registers left to move the reads away from their MFMAs: each MFMA group waited out the LDS latency of reads
    // issued a few instructions earlier -- the compute phase took 6,800 cycles per tile for 4,600 cycles of MFMA).
    // The transpose reads are issued as inline asm: behind a `buffer_load ... lds` the compiler puts s_waitcnt vmcnt(0) in front
    // of the next LDS read it knows about (the DMA might alias it), so the wave that had just issued the fill of tile t + 1
    // sat out the whole HBM latency of that fill in the middle of its MFMAs -- once per tile, every wave.  The fill goes to the
    // OTHER stage; the waits are explicit: lgkmcnt(0) behind the 18 MFMAs of a K step for the reads issued in front of them
    // (tied to their destination registers), vmcnt(0) + barrier once per tile.
    auto rd_tr = [&](int addr, int off) -> u32x2_t {
        u32x2_t v;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
        return v;
    };
    // B operand of tap (kh, kw) in K step p = patch rows 2p + kh (K half 0) and 2p + kh + 1 (K half 1) at column shift kw: one
    // 4-register group U(r, kw) = {row r, row r + 1} with r = 2p + kh.  Step p multiplies U(2p), U(2p + 1), U(2p + 2); step p + 1
    // needs U(2p + 2) again (tap kh = 2 of one step IS tap kh = 0 of the next) plus two new groups.  Round 5: every group is
    // read from LDS straight into its own register quadruple (two ds_read_b64_tr_b16 each, 12 + 4 reads per step instead of
    // 6 + 4) -- until round 4 a patch row was read once and the nine groups of a step were assembled from row PAIRS with ~12
    // v_mov per step, each in front of the MFMA that consumed it (VALU write -> MFMA read hazard on the critical path of an
    // in-order wave; compute-only form of the loop, SZN_WGT_ABLATE=1: 0.56 of the matrix peak).  LDS traffic 1.6 x, ~100 B/clk.
    u32x4_t U[3][3];
    u32x4_t Af[2][2];
    // Round 5: the K steps run on ACROSS tile boundaries.  Until round 4 a tile began with vmcnt(0) + barrier, then all eight
    // waves issued the fragment reads of K step 0 and waited for them with the matrix pipes idle.  Now the barrier of tile t sits
    // between its K steps 6 and 7: by then every wave holds the fragments of step 7 in registers (reads run one step ahead), so
    // nobody reads stage t any more, and the fill of tile t + 1 -- issued during steps 0 .. 3 -- has had three steps to land.
    // Behind it step 7 issues the reads of tile t + 1's step 0 (from the other stage) in front of its own 18 MFMAs, exactly like
    // steps 0 .. 6 do for their successor.  The fill of tile t + 2 goes into stage t during tile t + 1: behind this barrier too.
    auto rdU = [&](int base, int r, int kw) -> u32x4_t {      // base = LDS address of the stage's patch image (without the per-lane part)
        const int s0 = r * PWt + kw, s1 = (r + 1) * PWt + kw;
        const u32x2_t lo = rd_tr(base + Xb[s0 & 1][(s0 >> 1) & 3], s0 * 128), hi = rd_tr(base + Xb[s1 & 1][(s1 >> 1) & 3], s1 * 128);
        return u32x4_t{lo.x, lo.y, hi.x, hi.y};
    };
    auto rdA0 = [&](int base, int i) -> u32x4_t {
        const u32x2_t l2 = rd_tr(base + offA[i], 0), h2 = rd_tr(base + offA[i], 2048);
        return u32x4_t{l2.x, l2.y, h2.x, h2.y};
    };
    if (first < last) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) U[kh][kw] = rdU(smem_lds + DOUTB, kh, kw);
        Af[0][0] = rdA0(smem_lds, 0); Af[0][1] = rdA0(smem_lds, 1);
        // the waits name the registers the reads are landing in ("+v"): nothing -- not even a register copy -- may touch them earlier
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(U[0][0]), "+v"(U[0][1]), "+v"(U[0][2]), "+v"(U[1][0]), "+v"(U[1][1]), "+v"(U[1][2]),
                       "+v"(U[2][0]), "+v"(U[2][1]), "+v"(U[2][2]), "+v"(Af[0][0]), "+v"(Af[0][1]));
    }
    for (int t = first; t < last; ++t) {
        if (a.ablate == 9) c0 = clock64();
        const bool fill = t + 1 < last && a.ablate != 1;
        const int sdo = smem_lds + stage * STAGEt, sdn = smem_lds + (stage ^ 1) * STAGEt;
        if (a.ablate != 2)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            // The fill of tile t + 1 is issued by wave pair p at K step p (p < 4): ten 1-KiB LDS-DMA loads stall their wave
            // at VMEM issue for a few hundred cycles (SZN_WGT_ABLATE=9), so the waves take turns and the other waves of the
            // CU -- in particular the SIMD partner w +- 4 -- keep the MFMA pipe busy meanwhile.
            if (p < 4 && fill && (a.fill_mode ? (p < 2 && (w >> 2) == p) : (w >> 1) == p)) {
                prepare(t + 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) fireA(q, stage ^ 1);
#pragma unroll
                for (int q = 0; q < 6; ++q) fireB(q, stage ^ 1);
            }
            // the groups and dout fragments of the NEXT step, fetched while this step's 18 MFMAs run
            u32x4_t N[3][3], An[2];
            const bool nxt = p < 7 || fill;
            if (p < 7) {
#pragma unroll
                for (int kh = 1; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) N[kh][kw] = rdU(sdo + DOUTB, 2 * (p + 1) + kh, kw);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const u32x2_t l2 = rd_tr(sdo + offA[i], (p + 1) * 4096), h2 = rd_tr(sdo + offA[i], (p + 1) * 4096 + 2048);
                    An[i] = u32x4_t{l2.x, l2.y, h2.x, h2.y};
                }
            } else if (fill) {
                // every wave's share of tile t + 1 has landed and nobody reads this stage any more (see above)
                if (a.ablate == 9) c1 = clock64();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (a.ablate == 9) tw += clock64() - c1;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) N[kh][kw] = rdU(sdn + DOUTB, kh, kw);
                An[0] = rdA0(sdn, 0); An[1] = rdA0(sdn, 1);
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][kh * 3 + kw] = mfma16<T>(Af[p & 1][i], U[kh][kw], acc[i][kh * 3 + kw]);
            if (p < 7) {
                // behind this step's MFMAs (acc[1][8] is the last one's result): the reads issued in front of them have landed
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(N[1][0]), "+v"(N[1][1]), "+v"(N[1][2]), "+v"(N[2][0]), "+v"(N[2][1]), "+v"(N[2][2]),
                               "+v"(An[0]), "+v"(An[1]), "+v"(acc[1][8]));
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) { U[0][kw] = U[2][kw]; U[1][kw] = N[1][kw]; U[2][kw] = N[2][kw]; }
                Af[(p + 1) & 1][0] = An[0]; Af[(p + 1) & 1][1] = An[1];
            } else if (fill) {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(N[0][0]), "+v"(N[0][1]), "+v"(N[0][2]), "+v"(N[1][0]), "+v"(N[1][1]), "+v"(N[1][2]),
                               "+v"(N[2][0]), "+v"(N[2][1]), "+v"(N[2][2]), "+v"(An[0]), "+v"(An[1]), "+v"(acc[1][8]));
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) U[kh][kw] = N[kh][kw];
                Af[0][0] = An[0]; Af[0][1] = An[1];
            }
            (void)nxt;
        }
        if (a.ablate == 9) tc += clock64() - c0;
        stage ^= 1;
    }
    (void)ti; (void)c2;

    // ---- partial -> slab [wave][fragment f = 9 i + tap][e][lane], 256 contiguous bytes per store instruction ----
    float* slab = a.ws + (size_t)slab_id * SLAB + (size_t)w * (18 * 256) + lane;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) slab[((i * 9 + k) * 4 + e) * 64] = acc[i][k][e];
    if (a.ablate == 9 && tid == 0) {                  // debug: the block's cycle split replaces the head of its slab
        float* dbg = a.ws + (size_t)slab_id * SLAB;
        dbg[0] = (float)tw; dbg[1] = (float)ti; dbg[2] = (float)tc; dbg[3] = (float)(last - first);
    }
#endif
}

// dw[co][tap][ci] (+)= sum over the pixel splits of one (cot, cit) combo.  A block owns 64 groups of 4 consecutive slab elements
// (= 4 consecutive cins); its four waves each add a quarter of the splits with four independent running sums, and wave 0 combines the
// four partial sums in a fixed order (bit-reproducible).  (One thread per group summing ALL splits -- 256 of them for a 64-channel
// layer, in 36 blocks -- took 21 us for conv1_2.)
__global__ __launch_bounds__(256) void wgrad_taps_reduce(WtArgs a) {
    __shared__ f32x4_t part[4][64];
    const int ncombo = a.cotiles * a.citiles;
    const long total4 = (long)ncombo * (SLAB / 4);
    const size_t stride = (size_t)ncombo * SLAB;                     // block id = split * ncombo + combo
    const int grp = threadIdx.x >> 6, l64 = threadIdx.x & 63;
    const int sp0 = (int)((long)a.nsplit * grp / 4), sp1 = (int)((long)a.nsplit * (grp + 1) / 4);
    for (long base = (long)blockIdx.x * 64; base < total4; base += (long)gridDim.x * 64) {
        const long idx = base + l64;
        const bool ok = idx < total4;
        const int combo = ok ? (int)(idx / (SLAB / 4)) : 0, el = ok ? (int)(idx - (long)combo * (SLAB / 4)) * 4 : 0;
        const int cit = combo % a.citiles, cot = combo / a.citiles;
        const float* p = a.ws + (size_t)combo * SLAB + el;
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
        if (ok) {
            int sp = sp0;
            for (; sp + 4 <= sp1; sp += 4) {
                s0 += *(const f32x4_t*)(p + (size_t)sp * stride);
                s1 += *(const f32x4_t*)(p + (size_t)(sp + 1) * stride);
                s2 += *(const f32x4_t*)(p + (size_t)(sp + 2) * stride);
                s3 += *(const f32x4_t*)(p + (size_t)(sp + 3) * stride);
            }
            for (; sp < sp1; ++sp) s0 += *(const f32x4_t*)(p + (size_t)sp * stride);
        }
        part[grp][l64] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        const f32x4_t s = (part[0][l64] + part[1][l64]) + (part[2][l64] + part[3][l64]);
        __syncthreads();
        if (grp != 0 || !ok) continue;
        const int w = el / 4608, f = (el >> 8) % 18, e = (el >> 6) & 3, lane = el & 63;      // lane .. lane + 3: same row g
        const int h = w >> 2, c = w & 3, i = f / 9, tap = f - i * 9;
        const int co = cot * 64 + h * 32 + i * 16 + (lane >> 4) * 4 + e, ci = cit * 64 + c * 16 + (lane & 15);
        float* dst = a.dw + ((long)co * 9 + tap) * a.Ci + ci;
        f32x4_t sv = s;
        if (a.cb.on) {                                 // the skipped tiles: (their column sum of dout) x (the constant input pixel)
            const uint16_t* xr = (const uint16_t*)(a.in + a.cref) + ci;
            const float sc = a.csum[co];
#pragma unroll
            for (int q = 0; q < 4; ++q) sv[q] += sc * (a.is_f16 ? f16_bits_to_f32(xr[q]) : bf16_bits_to_f32(xr[q]));
        }
        if (a.dw_lp) {                                 // the data-parallel wire image: rounded once, from the fp32 sum
            uint2 pk;
            if (a.lp_f16) { pk.x = pack2<f16_raw>(sv[0], sv[1]); pk.y = pack2<f16_raw>(sv[2], sv[3]); }
            else { pk.x = pack2<bf16_raw>(sv[0], sv[1]); pk.y = pack2<bf16_raw>(sv[2], sv[3]); }
            *(uint2*)(a.dw_lp + ((long)co * 9 + tap) * a.Ci + ci) = pk;
        } else if (a.accumulate) {
            const f32x4_t o = *(const f32x4_t*)dst;
            *(f32x4_t*)dst = o + sv;
        } else {
            *(f32x4_t*)dst = sv;
        }
    }
}

// Column sums of dout over the tiles conv_wgrad_taps skips under the constant-border hint.  The skipped tiles of an image are
// numbered row-major (top band, the two side bands of the window rows, bottom band); a block takes `per` consecutive (image, tile)
// units -- 256 pixels each, 16-B chunks of 8 channels per thread, pixels strided over the block, eight independent loads in flight
// per thread -- and writes its sums to crow[block][Co] (fixed order: bit-reproducible).  HBM-bound: the skipped share of dout is read once.
struct CbSkip { int ntop, nmid, per_image, nfw, nsw, nleft; };      // counts of the numbering above (host: cb_skip_counts)
__host__ __device__ inline CbSkip cb_skip_counts(const CbGeom& c) {
    CbSkip k;
    k.nfw = c.fx1 - c.fx0; k.nleft = c.wx0 - c.fx0; k.nsw = k.nfw - (c.wx1 - c.wx0);
    k.ntop = (c.wy0 - c.fy0) * k.nfw; k.nmid = (c.wy1 - c.wy0) * k.nsw;
    k.per_image = k.ntop + k.nmid + (c.fy1 - c.wy1) * k.nfw;
    return k;
}
__device__ __forceinline__ void cb_skip_decode(const CbGeom& c, const CbSkip& k, int v, int& ty, int& tx) {
    if (v < k.ntop) { const int q = v / k.nfw; ty = c.fy0 + q; tx = c.fx0 + (v - q * k.nfw); return; }
    v -= k.ntop;
    if (v < k.nmid) {
        const int q = v / k.nsw, j = v - q * k.nsw;
        ty = c.wy0 + q; tx = j < k.nleft ? c.fx0 + j : c.wx1 + (j - k.nleft);
        return;
    }
    v -= k.nmid;
    const int q = v / k.nfw; ty = c.wy1 + q; tx = c.fx0 + (v - q * k.nfw);
}

__global__ __launch_bounds__(256) void wgrad_cb_colsum(WtArgs a, CbSkip k, int per, int units) {
    __shared__ float red[256][9];
    const int chunks = a.Co >> 3, ppi = 256 / chunks;                      // pixels per iteration (Co <= 2048)
    const int ch = threadIdx.x % chunks, pl = threadIdx.x / chunks;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool is16 = a.is_f16 != 0;
    auto add = [&](const uint4& v) {
        const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint16_t lo = (uint16_t)(wv[e] & 0xffffu), hi = (uint16_t)(wv[e] >> 16);
            s[2 * e] += is16 ? f16_bits_to_f32(lo) : bf16_bits_to_f32(lo);
            s[2 * e + 1] += is16 ? f16_bits_to_f32(hi) : bf16_bits_to_f32(hi);
        }
    };
    const int u0 = blockIdx.x * per, u1 = min(u0 + per, units);
    for (int u = u0; u < u1; ++u) {
        const int b = u / k.per_image;
        int ty, tx;
        cb_skip_decode(a.cb, k, u - b * k.per_image, ty, tx);
        const char* base = a.dout + ((size_t)((b * a.Ho + ty * 16) * a.Wo + tx * 16) * a.ldd + ch * 8) * 2;
        auto addr = [&](int q) -> const uint4* { return (const uint4*)(base + (size_t)((q >> 4) * a.Wo + (q & 15)) * a.ldd * 2); };
        if (pl < ppi) {
            int q = pl;
            for (; q + 7 * ppi < 256; q += 8 * ppi) {
                uint4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *addr(q + i * ppi);
#pragma unroll
                for (int i = 0; i < 8; ++i) add(v[i]);
            }
            for (; q < 256; q += ppi) add(*addr(q));
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
    __syncthreads();
    for (int co = threadIdx.x; co < a.Co; co += 256) {
        const int c8 = co >> 3, e = co & 7;
        float t = 0.f;
        for (int p = 0; p < ppi; ++p) t += red[p * chunks + c8][e];
        a.crow[(size_t)blockIdx.x * a.Co + co] = t;
    }
}

// csum[co] = sum of the rows of crow: a block owns 8 channels, 32 thread groups take the rows r = group (mod 32) with four running
// sums each, then a fixed-order tree over the groups
__global__ __launch_bounds__(256) void wgrad_cb_colsum_reduce(WtArgs a, int rows) {
    __shared__ float part[32][8];
    const int cl = threadIdx.x & 7, grp = threadIdx.x >> 3, co = blockIdx.x * 8 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (co < a.Co) {
        int r = grp;
        for (; r + 96 < rows; r += 128) {
            s0 += a.crow[(size_t)r * a.Co + co]; s1 += a.crow[(size_t)(r + 32) * a.Co + co];
            s2 += a.crow[(size_t)(r + 64) * a.Co + co]; s3 += a.crow[(size_t)(r + 96) * a.Co + co];
        }
        for (; r < rows; r += 32) s0 += a.crow[(size_t)r * a.Co + co];
    }
    part[grp][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x < 8 && co < a.Co) {
        float t = 0.f;
        for (int g2 = 0; g2 < 32; ++g2) t += part[g2][cl];
        a.csum[co] = t;
    }
}

}  // namespace

// The constant-border hint of a weight-gradient call (szn_conv_desc_t.cb_on): cb_rect = the input rows x columns the image can influence,
// cb_const = the rows x columns outside of which the zero padding of the layers so far is felt (include/szn.h).  A tile is skipped when its
// input patch -- rows [16 ty - pad, 16 ty + 18 - pad) -- lies inside cb_const, does not meet cb_rect, and the tile is a full 16 x 16 one.
// -> true + the tile bookkeeping + the reference pixel (top-left pixel of the first skipped tile's patch) when the hint is worth taking.
static bool taps_cb_geometry(const szn_conv_desc_t* d, int ncombo, int ncu, int min_tiles_per_block, CbGeom& out, int& ry, int& rx) {
    static const int cbon = szn_knob("SZN_WGT_CB", 1);
    if (!cbon || !d->cb_on) return false;
    auto fdiv = [](int x, int y) { return x >= 0 ? x / y : -((-x + y - 1) / y); };       // floor
    auto cdiv = [&](int x, int y) { return -fdiv(-x, y); };                              // ceil
    CbGeom c;
    c.on = 1; c.tiles_y = szn_div_up(d->Ho, 16); c.tiles_x = szn_div_up(d->Wo, 16);
    const int p = d->pad;
    c.fy0 = std::max(cdiv(d->cb_const[0] + p, 16), 0);
    c.fy1 = std::min(fdiv(d->cb_const[1] - 18 + p, 16) + 1, d->Ho / 16);
    c.fx0 = std::max(cdiv(d->cb_const[2] + p, 16), 0);
    c.fx1 = std::min(fdiv(d->cb_const[3] - 18 + p, 16) + 1, d->Wo / 16);
    c.wy0 = fdiv(d->cb_rect[0] - 18 + p, 16) + 1; c.wy1 = cdiv(d->cb_rect[1] + p, 16);
    c.wx0 = fdiv(d->cb_rect[2] - 18 + p, 16) + 1; c.wx1 = cdiv(d->cb_rect[3] + p, 16);
    if (!(c.fy1 > c.fy0 && c.fx1 > c.fx0 && d->cb_rect[1] > d->cb_rect[0] && d->cb_rect[3] > d->cb_rect[2])) return false;
    cb_finish(c);
    const long all = (long)c.tiles_y * c.tiles_x, skipped = all - c.per_image;
    ry = rx = -1;
    for (int ty = c.fy0; ty < c.fy1 && ry < 0; ++ty)
        for (int tx = c.fx0; tx < c.fx1; ++tx)
            if (cb_skippable(c, ty, tx)) { ry = ty * 16 - p; rx = tx * 16 - p; break; }
    // (not when the tiles that are left would be too few for this kernel: the dense run then beats conv_wgrad_v2)
    const long left = (long)d->B * c.per_image / (min_tiles_per_block < 1 ? 1 : min_tiles_per_block);
    const bool enough = std::min(left, (long)(ncu / ncombo)) * ncombo >= 32;
    if (!(skipped * 20 >= all && ry >= 0 && rx >= 0 && d->Co <= 2048 && enough)) return false;      // worth the bookkeeping
    out = c;
    return true;
}

// Called by szn_conv2d_wgrad after validation.  Returns 1 if the layer / workspace does not fit this kernel.
int szn_conv_wgrad_taps_try(const szn_conv_desc_t* d, const void* in, const void* dout, float* dw, int accumulate,
                            int min_tiles_per_block, szn_stream_t stream) {
    if (!szn_is16(d->dtype) || d->KH != 3 || d->KW != 3 || (d->Ci & 63) || (d->Co & 63) || d->pad > 2 || !d->workspace)
        return 1;
    if ((d->ldi & 7) || (d->ldo & 7) || ((uintptr_t)dw & 15) || ((uintptr_t)d->workspace & 15)) return 1;
    WtArgs a;
    a.cotiles = d->Co / 64; a.citiles = d->Ci / 64;
    const int ncombo = a.cotiles * a.citiles;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ncu = p.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    if (ncombo > ncu) return 1;
    a.tiles_x = szn_div_up(d->Wo, 16); a.tiles_y = szn_div_up(d->Ho, 16);
    long nt = (long)d->B * a.tiles_y * a.tiles_x;
    if (nt >= (1L << 30)) return 1;
    // constant-border hint: cb_rect = the input rows x columns the image can influence, cb_const = the rows x columns outside of which
    // the zero padding of the layers so far is felt (include/szn.h).  A tile is skipped when its input patch -- rows [16 ty - pad,
    // 16 ty + 18 - pad) -- lies inside cb_const, does not meet cb_rect, and the tile is a full 16 x 16 one.
    a.cb.on = 0; a.crow = a.csum = nullptr; a.cref = 0; a.is_f16 = d->dtype == SZN_F16;
    {
        int ry = 0, rx = 0;
        if (!accumulate && taps_cb_geometry(d, ncombo, ncu, min_tiles_per_block, a.cb, ry, rx)) {
            a.cref = (unsigned)(((size_t)ry * d->Wi + rx) * d->ldi * 2);
            nt = (long)d->B * a.cb.per_image;
        } else {
            a.cb.on = 0;
        }
    }
    a.ntiles = (int)nt;
    // pixel splits: one block per CU at most, at least min_tiles_per_block tiles each (the slab write + reduction
    // must amortise), slabs must fit the workspace; too little parallelism left -> conv_wgrad_v2
    // SZN_WGT_OVERSUB = k (default 1): k blocks per CU instead of one.  One block per CU is fastest on an idle GPU but its
    // static partition has a full-kernel tail whenever another queue (an RCCL all-reduce running under the backward
    // pass) holds CUs; k = 2 halves the late blocks.  Measured with a 32-CU stand-in hog (profiles/r01_ablations.txt): 12.2 vs 12.3
    // ms/step under contention, 11.7 vs 11.9 without -- no net gain, so nothing sets it by default.
    const int oversub = 1; /* (was SZN_WGT_OVERSUB) */
    // reserved_cus: CUs left to another queue (the RCCL all-reduce under the backward pass), see szn_conv_desc_t
    const int cus = (d->reserved_cus > 0 && ncu - d->reserved_cus >= ncombo) ? ncu - d->reserved_cus : ncu;
    long ns = (long)cus * oversub / ncombo;
    if (min_tiles_per_block < 1) min_tiles_per_block = 1;
    if (ns > nt / min_tiles_per_block) {
        // few tiles (conv5_x of ONE image: 9 tiles, 64 combos): the min_tiles rule would leave three quarters of the chip idle while 64
        // blocks walk 9 tiles each; down to two tiles per block the extra slabs cost less than the idle CUs (round 5, B = 1)
        const int small = 1; /* (was SZN_WGT_SMALLSPLIT) */
        const long relaxed = small ? std::max<long>(nt / min_tiles_per_block, std::min<long>(ns, nt / 2)) : nt / min_tiles_per_block;
        ns = relaxed;
    }
    const size_t slab_bytes = (size_t)ncombo * SLAB * sizeof(float);
    int cb_rows = 0, cb_per = 1, cb_units = 0;
    CbSkip ck = {};
    if (a.cb.on) {          // ~1500 blocks of <= 8 tiles
        ck = cb_skip_counts(a.cb);
        cb_units = d->B * ck.per_image;
        cb_per = std::max(1, std::min(8, cb_units / 1024));
        cb_rows = szn_div_up(cb_units, cb_per);
    }
    const size_t cb_bytes = a.cb.on ? ((size_t)cb_rows + 1) * d->Co * sizeof(float) : 0;
    if (cb_bytes + slab_bytes > d->workspace_bytes) return 1;
    if (ns > (long)((d->workspace_bytes - cb_bytes) / slab_bytes)) ns = (long)((d->workspace_bytes - cb_bytes) / slab_bytes);
    const int minblk = 32; /* (was SZN_WGT_MINBLOCKS) */
    if (ns < 1 || ns * ncombo < minblk) return 1;
    a.nsplit = (int)ns;
    const bool own_sum = a.cb.on && !d->colsum;        // d->colsum: the producer of dout already summed the skipped tiles (include/szn.h)
    if (a.cb.on) {                                     // behind the slabs
        a.crow = (float*)d->workspace + (size_t)ns * ncombo * SLAB;
        a.csum = own_sum ? a.crow + (size_t)cb_rows * d->Co : (float*)d->colsum;
    }
    {
        const int xm = 1; /* (was SZN_WGT_XCD) */
        a.xcd_mode = 0;
        if (xm && ncombo >= 4) {        // (two combos: measured 3 % slower than launch order, conv2_1)
            if (ns % 8 == 0) a.xcd_mode = 1;
            else if (ns <= 8 && 8 % ns == 0 && ncombo % (8 / ns) == 0) a.xcd_mode = 2;
        }
    }
    a.dout = (const char*)dout; a.in = (const char*)in; a.dw = dw; a.ws = (float*)d->workspace;
    a.dw_lp = accumulate ? nullptr : (uint16_t*)d->dw_lp; a.lp_f16 = d->dw_lp_dtype == SZN_F16;
    a.dout_bytes = (unsigned)((size_t)d->B * d->Ho * d->Wo * d->ldo * 2);
    a.in_bytes = (unsigned)((size_t)d->B * d->Hi * d->Wi * d->ldi * 2);
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co; a.pad = d->pad;
    a.ldi = d->ldi; a.ldd = d->ldo; a.accumulate = accumulate;
    { static int abl = -1; if (abl < 0) { abl = szn_ablate_env("SZN_WGT_ABLATE"); } a.ablate = abl; }
    { const int fm = 0; /* (was SZN_WGT_FILL) */ a.fill_mode = fm; }
    hipStream_t st = (hipStream_t)stream;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_taps<bf16_raw>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WT);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_taps<f16_raw>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WT);
        attr_done = true;
    }
    if (d->dtype == SZN_F16) hipLaunchKernelGGL(conv_wgrad_taps<f16_raw>, dim3((unsigned)(ns * ncombo)), dim3(512), LDS_WT, st, a);
    else hipLaunchKernelGGL(conv_wgrad_taps<bf16_raw>, dim3((unsigned)(ns * ncombo)), dim3(512), LDS_WT, st, a);
    SZN_CHECK_LAUNCH("conv_wgrad_taps");
    if (a.cb.on) {
        if (own_sum) {
            hipLaunchKernelGGL(wgrad_cb_colsum, dim3((unsigned)cb_rows), dim3(256), 0, st, a, ck, cb_per, cb_units);
            hipLaunchKernelGGL(wgrad_cb_colsum_reduce, dim3((unsigned)szn_div_up(d->Co, 8)), dim3(256), 0, st, a, cb_rows);
            SZN_CHECK_LAUNCH("wgrad_cb_colsum");
        }
        szn_note_work_fraction((float)a.cb.per_image / (float)(a.tiles_y * a.tiles_x));
    }
    const long total4 = (long)ncombo * (SLAB / 4);
    hipLaunchKernelGGL(wgrad_taps_reduce, dim3((unsigned)((total4 + 63) / 64)), dim3(256), 0, st, a);
    SZN_CHECK_LAUNCH("wgrad_taps_reduce");
    return SZN_OK;
}

// Which part of dout a szn_conv2d_wgrad call with this descriptor (constant-border hint set, accumulate = 0) replaces by the rank-one term:
// region[8] = pixel rows / columns {fy0, fy1, fx0, fx1, wy0, wy1, wx0, wx1} (multiples of 16) -- the pixels inside [fy0, fy1) x [fx0, fx1) and
// outside [wy0, wy1) x [wx0, wx1).  Returns 1 (region filled) or 0 (the call runs dense).  For the producer of dout, which can sum that
// region while it writes it (szn_maxpool2x2_ceil_bwd_code_cb) and hand the result to the call as szn_conv_desc_t.colsum.
extern "C" int szn_conv2d_wgrad_cb_region(const szn_conv_desc_t* d, int region[8]) {
    if (!d || !region || !szn_is16(d->dtype) || d->KH != 3 || d->KW != 3 || (d->Ci & 63) || (d->Co & 63) || d->pad > 2 || !d->workspace ||
        (d->ldi & 7) || (d->ldo & 7))
        return 0;
    static int ncu = 0;                              // (cached: this query sits on the host-bound one-image path)
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ncu = p.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    const int ncombo = (d->Co / 64) * (d->Ci / 64);
    if (ncombo > ncu) return 0;
    static const int taps_min = szn_knob("SZN_WGT_MINTILES", 8);
    CbGeom c;
    int ry, rx;
    if (!taps_cb_geometry(d, ncombo, ncu, taps_min, c, ry, rx)) return 0;
    region[0] = c.fy0 * 16; region[1] = c.fy1 * 16; region[2] = c.fx0 * 16; region[3] = c.fx1 * 16;
    region[4] = c.wy0 * 16; region[5] = c.wy1 * 16; region[6] = c.wx0 * 16; region[7] = c.wx1 * 16;
    return 1;
}
