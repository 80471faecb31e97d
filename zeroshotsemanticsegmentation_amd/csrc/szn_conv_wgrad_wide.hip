// szn_conv_wgrad_wide.hip -- 256 x 256 tile weight-gradient kernel (bf16) for the layers with many channels and few
// pixels: fc6 (4096 x 49 x 512, 2312 pixels at B = 8) and fc7 (4096 x 4096).
//
// conv_wgrad_v2 (128 x 128 tile, 4 waves) stages 256 B of operands per MFMA and sits at 460-640 TFLOP/s on these
// layers.  Here: 512 threads = 8 waves (2 x 4), wave = 128 couts x 64 cins (8 x 4 accumulator fragments = 128 VGPRs),
// K step = 32 pixels per barrier, 128 B of LDS fill per MFMA.  dout / input slices are used by few blocks, so most of
// the fill misses L2 and sees HBM / MALL latency (~3 us measured): the fill rate is (bytes in flight) / latency, hence a
// 4-stage ring of 32-KiB stages rather than two 64-KiB stages with one in flight (two stages = 64 KiB stay in flight since the
// wave groups are phase-shifted: see the K loop).
//   * per filter tap a GEMM D[co][ci] = A^T B, A = dout [pixel][256 co], B = shifted input [pixel][256 ci]; both
//     staged pixel-major ([32 px][512 B]) by LDS-DMA into a 4-stage ring (counted vmcnt), fragments by ds_read_b64_tr_b16 (inline asm:
//     behind an LDS-DMA load the compiler would put s_waitcnt vmcnt(0) in front of an LDS read it knows about);
//   * 16-B chunk index XOR-swizzled by (row & 7) << 1 on the DMA source side (8 consecutive 512-B rows of a transpose
//     read land in 8 distinct 32-B bank groups);
//   * out-of-image taps / tile edges / the pixel tail are out-of-range buffer offsets (zeros); pixel coordinates advance
//     incrementally;
//   * one pixel split: a tile is written once (plain stores, or read-add-write when accumulating) through an LDS-staged
//     epilogue in four 64-row passes -- deterministic, no atomics, no memset.
#include "szn_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((address_space(3))) void* ldsptr_t;

#ifndef SZN_WGW_ADAM_AUX
#define SZN_WGW_ADAM_AUX 2      // cache policy of the update's loads / stores: 2 = non-temporal (0 = default policy: 0.075 ms per step slower,
                                // profiles/r04_ablations.txt 18 -- 2.7 GB that nobody reads again soon stays out of the caches' way)
#endif

namespace {

struct WgwArgs {
    const char* dout; const char* in; float* dw;
    uint16_t* dw_lp; int lp_f16;       // szn_conv_desc_t.dw_lp (<T, false> only): the epilogue stores the gradient as a 16-bit image instead
    unsigned dout_bytes, in_bytes;
    int B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, pad;
    int ldi, ldd;
    int M;
    int cotiles, citiles;
    int accumulate;
    int ablate;                // debug (env SZN_WGW_ABLATE, wrong results): 1 = no LDS-DMA in the loop, 2 = no reads / MFMA
    int use_tab;               // 1: pixel -> input-offset table of this block's tap in LDS behind the ring (M <= kTabMax)
    int shift;                 // 1: phase-shifted wave groups (SZN_WGW_SHIFT=0: lockstep)
    int xcd_order;             // 1: an XCD takes a contiguous share of the tiles, cout tile fastest (the B operand is the large one)
    // conv_wgrad_wide<T, true> (szn_conv2d_wgrad_adam): the Adam step of this layer's weights in the epilogue -- fp32 master, both
    // moments and the 16-bit weight image, all in the gradient's OHWI order; dw may then be NULL (gradient not stored)
    float* p; float* m1; float* m2; uint16_t* wlp;
    float b1, b2, eps, wd, step_size, inv_bc2_sqrt, gscale;
    int stagger;               // <T, true>: s_sleep(127) units (~4 us) per phase step of the first-round blocks, see the kernel
    int ncu;                   // compute units of the device (the first-round blocks are blockIdx < ncu)
};

constexpr unsigned kOOBg = 0x80000000u;
constexpr int KPg = 32;                              // pixels per stage
constexpr int STAGEg = KPg * 1024;                   // A rows 512 B + B rows 512 B
constexpr int NSTg = 4;
constexpr int LDS_WGW = NSTg * STAGEg;               // 128 KiB
constexpr int kTabMax = 8000;                        // pixels whose 4-B table entries fit behind the ring (160 KiB LDS)

template <typename T, bool ADAM>      // T = bf16_raw | f16_raw: only the MFMA opcode (and the weight image's packing) differs
__global__ __launch_bounds__(512) void conv_wgrad_wide(WgwArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) bf16x4_t* lp_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;               // wave -> couts 128 wm .. + 127, cins 64 wn .. + 63
    const int g = lane >> 4, r16 = lane & 15;

    int bid = blockIdx.x;
    int cit, cot, tap;
    if (a.xcd_order) {
        // Blocks go to the 8 XCDs round-robin.  When B (the "input": for fc6's dgrad on the forward layout the 205 MB filter bank) is far
        // larger than A, the cout tiles that share one B slice should run on ONE XCD at the same time, so that the slice crosses the
        // fabric once instead of once per cout tile: XCD x takes the contiguous range of tiles, cout tile fastest
        const int nwg = (int)gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        if (a.xcd_order == 2) {           // tap fastest: the 32 CUs of an XCD share ONE (cout tile, cin tile) pair's operand slices
            const int ntap = a.KH * a.KW;
            tap = lid % ntap; lid /= ntap;
            cot = lid % a.cotiles; cit = lid / a.cotiles;
        } else {
            cot = lid % a.cotiles; lid /= a.cotiles;
            cit = lid % a.citiles; tap = lid / a.citiles;
        }
    } else {
        cit = bid % a.citiles; bid /= a.citiles;
        cot = bid % a.cotiles; tap = bid / a.cotiles;
    }
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int co0 = cot * 256, ci0 = cit * 256;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.dout, 0, (int)a.dout_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);

    // ---- LDS-DMA slots: instruction i (0..1) of wave w = piece 2 w + i = image rows 2 piece, 2 piece + 1 (512 B each);
    //      lane -> row 4 w + 2 i + (lane >> 5), 16-B slot lane & 31; source chunk = slot ^ ((row & 7) << 1).
    //      Offsets of a step are PREPARED (pixel -> (b, oh, ow) by reciprocal multiplication, no loops / branches) one
    //      step ahead, so that behind the barrier only the eight buffer loads remain to be issued. ----
    const int hrow = lane >> 5, slot = lane & 31;
    const int HW = a.Ho * a.Wo;
    const float invHW = 1.0f / (float)HW, invW = 1.0f / (float)a.Wo;
    auto divq = [](int m, int d, float inv) -> int {               // floor(m / d) for 0 <= m < 2^22
        int q = (int)((float)m * inv);
        q -= (q * d > m) ? 1 : 0;
        q += ((q + 1) * d <= m) ? 1 : 0;
        return q;
    };
    const int smem_lds = (int)(uintptr_t)(__attribute__((address_space(3))) char*)smem;      // LDS byte address of the dynamic segment
    // Pixel -> byte offset of the input pixel under THIS block's tap (or out of range), built once: the per-step address work
    // (two float reciprocal divisions, bounds tests and multiplies per slot, ~100 VALU operations per wave per 32 MFMAs) sat in
    // the issue slots of the MFMA phase.  With the table a slot costs one ds_read_b32, issued in front of the fragment reads and
    // covered by their wait.
    auto tap_offset = [&](int m) -> unsigned {
        const int b = divq(m, HW, invHW), r = m - b * HW;
        const int oh = divq(r, a.Wo, invW), ow = r - oh * a.Wo;
        const int ih = oh + kh - a.pad, iw = ow + kw - a.pad;
        const bool ok = (unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi;
        return ok ? (unsigned)((b * a.Hi + ih) * a.Wi + iw) * (unsigned)(a.ldi * 2) : kOOBg;
    };
    if (a.use_tab) {
        unsigned* tab = (unsigned*)(smem + LDS_WGW);
        for (int m = tid; m < a.M; m += 512) tab[m] = tap_offset(m);
        __syncthreads();
    }
    unsigned vA[2], vB[2], vAn[2], tb[2];          // vAn: dout offsets of the step being prepared (vA / vB are live until fire())
    int mnext = 4 * w + hrow;                                       // pixel of slot 0 in the step being prepared
    int mcur[2];
    auto prepare_issue = [&]() {                                    // vA; table reads for vB (asm: see the K loop)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mnext + 2 * i;
            mcur[i] = m;
            const int chunk = slot ^ (((4 * (w & 1) + 2 * i + hrow) & 7) << 1);     // row & 7
            const int co = co0 + chunk * 8;
            vAn[i] = (m < a.M && co < a.Co && co + 8 <= a.ldd) ? (unsigned)m * (unsigned)(a.ldd * 2) + (unsigned)(co * 2) : kOOBg;
            if (a.use_tab) {
                const int addr = smem_lds + LDS_WGW + 4 * min(m, a.M - 1);
                asm volatile("ds_read_b32 %0, %1" : "=v"(tb[i]) : "v"(addr));
            }
        }
        mnext += KPg;
    };
    auto prepare_finish = [&]() {                                   // (behind a wait that covers the table reads)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mcur[i];
            const int chunk = slot ^ (((4 * (w & 1) + 2 * i + hrow) & 7) << 1);
            const int ci = ci0 + chunk * 8;
            const unsigned t = a.use_tab ? tb[i] : tap_offset(min(m, a.M - 1));
            vB[i] = (m < a.M && ci < a.Ci && t != kOOBg) ? t + (unsigned)(ci * 2) : kOOBg;
            vA[i] = vAn[i];
        }
    };
    auto prepare = [&]() {
        prepare_issue();
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb[0]), "+v"(tb[1]));
        prepare_finish();
    };
    auto fire = [&](int stage) {
        char* sb = smem + stage * STAGEg;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(sb + (2 * w + i) * 1024), 16, vA[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)(sb + KPg * 512 + (2 * w + i) * 1024), 16, vB[i], 0, 0, 0);
    };

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // per-lane transpose-read offsets: this lane supplies row kk (of a 16-row block) and 8 B = 4 channels
    const int kk = g * 4 + (r16 >> 2);
    const int sub = (r16 & 3) * 8;
    const int sw = (kk & 7) << 1;                    // same for kk + 16
    int offA[8], offB[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) offA[i] = kk * 512 + (((wm * 16 + i * 2) ^ sw) << 4) + sub;
#pragma unroll
    for (int j = 0; j < 4; ++j) offB[j] = KPg * 512 + kk * 512 + (((wn * 8 + j * 2) ^ sw) << 4) + sub;

    const int nK = (a.M + KPg - 1) / KPg;
    if (ADAM && a.stagger > 0 && (int)blockIdx.x < a.ncu) {
        // With the update in the epilogue a tile is a K loop (MFMA-bound, HBM idle) followed by 1.7 MB of master / moment traffic
        // (HBM-bound, MFMA idle).  All CUs start together and every tile takes the same time, so the whole chip alternated between
        // the two phases in lockstep and the launch took the SUM of both (940 us for fc6 at B = 8 against 476 + 523 separately).  The
        // first block of every CU therefore starts 0 .. 7 eighths of a K loop late (blocks are dealt to the XCDs round-robin, so
        // blockIdx >> 3 walks the CUs of an XCD): from then on an eighth of the CUs is in each phase of the cycle at any time.
        const int phase = (blockIdx.x >> 3) & 7;
        for (int i = 0; i < phase * a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
#ifdef SZN_ABLATE_BUILD
    long long pt[12];                                 // SZN_WGW_ABLATE=9: clock64 at the phase boundaries of the tile (tools/probe_wgw_cycles.py)
#pragma unroll
    for (int i = 0; i < 12; ++i) pt[i] = 0;
    if (a.ablate == 9) pt[0] = clock64();
#endif
    prepare(); fire(0);
    prepare(); if (nK > 1) fire(1);
    prepare(); if (nK > 2) fire(2);
    prepare();                                        // offsets of step 3
    // Phase shift: the two waves of a SIMD (w, w + 4) used to run the same phase at the same time (24 transpose reads -- 96 KiB per
    // CU per step through the LDS port -- then 32 MFMAs).  Waves 4 .. 7 (group B) take the per-step barrier between their reads and
    // their MFMAs, waves 0 .. 3 (group A) in front of their reads as before:
    //     A:  W | fire reads(k) MFMA(k)  W | fire reads(k+1) MFMA(k+1) ...      B:  W reads(k) | MFMA(k) fire  W reads(k+1) | MFMA(k+1) fire ...
    // so between two barriers every SIMD has one wave reading and one multiplying.  Rules of the ring under this order:
    //   * W at the top of step k waits for the wave's own pieces of stage k + 1 (B reads stage k + 1 behind barrier k), so two
    //     stages stay in flight instead of three; a prologue barrier covers stage 0;
    //   * a wave fires stage k + 3 (the buffer of step k - 1) only behind barrier k, when everybody has finished reading step k - 1.
    const bool grpB = a.shift && w >= 4;
    // stage 0 has landed for every wave before group B reads it (the prologue fired min(nK, 3) stages of four loads each)
    if (nK > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nK > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int stage = 0;
    for (int kc = 0; kc < nK; ++kc) {
        // four LDS-DMA instructions per wave per stage: only stage kc + 2 may stay in flight here
        if (kc + 2 < nK) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!grpB) __builtin_amdgcn_s_barrier();
        if (!grpB && kc + 3 < nK && a.ablate != 1) fire((stage + 3) & 3);       // the stage drained in step kc - 1
        prepare_issue();                              // offsets of step kc + 4 (table reads go out in front of the fragment reads)
        // Transpose reads as inline asm with explicit waits: behind the `buffer_load ... lds` of fire() the compiler puts
        // s_waitcnt vmcnt(0) in front of the next LDS read it knows about (the DMA might alias it) -- every K step then waited for
        // the stage it had JUST issued, i.e. the ring never had anything in flight (0.32 of peak).  The fill goes to another stage;
        // the counted vmcnt above is the only wait on it.
        const int sbo = smem_lds + stage * STAGEg;
        if (a.ablate != 2) {
            auto rd_tr = [&](int addr, int off) -> u32x2_t {
                u32x2_t v;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
                return v;
            };
            u32x2_t dl[8], dh[8], xl[4], xh[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { xl[j] = rd_tr(sbo + offB[j], 0); xh[j] = rd_tr(sbo + offB[j], 16 * 512); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { dl[i] = rd_tr(sbo + offA[i], 0); dh[i] = rd_tr(sbo + offA[i], 16 * 512); }
#pragma unroll
            for (int i = 4; i < 8; ++i) { dl[i] = rd_tr(sbo + offA[i], 0); dh[i] = rd_tr(sbo + offA[i], 16 * 512); }
            // first half: the pixel fragments and dout fragments 0 .. 3 have landed (8 reads may still be in flight)
            asm volatile("s_waitcnt lgkmcnt(8)"
                         : "+v"(tb[0]), "+v"(tb[1]), "+v"(xl[0]), "+v"(xh[0]), "+v"(xl[1]), "+v"(xh[1]), "+v"(xl[2]), "+v"(xh[2]), "+v"(xl[3]), "+v"(xh[3]),
                           "+v"(dl[0]), "+v"(dh[0]), "+v"(dl[1]), "+v"(dh[1]), "+v"(dl[2]), "+v"(dh[2]), "+v"(dl[3]), "+v"(dh[3]));
            if (grpB) __builtin_amdgcn_s_barrier();
            u32x4_t xf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[j] = u32x4_t{xl[j].x, xl[j].y, xh[j].x, xh[j].y};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4_t df = u32x4_t{dl[i].x, dl[i].y, dh[i].x, dh[i].y};
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<T>(df, xf[j], acc[i][j]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(dl[4]), "+v"(dh[4]), "+v"(dl[5]), "+v"(dh[5]), "+v"(dl[6]), "+v"(dh[6]), "+v"(dl[7]), "+v"(dh[7]),
                           "+v"(acc[3][3]));            // (behind the 16 MFMAs of the first half)
#pragma unroll
            for (int i = 4; i < 8; ++i) {
                const u32x4_t df = u32x4_t{dl[i].x, dl[i].y, dh[i].x, dh[i].y};
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<T>(df, xf[j], acc[i][j]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb[0]), "+v"(tb[1]) : : "memory");
        if (grpB && a.ablate == 2) __builtin_amdgcn_s_barrier();
        if (grpB && kc + 3 < nK && a.ablate != 1) fire((stage + 3) & 3);        // (behind B's barrier)
        prepare_finish();
        stage = (stage + 1) & 3;
    }

#ifdef SZN_ABLATE_BUILD
    if (a.ablate == 9) pt[1] = clock64();
#endif
    // ---- epilogue: D[co][ci] (lane: rows co = 4 g + e, column ci = r16) staged through LDS in four 64-row passes so that
    //      every store instruction covers whole 1-KiB rows of the OHWI gradient ----
    constexpr int PT = 256 + 4;
    const unsigned nwb = (unsigned)a.Co * (unsigned)(a.KH * a.KW * a.Ci) * 4u;           // bytes of one f32 array of this layer
    const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void*)a.p, 0, ADAM ? (int)nwb : 0, 0x00020000);
    const auto rsM = __builtin_amdgcn_make_buffer_rsrc((void*)a.m1, 0, ADAM ? (int)nwb : 0, 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)a.m2, 0, ADAM ? (int)nwb : 0, 0x00020000);
    const auto rsL = __builtin_amdgcn_make_buffer_rsrc((void*)a.wlp, 0, ADAM ? (int)(nwb >> 1) : 0, 0x00020000);
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc((void*)a.dw, 0, ADAM ? (int)nwb : 0, 0x00020000);
    float* tile = (float*)smem;                       // 64 x 260 x 4 B = 66,560 B
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        // ADAM: the 16-B groups of master / moments this thread updates in this pass (8 of the pass's 4096) are requested before the
        // tile is staged, so that their latency runs under the LDS round trip; the requests of a block overlap with the K loops of
        // the other CUs (the launch as a whole becomes HBM-bound: 26 B per weight instead of 4)
        u32x4_t pq[8], mq[8], vq[8];
        int eo[8];                                    // element offset in the OHWI gradient (< 2^29: checked by the launcher), -1 = outside
        if (ADAM) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int idx = tid + it * 512;
                const int r = idx >> 6, c4 = (idx & 63) * 4;
                const int co = co0 + pass * 64 + r, ci = ci0 + c4;
                eo[it] = (co < a.Co && ci < a.Ci) ? ((co * a.KH + kh) * a.KW + kw) * a.Ci + ci : -1;       // Ci % 8 == 0: whole groups
                // (buffer loads with a 32-bit byte offset: no 64-bit address pair per group has to stay live until the stores;
                //  an outside group reads offset 0xFFFFFFF0 = out of range = zeros, and is not stored)
                const unsigned off = eo[it] >= 0 ? (unsigned)eo[it] * 4u : 0xFFFFFFF0u;
                pq[it] = __builtin_amdgcn_raw_buffer_load_b128(rsP, off, 0, SZN_WGW_ADAM_AUX);
                mq[it] = __builtin_amdgcn_raw_buffer_load_b128(rsM, off, 0, SZN_WGW_ADAM_AUX);
                vq[it] = __builtin_amdgcn_raw_buffer_load_b128(rsV, off, 0, SZN_WGW_ADAM_AUX);
            }
        }
#ifdef SZN_ABLATE_BUILD
        if (a.ablate == 9 && pass == 1) pt[2] = clock64();            // pass 1: loads issued
#endif
        __syncthreads();
#ifdef SZN_ABLATE_BUILD
        if (a.ablate == 9 && pass == 1) pt[3] = clock64();            // ... first barrier passed
#endif
        if (wm == (pass >> 1)) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        tile[(ii * 16 + g * 4 + e) * PT + wn * 64 + j * 16 + r16] = acc[(pass & 1) * 4 + ii][j][e];
        }
        __syncthreads();
#ifdef SZN_ABLATE_BUILD
        if (a.ablate == 9 && pass == 1) pt[4] = clock64();            // ... gradient tile staged
#endif
        if (ADAM) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
#ifdef SZN_ABLATE_BUILD
                if (a.ablate == 9 && pass == 1 && it == 1) pt[5] = clock64();     // ... first group updated and stored (its loads had landed)
                if (a.ablate == 9 && pass == 1 && it == 7) pt[6] = clock64();
#endif
                if (eo[it] < 0) continue;
                const int idx = tid + it * 512;
                const int r = idx >> 6, c4 = (idx & 63) * 4;
                const f32x4_t gq = *(const f32x4_t*)(tile + r * PT + c4);
                u32x4_t po, mo, vo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = __uint_as_float(pq[it][e]), me = __uint_as_float(mq[it][e]), ve = __uint_as_float(vq[it][e]);
                    adam_elem(pe, gq[e], me, ve, a.b1, a.b2, a.eps, a.wd, a.step_size, a.inv_bc2_sqrt, a.gscale);
                    po[e] = __float_as_uint(pe); mo[e] = __float_as_uint(me); vo[e] = __float_as_uint(ve);
                }
                const unsigned off = (unsigned)eo[it] * 4u;
                __builtin_amdgcn_raw_buffer_store_b128(mo, rsM, off, 0, SZN_WGW_ADAM_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(vo, rsV, off, 0, SZN_WGW_ADAM_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(po, rsP, off, 0, SZN_WGW_ADAM_AUX);
                if (a.wlp) {
                    u32x2_t pk;
                    pk.x = pack2<T>(__uint_as_float(po[0]), __uint_as_float(po[1]));
                    pk.y = pack2<T>(__uint_as_float(po[2]), __uint_as_float(po[3]));
                    __builtin_amdgcn_raw_buffer_store_b64(pk, rsL, off >> 1, 0, 0);
                }
                if (a.dw) {
                    const u32x4_t go = u32x4_t{__float_as_uint(gq[0]), __float_as_uint(gq[1]), __float_as_uint(gq[2]), __float_as_uint(gq[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(go, rsG, off, 0, 0);
                }
            }
#ifdef SZN_ABLATE_BUILD
            if (a.ablate == 9 && pass == 1) pt[7] = clock64();        // pass 1 done (stores issued)
            if (a.ablate == 9 && pass == 3 && tid == 0 && a.dw_lp) {  // the block's split: cycles in {prologue + K loop, pass 0, pass 1 parts, passes 2-3}
                pt[8] = clock64();
                float* dbg = (float*)a.dw_lp + (size_t)blockIdx.x * 16;   // (probe runs hand a scratch buffer over in szn_conv_desc_t.dw_lp)
                dbg[0] = (float)(pt[1] - pt[0]); dbg[1] = (float)(pt[2] - pt[1]); dbg[2] = (float)(pt[3] - pt[2]); dbg[3] = (float)(pt[4] - pt[3]);
                dbg[4] = (float)(pt[5] - pt[4]); dbg[5] = (float)(pt[6] - pt[5]); dbg[6] = (float)(pt[7] - pt[6]); dbg[7] = (float)(pt[8] - pt[7]);
                dbg[8] = (float)(pt[8] - pt[0]);
            }
#endif
            continue;
        }
        for (int idx = tid; idx < 64 * 64; idx += 512) {              // 64 rows x 64 float4
            const int r = idx >> 6, c4 = (idx & 63) * 4;
            const int co = co0 + pass * 64 + r, ci = ci0 + c4;
            if (co < a.Co && ci < a.Ci) {
                float* dst = a.dw + ((long)(co * a.KH + kh) * a.KW + kw) * a.Ci + ci;
                f32x4_t v = *(const f32x4_t*)(tile + r * PT + c4);
                if (a.dw_lp) {                                            // Ci % 8 == 0: whole groups; 8-B aligned image
                    uint2 pk;
                    if (a.lp_f16) { pk.x = pack2<f16_raw>(v[0], v[1]); pk.y = pack2<f16_raw>(v[2], v[3]); }
                    else { pk.x = pack2<bf16_raw>(v[0], v[1]); pk.y = pack2<bf16_raw>(v[2], v[3]); }
                    *(uint2*)(a.dw_lp + ((long)(co * a.KH + kh) * a.KW + kw) * a.Ci + ci) = pk;
                    continue;
                }
                if (ci + 4 <= a.Ci && ((((uintptr_t)dst) & 15) == 0)) {
                    if (a.accumulate) { const f32x4_t o = *(const f32x4_t*)dst; v += o; }
                    *(f32x4_t*)dst = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ci + e < a.Ci) dst[e] = a.accumulate ? dst[e] + v[e] : v[e];
                }
            }
        }
    }
#endif
}


// ---- conv_wgrad_half<T>: the Adam form on 128 x 256 tiles, TWO blocks per CU (round 5) ---------------------------------------------
// conv_wgrad_wide<T, true> runs one block per CU (160 KiB of LDS, 2 waves per SIMD x ~207 VGPRs): a CU alternates between a K loop
// (matrix pipes busy, memory idle) and 1.7 MB of master / moment / image traffic per tile (memory busy, matrix pipes idle), and the
// launch takes nearly the sum of the two (fc6 at B = 8: 447 us of K loops + ~350 us of update traffic = 792 us; the update alone
// streams at 4.4 TB/s at B = 1).  Here a block is FOUR waves (one per SIMD) on a 128 cout x 256 cin tile -- each wave still owns
// 128 x 64 outputs (8 x 4 accumulator fragments, the same per-wave body and the same summation order: bit-identical gradients) --
// with a 3-stage ring of 24-KiB stages (72 KiB), so two blocks share a CU: while one streams its update the other multiplies.
//   * A = dout rows [32 px][128 co] (256 B per pixel), B = shifted input rows [32 px][256 ci] (512 B); 16-B chunk index XOR
//     (row & 7) << 1 on the DMA source side for both; six 1-KiB LDS-DMA pieces per wave and stage (2 A + 4 B);
//   * four waves in lockstep: per K step { vmcnt: stage k landed; barrier; fire stage k + 2 into the buffer of step k - 1;
//     24 transpose reads (inline asm); 32 MFMA };
//   * pixel -> input-pixel-index table of the block's tap as uint16 behind the ring (M <= 3840, B Hi Wi < 65535), else computed;
//   * epilogue: two 64-row passes, each in four batches of 4 x 16-B groups per thread with the next batch's master / moment loads
//     in flight while the current one is updated (adam_elem, non-temporal policy as in <T, true>);
//   * the blocks that fill the SECOND slot of every CU (blockIdx 256 .. 511) start half a K loop late, so that a CU's two
//     blocks alternate between the two phases from the first tile on.
constexpr int STAGEh = KPg * (256 + 512);            // 24 KiB
constexpr int NSTh = 3;
constexpr int LDS_WGH = NSTh * STAGEh;               // 72 KiB
constexpr int kTabMaxH = 3840;                       // uint16 entries behind the ring: 72 KiB + 7.5 KiB < 80 KiB

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_wgrad_half(WgwArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave -> cins 64 w .. + 63 (all 128 couts)
    const int g = lane >> 4, r16 = lane & 15;

    int bid = blockIdx.x;
    int cit, cot, tap;
    if (a.xcd_order == 2) {
        const int nwg = (int)gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        const int ntap = a.KH * a.KW;
        tap = lid % ntap; lid /= ntap;
        cot = lid % a.cotiles; cit = lid / a.cotiles;
    } else {
        cit = bid % a.citiles; bid /= a.citiles;
        cot = bid % a.cotiles;
        tap = bid / a.cotiles;
    }
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int co0 = cot * 128, ci0 = cit * 256;

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.dout, 0, (int)a.dout_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);

    const int HW = a.Ho * a.Wo;
    const float invHW = 1.0f / (float)HW, invW = 1.0f / (float)a.Wo;
    auto divq = [](int m, int d, float inv) -> int {               // floor(m / d) for 0 <= m < 2^22
        int q = (int)((float)m * inv);
        q -= (q * d > m) ? 1 : 0;
        q += ((q + 1) * d <= m) ? 1 : 0;
        return q;
    };
    const int smem_lds = (int)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto tap_pixel = [&](int m) -> unsigned {                      // input pixel index under this block's tap, 0xFFFF = outside
        const int b = divq(m, HW, invHW), r = m - b * HW;
        const int oh = divq(r, a.Wo, invW), ow = r - oh * a.Wo;
        const int ih = oh + kh - a.pad, iw = ow + kw - a.pad;
        const bool ok = (unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi;
        return ok ? (unsigned)((b * a.Hi + ih) * a.Wi + iw) : 0xFFFFu;
    };
    if (a.use_tab) {
        uint16_t* tab = (uint16_t*)(smem + LDS_WGH);
        for (int m = tid; m < a.M; m += 256) tab[m] = (uint16_t)tap_pixel(m);
        __syncthreads();
    }
    // DMA slots.  A piece i (0..1) of wave w = stage rows 8 w + 4 i + (lane >> 4), 16 slots of 16 B; B piece i (0..3) = rows
    // 8 w + 2 i + (lane >> 5), 32 slots.  (row & 7) = 4 i + (lane >> 4) resp. 2 i + (lane >> 5).
    unsigned vA[2], vB[4], vAn[2], tb[4];
    int mA[2], mB[4];
    int kbase = 0;                                                  // first pixel of the step being prepared
    const unsigned pxb = (unsigned)(a.ldi * 2);
    auto prepare_issue = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rr = 4 * i + (lane >> 4);
            const int m = kbase + 8 * w + rr;
            mA[i] = m;
            const int chunk = (lane & 15) ^ ((rr & 7) << 1);
            const int co = co0 + chunk * 8;
            vAn[i] = (m < a.M && co < a.Co && co + 8 <= a.ldd) ? (unsigned)m * (unsigned)(a.ldd * 2) + (unsigned)(co * 2) : kOOBg;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = kbase + 8 * w + 2 * i + (lane >> 5);
            mB[i] = m;
            if (a.use_tab) {
                const int addr = smem_lds + LDS_WGH + 2 * min(m, a.M - 1);
                asm volatile("ds_read_u16 %0, %1" : "=v"(tb[i]) : "v"(addr));
            }
        }
        kbase += KPg;
    };
    auto prepare_finish = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = 2 * i + (lane >> 5);
            const int m = mB[i];
            const int chunk = (lane & 31) ^ ((rr & 7) << 1);
            const int ci = ci0 + chunk * 8;
            const unsigned t = a.use_tab ? tb[i] : tap_pixel(min(m, a.M - 1));
            vB[i] = (m < a.M && ci < a.Ci && t != 0xFFFFu) ? t * pxb + (unsigned)(ci * 2) : kOOBg;
        }
        vA[0] = vAn[0]; vA[1] = vAn[1];
    };
    auto prepare = [&]() {
        prepare_issue();
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tb[0]), "+v"(tb[1]), "+v"(tb[2]), "+v"(tb[3]));
        prepare_finish();
    };
    auto fire = [&](int stage) {
        char* sb = smem + stage * STAGEh;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(sb + (2 * w + i) * 1024), 16, vA[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)(sb + KPg * 256 + (4 * w + i) * 1024), 16, vB[i], 0, 0, 0);
    };

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // per-lane transpose-read offsets: this lane supplies row kk (of a 16-row block) and 8 B = 4 channels
    const int kk = g * 4 + (r16 >> 2);
    const int sub = (r16 & 3) * 8;
    const int sw = (kk & 7) << 1;                    // same for kk + 16
    int offA[8], offB[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) offA[i] = kk * 256 + (((i * 2) ^ sw) << 4) + sub;
#pragma unroll
    for (int j = 0; j < 4; ++j) offB[j] = KPg * 256 + kk * 512 + (((w * 8 + j * 2) ^ sw) << 4) + sub;

    const int nK = (a.M + KPg - 1) / KPg;
    if (a.stagger > 0 && (int)blockIdx.x >= a.ncu && (int)blockIdx.x < 2 * a.ncu)
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    prepare(); fire(0);
    prepare(); if (nK > 1) fire(1);
    prepare();                                        // offsets of step 2
    int stage = 0;
    for (int kc = 0; kc < nK; ++kc) {
        // six LDS-DMA instructions per wave per stage: only stage kc + 1 may stay in flight here
        if (kc + 1 < nK) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kc + 2 < nK) fire(stage == 0 ? 2 : stage - 1);          // (stage + 2) % 3: the buffer of step kc - 1
        prepare_issue();                              // offsets of step kc + 3
        const int sbo = smem_lds + stage * STAGEh;
        auto rd_tr = [&](int addr, int off) -> u32x2_t {
            u32x2_t v;
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off));
            return v;
        };
        u32x2_t dl[8], dh[8], xl[4], xh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { xl[j] = rd_tr(sbo + offB[j], 0); xh[j] = rd_tr(sbo + offB[j], 16 * 512); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { dl[i] = rd_tr(sbo + offA[i], 0); dh[i] = rd_tr(sbo + offA[i], 16 * 256); }
#pragma unroll
        for (int i = 4; i < 8; ++i) { dl[i] = rd_tr(sbo + offA[i], 0); dh[i] = rd_tr(sbo + offA[i], 16 * 256); }
        asm volatile("s_waitcnt lgkmcnt(8)"
                     : "+v"(tb[0]), "+v"(tb[1]), "+v"(tb[2]), "+v"(tb[3]), "+v"(xl[0]), "+v"(xh[0]), "+v"(xl[1]), "+v"(xh[1]), "+v"(xl[2]),
                       "+v"(xh[2]), "+v"(xl[3]), "+v"(xh[3]), "+v"(dl[0]), "+v"(dh[0]), "+v"(dl[1]), "+v"(dh[1]), "+v"(dl[2]), "+v"(dh[2]),
                       "+v"(dl[3]), "+v"(dh[3]));
        u32x4_t xf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[j] = u32x4_t{xl[j].x, xl[j].y, xh[j].x, xh[j].y};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4_t df = u32x4_t{dl[i].x, dl[i].y, dh[i].x, dh[i].y};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<T>(df, xf[j], acc[i][j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(dl[4]), "+v"(dh[4]), "+v"(dl[5]), "+v"(dh[5]), "+v"(dl[6]), "+v"(dh[6]), "+v"(dl[7]), "+v"(dh[7]),
                       "+v"(acc[3][3]));
#pragma unroll
        for (int i = 4; i < 8; ++i) {
            const u32x4_t df = u32x4_t{dl[i].x, dl[i].y, dh[i].x, dh[i].y};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<T>(df, xf[j], acc[i][j]);
        }
        prepare_finish();
        stage = stage == 2 ? 0 : stage + 1;
    }

    // ---- epilogue: two passes of 64 rows x 256 columns; thread -> 16 groups of 4 columns per pass, in four batches of four ----
    constexpr int PT = 256 + 4;
    const unsigned nwb = (unsigned)a.Co * (unsigned)(a.KH * a.KW * a.Ci) * 4u;
    const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void*)a.p, 0, (int)nwb, 0x00020000);
    const auto rsM = __builtin_amdgcn_make_buffer_rsrc((void*)a.m1, 0, (int)nwb, 0x00020000);
    const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)a.m2, 0, (int)nwb, 0x00020000);
    const auto rsL = __builtin_amdgcn_make_buffer_rsrc((void*)a.wlp, 0, (int)(nwb >> 1), 0x00020000);
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc((void*)a.dw, 0, (int)nwb, 0x00020000);
    float* tile = (float*)smem;                       // 64 x 260 x 4 B = 66,560 B (the ring is drained: vmcnt(0) + the barriers below)
    u32x4_t pq[2][4], mq[2][4], vq[2][4];
    int eo[2][4];
    auto load_batch = [&](int pass, int bt, int buf) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + (bt * 4 + it) * 256;
            const int r = idx >> 6, c4 = (idx & 63) * 4;
            const int co = co0 + pass * 64 + r, ci = ci0 + c4;
            eo[buf][it] = (co < a.Co && ci < a.Ci) ? ((co * a.KH + kh) * a.KW + kw) * a.Ci + ci : -1;
            const unsigned off = eo[buf][it] >= 0 ? (unsigned)eo[buf][it] * 4u : 0xFFFFFFF0u;
            pq[buf][it] = __builtin_amdgcn_raw_buffer_load_b128(rsP, off, 0, SZN_WGW_ADAM_AUX);
            mq[buf][it] = __builtin_amdgcn_raw_buffer_load_b128(rsM, off, 0, SZN_WGW_ADAM_AUX);
            vq[buf][it] = __builtin_amdgcn_raw_buffer_load_b128(rsV, off, 0, SZN_WGW_ADAM_AUX);
        }
    };
    auto update_batch = [&](int bt, int buf) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if (eo[buf][it] < 0) continue;
            const int idx = tid + (bt * 4 + it) * 256;
            const int r = idx >> 6, c4 = (idx & 63) * 4;
            const f32x4_t gq = *(const f32x4_t*)(tile + r * PT + c4);
            u32x4_t po, mo, vo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = __uint_as_float(pq[buf][it][e]), me = __uint_as_float(mq[buf][it][e]), ve = __uint_as_float(vq[buf][it][e]);
                adam_elem(pe, gq[e], me, ve, a.b1, a.b2, a.eps, a.wd, a.step_size, a.inv_bc2_sqrt, a.gscale);
                po[e] = __float_as_uint(pe); mo[e] = __float_as_uint(me); vo[e] = __float_as_uint(ve);
            }
            const unsigned off = (unsigned)eo[buf][it] * 4u;
            __builtin_amdgcn_raw_buffer_store_b128(mo, rsM, off, 0, SZN_WGW_ADAM_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(vo, rsV, off, 0, SZN_WGW_ADAM_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(po, rsP, off, 0, SZN_WGW_ADAM_AUX);
            if (a.wlp) {
                u32x2_t pk;
                pk.x = pack2<T>(__uint_as_float(po[0]), __uint_as_float(po[1]));
                pk.y = pack2<T>(__uint_as_float(po[2]), __uint_as_float(po[3]));
                __builtin_amdgcn_raw_buffer_store_b64(pk, rsL, off >> 1, 0, 0);
            }
            if (a.dw) {
                const u32x4_t go = u32x4_t{__float_as_uint(gq[0]), __float_as_uint(gq[1]), __float_as_uint(gq[2]), __float_as_uint(gq[3])};
                __builtin_amdgcn_raw_buffer_store_b128(go, rsG, off, 0, 0);
            }
        }
    };
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        load_batch(pass, 0, 0);                       // in flight across the LDS round trip of the gradient tile
        __syncthreads();                              // ring / previous pass fully consumed
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    tile[(ii * 16 + g * 4 + e) * PT + w * 64 + j * 16 + r16] = acc[pass * 4 + ii][j][e];
        __syncthreads();
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) {
            if (bt + 1 < 4) load_batch(pass, bt + 1, (bt + 1) & 1);
            update_batch(bt, bt & 1);
        }
    }
#endif
}

}  // namespace

// Called by szn_conv2d_wgrad after validation.  Returns 1 if the layer does not fit this kernel.
int szn_conv_wgrad_wide_try(const szn_conv_desc_t* d, const void* in, const void* dout, float* dw, int accumulate,
                            int min_tiles, szn_stream_t stream, const szn_adam_args_t* opt) {
    if (!szn_is16(d->dtype) || d->Co < 256 || d->Ci < 256 || (d->ldi & 7) || (d->ldo & 7) || (d->Ci & 7)) return 1;
    WgwArgs a;
    a.p = a.m1 = a.m2 = nullptr; a.wlp = nullptr;
    a.b1 = a.b2 = a.eps = a.wd = a.step_size = a.inv_bc2_sqrt = a.gscale = 0.f;
    if (opt) {
        if ((long)d->Co * d->KH * d->KW * d->Ci >= (1L << 29)) return 1;      // 32-bit byte offsets into the f32 arrays (< 2 GiB each)
        a.p = opt->param; a.m1 = opt->exp_avg; a.m2 = opt->exp_avg_sq; a.wlp = (uint16_t*)opt->w_lp;
        a.b1 = opt->beta1; a.b2 = opt->beta2; a.eps = opt->eps; a.wd = opt->weight_decay; a.gscale = opt->grad_scale;
        szn_adam_scalars(opt->lr, opt->beta1, opt->beta2, opt->step, &a.step_size, &a.inv_bc2_sqrt);
    }
    a.cotiles = szn_div_up(d->Co, 256); a.citiles = szn_div_up(d->Ci, 256);
    const long tiles = (long)a.cotiles * a.citiles * d->KH * d->KW;
    if (tiles < min_tiles || tiles >= (1L << 31) || (long)d->B * d->Ho * d->Wo >= (1L << 22)) return 1;
    // padding waste of the last tiles must stay small (or the launch is a single round of the chip anyway: the native dgrad at B = 1)
    if ((long)a.cotiles * 256 * a.citiles * 256 > (long)d->Co * d->Ci * 5 / 4 && !(min_tiles <= 1 && tiles <= 256)) return 1;
    a.dout = (const char*)dout; a.in = (const char*)in; a.dw = dw;
    a.dw_lp = (accumulate || opt) ? nullptr : (uint16_t*)d->dw_lp; a.lp_f16 = d->dw_lp_dtype == SZN_F16;
    a.dout_bytes = (unsigned)((size_t)d->B * d->Ho * d->Wo * d->ldo * 2);
    a.in_bytes = (unsigned)((size_t)d->B * d->Hi * d->Wi * d->ldi * 2);
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co;
    a.KH = d->KH; a.KW = d->KW; a.pad = d->pad; a.ldi = d->ldi; a.ldd = d->ldo;
    a.M = d->B * d->Ho * d->Wo;
    a.accumulate = accumulate;
    { static int abl = -1; if (abl < 0) { abl = szn_ablate_env("SZN_WGW_ABLATE"); } a.ablate = abl; }
    if (a.ablate == 9 && opt) a.dw_lp = (uint16_t*)d->dw_lp;      // (ablation build: the cycle probe's output buffer, >= 64 B per tile)
    { const int tab = 1; /* (was SZN_WGW_TAB) */ a.use_tab = (tab && a.M <= kTabMax) ? 1 : 0; }
    { const int sh = 1; /* (was SZN_WGW_SHIFT) */ a.shift = sh; }
    a.stagger = 0;
    {
        static int ncu = 0;
        if (!ncu) {
            int dev = 0; hipDeviceProp_t p;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ncu = p.multiProcessorCount;
            if (ncu <= 0) ncu = 256;
        }
        a.ncu = ncu;
    }
    if (opt) {
        // one K step (32 pixels) takes ~1 us per CU; the full stagger spans ~7/8 of a K loop: phase x nK / 32 sleeps of ~4 us
        static const int sg = szn_knob("SZN_WGW_STAGGER", -1);
        const int nK = (a.M + KPg - 1) / KPg;
        a.stagger = tiles < 3L * a.ncu ? 0 : (sg >= 0 ? sg : (nK + 8) / 16);     // (a one-round launch would only start late)
    }
    { static const int xo = szn_knob("SZN_WGW_XCD", 1);
      a.xcd_order = (xo && (size_t)a.in_bytes > 4 * (size_t)a.dout_bytes) ? 1 : 0; }
    const int lds = LDS_WGW + (a.use_tab ? kTabMax * 4 : 0);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_wide<bf16_raw, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WGW + kTabMax * 4);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_wide<f16_raw, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WGW + kTabMax * 4);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_wide<bf16_raw, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WGW + kTabMax * 4);
        (void)hipFuncSetAttribute((const void*)conv_wgrad_wide<f16_raw, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WGW + kTabMax * 4);
        attr_done = true;
    }
    const dim3 grid((unsigned)tiles);
    hipStream_t st = (hipStream_t)stream;
    if (opt) {
        if (opt->grad_optional) a.dw = nullptr;
        // round 5: 128 x 256 tiles, two four-wave blocks per CU (conv_wgrad_half).  Measured (profiles/r05_ablations.txt 3): the
        // update streams better (fc6 at B = 1, where the K loop is 289 pixels: 550 -> 517 us, the one-image step 2.85 -> 2.82 ms)
        // but with a long K loop the smaller tile's extra operand traffic costs what the overlap gains (B = 8: 842 -> 864 us), so it
        // takes the launches whose K loop is short.  SZN_WGW_HALF = 0 / 1 forces conv_wgrad_wide<T, true> / this kernel
        // (read per call: the tests run both).
        const int eh = szn_knob_live("SZN_WGW_HALF", -1);
        const long cot_h = szn_div_up(d->Co, 128);
        const bool want_half = eh >= 0 ? eh != 0 : a.M <= 1024;
        if (want_half && cot_h * a.citiles * d->KH * d->KW < (1L << 31)) {
            WgwArgs h = a;
            h.cotiles = (int)cot_h;
            h.use_tab = (a.M <= kTabMaxH && (long)d->B * d->Hi * d->Wi < 65535) ? a.use_tab : 0;
            const long tiles_h = cot_h * a.citiles * d->KH * d->KW;
            // the second-slot blocks start half a K loop late: one K step of a lone four-wave block ~0.25 us, s_sleep(127) ~3.4 us
            static const int sgh = szn_knob("SZN_WGH_STAGGER", -1);
            const int nKh = (a.M + KPg - 1) / KPg;
            h.stagger = tiles_h < 4L * a.ncu ? 0 : (sgh >= 0 ? sgh : (nKh + 16) / 32);
            const int ldsh = LDS_WGH + (h.use_tab ? kTabMaxH * 2 : 0);
            static bool attr_h = false;
            if (!attr_h) {
                (void)hipFuncSetAttribute((const void*)conv_wgrad_half<bf16_raw>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WGH + kTabMaxH * 2);
                (void)hipFuncSetAttribute((const void*)conv_wgrad_half<f16_raw>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WGH + kTabMaxH * 2);
                attr_h = true;
            }
            if (d->dtype == SZN_F16) hipLaunchKernelGGL((conv_wgrad_half<f16_raw>), dim3((unsigned)tiles_h), dim3(256), ldsh, st, h);
            else hipLaunchKernelGGL((conv_wgrad_half<bf16_raw>), dim3((unsigned)tiles_h), dim3(256), ldsh, st, h);
            SZN_CHECK_LAUNCH("conv_wgrad_half_adam");
            return SZN_OK;
        }
        if (d->dtype == SZN_F16) hipLaunchKernelGGL((conv_wgrad_wide<f16_raw, true>), grid, dim3(512), lds, st, a);
        else hipLaunchKernelGGL((conv_wgrad_wide<bf16_raw, true>), grid, dim3(512), lds, st, a);
        SZN_CHECK_LAUNCH("conv_wgrad_wide_adam");
        return SZN_OK;
    }
    if (d->dtype == SZN_F16) hipLaunchKernelGGL((conv_wgrad_wide<f16_raw, false>), grid, dim3(512), lds, st, a);
    else hipLaunchKernelGGL((conv_wgrad_wide<bf16_raw, false>), grid, dim3(512), lds, st, a);
    SZN_CHECK_LAUNCH("conv_wgrad_wide");
    return SZN_OK;
}
