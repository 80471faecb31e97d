// gemm_8phase.hip -- the CDNA guide's "256^2 8-phase" bf16 GEMM template (cdna_hip_programming.md section 5, "The 256^2 8-phase
// template"), rebuilt from the guide's description (its examples/gemm_256sq_8phase_bf16.cpp is not shipped in this image), as a
// MICRO-BENCH: the reference point VERDICT r03 asked for ("turn the MFMA ceiling into evidence").  It is not part of libszn_hip.so.
//
//   C[M][N] (bf16) = A[M][K] x B[N][K]^T, bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16).
//
// Geometry as the guide's table: 256 x 256 tile, BK = 64, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 C-quadrants of
// 64 x 32, LDS 128 KiB = 2 buffers x 4 half-tiles of 128 rows x 128 B, operands HBM/L2 -> LDS by `buffer_load ... lds` (16 B per
// lane, two instructions per wave per half-tile), 4 phases per K tile / 8 per loop iteration, each phase
//     { ds_read the phase's register sub-tile ; issue ONE half-tile of prefetch ; counted s_waitcnt vmcnt ; s_barrier ;
//       s_waitcnt lgkmcnt(0) ; s_setprio 1 ; 16 MFMA (one C-quadrant x K = 64) ; s_setprio 0 ; s_barrier }
// with the two wave groups (wr = 0 / 1: waves w and w + 4 share a SIMD) running one barrier apart, so that on every SIMD one
// wave reads / issues loads while its partner multiplies.  vmcnt is never 0 in the main loop.
//
// Where this file had to choose (the guide gives the structure, not every detail):
//   * half-tiles are interleaved so that every wave reads a half-tile in ONE phase: A0 / A1 = the lower / upper 64 rows of each
//     wave row group, B0 / B1 = the lower / upper 32 columns of each wave column group.  Quadrant order (A0,B0) (A0,B1) (A1,B1)
//     (A1,B0): phase reads 12 / 4 / 8 / 0 ds_read_b128 (both B sub-tiles stay in registers).
//   * software pipeline: the half-tile issued in phase p is waited for in phase p + 4 (s_waitcnt vmcnt(8): four half-tiles = 64 KiB
//     per CU in flight) and read from phase p + 5 on; a slot is re-staged >= 2 phases after its last ds_read (the guide's WAR rule
//     for staggered groups).  Staging order per K tile t (buffer b = t & 1): P1 -> B1 of t + 1, P2 -> A1 of t + 1, P3 -> A0 of
//     t + 2, P4 -> B0 of t + 2.
//   * LDS swizzle: 16-B chunk ^ (row & 7) on 128-B rows, applied on the SOURCE address of the LDS-DMA and on the read (conflict-free
//     for ds_read_b128: the 16 lanes of a read group land on 16 different 16-B slots of the 256-B bank row).  VARIANT 1 uses the
//     guide's st_16x32 instead (byte ^= ((byte >> 9) & 1) << 5 on the linear 128-B-row image), VARIANT 2 no swizzle.
//   * epilogue: the register epilogue of libszn_hip.so (szn_epilogue.h): v_permlane16_swap pairs, one 16-B store per piece.
#include "../../zeroshotsemanticsegmentation_amd/csrc/szn_common.h"
#include "../../zeroshotsemanticsegmentation_amd/csrc/szn_epilogue.h"

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((address_space(3))) void* ldsptr_t;

namespace {

struct G8Args {
    const char* in; const char* w; char* out;          // A [M][K], B [N][K], C [M][ldo] (names as the epilogue expects them)
    const float* bias; const char* gate; const float* cscale; float* colsum; float* cslab;
    int M, Co, K, ldo, ldg, relu, out_f32, HoWo, abl_ep;
    unsigned in_bytes, w_bytes;
    int mtiles, ntiles;
    int flags;                                        // bit 0: no s_setprio, bit 1: no stagger (both groups in lockstep)
};

constexpr unsigned kOOB = 0x80000000u;
constexpr int SLOT = 16384;                           // half-tile: 128 rows x 128 B
constexpr int BUF = 4 * SLOT;                         // A0 | A1 | B0 | B1

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int OFF> __device__ __forceinline__ void dsr(u32x4_t& v, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

// swizzled byte offset of 16-B chunk c (0..7) in row r of a 128-B-row image
template <int VARIANT> __device__ __forceinline__ int swz(int r, int c) {
    if constexpr (VARIANT == 0) return ((c ^ (r & 7)) << 4);
    else if constexpr (VARIANT == 1) return ((c ^ (((r >> 2) & 1) << 1)) << 4);      // st_16x32: bit 9 of r * 128 + 16 c  ->  flip bit 5
    else return c << 4;
}

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
// Round 6 experiment (MMA32, TIMING ONLY -- the results are wrong): the same schedule, the same LDS reads and the same registers, but every quadrant's
// 16 v_mfma_f32_16x16x32_bf16 replaced by 8 v_mfma_f32_32x32x16_bf16 on the same operand registers (the guide's micro-benchmark table has the
// 32 x 32 shape at 2,382 TF against 2,075 TF for 16 x 16: is the matrix INSTRUCTION part of what keeps this kernel at 0.53?)
// PERSIST (round 6 experiment): 0 = one tile per block (the template); 1 = a block walks tiles blockIdx.x, + gridDim.x, ... and issues the NEXT
// tile's six prologue half-tiles before the epilogue stores of the finished one (the LDS ring is idle during the register epilogue); 2 = the same
// loop with the prologue behind the epilogue (what persistence alone buys: no block dispatch between tiles)
template <int VARIANT, bool PRIO, bool STAGGER, bool MMA32 = false, int PERSIST = 0>
__global__ __launch_bounds__(512, 2) void gemm_256sq_8phase(G8Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int g = lane >> 4, r16 = lane & 15;
    constexpr bool stagger = STAGGER, prio = PRIO;

    const int nwg = a.mtiles * a.ntiles;
    int m0, n0;
    auto tile_of = [&](int bid) { const int lid = xcd_remap(bid, nwg); m0 = (lid / a.ntiles) * 256; n0 = (lid % a.ntiles) * 256; };
    tile_of(blockIdx.x);

    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.w_bytes, 0x00020000);

    // ---- staging: a half-tile is 16 wave-instructions of 8 rows x 128 B; wave w issues instructions 2 w and 2 w + 1
    // voff[kind][i]: kind 0 = A0, 1 = A1, 2 = B0, 3 = B1
    unsigned voff[4][2];
    auto setup = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (2 * w + i) * 8 + (lane >> 3);           // row of the half-tile image
        const int c = lane & 7;                                  // LDS chunk this lane fills; source chunk = its pre-image
        int sc;
        if constexpr (VARIANT == 0) sc = c ^ (row & 7);
        else if constexpr (VARIANT == 1) sc = c ^ (((row >> 2) & 1) << 1);
        else sc = c;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = m0 + (row >> 6) * 128 + h * 64 + (row & 63);
            voff[h][i] = m < a.M ? (unsigned)(((size_t)m * a.K + sc * 8) * 2) : kOOB;
            const int n = n0 + (row >> 5) * 64 + h * 32 + (row & 31);
            voff[2 + h][i] = n < a.Co ? (unsigned)(((size_t)n * a.K + sc * 8) * 2) : kOOB;
        }
    }
    };
    setup();
    const int nkt = a.K / 64;
    auto stage = [&](int kind, int buf, int kt) {               // kind / buf are compile-time after unrolling
        const unsigned kill = kt < nkt ? 0u : kOOB;               // beyond K: zeros into a slot nobody reads (keeps vmcnt uniform)
        char* dst = smem + buf * BUF + kind * SLOT + (2 * w) * 1024;
        const int soff = kt * 128;
        if (kind < 2) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)dst, 16, voff[kind][0] | kill, soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (ldsptr_t)(dst + 1024), 16, voff[kind][1] | kill, soff, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)dst, 16, voff[kind][0] | kill, soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ldsptr_t)(dst + 1024), 16, voff[kind][1] | kill, soff, 0, 0);
        }
    };

    // ---- fragment read addresses (bytes from the start of smem, buffer 0): row base + swizzled chunk of K half s
    // A half-tile image row of (wave row group wr, fragment j, lane row r16) = wr * 64 + 16 j + r16; B: wc * 32 + 16 i + r16
    unsigned adA[2][2], adB[2][2];                                // [buffer][K half]: ds_read offsets are 16-bit immediates, a buffer is 64 KiB
    {
        const int rowA = wr * 64 + r16, rowB = wc * 32 + r16;    // (+ 16 j / 16 i: multiples of 16 leave row & 7 and (row >> 2) & 1 ... see below)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            adA[0][s] = (unsigned)(rowA * 128 + swz<VARIANT>(rowA, 4 * s + g));
            adB[0][s] = (unsigned)(2 * SLOT + rowB * 128 + swz<VARIANT>(rowB, 4 * s + g));
            adA[1][s] = adA[0][s] + BUF;
            adB[1][s] = adB[0][s] + BUF;
        }
    }
    // (the swizzles depend on row bits 0..2 only, and fragments are 16 rows apart: the fragment offset is a plain immediate)

    f32x4_t acc[4][2][4];                                         // [quadrant][n fragment i][m fragment j]

    f32x16_t acc32[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc32[q][mb][e] = 0.f;
    u32x4_t fa[4][2], fb0[2][2], fb1[2][2];                       // A sub-tile [j][s]; B0 / B1 sub-tiles [i][s]

    // ---- prologue: A0 B0 B1 A1 of tile 0, A0 B0 of tile 1 (what phases -6 .. -1 of the steady state would have issued)
    auto prologue = [&]() { stage(0, 0, 0); stage(2, 0, 0); stage(3, 0, 0); stage(1, 0, 0); stage(0, 1, 1); stage(2, 1, 1); };
    prologue();
    for (int bid = blockIdx.x; bid < nwg; bid += (PERSIST ? (int)gridDim.x : nwg)) {
    const int m0c = m0, n0c = n0;                                 // the tile this iteration finishes (the epilogue's coordinates)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[q][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (PERSIST && bid != (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (behind epilogue stores: every older operation, loads and stores)
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // A0, B0 of tile 0 landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                                 // ... everyone's
    if (stagger && wr == 1) __builtin_amdgcn_s_barrier();         // group 1 runs one barrier behind group 0

#define G8_READ_A(KIND, BUFI)                                                                                        \
    {                                                                                                                \
        constexpr int o_ = (KIND) * SLOT;                                                                            \
        dsr<o_ + 0 * 2048>(fa[0][0], adA[BUFI][0]); dsr<o_ + 1 * 2048>(fa[1][0], adA[BUFI][0]);                      \
        dsr<o_ + 2 * 2048>(fa[2][0], adA[BUFI][0]); dsr<o_ + 3 * 2048>(fa[3][0], adA[BUFI][0]);                      \
        dsr<o_ + 0 * 2048>(fa[0][1], adA[BUFI][1]); dsr<o_ + 1 * 2048>(fa[1][1], adA[BUFI][1]);                      \
        dsr<o_ + 2 * 2048>(fa[2][1], adA[BUFI][1]); dsr<o_ + 3 * 2048>(fa[3][1], adA[BUFI][1]);                      \
    }
#define G8_READ_B(FB, KIND, BUFI)                                                                                    \
    {                                                                                                                \
        constexpr int o_ = ((KIND) - 2) * SLOT;                                                                      \
        dsr<o_ + 0>(FB[0][0], adB[BUFI][0]); dsr<o_ + 2048>(FB[1][0], adB[BUFI][0]);                                 \
        dsr<o_ + 0>(FB[0][1], adB[BUFI][1]); dsr<o_ + 2048>(FB[1][1], adB[BUFI][1]);                                 \
    }
#define G8_SYNC_AND_MMA(Q, FB)                                                                                       \
    {                                                                                                                \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                             \
        __builtin_amdgcn_s_barrier();                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
                     : "+v"(fa[0][0]), "+v"(fa[1][0]), "+v"(fa[2][0]), "+v"(fa[3][0]), "+v"(fa[0][1]), "+v"(fa[1][1]), \
                       "+v"(fa[2][1]), "+v"(fa[3][1]), "+v"(FB[0][0]), "+v"(FB[1][0]), "+v"(FB[0][1]), "+v"(FB[1][1])); \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if (prio) __builtin_amdgcn_s_setprio(1);                                                                     \
        if constexpr (MMA32) {                                                                                       \
            _Pragma("unroll") for (int kq = 0; kq < 4; ++kq)                                                         \
                _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                     \
                    acc32[Q][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                          \
                        __builtin_bit_cast(bf16x8_t, FB[kq & 1][kq >> 1]), __builtin_bit_cast(bf16x8_t, fa[2 * mb + (kq & 1)][kq >> 1]), \
                        acc32[Q][mb], 0, 0, 0);                                                                      \
        } else {                                                                                                     \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                            \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
                    acc[Q][i][j] = mfma16<bf16_raw>(FB[i][s], fa[j][s], acc[Q][i][j]);                               \
        }                                                                                                            \
        if (prio) __builtin_amdgcn_s_setprio(0);                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                \
    }
// one K tile from buffer BUFI (tile index t): four phases
#define G8_TILE(BUFI, t)                                                                                             \
    {                                                                                                                \
        G8_READ_B(fb0, 2, BUFI) G8_READ_A(0, BUFI) stage(3, (BUFI) ^ 1, (t) + 1); G8_SYNC_AND_MMA(0, fb0)            \
        G8_READ_B(fb1, 3, BUFI) stage(1, (BUFI) ^ 1, (t) + 1); G8_SYNC_AND_MMA(1, fb1)                               \
        G8_READ_A(1, BUFI) stage(0, BUFI, (t) + 2); G8_SYNC_AND_MMA(2, fb1)                                          \
        stage(2, BUFI, (t) + 2); G8_SYNC_AND_MMA(3, fb0)                                                             \
    }
    int t = 0;
    for (; t + 1 < nkt; t += 2) {
        G8_TILE(0, t)
        G8_TILE(1, t + 1)
    }
    if (t < nkt) G8_TILE(0, t)
    if (stagger && wr == 0) __builtin_amdgcn_s_barrier();         // group 0 waits for group 1's last phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the dead prefetches of the tail
    const bool has_next = PERSIST && bid + (int)gridDim.x < nwg;
    if (PERSIST == 1 && has_next) {                               // nobody reads the ring any more: the next tile's first stages go out now
        __builtin_amdgcn_s_barrier();                             // (... every wave's dead prefetches have landed, not only this wave's)
        tile_of(bid + (int)gridDim.x); setup(); prologue();
    }

    // ---- epilogue: each quadrant is a 64 x 32 block = acc[2][4] of the register epilogue (pairs swapped by v_permlane16_swap)
    if constexpr (MMA32) {                                        // (hand the 32 x 32 accumulators to the 16 x 16 epilogue as they are: timing only)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[q][i][j][e] = acc32[q][i][j * 4 + e];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int mh = q >> 1, nh = (q == 1 || q == 2) ? 1 : 0;   // (A0,B0) (A0,B1) (A1,B1) (A1,B0)
        tile_epilogue_direct<bf16_raw, 2, false, false>(a, acc[q], smem, tid, 0, 0, g, r16, m0c + wr * 128 + mh * 64,
                                                        n0c + wc * 64 + nh * 32);
    }
    if (PERSIST == 2 && has_next) {
        __builtin_amdgcn_s_barrier();
        tile_of(bid + (int)gridDim.x); setup(); prologue();
    }
    }   // tiles of this block
#endif
}

template <int VARIANT, bool PRIO, bool STAGGER>
int launch(const G8Args& a, hipStream_t st) {
    (void)hipFuncSetAttribute((const void*)gemm_256sq_8phase<VARIANT, PRIO, STAGGER>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
    hipLaunchKernelGGL((gemm_256sq_8phase<VARIANT, PRIO, STAGGER>), dim3(a.mtiles * a.ntiles), dim3(512), 2 * BUF, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

// C [M][ldc] bf16 = A [M][K] x B [N][K]^T.  K a multiple of 64, N a multiple of 8, 16-B aligned pointers, A / B below 2 GiB.
// variant: 0 = XOR-8 swizzle (conflict-free), 1 = the guide's st_16x32, 2 = linear.  flags: 1 = no setprio, 2 = no stagger.
extern "C" int gemm8_bf16(long M, int N, int K, int ldc, const void* A, const void* B, void* C, int variant, int flags, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % 64) || (N % 8) || (ldc % 8) || (size_t)M * K * 2 >= 0x7fff0000ul || (size_t)N * K * 2 >= 0x7fff0000ul)
        return -1;
    G8Args a = {};
    a.in = (const char*)A; a.w = (const char*)B; a.out = (char*)C;
    a.M = (int)M; a.Co = N; a.K = K; a.ldo = ldc; a.HoWo = 1;
    a.in_bytes = (unsigned)((size_t)M * K * 2); a.w_bytes = (unsigned)((size_t)N * K * 2);
    a.mtiles = (int)((M + 255) / 256); a.ntiles = (N + 255) / 256;
    a.flags = flags;
    hipStream_t st = (hipStream_t)stream;
    if (variant == 4 || variant == 5) {
        int ncu = 256;
        { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount; }
        const int nwg_ = a.mtiles * a.ntiles, grid = nwg_ < ncu ? nwg_ : ncu;
        if (variant == 4) {
            (void)hipFuncSetAttribute((const void*)gemm_256sq_8phase<0, true, true, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
            hipLaunchKernelGGL((gemm_256sq_8phase<0, true, true, false, 1>), dim3(grid), dim3(512), 2 * BUF, st, a);
        } else {
            (void)hipFuncSetAttribute((const void*)gemm_256sq_8phase<0, true, true, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
            hipLaunchKernelGGL((gemm_256sq_8phase<0, true, true, false, 2>), dim3(grid), dim3(512), 2 * BUF, st, a);
        }
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (variant == 3) {
        (void)hipFuncSetAttribute((const void*)gemm_256sq_8phase<0, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
        hipLaunchKernelGGL((gemm_256sq_8phase<0, true, true, true>), dim3(a.mtiles * a.ntiles), dim3(512), 2 * BUF, st, a);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (variant == 1) return launch<1, true, true>(a, st);
    if (variant == 2) return launch<2, true, true>(a, st);
    if (flags == 1) return launch<0, false, true>(a, st);
    if (flags == 2) return launch<0, true, false>(a, st);
    if (flags == 3) return launch<0, false, false>(a, st);
    return launch<0, true, true>(a, st);
}
