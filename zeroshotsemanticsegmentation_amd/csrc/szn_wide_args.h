// szn_wide_args.h -- argument block shared by the 256-pixel x 256-cout tile kernels (szn_conv_wide.hip, szn_conv_8ph.hip)
#pragma once
#include "szn_common.h"

namespace {

struct WideArgs {
    const char* in; const char* w; const float* bias; const char* gate; const float* cscale; char* out;
    float* colsum;
    float* cslab;              // optional [mtiles][Co]: the column sums of a pixel tile go to its row instead of fp32 atomics on colsum
    unsigned in_bytes, w_bytes;
    int B, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, pad;
    int ldi, ldo, ldg, relu, out_f32;
    int M, HoWo, mtiles, ntiles, nmajor;
    float* ws;                 // split-K: fp32 slabs [nsplit][M][Co] (plain stores, no epilogue); nullptr = single pass
    int nsplit, chunks_per_split;
    int stagger;               // 1: wave pairs take turns issuing the LDS-DMA loads of a chunk (SZN_WIDE_STAGGER=0: all at once)
    int gate_prefetch;         // 1: the epilogue fetches the ReLU-gate rows one pass ahead (SZN_WIDE_GATEPF=0: inside the store loop)
    int direct_ep;             // 1: epilogue straight from the accumulator registers (wide_epilogue_direct)
    int abl_ep;                // ablation builds only (SZN_WIDE_EPABL): 1 = epilogue without global stores / gate loads, 2 = no epilogue
    int proj_abl;              // ablation builds only (SZN_PROJ_ABLATE): 1 = every block of proj_gemm_stream streams the rows of block 0
};

constexpr unsigned kOOBx = 0x80000000u;

__device__ __forceinline__ int xcd_remap_w(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace
