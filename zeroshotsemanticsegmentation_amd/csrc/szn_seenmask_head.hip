// szn_seenmask_head.hip -- the phase-2 (seen-mask) head fused from the 1/32 map, gfx950.
//
// Reference chain (trainer_seenmask.py:50-70 on top of models.py:149-151):
//   seenmask_upscore = ConvTranspose2d(2, 2, 64, stride 32, bias=False) with a LEARNED kernel  -> crop [19:19+H]
//   target = np.in1d(label, seen)                (unlabelled pixels become 0 = "unseen", :55-56)
//   loss   = cross_entropy2d(score, target, size_average=True)      (utils.py:19-48)
//   pred   = score.max(1)[1]
//   backward into seenmask_upscore.weight and (through the 2-channel coarse map) seenmask_score
//
// The (B,2,H,W) score and its gradient never exist in HBM: an output pixel (Y,X) = (y+crop, x+crop) reads the four coarse
// cells (Y>>5 - 1 + di, X>>5 - 1 + dj) with filter taps ((Y&31) + 32 - 32 di, (X&31) + 32 - 32 dj), so all pixels of one
// 32x32 "pixel cell" share the same 4 x 2 coarse values and use every filter tap exactly once.  A persistent block owns
// one (ty, tx) position set -- thread = 4 pixels of the cell -- keeps ITS 64 filter taps and ITS 64 weight-gradient
// accumulators in registers for the whole launch and walks the pixel cells of the batch; per cell it reduces the 8
// d(coarse) contributions over the block.  Everything that crosses blocks goes through fixed-order slabs
// (bit-reproducible; no atomics): loss / count partials per block, d(coarse) partials per pixel cell, weight-gradient
// slabs per block, all combined by smh_finalize_kernel, which also applies the 1/N of size_average (the gradient is
// linear in it, so the main pass runs with N = 1).
//
// Arithmetic order per pixel is that of deconv_fwd_kernel / ce_fwd_kernel / ce_bwd_kernel (szn_head.hip): the fused score,
// loss terms and class decisions are bit-identical to the materialised path.
#include "szn_common.h"

namespace {

struct SmhGeom {
    int B, h, w, ldc, c0, H, W, crop;
    int i1lo, j1lo, nci, ncj;          // pixel cells: i1 in [i1lo, i1lo + nci), j1 likewise
};

constexpr int kTaps = 2 * 2 * 64 * 64;  // elements of the (2,2,64,64) filter bank

__device__ __forceinline__ void block_sum8(float (&v)[8], float (*red)[8], float* out) {
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = wave_sum(v[q]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                                   // previous cell's readers are done with red
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) red[wave][q] = v[q];
    }
    __syncthreads();
    if (threadIdx.x < 8) out[threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// part[blk] = {sum of loss terms, valid pixels, conf[0..3]} (doubles); cellpart[cell][di][dj][ci]; slab[blk][kTaps]
template <bool GRAD>
__global__ __launch_bounds__(256) void smh_cell_kernel(const float* __restrict__ coarse, const float* __restrict__ wt,
                                                       const int64_t* __restrict__ target, int n_class, ClassBits seen,
                                                       int64_t* __restrict__ pred, double* __restrict__ part,
                                                       float* __restrict__ cellpart, float* __restrict__ slab, SmhGeom g) {
    __shared__ float red[4][8];
    __shared__ double dred[4][6];
    const int t = threadIdx.x, tx = t & 31, ty0 = t >> 5;
    float wr[4][2][2][2][2];                   // [r][ci][co][di][dj]: the taps of this thread's pixel positions
    float dw[4][2][2][2][2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int co = 0; co < 2; ++co)
#pragma unroll
                for (int di = 0; di < 2; ++di)
#pragma unroll
                    for (int dj = 0; dj < 2; ++dj) {
                        const int ky = ty0 + 8 * r + 32 - 32 * di, kx = tx + 32 - 32 * dj;
                        wr[r][ci][co][di][dj] = wt[((ci * 2 + co) * 64 + ky) * 64 + kx];
                        dw[r][ci][co][di][dj] = 0.f;
                    }
    double lsum = 0.0, lcnt = 0.0, cf[4] = {0.0, 0.0, 0.0, 0.0};
    const int ncell = g.B * g.nci * g.ncj;
    for (int cell = blockIdx.x; cell < ncell; cell += gridDim.x) {
        const int jc = cell % g.ncj;
        const int tt = cell / g.ncj;
        const int ic = tt % g.nci, b = tt / g.nci;
        const int i1 = g.i1lo + ic, j1 = g.j1lo + jc;
        float c[2][2][2];                      // [di][dj][ci]; cells outside the map contribute exact zeros
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                const int i = i1 - 1 + di, j = j1 - 1 + dj;
                const bool ok = i >= 0 && i < g.h && j >= 0 && j < g.w;
                const float* p = coarse + (((long)b * g.h + (ok ? i : 0)) * g.w + (ok ? j : 0)) * g.ldc + g.c0;
                c[di][dj][0] = ok ? p[0] : 0.f;
                c[di][dj][1] = ok ? p[1] : 0.f;
            }
        float dc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // [(di*2+dj)*2+ci]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = 32 * i1 + ty0 + 8 * r - g.crop, x = 32 * j1 + tx - g.crop;
            if (y < 0 || y >= g.H || x < 0 || x >= g.W) continue;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int di = 0; di < 2; ++di)
#pragma unroll
                    for (int dj = 0; dj < 2; ++dj) {
                        s0 = fmaf(c[di][dj][ci], wr[r][ci][0][di][dj], s0);
                        s1 = fmaf(c[di][dj][ci], wr[r][ci][1][di][dj], s1);
                    }
            const long pix = ((long)b * g.H + y) * g.W + x;
            const int am = s1 > s0 ? 1 : 0;                  // first maximum on ties (torch max / ce_fwd_kernel)
            if (pred) pred[pix] = am;
            const long lbl = target[pix];
            int tb;
            bool valid = true;
            // labels below -1 are batch PADDING (datasets.pad_collate writes -2): not a pixel of any image, so it is neither a
            // target nor counted -- unlike -1 ("unlabelled"), which the reference turns into target 0 (trainer_seenmask.py:55-56)
            if (n_class > 0) { valid = lbl >= -1; tb = (lbl < n_class && in_set(seen, lbl)) ? 1 : 0; }
            else { valid = lbl >= 0 && lbl < 2; tb = (int)lbl; }
            if (!valid) continue;
            const float mx = am ? s1 : s0;
            const float e0 = expf(s0 - mx), e1 = expf(s1 - mx);
            float se = 0.f;
            se += e0; se += e1;
            lsum += (double)(-((tb ? s1 : s0) - mx - logf(se)));
            lcnt += 1.0;
            cf[tb * 2 + am] += 1.0;
            if (GRAD) {
                const float inv = 1.f / se;
                const float d0 = e0 * inv - (tb == 0 ? 1.f : 0.f), d1 = e1 * inv - (tb == 1 ? 1.f : 0.f);
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int di = 0; di < 2; ++di)
#pragma unroll
                        for (int dj = 0; dj < 2; ++dj) {
                            dw[r][ci][0][di][dj] = fmaf(c[di][dj][ci], d0, dw[r][ci][0][di][dj]);
                            dw[r][ci][1][di][dj] = fmaf(c[di][dj][ci], d1, dw[r][ci][1][di][dj]);
                            float& a = dc[(di * 2 + dj) * 2 + ci];
                            a = fmaf(d0, wr[r][ci][0][di][dj], a);
                            a = fmaf(d1, wr[r][ci][1][di][dj], a);
                        }
            }
        }
        if (GRAD) block_sum8(dc, red, cellpart + (long)cell * 8);
    }
    if (GRAD) {
        float* sl = slab + (long)blockIdx.x * kTaps;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int co = 0; co < 2; ++co)
#pragma unroll
                    for (int di = 0; di < 2; ++di)
#pragma unroll
                        for (int dj = 0; dj < 2; ++dj) {
                            const int ky = ty0 + 8 * r + 32 - 32 * di, kx = tx + 32 - 32 * dj;
                            sl[((ci * 2 + co) * 64 + ky) * 64 + kx] = dw[r][ci][co][di][dj];
                        }
    }
    double v[6] = {lsum, lcnt, cf[0], cf[1], cf[2], cf[3]};
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = wave_sum_d(v[q]);
    if ((t & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) dred[t >> 6][q] = v[q];
    }
    __syncthreads();
    if (t < 6) part[(long)blockIdx.x * 6 + t] = (dred[0][t] + dred[1][t]) + (dred[2][t] + dred[3][t]);
}

// every block re-derives S = sum of loss terms and N = valid pixels in the same fixed order, then takes its share of the
// outputs: weight gradient (kTaps elements), d(coarse) gather (B*h*w*2 elements).  Block 0 also writes loss / stats / conf.
__global__ __launch_bounds__(256) void smh_finalize_kernel(const double* __restrict__ part, int G,
                                                           const float* __restrict__ cellpart, const float* __restrict__ slab,
                                                           float* __restrict__ loss, float* __restrict__ stats,
                                                           int64_t* __restrict__ conf, float* __restrict__ dsc,
                                                           float* __restrict__ dweight, SmhGeom g) {
    __shared__ double sh[4][6];
    const int t = threadIdx.x;
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k = t; k < G; k += 256)
#pragma unroll
        for (int q = 0; q < 6; ++q) v[q] += part[(long)k * 6 + q];
#pragma unroll
    for (int q = 0; q < 6; ++q) v[q] = wave_sum_d(v[q]);
    if ((t & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) sh[t >> 6][q] = v[q];
    }
    __syncthreads();
    const double S = (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]);
    const double N = (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]);
    if (blockIdx.x == 0) {
        if (t == 0) {
            loss[0] = (float)(S / N);
            if (stats) { stats[0] = (float)S; stats[1] = (float)N; }
        }
        if (conf && t < 4) conf[t] += (int64_t)((sh[0][2 + t] + sh[1][2 + t]) + (sh[2][2 + t] + sh[3][2 + t]));
    }
    const float gs = (float)(1.0 / N);
    // weight gradient: 64 elements per block, the G slabs split over four thread groups (four independent chains of loads per
    // element instead of one 256-long one; 256 blocks instead of 64), combined in a fixed order
    __shared__ float wsum[4][64];
    const int nwblk = kTaps / 64;
    if ((int)blockIdx.x < nwblk) {
        if (dweight) {
            const int e = blockIdx.x * 64 + (t & 63), q = t >> 6;
            const int k0 = q * ((G + 3) / 4), k1 = min(G, k0 + (G + 3) / 4);
            float a = 0.f;
            for (int k = k0; k < k1; ++k) a += slab[(long)k * kTaps + e];
            wsum[q][t & 63] = a;
            __syncthreads();
            if (t < 64) dweight[blockIdx.x * 64 + t] = ((wsum[0][t] + wsum[1][t]) + (wsum[2][t] + wsum[3][t])) * gs;
        }
        return;
    }
    const long e2 = (long)(blockIdx.x - nwblk) * 256 + t;
    if (!dsc || e2 >= (long)g.B * g.h * g.w * 2) return;
    const int ci = (int)(e2 & 1);
    long m = e2 >> 1;
    const int j = (int)(m % g.w); m /= g.w;
    const int i = (int)(m % g.h);
    const int b = (int)(m / g.h);
    float a = 0.f;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            const int ic = i + 1 - di - g.i1lo, jc = j + 1 - dj - g.j1lo;
            if (ic < 0 || ic >= g.nci || jc < 0 || jc >= g.ncj) continue;
            a += cellpart[(((long)b * g.nci + ic) * g.ncj + jc) * 8 + (di * 2 + dj) * 2 + ci];
        }
    dsc[e2] = a * gs;
}

// ---- seenmask_score (Conv2d(4096, 2, 1), models.py:97) weight / bias gradient from the compact d(coarse) -----------------
// dw[c][k] = sum_m dsc[m][c] * feat[m][k]: lane = 8 consecutive channels, block = `rows` consecutive pixels, fp32 slabs
template <typename T> struct ld8;
template <> struct ld8<float> {
    __device__ static __forceinline__ void load(const float* p, float (&o)[8]) {
        const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
};
template <typename T> struct ld8 {
    __device__ static __forceinline__ void load(const T* p, float (&o)[8]) {
        const uint4 u = *(const uint4*)p;
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o[2 * q] = from_bits16<T>((uint16_t)(w[q] & 0xffffu));
            o[2 * q + 1] = from_bits16<T>((uint16_t)(w[q] >> 16));
        }
    }
};

template <typename T>
__global__ __launch_bounds__(512) void smh_score_wgrad_kernel(const T* __restrict__ feat, const float* __restrict__ dsc,
                                                              float* __restrict__ slab, long M, int F, int ldf, int rows) {
    const int k0 = (blockIdx.y * 512 + threadIdx.x) * 8;
    if (k0 >= F) return;
    const long m0 = (long)blockIdx.x * rows, m1 = (m0 + rows < M) ? m0 + rows : M;
    float a0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, a1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long m = m0; m < m1; ++m) {
        const float d0 = dsc[2 * m], d1 = dsc[2 * m + 1];
        float x[8];
        ld8<T>::load(feat + m * ldf + k0, x);
#pragma unroll
        for (int q = 0; q < 8; ++q) { a0[q] = fmaf(d0, x[q], a0[q]); a1[q] = fmaf(d1, x[q], a1[q]); }
    }
    float* s0 = slab + ((long)blockIdx.x * 2) * F + k0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { s0[q] = a0[q]; s0[F + q] = a1[q]; }
}

__global__ __launch_bounds__(256) void smh_score_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ dsc,
                                                               float* __restrict__ dw, float* __restrict__ db, long M, int F,
                                                               int S) {
    // 64 outputs per block, the S slabs split over four thread groups, combined in a fixed order
    __shared__ float part[4][64];
    const int t = threadIdx.x, q = t >> 6;
    const long e = (long)blockIdx.x * 64 + (t & 63);
    const int s0 = q * ((S + 3) / 4), s1 = min(S, s0 + (S + 3) / 4);
    float a = 0.f;
    if (e < 2L * F)
        for (int s = s0; s < s1; ++s) a += slab[(long)s * 2 * F + e];
    part[q][t & 63] = a;
    __syncthreads();
    if (t < 64 && e < 2L * F) dw[e] = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
    if (blockIdx.x == gridDim.x - 1 && t < 128 && db) {     // bias: one wave per channel, fixed order
        const int c = t >> 6, lane = t & 63;
        float b = 0.f;
        for (long m = lane; m < M; m += 64) b += dsc[2 * m + c];
        b = wave_sum(b);
        if (lane == 0) db[c] = b;
    }
}

SmhGeom make_geom(int B, int h, int w, int ldc, int c0, int H, int W, int crop) {
    SmhGeom g;
    g.B = B; g.h = h; g.w = w; g.ldc = ldc; g.c0 = c0; g.H = H; g.W = W; g.crop = crop;
    g.i1lo = crop >> 5; g.j1lo = crop >> 5;
    g.nci = ((crop + H - 1) >> 5) - g.i1lo + 1;
    g.ncj = ((crop + W - 1) >> 5) - g.j1lo + 1;
    return g;
}

int smh_grid(const SmhGeom& g) {
    const long ncell = (long)g.B * g.nci * g.ncj;
    return (int)(ncell < 256 ? ncell : 256);
}

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t szn_seenmask_head_workspace_bytes(int B, int h, int w, int H, int W, int crop) {
    const SmhGeom g = make_geom(B, h, w, 0, 0, H, W, crop);
    const long ncell = (long)g.B * g.nci * g.ncj;
    return align256(256 * 6 * sizeof(double)) + align256((size_t)ncell * 8 * sizeof(float)) + (size_t)256 * kTaps * sizeof(float);
}

namespace {
int seenmask_head_impl(int B, int h, int w, int ldc, int c0, int H, int W, int crop, const float* coarse, const float* weight,
                       const int64_t* target, int n_class, const ClassBits& seen_bits, float* loss, float* stats, int64_t* conf,
                       int64_t* pred, float* dscore2, float* dweight, void* workspace, szn_stream_t stream) {
    if (B < 1 || h < 1 || w < 1 || H < 1 || W < 1 || crop < 0) SZN_FAIL(SZN_ERR_ARG, "szn_seenmask_head: bad geometry");
    if (!coarse || !weight || !target || !loss || !workspace) SZN_FAIL(SZN_ERR_ARG, "szn_seenmask_head: null pointer");
    if (n_class < 0 || n_class > SZN_MAX_CLASSES) SZN_FAIL(SZN_ERR_UNSUPPORTED, "szn_seenmask_head: n_class %d > %d", n_class, SZN_MAX_CLASSES);
    if ((dscore2 == nullptr) != (dweight == nullptr)) SZN_FAIL(SZN_ERR_ARG, "szn_seenmask_head: dscore2 and dweight go together");
    // every tap of a pixel must exist or be outside the map on the low side only when crop says so: (Y>>5) <= h
    if (((crop + H - 1) >> 5) > h || ((crop + W - 1) >> 5) > w)
        SZN_FAIL(SZN_ERR_ARG, "szn_seenmask_head: %dx%d output (+crop %d) does not fit a %dx%d map at stride 32", H, W, crop, h, w);
    const SmhGeom g = make_geom(B, h, w, ldc, c0, H, W, crop);
    const long ncell = (long)g.B * g.nci * g.ncj;
    const int G = smh_grid(g);
    char* ws = (char*)workspace;
    double* part = (double*)ws;
    float* cellpart = (float*)(ws + align256(256 * 6 * sizeof(double)));
    float* slab = (float*)(ws + align256(256 * 6 * sizeof(double)) + align256((size_t)ncell * 8 * sizeof(float)));
    hipStream_t st = (hipStream_t)stream;
    const bool grad = dweight != nullptr;
    if (grad) smh_cell_kernel<true><<<G, 256, 0, st>>>(coarse, weight, target, n_class, seen_bits, pred, part, cellpart, slab, g);
    else smh_cell_kernel<false><<<G, 256, 0, st>>>(coarse, weight, target, n_class, seen_bits, pred, part, cellpart, slab, g);
    SZN_CHECK_LAUNCH("smh_cell_kernel");
    const int fblocks = grad ? kTaps / 64 + szn_div_up((long)B * h * w * 2, 256) : 1;
    smh_finalize_kernel<<<fblocks, 256, 0, st>>>(part, G, cellpart, slab, loss, stats, conf, grad ? dscore2 : nullptr,
                                                               grad ? dweight : nullptr, g);
    SZN_CHECK_LAUNCH("smh_finalize_kernel");
    return SZN_OK;
}
}  // namespace

extern "C" int szn_seenmask_head(int B, int h, int w, int ldc, int c0, int H, int W, int crop, const float* coarse,
                                 const float* weight, const int64_t* target, int n_class, uint64_t seen_bits, float* loss,
                                 float* stats, int64_t* conf, int64_t* pred, float* dscore2, float* dweight, void* workspace,
                                 szn_stream_t stream) {
    if (n_class > 64) SZN_FAIL(SZN_ERR_UNSUPPORTED, "szn_seenmask_head: n_class %d > 64 needs szn_seenmask_head_k (szn_class_set)", n_class);
    return seenmask_head_impl(B, h, w, ldc, c0, H, W, crop, coarse, weight, target, n_class, class_bits64(seen_bits), loss, stats, conf,
                              pred, dscore2, dweight, workspace, stream);
}

extern "C" int szn_seenmask_head_k(int B, int h, int w, int ldc, int c0, int H, int W, int crop, const float* coarse,
                                   const float* weight, const int64_t* target, int n_class, const szn_class_set* seen, float* loss,
                                   float* stats, int64_t* conf, int64_t* pred, float* dscore2, float* dweight, void* workspace,
                                   szn_stream_t stream) {
    return seenmask_head_impl(B, h, w, ldc, c0, H, W, crop, coarse, weight, target, n_class, class_bits(seen), loss, stats, conf, pred,
                              dscore2, dweight, workspace, stream);
}

extern "C" size_t szn_seenmask_score_wgrad_workspace_bytes(long M, int F) {
    const int rows = 32;
    return (size_t)szn_div_up(M, rows) * 2 * F * sizeof(float);
}

extern "C" int szn_seenmask_score_wgrad(int dtype, long M, int F, int ldf, const void* feat, const float* dscore2, float* dw,
                                        float* db, void* workspace, szn_stream_t stream) {
    if (M < 1 || F < 8 || (F % 8) || ldf < F || (ldf % 8)) SZN_FAIL(SZN_ERR_ARG, "szn_seenmask_score_wgrad: M %ld F %d ldf %d", M, F, ldf);
    if (!feat || !dscore2 || !dw || !workspace) SZN_FAIL(SZN_ERR_ARG, "szn_seenmask_score_wgrad: null pointer");
    const int rows = 32;
    const int S = szn_div_up(M, rows);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(S, szn_div_up(F, 512 * 8));
    float* slab = (float*)workspace;
    if (dtype == SZN_F32) smh_score_wgrad_kernel<float><<<grid, 512, 0, st>>>((const float*)feat, dscore2, slab, M, F, ldf, rows);
    else if (dtype == SZN_BF16) smh_score_wgrad_kernel<bf16_raw><<<grid, 512, 0, st>>>((const bf16_raw*)feat, dscore2, slab, M, F, ldf, rows);
    else if (dtype == SZN_F16) smh_score_wgrad_kernel<f16_raw><<<grid, 512, 0, st>>>((const f16_raw*)feat, dscore2, slab, M, F, ldf, rows);
    else SZN_FAIL(SZN_ERR_ARG, "szn_seenmask_score_wgrad: dtype %d", dtype);
    SZN_CHECK_LAUNCH("smh_score_wgrad_kernel");
    smh_score_reduce_kernel<<<szn_div_up(2L * F, 64), 256, 0, st>>>(slab, dscore2, dw, db, M, F, S);
    SZN_CHECK_LAUNCH("smh_score_reduce_kernel");
    return SZN_OK;
}
