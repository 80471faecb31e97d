// szn_head.hip -- the per-pixel head of the SZN path (HBM-bound): bilinear x32 upsample + crop, the
// learned seen-mask deconvolution, cosine / MSE / cross-entropy losses with their gradients, the
// nearest-class-embedding argmax (infer_lbl family) and the confusion histogram.
//
// Reference sites: models.py:11-24,94,98,146-151 (upscore / seenmask_upscore + crop),
// utils.py:19-48 (cross_entropy2d), :50-73 (mse_loss), :75-102 (cosine_loss), :104-154 (metrics),
// :159-205 (infer_lbl, infer_lbl_forced_unseen, infer_lbl_szn, stich_seen_unseen_with_mask).
//
// Arithmetic contract shared with oracle/szn_oracle.c (bit-exact argmax): every per-pixel dot product
// and squared norm is an fmaf chain over the channel index in ascending order, norms use correctly
// rounded sqrtf, similarities use IEEE division; this file is compiled with -ffp-contract=off.
#include "szn_common.h"

namespace {

// 1-D bilinear tap of get_upsampling_weight(k = 2 S): factor S, center S - 0.5 (models.py:13-20), in double
template <int S>
__device__ __forceinline__ double bil1d(int t) { return 1.0 - fabs((double)t - ((double)S - 0.5)) / (double)S; }

// ---- upscore forward ---------------------------------------------------------------------------------------------------
// Fixed bilinear ConvTranspose2d(E, E, 2 S, stride S) + crop; S = 32 is FCN32s' upscore (models.py:94,146-147), S = 8 the last
// stage of the FCN8s head.  block = (256 / S cells of S columns, one cell row I, image b); thread = one column X of the uncropped
// S(h+1) x S(w+1) deconv output (x = X - crop).  The 2 x (256/S + 2) coarse vectors the segment blends are staged in LDS once and
// reused for the cell row's S output rows; the four bilinear weights of a pixel are formed once per row (double product rounded
// to float, models.py:13-24) and reused for all E channels.  Per element: 4 LDS reads + 4 fmaf (tap order (i-1,j-1), (i-1,j),
// (i,j-1), (i,j): bit-identical to the gather form) + one coalesced store -- the kernel is bound by its B*E*H*W*4 B of stores.
template <int S>
__global__ __launch_bounds__(256) void up_fwd_kernel(const float* __restrict__ coarse, float* __restrict__ out, int B,
                                                     int h, int w, int E, int ldc, int c0, int H, int W, int crop) {
    constexpr int NC = 256 / S, NJ = NC + 2;     // cells per segment; coarse columns it touches: NC seg - 1 .. NC seg + NC
    extern __shared__ __attribute__((aligned(16))) float taps[];      // [2][NJ][E]
    const int seg = blockIdx.x, I = blockIdx.y, b = blockIdx.z;
    const int jbase = NC * seg - 1;
    for (int idx = threadIdx.x; idx < 2 * NJ * E; idx += 256) {
        const int r = idx / (NJ * E), rem = idx - r * NJ * E;
        const int jj = rem / E, c = rem - jj * E;
        const int i = I - 1 + r, j = jbase + jj;
        taps[idx] = (i >= 0 && i < h && j >= 0 && j < w) ? coarse[(((long)b * h + i) * w + j) * ldc + c0 + c] : 0.f;
    }
    __syncthreads();
    const int X = 256 * seg + threadIdx.x, x = X - crop;
    if (x < 0 || x >= W) return;
    const int jl = threadIdx.x / S, tx = threadIdx.x % S;            // cell inside the segment (J = NC seg + jl), column inside it
    const double fx1 = bil1d<S>(tx), fx0 = bil1d<S>(tx + S);
    const float* t00 = taps + (0 * NJ + jl) * E;                      // (I-1, J-1)
    const float* t01 = t00 + E;                                       // (I-1, J)
    const float* t10 = taps + (1 * NJ + jl) * E;                      // (I,   J-1)
    const float* t11 = t10 + E;                                       // (I,   J)
    for (int ty = 0; ty < S; ++ty) {
        const int y = S * I + ty - crop;
        if (y < 0 || y >= H) continue;
        const double fy1 = bil1d<S>(ty), fy0 = bil1d<S>(ty + S);
        const float w00 = (float)(fy0 * fx0), w01 = (float)(fy0 * fx1), w10 = (float)(fy1 * fx0), w11 = (float)(fy1 * fx1);
        float* op = out + ((long)b * E * H + y) * W + x;
        for (int c = 0; c < E; ++c) {
            float acc = fmaf(t00[c], w00, 0.f);
            acc = fmaf(t01[c], w01, acc);
            acc = fmaf(t10[c], w10, acc);
            acc = fmaf(t11[c], w11, acc);
            op[(long)c * H * W] = acc;
        }
    }
}

// ---- upscore backward ----------------------------------------------------------------------------------------------------
// dcoarse[b][i][j][c] = sum over the 2S x 2S window of (i, j).  block = (coarse row i, image b, channel slice); thread = column X
// of the uncropped output.  Per channel a thread forms ONE weighted column sum over the window's 2S rows (ky = Y - S i), then
// its column contributes bil1d(tx) * col to output J (left half of J's window) and bil1d(S + tx) * col to output J - 1 (right
// half): two S-lane reductions per cell, combined through LDS, plain stores -- deterministic, every dscore element is read by
// 2 blocks (the row overlap) instead of 4 waves.
template <int S>
__global__ __launch_bounds__(256) void up_bwd_kernel(const float* __restrict__ dscore, float* __restrict__ dcoarse,
                                                     int B, int h, int w, int E, int ldc, int c0, int H, int W,
                                                     int crop, int cper) {
    constexpr int NC = 256 / S;
    extern __shared__ __attribute__((aligned(16))) float red[];       // [2][cper][ncell]: R (own cell) | L (for the cell to the left)
    const int i = blockIdx.x, b = blockIdx.y, cbeg = blockIdx.z * cper;
    const int cend = min(E, cbeg + cper);
    const int ncell = w + 1;
    const int laneS = threadIdx.x % S;
    const float fL = (float)bil1d<S>(laneS), fR = (float)bil1d<S>(laneS + S);
    __shared__ float wy[2 * S];
    if (threadIdx.x < 2 * S) wy[threadIdx.x] = (float)bil1d<S>(threadIdx.x);
    __syncthreads();
    for (int seg = 0; seg * NC < ncell; ++seg) {
        const int X = 256 * seg + threadIdx.x, x = X - crop;
        const int J = X / S;
        const bool okx = x >= 0 && x < W && J < ncell;
        // four channels at a time: four independent accumulation chains, 16 loads in flight per lane
        const int ky_lo = max(0, crop - S * i), ky_hi = min(2 * S, H + crop - S * i);
        for (int c = cbeg; c < cend; c += 4) {
            float col[4] = {0.f, 0.f, 0.f, 0.f};
            if (okx) {
                const float* plane = dscore + ((long)b * E + c) * H * W + x;
                const long cs = (long)H * W;
                const int nc = min(4, cend - c);
                for (int ky = ky_lo; ky < ky_hi; ++ky) {
                    const float wv = wy[ky];
                    const float* rp = plane + (long)(S * i + ky - crop) * W;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (u < nc) col[u] = fmaf(rp[u * cs], wv, col[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float pl = col[u] * fL, pr = col[u] * fR;      // this column inside the window of output J / of output J - 1
#pragma unroll
                for (int o = S / 2; o > 0; o >>= 1) { pl += __shfl_xor(pl, o, 64); pr += __shfl_xor(pr, o, 64); }
                if (laneS == 0 && J < ncell && c + u < cend) {
                    red[(c + u - cbeg) * ncell + J] = pl;
                    red[(cper + c + u - cbeg) * ncell + J] = pr;
                }
            }
        }
    }
    __syncthreads();
    // output j = (columns of cell j, left half of the window) + (columns of cell j + 1, right half)
    for (int idx = threadIdx.x; idx < (cend - cbeg) * w; idx += 256) {
        const int j = idx / (cend - cbeg), cc = idx - j * (cend - cbeg);
        dcoarse[(((long)b * h + i) * w + j) * ldc + c0 + cbeg + cc] = red[cc * ncell + j] + red[(cper + cc) * ncell + j + 1];
    }
}

// ---- FCN8s skip fusion: x2 bilinear ConvTranspose2d(E, E, 4, stride 2) between coarse NHWC maps (a few MB) ---------------------
// out (B, 2h+2, 2w+2, ld) [c] = sum_{i,j} in[i][j][c] f[Y - 2i] f[X - 2j], f = (0.25, 0.75, 0.75, 0.25): every output blends
// <= 2 x 2 inputs, tap order (i-1,j-1), (i-1,j), (i,j-1), (i,j) with i = Y >> 1
__global__ __launch_bounds__(256) void up2_nhwc_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int h,
                                                           int w, int C, int ld) {
    const int Ho = 2 * h + 2, Wo = 2 * w + 2;
    const long total = (long)B * Ho * Wo * C;
    for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
        const int c = (int)(gid % C);
        long t = gid / C;
        const int X = (int)(t % Wo); t /= Wo;
        const int Y = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const int i1 = Y >> 1, j1 = X >> 1, ty = Y & 1, tx = X & 1;
        float acc = 0.f;
#pragma unroll
        for (int di = 0; di < 2; ++di) {
            const int i = i1 - 1 + di;
            if (i < 0 || i >= h) continue;
            const double fy = bil1d<2>(ty + 2 - 2 * di);
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                const int j = j1 - 1 + dj;
                if (j < 0 || j >= w) continue;
                const float wt = (float)(fy * bil1d<2>(tx + 2 - 2 * dj));
                acc = fmaf(in[(((long)b * h + i) * w + j) * ld + c], wt, acc);
            }
        }
        out[(((long)b * Ho + Y) * Wo + X) * ld + c] = acc;
    }
}

// din[i][j][c] = sum over the 4 x 4 window (ky, kx) of dout[2i + ky][2j + kx][c] f[ky] f[kx], ascending (ky, kx)
__global__ __launch_bounds__(256) void up2_nhwc_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int B, int h,
                                                           int w, int C, int ld) {
    const int Wo = 2 * w + 2, Ho = 2 * h + 2;
    const long total = (long)B * h * w * C;
    for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
        const int c = (int)(gid % C);
        long t = gid / C;
        const int j = (int)(t % w); t /= w;
        const int i = (int)(t % h);
        const int b = (int)(t / h);
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const float wt = (float)(bil1d<2>(ky) * bil1d<2>(kx));
                acc = fmaf(dout[(((long)b * Ho + 2 * i + ky) * Wo + 2 * j + kx) * ld + c], wt, acc);
            }
        din[(((long)b * h + i) * w + j) * ld + c] = acc;
    }
}

// ---- seenmask_upscore (dense learned ConvTranspose2d, C <= 4) -----------------------------------
// weight: torch layout (Cin, Cout, 64, 64)
__global__ __launch_bounds__(256) void deconv_fwd_kernel(const float* __restrict__ coarse, const float* __restrict__ wt,
                                                         float* __restrict__ out, int B, int h, int w, int C, int ldc,
                                                         int c0, int H, int W, int crop) {
    const long total = (long)B * C * H * W;
    for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
        const int x = (int)(gid % W);
        long t = gid / W;
        const int y = (int)(t % H); t /= H;
        const int co = (int)(t % C);
        const int b = (int)(t / C);
        const int Y = y + crop, X = x + crop;
        const int i1 = Y >> 5, j1 = X >> 5, ty = Y & 31, tx = X & 31;
        float acc = 0.f;
        for (int ci = 0; ci < C; ++ci) {
            const float* wk = wt + ((long)ci * C + co) * 4096;
            const float* base = coarse + (long)b * h * w * ldc + c0 + ci;
#pragma unroll
            for (int di = 0; di < 2; ++di) {
                const int i = i1 - 1 + di, ky = ty + 32 - 32 * di;
                if (i < 0 || i >= h) continue;
#pragma unroll
                for (int dj = 0; dj < 2; ++dj) {
                    const int j = j1 - 1 + dj, kx = tx + 32 - 32 * dj;
                    if (j < 0 || j >= w) continue;
                    acc = fmaf(base[((long)i * w + j) * ldc], wk[ky * 64 + kx], acc);
                }
            }
        }
        out[gid] = acc;
    }
}

// dcoarse[b][i][j][ci] = sum_co sum_window dout[b][co][y][x] * wt[ci][co][ky][kx]; wave per (b,i,j,ci)
__global__ __launch_bounds__(256) void deconv_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ wt,
                                                           float* __restrict__ dcoarse, int B, int h, int w, int C,
                                                           int ldc, int c0, int H, int W, int crop) {
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long total = (long)B * h * w * C;
    if (wid >= total) return;
    const int ci = (int)(wid % C);
    long t = wid / C;
    const int j = (int)(t % w); t /= w;
    const int i = (int)(t % h);
    const int b = (int)(t / h);
    const int x = 32 * j - crop + lane;
    float acc = 0.f;
    if (x >= 0 && x < W) {
        for (int co = 0; co < C; ++co) {
            const float* plane = dout + ((long)b * C + co) * H * W;
            const float* wk = wt + ((long)ci * C + co) * 4096;
            for (int ky = 0; ky < 64; ++ky) {
                const int y = 32 * i - crop + ky;
                if (y < 0 || y >= H) continue;
                acc = fmaf(plane[(long)y * W + x], wk[ky * 64 + lane], acc);
            }
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) dcoarse[(((long)b * h + i) * w + j) * ldc + c0 + ci] = acc;
}

// dwt[ci][co][ky][kx] (+)= sum_{b,i,j} coarse[b][i][j][ci] * dout[b][co][32i+ky-crop][32j+kx-crop]
// block = one filter row (ci, co, ky): 64 kx lanes x 4 parts; part q sums the coarse rows i = q, q + 4, ... (four independent
// chains per output instead of one 2,312-term chain on 64 blocks), the parts are combined in a fixed order: deterministic.
__global__ __launch_bounds__(256) void deconv_wgrad_kernel(const float* __restrict__ coarse, const float* __restrict__ dout,
                                                           float* __restrict__ dwt, int B, int h, int w, int C, int ldc,
                                                           int c0, int H, int W, int crop, int accumulate) {
    __shared__ float part[4][64];
    const int kx = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int ky = blockIdx.x & 63, co = (blockIdx.x >> 6) % C, ci = (blockIdx.x >> 6) / C;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* plane = dout + ((long)b * C + co) * H * W;
        for (int i = q; i < h; i += 4) {
            const int y = 32 * i - crop + ky;
            if (y < 0 || y >= H) continue;
            const float* cr = coarse + (((long)b * h + i) * w) * ldc + c0 + ci;
            const float* row = plane + (long)y * W - crop + kx;
            for (int j = 0; j < w; ++j) {
                const int x = 32 * j - crop + kx;
                if (x < 0 || x >= W) continue;
                acc = fmaf(cr[(long)j * ldc], row[32 * j], acc);
            }
        }
    }
    part[q][kx] = acc;
    __syncthreads();
    if (q == 0) {
        const float v = (part[0][kx] + part[1][kx]) + (part[2][kx] + part[3][kx]);
        const int gid = ((ci * C + co) * 64 + ky) * 64 + kx;
        dwt[gid] = accumulate ? dwt[gid] + v : v;
    }
}

// ---- block reduction of (sum, count) into a per-(image, block) partial --------------------------
__device__ __forceinline__ void block_partial(double v, double n, double* part /* [2] */) {
    __shared__ double sv[4], sn[4];
    v = wave_sum_d(v); n = wave_sum_d(n);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[wave] = v; sn[wave] = n; }
    __syncthreads();
    if (threadIdx.x == 0) { part[0] = sv[0] + sv[1] + sv[2] + sv[3]; part[1] = sn[0] + sn[1] + sn[2] + sn[3]; }
}

// mode 0: cosine  loss = mean_b (N_b - S_b) / N_b ; mode 1: mse  loss = mean_b S_b / N_b ;
// mode 2: CE      loss = sum_b S_b  (/ sum_b N_b if size_average)
__global__ void loss_finalize_kernel(const double* __restrict__ part, int B, int nblk, int mode, int size_average,
                                     float* __restrict__ loss, float* __restrict__ stats) {
    const int lane = threadIdx.x;   // 64 threads
    double tot = 0.0, totn = 0.0, acc = 0.0;
    for (int b = 0; b < B; ++b) {
        double s = 0.0, n = 0.0;
        for (int k = lane; k < nblk; k += 64) { s += part[((long)b * nblk + k) * 2]; n += part[((long)b * nblk + k) * 2 + 1]; }
        s = wave_sum_d(s); n = wave_sum_d(n);
        if (lane == 0) { stats[2 * b] = (float)s; stats[2 * b + 1] = (float)n; }
        tot += s; totn += n;
        if (mode == 0) acc += (n - s) / n;
        else if (mode == 1) acc += s / n;
    }
    if (lane == 0) {
        if (mode == 2) loss[0] = (float)(size_average ? tot / totn : tot);
        else loss[0] = (float)(acc / B);
    }
}

// ---- cosine / mse forward: thread = pixel, grid (blocks per image, B) -----------------------------
template <int MODE>   // 0 cosine, 1 mse
__global__ __launch_bounds__(256) void embed_loss_fwd_kernel(const float* __restrict__ score, const int64_t* __restrict__ target,
                                                             const float* __restrict__ embed, const float* __restrict__ tembed,
                                                             double* __restrict__ part, int E, int HW, int K) {
    extern __shared__ __attribute__((aligned(16))) float emb_l[];   // [K][E] when gathering
    if (!tembed) {
        for (int i = threadIdx.x; i < K * E; i += 256) emb_l[i] = embed[i];
        __syncthreads();
    }
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    double term = 0.0, cnt = 0.0;
    if (p < HW) {
        const long lbl = target[(long)b * HW + p];
        if (lbl >= 0) {
            const float* sp = score + (long)b * E * HW + p;
            const float* tp = tembed ? tembed + (long)b * E * HW + p : nullptr;
            const float* er = emb_l + (lbl < K ? lbl : 0) * E;
            float ss = 0.f, st = 0.f, tt = 0.f;
            for (int c = 0; c < E; ++c) {
                const float s = sp[(long)c * HW];
                const float t = tp ? tp[(long)c * HW] : er[c];
                if (MODE == 0) { ss = fmaf(s, s, ss); st = fmaf(s, t, st); tt = fmaf(t, t, tt); }
                else { const float d = s - t; ss = fmaf(d, d, ss); }
            }
            term = (MODE == 0) ? (double)(st / (sqrtf(ss) * sqrtf(tt))) : (double)ss;
            cnt = 1.0;
        }
    }
    block_partial(term, cnt, part + ((long)b * gridDim.x + blockIdx.x) * 2);
}

template <int MODE>
__global__ __launch_bounds__(256) void embed_loss_bwd_kernel(const float* __restrict__ score, const int64_t* __restrict__ target,
                                                             const float* __restrict__ embed, const float* __restrict__ tembed,
                                                             const float* __restrict__ stats, const float* __restrict__ gout,
                                                             float* __restrict__ dscore, int B, int E, int HW, int K) {
    extern __shared__ __attribute__((aligned(16))) float emb_l[];
    if (!tembed) {
        for (int i = threadIdx.x; i < K * E; i += 256) emb_l[i] = embed[i];
        __syncthreads();
    }
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const long lbl = target[(long)b * HW + p];
    const float* sp = score + (long)b * E * HW + p;
    float* dp = dscore + (long)b * E * HW + p;
    if (lbl < 0) {
        for (int c = 0; c < E; ++c) dp[(long)c * HW] = 0.f;
        return;
    }
    const float* tp = tembed ? tembed + (long)b * E * HW + p : nullptr;
    const float* er = emb_l + (lbl < K ? lbl : 0) * E;
    const float g = (gout ? gout[0] : 1.f) / ((float)B * stats[2 * b + 1]);
    if (MODE == 0) {
        float ss = 0.f, st = 0.f, tt = 0.f;
        for (int c = 0; c < E; ++c) {
            const float s = sp[(long)c * HW];
            const float t = tp ? tp[(long)c * HW] : er[c];
            ss = fmaf(s, s, ss); st = fmaf(s, t, st); tt = fmaf(t, t, tt);
        }
        const float ns = sqrtf(ss), nt = sqrtf(tt);
        const float cosv = st / (ns * nt);
        const float a = g / (ns * nt);       // coefficient of t_c
        const float bq = g * cosv / ss;      // coefficient of s_c
        for (int c = 0; c < E; ++c) {
            const float s = sp[(long)c * HW];
            const float t = tp ? tp[(long)c * HW] : er[c];
            dp[(long)c * HW] = bq * s - a * t;
        }
    } else {
        for (int c = 0; c < E; ++c) {
            const float s = sp[(long)c * HW];
            const float t = tp ? tp[(long)c * HW] : er[c];
            dp[(long)c * HW] = 2.f * g * (s - t);
        }
    }
}

// ---- cross_entropy2d --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ score, const int64_t* __restrict__ target,
                                                     const float* __restrict__ cweight, double* __restrict__ part,
                                                     int64_t* __restrict__ pred, int C, int HW) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    double term = 0.0, cnt = 0.0;
    if (p < HW) {
        const float* sp = score + (long)b * C * HW + p;
        float mx = sp[0];
        int am = 0;
        for (int c = 1; c < C; ++c) { const float s = sp[(long)c * HW]; if (s > mx) { mx = s; am = c; } }
        if (pred) pred[(long)b * HW + p] = am;
        const long lbl = target[(long)b * HW + p];
        if (lbl >= 0 && lbl < C) {
            float se = 0.f;
            for (int c = 0; c < C; ++c) se += expf(sp[(long)c * HW] - mx);
            term = (double)(-(sp[lbl * (long)HW] - mx - logf(se)));
            if (cweight) term = (double)(cweight[lbl] * (float)term);      // F.nll_loss(weight=): w[target] * nll, fp32 product
            cnt = 1.0;
        }
    }
    block_partial(term, cnt, part + ((long)b * gridDim.x + blockIdx.x) * 2);
}

__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ score, const int64_t* __restrict__ target,
                                                     const float* __restrict__ cweight, const float* __restrict__ stats,
                                                     const float* __restrict__ gout, float* __restrict__ dscore, int B, int C,
                                                     int HW, int size_average) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float* sp = score + (long)b * C * HW + p;
    float* dp = dscore + (long)b * C * HW + p;
    const long lbl = target[(long)b * HW + p];
    if (lbl < 0 || lbl >= C) {
        for (int c = 0; c < C; ++c) dp[(long)c * HW] = 0.f;
        return;
    }
    float g = gout ? gout[0] : 1.f;
    if (size_average) {
        float n = 0.f;
        for (int i = 0; i < B; ++i) n += stats[2 * i + 1];
        g /= n;
    }
    if (cweight) g *= cweight[lbl];
    float mx = sp[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, sp[(long)c * HW]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(sp[(long)c * HW] - mx);
    const float inv = 1.f / se;
    for (int c = 0; c < C; ++c) {
        const float sm = expf(sp[(long)c * HW] - mx) * inv;
        dp[(long)c * HW] = g * (sm - (c == lbl ? 1.f : 0.f));
    }
}

// ---- nearest-class-embedding argmax ---------------------------------------------------------------------------------
// thread = TWO pixels (p, p + 256 of a 512-pixel tile), 2 x KP accumulators in registers; the class matrix sits transposed in LDS
// ([E][KP], 76.8 KB at E = 300, KP = 64) and every 16-B broadcast read of it feeds 8 FMAs (it fed 4 with one pixel per thread:
// the kernel was LDS-issue- and latency-bound at 2 waves per SIMD, 0.7 TB/s).  A block walks several tiles so that the table is
// staged once per block, not once per 256 pixels.  Arithmetic per (pixel, class) is unchanged: one fmaf chain over ascending
// channels, sqrtf, IEEE division -- bit-identical to szo_embed_argmax.  At K = 59 the work is 2*E*K = 35.4 kFLOP per 1.2 KB pixel
// (29.5 FLOP/B): above the fp32-vector ridge of the chip (157 TF / 8 TB/s = 19.6), i.e. VALU-bound; at K = 21 it is HBM-bound.
// running first-index argmax over class chunks: (bs, is) over the seen-only similarities (all classes in mode 0), (bu, iu) over the
// unseen-only ones.  One pass over all K similarities gives both: a zeroed row scores (0 / (sn * 1)) and still competes, exactly like
// utils.py:173-179.  Classes are visited in ascending order and only a strictly larger value replaces the best: first index wins.
struct Best {
    float bs, bu;
    int is, iu;
};
template <int KP>
__device__ __forceinline__ void argmax_update(Best& r, const float (&acc)[KP], float sn, const float* __restrict__ en, int k0, int K,
                                              int mode, uint64_t word) {
    const float zero_sim = 0.f / (sn * 1.f);
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (k0 + k < K) {
            const float sim = acc[k] / (sn * en[k]);
            if (mode == 0) {
                if (k0 + k == 0 || sim > r.bs) { r.bs = sim; r.is = k0 + k; }
            } else {
                const bool un = (word >> (k & 63)) & 1ull;
                const float vs = un ? zero_sim : sim, vu = un ? sim : zero_sim;
                if (k0 + k == 0 || vs > r.bs) { r.bs = vs; r.is = k0 + k; }
                if (k0 + k == 0 || vu > r.bu) { r.bu = vu; r.iu = k0 + k; }
            }
        }
    }
}
__device__ __forceinline__ int argmax_pick(const Best& r, int mode, const ClassBits& unseen, const float* __restrict__ seenmask,
                                           const int64_t* __restrict__ target, int b, long p, int HW) {
    if (mode == 0) return r.is;
    bool take_unseen;
    if (seenmask) {
        const float s0 = seenmask[((long)b * 2 + 0) * HW + p], s1 = seenmask[((long)b * 2 + 1) * HW + p];
        take_unseen = !(s1 > s0);          // argmax over 2 channels == 0  (utils.py:197-198)
    } else {
        take_unseen = in_set(unseen, target[(long)b * HW + p]);   // np.in1d(target, unseen)
    }
    return take_unseen ? r.iu : r.is;
}

// CHUNKS == false: K <= KP, the table is staged once and a block walks several 512-pixel tiles.  CHUNKS == true (K > 64, KP = 64):
// one tile per block, the class matrix goes through LDS in chunks of 64 rows and the two pixels of a thread keep their running best
// across chunks (the score vector of a pixel is re-read per chunk: L2 hits, 1.2 KB per pixel).
template <int KP, bool CHUNKS>
__global__ __launch_bounds__(256) void embed_argmax_kernel(const float* __restrict__ score, const float* __restrict__ embed,
                                                           const float* __restrict__ seenmask,
                                                           const int64_t* __restrict__ target, int64_t* __restrict__ pred,
                                                           int E, int HW, int K, int mode, ClassBits unseen) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // embT [E][KP] | en [KP]
    float* embT = sm;
    float* en = sm + (long)E * KP;
    const int b = blockIdx.y;
    const float* sb = score + (long)b * E * HW;
    auto stage = [&](int k0) {
        for (int i = threadIdx.x; i < E * KP; i += 256) {
            const int k = i % KP, c = i / KP;                    // conflict-free LDS writes; the 70 KB matrix is L2-resident
            embT[i] = (k0 + k < K) ? embed[(long)(k0 + k) * E + c] : 0.f;
        }
        __syncthreads();
        if (threadIdx.x < KP) {
            float s = 0.f;
            for (int c = 0; c < E; ++c) { const float v = embT[c * KP + threadIdx.x]; s = fmaf(v, v, s); }
            const float n = sqrtf(s);
            en[threadIdx.x] = (n == 0.f) ? 1.f : n;     // utils.py:175
        }
        __syncthreads();
    };
    if (!CHUNKS) stage(0);
    for (long p0 = (long)blockIdx.x * 512 + threadIdx.x; CHUNKS ? (p0 < (long)blockIdx.x * 512 + 256) : (p0 < HW);
         p0 += (long)gridDim.x * 512) {
        const long p1 = p0 + 256;
        const bool ok0 = p0 < HW, ok1 = p1 < HW;
        const float* sp0 = sb + (ok0 ? p0 : 0);
        const float* sp1 = sb + (ok1 ? p1 : (ok0 ? p0 : 0));
        Best r0{0.f, 0.f, 0, 0}, r1{0.f, 0.f, 0, 0};
        float sn0 = 0.f, sn1 = 0.f;
        for (int k0 = 0; k0 < (CHUNKS ? K : 1); k0 += KP) {
            if (CHUNKS) {
                __syncthreads();                                 // the previous chunk's readers are done with the table
                stage(k0);
            }
            float acc0[KP], acc1[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) { acc0[k] = 0.f; acc1[k] = 0.f; }
            float ss0 = 0.f, ss1 = 0.f;
            for (int c = 0; c < E; ++c) {
                const float s0 = sp0[(long)c * HW], s1 = sp1[(long)c * HW];
                ss0 = fmaf(s0, s0, ss0);
                ss1 = fmaf(s1, s1, ss1);
                const float4* er = (const float4*)(embT + c * KP);
#pragma unroll
                for (int k4 = 0; k4 < KP / 4; ++k4) {
                    const float4 e = er[k4];
                    acc0[4 * k4 + 0] = fmaf(s0, e.x, acc0[4 * k4 + 0]);
                    acc0[4 * k4 + 1] = fmaf(s0, e.y, acc0[4 * k4 + 1]);
                    acc0[4 * k4 + 2] = fmaf(s0, e.z, acc0[4 * k4 + 2]);
                    acc0[4 * k4 + 3] = fmaf(s0, e.w, acc0[4 * k4 + 3]);
                    acc1[4 * k4 + 0] = fmaf(s1, e.x, acc1[4 * k4 + 0]);
                    acc1[4 * k4 + 1] = fmaf(s1, e.y, acc1[4 * k4 + 1]);
                    acc1[4 * k4 + 2] = fmaf(s1, e.z, acc1[4 * k4 + 2]);
                    acc1[4 * k4 + 3] = fmaf(s1, e.w, acc1[4 * k4 + 3]);
                }
            }
            sn0 = sqrtf(ss0);
            sn1 = sqrtf(ss1);
            const uint64_t word = class_word(unseen, k0 >> 6);
            argmax_update<KP>(r0, acc0, sn0, en, k0, K, mode, word);
            argmax_update<KP>(r1, acc1, sn1, en, k0, K, mode, word);
        }
        if (ok0) pred[(long)b * HW + p0] = argmax_pick(r0, mode, unseen, seenmask, target, b, p0, HW);
        if (ok1) pred[(long)b * HW + p1] = argmax_pick(r1, mode, unseen, seenmask, target, b, p1, HW);
    }
}

// ---- confusion histogram -------------------------------------------------------------------------------
// LDS == true: per-block counters in LDS (nh K^2 x 4 B: K <= 64), merged into the int64 histogram at the end; LDS == false (more
// classes than that): the merged runs go to the global counters directly
template <bool LDS>
__global__ __launch_bounds__(256) void hist_kernel(const int64_t* __restrict__ lt, const int64_t* __restrict__ lp, long n,
                                                   int K, int nh, ClassBits unseen, unsigned long long* __restrict__ hist) {
    extern __shared__ unsigned int hl[];   // [nh][K*K]
    if (LDS) {
        for (int i = threadIdx.x; i < nh * K * K; i += 256) hl[i] = 0;
        __syncthreads();
    }
    // a thread takes 8 consecutive pixels and merges runs of equal (true, predicted) pairs into one atomic: label maps are
    // piecewise constant, and one atomic per pixel on a handful of hot counters serialised the block
    auto flush = [&](int idx, unsigned cnt, long t) {
        if (idx < 0 || !cnt) return;
        const int idx2 = (in_set(unseen, t) ? 2 : 1) * K * K + idx;
        if (LDS) {
            atomicAdd(&hl[idx], cnt);
            if (nh > 1) atomicAdd(&hl[idx2], cnt);
        } else {
            atomicAdd(&hist[idx], (unsigned long long)cnt);
            if (nh > 1) atomicAdd(&hist[idx2], (unsigned long long)cnt);
        }
    };
    for (long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 8; i0 < n; i0 += (long)gridDim.x * 256 * 8) {
        int cur = -1; unsigned cnt = 0; long curt = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long i = i0 + e;
            int idx = -1; long t = 0;
            if (i < n) {
                t = lt[i];
                const long p = lp[i];
                if (t >= 0 && t < K && p >= 0 && p < K) idx = (int)(t * K + p);
            }
            if (idx == cur) ++cnt;
            else { flush(cur, cnt, curt); cur = idx; cnt = 1; curt = t; }
        }
        flush(cur, cnt, curt);
    }
    if (LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < nh * K * K; i += 256)
            if (hl[i]) atomicAdd(&hist[i], (unsigned long long)hl[i]);
    }
}

inline int grid_for(long n, int cap) {
    long b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

int check_up(int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop, const void* a, const void* b, int S = 32) {
    if (B <= 0 || h <= 0 || w <= 0 || E <= 0 || ldc < c0 + E || c0 < 0 || H <= 0 || W <= 0 || crop < 0 || !a || !b)
        SZN_FAIL(SZN_ERR_ARG, "upsample: bad argument");
    if (H + crop > S * h + S || W + crop > S * w + S)
        SZN_FAIL(SZN_ERR_ARG, "upsample: crop window [%d,%d)+%d exceeds the %dx%d deconv output", H, W, crop, S * h + S,
                 S * w + S);
    return SZN_OK;
}

constexpr size_t kMaxDynLds = 160 * 1024 - 2048;

template <int S>
int launch_up_fwd(int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop, const float* coarse, float* score,
                  hipStream_t st) {
    const size_t lds = (size_t)2 * (256 / S + 2) * E * sizeof(float);
    if (lds > kMaxDynLds) SZN_FAIL(SZN_ERR_UNSUPPORTED, "up_fwd: E = %d too large for the LDS tap table", E);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)up_fwd_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    // uncropped columns crop .. crop + W - 1 in 256-wide segments, cell rows (crop / S) .. ((H - 1 + crop) / S)
    const dim3 grid((unsigned)((W + crop + 255) / 256), (unsigned)((H - 1 + crop) / S + 1), (unsigned)B);
    hipLaunchKernelGGL(up_fwd_kernel<S>, grid, dim3(256), lds, st, coarse, score, B, h, w, E, ldc, c0, H, W, crop);
    SZN_CHECK_LAUNCH("up_fwd_kernel");
    return SZN_OK;
}

template <int S>
int launch_up_bwd(int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop, const float* dscore, float* dcoarse,
                  hipStream_t st) {
    // channel slices so that ~2 blocks per CU exist; LDS = 2 x slice x (w + 1) floats
    int csplit = (1536 + B * h - 1) / (B * h);
    if (csplit < 1) csplit = 1;
    if (csplit > E) csplit = E;
    int cper = (E + csplit - 1) / csplit;
    if (cper > 32) cper = 32;            // bounds the LDS footprint (occupancy) when B * h alone already fills the chip (stride 8)
    csplit = (E + cper - 1) / cper;
    const size_t lds = (size_t)2 * cper * (w + 1) * sizeof(float);
    if (lds > kMaxDynLds) SZN_FAIL(SZN_ERR_UNSUPPORTED, "up_bwd: coarse row too wide for LDS (w = %d)", w);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)up_bwd_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(up_bwd_kernel<S>, dim3((unsigned)h, (unsigned)B, (unsigned)csplit), dim3(256), lds, st, dscore, dcoarse, B, h,
                       w, E, ldc, c0, H, W, crop, cper);
    SZN_CHECK_LAUNCH("up_bwd_kernel");
    return SZN_OK;
}

}  // namespace

extern "C" int szn_bilinear_up_crop_fwd(int stride, int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop,
                                        const float* coarse, float* score, szn_stream_t stream) {
    if (stride != 32 && stride != 8) SZN_FAIL(SZN_ERR_UNSUPPORTED, "bilinear_up_crop_fwd: stride %d (32 and 8 are built)", stride);
    int rc = check_up(B, h, w, E, ldc, c0, H, W, crop, coarse, score, stride);
    if (rc) return rc;
    return stride == 32 ? launch_up_fwd<32>(B, h, w, E, ldc, c0, H, W, crop, coarse, score, (hipStream_t)stream)
                        : launch_up_fwd<8>(B, h, w, E, ldc, c0, H, W, crop, coarse, score, (hipStream_t)stream);
}

extern "C" int szn_bilinear_up_crop_bwd(int stride, int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop,
                                        const float* dscore, float* dcoarse, szn_stream_t stream) {
    if (stride != 32 && stride != 8) SZN_FAIL(SZN_ERR_UNSUPPORTED, "bilinear_up_crop_bwd: stride %d (32 and 8 are built)", stride);
    int rc = check_up(B, h, w, E, ldc, c0, H, W, crop, dscore, dcoarse, stride);
    if (rc) return rc;
    return stride == 32 ? launch_up_bwd<32>(B, h, w, E, ldc, c0, H, W, crop, dscore, dcoarse, (hipStream_t)stream)
                        : launch_up_bwd<8>(B, h, w, E, ldc, c0, H, W, crop, dscore, dcoarse, (hipStream_t)stream);
}

extern "C" int szn_bilinear_up32_crop_fwd(int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop,
                                          const float* coarse, float* score, szn_stream_t stream) {
    return szn_bilinear_up_crop_fwd(32, B, h, w, E, ldc, c0, H, W, crop, coarse, score, stream);
}

extern "C" int szn_bilinear_up32_crop_bwd(int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop,
                                          const float* dscore, float* dcoarse, szn_stream_t stream) {
    return szn_bilinear_up_crop_bwd(32, B, h, w, E, ldc, c0, H, W, crop, dscore, dcoarse, stream);
}

extern "C" int szn_bilinear_up2_nhwc_fwd(int B, int h, int w, int C, int ld, const float* in, float* out, szn_stream_t stream) {
    if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || ld < C || !in || !out) SZN_FAIL(SZN_ERR_ARG, "bilinear_up2_nhwc_fwd: bad argument");
    hipLaunchKernelGGL(up2_nhwc_fwd_kernel, dim3(grid_for((long)B * (2 * h + 2) * (2 * w + 2) * C, 1 << 16)), dim3(256), 0,
                       (hipStream_t)stream, in, out, B, h, w, C, ld);
    SZN_CHECK_LAUNCH("up2_nhwc_fwd_kernel");
    return SZN_OK;
}

extern "C" int szn_bilinear_up2_nhwc_bwd(int B, int h, int w, int C, int ld, const float* dout, float* din, szn_stream_t stream) {
    if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || ld < C || !dout || !din) SZN_FAIL(SZN_ERR_ARG, "bilinear_up2_nhwc_bwd: bad argument");
    hipLaunchKernelGGL(up2_nhwc_bwd_kernel, dim3(grid_for((long)B * h * w * C, 1 << 16)), dim3(256), 0, (hipStream_t)stream, dout,
                       din, B, h, w, C, ld);
    SZN_CHECK_LAUNCH("up2_nhwc_bwd_kernel");
    return SZN_OK;
}

extern "C" int szn_deconv64s32_fwd(int B, int h, int w, int C, int ldc, int c0, int H, int W, int crop,
                                   const float* coarse, const float* weight, float* out, szn_stream_t stream) {
    int rc = check_up(B, h, w, C, ldc, c0, H, W, crop, coarse, out);
    if (rc) return rc;
    if (!weight || C > 4) SZN_FAIL(SZN_ERR_ARG, "deconv64s32_fwd: weight missing or C > 4");
    hipLaunchKernelGGL(deconv_fwd_kernel, dim3(grid_for((long)B * C * H * W, 1 << 20)), dim3(256), 0, (hipStream_t)stream,
                       coarse, weight, out, B, h, w, C, ldc, c0, H, W, crop);
    SZN_CHECK_LAUNCH("deconv_fwd_kernel");
    return SZN_OK;
}

extern "C" int szn_deconv64s32_dgrad(int B, int h, int w, int C, int ldc, int c0, int H, int W, int crop,
                                     const float* dout, const float* weight, float* dcoarse, szn_stream_t stream) {
    int rc = check_up(B, h, w, C, ldc, c0, H, W, crop, dout, dcoarse);
    if (rc) return rc;
    if (!weight || C > 4) SZN_FAIL(SZN_ERR_ARG, "deconv64s32_dgrad: weight missing or C > 4");
    const long waves = (long)B * h * w * C;
    hipLaunchKernelGGL(deconv_dgrad_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dout,
                       weight, dcoarse, B, h, w, C, ldc, c0, H, W, crop);
    SZN_CHECK_LAUNCH("deconv_dgrad_kernel");
    return SZN_OK;
}

extern "C" int szn_deconv64s32_wgrad(int B, int h, int w, int C, int ldc, int c0, int H, int W, int crop,
                                     const float* coarse, const float* dout, float* dweight, int accumulate,
                                     szn_stream_t stream) {
    int rc = check_up(B, h, w, C, ldc, c0, H, W, crop, coarse, dout);
    if (rc) return rc;
    if (!dweight || C > 4) SZN_FAIL(SZN_ERR_ARG, "deconv64s32_wgrad: dweight missing or C > 4");
    hipLaunchKernelGGL(deconv_wgrad_kernel, dim3(C * C * 64), dim3(256), 0, (hipStream_t)stream, coarse, dout, dweight, B, h,
                       w, C, ldc, c0, H, W, crop, accumulate);
    SZN_CHECK_LAUNCH("deconv_wgrad_kernel");
    return SZN_OK;
}

extern "C" size_t szn_loss_workspace_bytes(int B, int H, int W) {
    const long nblk = ((long)H * W + 255) / 256;
    return (size_t)B * nblk * 2 * sizeof(double);
}

namespace {
template <int MODE>
int embed_loss_fwd(int B, int E, int H, int W, int K, const float* score, const int64_t* target, const float* embed,
                   const float* tembed, float* loss, float* stats, void* ws, hipStream_t st, const char* name) {
    if (!score || !target || !loss || !stats || !ws || B <= 0 || E <= 0 || H <= 0 || W <= 0)
        SZN_FAIL(SZN_ERR_ARG, "%s: bad argument", name);
    if (!tembed && (!embed || K <= 0)) SZN_FAIL(SZN_ERR_ARG, "%s: need embed[K][E] or target_embed", name);
    const size_t lds = tembed ? 0 : (size_t)K * E * sizeof(float);
    if (lds > kMaxDynLds) SZN_FAIL(SZN_ERR_UNSUPPORTED, "%s: K*E*4 = %zu B exceeds the LDS budget", name, lds);
    const int HW = H * W, nblk = (HW + 255) / 256;
    auto kern = embed_loss_fwd_kernel<MODE>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(nblk, B), dim3(256), lds, st, score, target, embed, tembed, (double*)ws, E, HW, K);
    SZN_CHECK_LAUNCH(name);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, B, nblk, MODE, 0, loss, stats);
    SZN_CHECK_LAUNCH("loss_finalize_kernel");
    return SZN_OK;
}
template <int MODE>
int embed_loss_bwd(int B, int E, int H, int W, int K, const float* score, const int64_t* target, const float* embed,
                   const float* tembed, const float* stats, const float* gout, float* dscore, hipStream_t st,
                   const char* name) {
    if (!score || !target || !stats || !dscore || B <= 0 || E <= 0 || H <= 0 || W <= 0)
        SZN_FAIL(SZN_ERR_ARG, "%s: bad argument", name);
    if (!tembed && (!embed || K <= 0)) SZN_FAIL(SZN_ERR_ARG, "%s: need embed[K][E] or target_embed", name);
    const size_t lds = tembed ? 0 : (size_t)K * E * sizeof(float);
    if (lds > kMaxDynLds) SZN_FAIL(SZN_ERR_UNSUPPORTED, "%s: K*E*4 = %zu B exceeds the LDS budget", name, lds);
    const int HW = H * W, nblk = (HW + 255) / 256;
    auto kern = embed_loss_bwd_kernel<MODE>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(nblk, B), dim3(256), lds, st, score, target, embed, tembed, stats, gout, dscore, B, E, HW, K);
    SZN_CHECK_LAUNCH(name);
    return SZN_OK;
}
}  // namespace

extern "C" int szn_cosine_loss_fwd(int B, int E, int H, int W, int K, const float* score, const int64_t* target,
                                   const float* embed, const float* target_embed, float* loss, float* stats, void* ws,
                                   szn_stream_t stream) {
    return embed_loss_fwd<0>(B, E, H, W, K, score, target, embed, target_embed, loss, stats, ws, (hipStream_t)stream,
                             "cosine_loss_fwd");
}
extern "C" int szn_cosine_loss_bwd(int B, int E, int H, int W, int K, const float* score, const int64_t* target,
                                   const float* embed, const float* target_embed, const float* stats, const float* gout,
                                   float* dscore, szn_stream_t stream) {
    return embed_loss_bwd<0>(B, E, H, W, K, score, target, embed, target_embed, stats, gout, dscore, (hipStream_t)stream,
                             "cosine_loss_bwd");
}
extern "C" int szn_mse_loss_fwd(int B, int E, int H, int W, int K, const float* score, const int64_t* target,
                                const float* embed, const float* target_embed, float* loss, float* stats, void* ws,
                                szn_stream_t stream) {
    return embed_loss_fwd<1>(B, E, H, W, K, score, target, embed, target_embed, loss, stats, ws, (hipStream_t)stream,
                             "mse_loss_fwd");
}
extern "C" int szn_mse_loss_bwd(int B, int E, int H, int W, int K, const float* score, const int64_t* target,
                                const float* embed, const float* target_embed, const float* stats, const float* gout,
                                float* dscore, szn_stream_t stream) {
    return embed_loss_bwd<1>(B, E, H, W, K, score, target, embed, target_embed, stats, gout, dscore, (hipStream_t)stream,
                             "mse_loss_bwd");
}

extern "C" int szn_ce2d_fwd(int B, int C, int H, int W, const float* score, const int64_t* target, const float* weight,
                            int size_average, float* loss, float* stats, int64_t* pred, void* ws, szn_stream_t stream) {
    if (!score || !target || !loss || !stats || !ws || B <= 0 || C <= 0 || H <= 0 || W <= 0)
        SZN_FAIL(SZN_ERR_ARG, "ce2d_fwd: bad argument");
    const int HW = H * W, nblk = (HW + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(nblk, B), dim3(256), 0, st, score, target, weight, (double*)ws, pred, C, HW);
    SZN_CHECK_LAUNCH("ce_fwd_kernel");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, B, nblk, 2, size_average, loss,
                       stats);
    SZN_CHECK_LAUNCH("loss_finalize_kernel");
    return SZN_OK;
}

extern "C" int szn_ce2d_bwd(int B, int C, int H, int W, const float* score, const int64_t* target, const float* weight,
                            int size_average, const float* stats, const float* gout, float* dscore, szn_stream_t stream) {
    if (!score || !target || !stats || !dscore || B <= 0 || C <= 0 || H <= 0 || W <= 0)
        SZN_FAIL(SZN_ERR_ARG, "ce2d_bwd: bad argument");
    const int HW = H * W, nblk = (HW + 255) / 256;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, score, target, weight, stats, gout, dscore, B,
                       C, HW, size_average);
    SZN_CHECK_LAUNCH("ce_bwd_kernel");
    return SZN_OK;
}

namespace {
int embed_argmax_impl(int B, int E, int H, int W, int K, const float* score, const float* embed, int mode, const ClassBits& unseen,
                      const float* seenmask, const int64_t* target, int64_t* pred, szn_stream_t stream) {
    if (!score || !embed || !pred || B <= 0 || E <= 0 || H <= 0 || W <= 0 || K <= 0)
        SZN_FAIL(SZN_ERR_ARG, "embed_argmax: bad argument");
    if (K > SZN_MAX_CLASSES) SZN_FAIL(SZN_ERR_UNSUPPORTED, "embed_argmax: K=%d > %d", K, SZN_MAX_CLASSES);
    if (mode != 0 && mode != 1) SZN_FAIL(SZN_ERR_ARG, "embed_argmax: bad mode %d", mode);
    if (mode == 1 && !seenmask && !target) SZN_FAIL(SZN_ERR_ARG, "embed_argmax: mode 1 needs seenmask or target");
    if (!class_bits_fit(unseen, K)) SZN_FAIL(SZN_ERR_ARG, "embed_argmax: the unseen set names a class >= K = %d", K);
    const int HW = H * W;
    // 512-pixel tiles, ~2 blocks per CU in total (the LDS table allows two resident blocks): a block stages the table once.
    // K > 64: one tile per block (the table is re-staged per 64-class chunk, the running best stays in registers)
    const bool chunks = K > 64;
    int nblk = (HW + 511) / 512;
    const int cap = (512 + B - 1) / B;
    if (!chunks && nblk > cap) nblk = cap;
    const int KP = K <= 24 ? 24 : (K <= 40 ? 40 : 64);
    const size_t lds = ((size_t)E * KP + KP) * sizeof(float);
    if (lds > kMaxDynLds) SZN_FAIL(SZN_ERR_UNSUPPORTED, "embed_argmax: E*KP*4 = %zu B exceeds the LDS budget", lds);
    hipStream_t st = (hipStream_t)stream;
#define SZN_LAUNCH_AM(KPV, CH)                                                                                         \
    do {                                                                                                               \
        auto kern = embed_argmax_kernel<KPV, CH>;                                                                      \
        if (lds > 48 * 1024)                                                                                           \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
        hipLaunchKernelGGL(kern, dim3(nblk, B), dim3(256), lds, st, score, embed, seenmask, target, pred, E, HW, K,    \
                           mode, unseen);                                                                              \
    } while (0)
    if (chunks) SZN_LAUNCH_AM(64, true);
    else if (KP == 24) SZN_LAUNCH_AM(24, false);
    else if (KP == 40) SZN_LAUNCH_AM(40, false);
    else SZN_LAUNCH_AM(64, false);
#undef SZN_LAUNCH_AM
    SZN_CHECK_LAUNCH(chunks ? "embed_argmax_kernel_chunks" : "embed_argmax_kernel");
    return SZN_OK;
}

int confusion_hist_impl(long npix, int K, const int64_t* label_true, const int64_t* label_pred, const ClassBits& unseen,
                        int64_t* hist, szn_stream_t stream) {
    if (!label_true || !label_pred || !hist || npix <= 0 || K <= 0 || K > SZN_MAX_CLASSES)
        SZN_FAIL(SZN_ERR_ARG, "confusion_hist: bad argument");
    if (!class_bits_fit(unseen, K)) SZN_FAIL(SZN_ERR_ARG, "confusion_hist: the unseen set names a class >= K = %d", K);
    const int nh = class_bits_any(unseen) ? 3 : 1;
    const dim3 grid(grid_for((npix + 7) / 8, 256));
    if (K <= 64) {
        const size_t lds = (size_t)nh * K * K * sizeof(unsigned int);
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)hist_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(hist_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, label_true, label_pred, npix, K, nh, unseen,
                           (unsigned long long*)hist);
        SZN_CHECK_LAUNCH("hist_kernel");
    } else {
        hipLaunchKernelGGL(hist_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, label_true, label_pred, npix, K, nh, unseen,
                           (unsigned long long*)hist);
        SZN_CHECK_LAUNCH("hist_kernel_global");
    }
    return SZN_OK;
}
}  // namespace

extern "C" int szn_embed_argmax(int B, int E, int H, int W, int K, const float* score, const float* embed, int mode,
                                uint64_t unseen_bits, const float* seenmask, const int64_t* target, int64_t* pred,
                                szn_stream_t stream) {
    if (K > 64) SZN_FAIL(SZN_ERR_UNSUPPORTED, "embed_argmax: K=%d > 64 needs szn_embed_argmax_k (szn_class_set)", K);
    return embed_argmax_impl(B, E, H, W, K, score, embed, mode, class_bits64(unseen_bits), seenmask, target, pred, stream);
}

extern "C" int szn_embed_argmax_k(int B, int E, int H, int W, int K, const float* score, const float* embed, int mode,
                                  const szn_class_set* unseen, const float* seenmask, const int64_t* target, int64_t* pred,
                                  szn_stream_t stream) {
    return embed_argmax_impl(B, E, H, W, K, score, embed, mode, class_bits(unseen), seenmask, target, pred, stream);
}

extern "C" int szn_confusion_hist(long npix, int K, const int64_t* label_true, const int64_t* label_pred,
                                  uint64_t unseen_bits, int64_t* hist, szn_stream_t stream) {
    if (K > 64) SZN_FAIL(SZN_ERR_ARG, "confusion_hist: K=%d > 64 needs szn_confusion_hist_k (szn_class_set)", K);
    return confusion_hist_impl(npix, K, label_true, label_pred, class_bits64(unseen_bits), hist, stream);
}

extern "C" int szn_confusion_hist_k(long npix, int K, const int64_t* label_true, const int64_t* label_pred,
                                    const szn_class_set* unseen, int64_t* hist, szn_stream_t stream) {
    return confusion_hist_impl(npix, K, label_true, label_pred, class_bits(unseen), hist, stream);
}
