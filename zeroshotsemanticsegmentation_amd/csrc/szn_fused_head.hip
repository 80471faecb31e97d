// szn_fused_head.hip -- the SZN head evaluated from the 1/32-resolution projection map without ever
// materialising the (B,E,H,W) score: bilinear x32 upsample + crop (models.py:146-147), cosine loss
// (utils.py:75-102), nearest-class-embedding argmax (utils.py:159-185) and the gradient back to the
// coarse map, per 32x32 output cell.  The same kernels are instantiated for stride 8 (8x8 cells of the 1/8 map: the last
// stage of the FCN8s skip head, upscore8 + crop 31).
//
// Inside one cell (Y / S, X / S fixed) every pixel's score vector is a blend of the SAME four coarse
// vectors C_t with per-pixel bilinear weights w_t, so
//     s . e_k = sum_t w_t (C_t . e_k)            -> per-cell table G[4][K]
//     |s|^2   = sum_{t,t'} w_t w_t' (C_t . C_t') -> per-cell Gram matrix Q[4][4]
// and the gradient wrt the coarse vectors collapses to
//     dC_t = -sum_k A[t][k] e_k + sum_t' Bm[t][t'] C_t',  A[t][k] = sum_{px: label k} w_t / (|s||e_k|),
//                                                         Bm[t][t'] = sum_px w_t w_t' cos / |s|^2
// (all scaled by 1/(B N_b)).  HBM traffic: labels in, prediction out (16 B/px) + the coarse map.
//
// Kernel 1 (cell kernel): one block per (image, cell): builds G, Q, walks its <= 1024 pixels, writes pred,
//   the cell's loss partial, A and Bm.  All reductions are fixed-order (bit-reproducible).
// Kernel 2: loss_finalize (per-image sums, fixed order).  Kernel 3 (gather): one block per coarse
//   position sums the contributions of the <= 4 cells that use it as a tap and writes dcoarse.
#include "szn_common.h"

namespace {

template <int S>
__device__ __forceinline__ double bil1d(int t) { return 1.0 - fabs((double)t - ((double)S - 0.5)) / (double)S; }

struct FhArgs {
    const float* coarse; const float* embed; const int64_t* target;
    int64_t* pred; float* ws_f; double* part;
    int B, h, w, E, ldc, c0, H, W, crop, K, KP;
};

// workspace (floats): embT [E][KP] | en [KP] (0 -> 1, for the argmax) | ent [KP] (raw norms, for the loss)
//                     | per cell: A [4][KP] , Bm [16]
__host__ __device__ inline size_t ws_cell_off(int E, int KP) { return (size_t)E * KP + 2 * KP; }
__host__ __device__ inline size_t ws_cell_stride(int KP) { return (size_t)4 * KP + 16; }

__global__ __launch_bounds__(256) void fh_prep_kernel(const float* __restrict__ embed, float* __restrict__ ws, int E,
                                                      int K, int KP) {
    float* embT = ws;
    float* en = ws + (size_t)E * KP;
    float* ent = en + KP;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < E * KP; i += gridDim.x * 256) {
        const int k = i % KP, c = i / KP;
        embT[i] = (k < K) ? embed[(size_t)k * E + c] : 0.f;
    }
    if (blockIdx.x == 0) {
        // norms: thread k walks its row in ascending order (the order is part of the arithmetic contract with the oracle).  The
        // rows are staged through LDS in slices first: as a chain of dependent global loads the 300 steps took ~25 us.
        __shared__ float row[64][65];
        float s = 0.f;
        const int k = threadIdx.x;                 // class k (K <= 256 = the block)
        for (int g0 = 0; g0 < K; g0 += 64)         // 64 classes at a time through the LDS tile
            for (int c0 = 0; c0 < E; c0 += 64) {
                const int n = min(64, E - c0);
                __syncthreads();
                for (int i = threadIdx.x; i < 64 * 64; i += 256) {
                    const int kk = i >> 6, c = i & 63;
                    row[kk][c] = (g0 + kk < K && c < n) ? embed[(size_t)(g0 + kk) * E + c0 + c] : 0.f;
                }
                __syncthreads();
                if (k < K && (k >> 6) == (g0 >> 6))
                    for (int c = 0; c < n; ++c) s = fmaf(row[k & 63][c], row[k & 63][c], s);
            }
        if (k < KP) {
            const float nrm = sqrtf(s);
            en[k] = (nrm == 0.f) ? 1.f : nrm;
            ent[k] = nrm;
        }
    }
}

template <int KP, int S>
__global__ __launch_bounds__(256) void fh_cell_kernel(FhArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Ct = sm;                       // [4][E]
    float* G = Ct + 4 * a.E;              // [4][KP]
    float* Q = G + 4 * KP;                // [16]
    float* Aw = Q + 16;                   // [4 waves][4][KP]
    float* red = Aw + 16 * KP;            // [4 waves][16]
    double* dred = (double*)(red + 64);   // [4 waves][2]   (offset is a multiple of 8 B: all terms are multiples of 4 floats... see host check)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cells_w = a.w + 1, cells = (a.h + 1) * cells_w;
    const int b = blockIdx.x / cells, cell = blockIdx.x % cells;
    const int I = cell / cells_w, J = cell % cells_w;

    const float* embT = a.ws_f;
    const float* en = a.ws_f + (size_t)a.E * KP;
    const float* ent = en + KP;

    // ---- the four tap vectors (missing taps are zero) ----
    for (int i = tid; i < 4 * a.E; i += 256) {
        const int t = i / a.E, c = i - t * a.E;
        const int ci = I - 1 + (t >> 1), cj = J - 1 + (t & 1);
        float v = 0.f;
        if (ci >= 0 && ci < a.h && cj >= 0 && cj < a.w)
            v = a.coarse[(((size_t)b * a.h + ci) * a.w + cj) * a.ldc + a.c0 + c];
        Ct[i] = v;
    }
    for (int i = tid; i < 16 * KP; i += 256) Aw[i] = 0.f;
    __syncthreads();
    // G[t][k]: wave t, lane k (+ 64, + 128, + 192 when KP > 64)
    for (int k = lane; k < KP; k += 64) {
        float g = 0.f;
        const float* ct = Ct + wave * a.E;
        // (the chain is sequential by contract; 20 independent L2 loads per batch keep it fed)
        int c = 0;
        for (; c + 20 <= a.E; c += 20) {
            float ev[20];
#pragma unroll
            for (int u = 0; u < 20; ++u) ev[u] = embT[(size_t)(c + u) * KP + k];
#pragma unroll
            for (int u = 0; u < 20; ++u) g = fmaf(ct[c + u], ev[u], g);
        }
        for (; c < a.E; ++c) g = fmaf(ct[c], embT[(size_t)c * KP + k], g);
        G[wave * KP + k] = g;
    }
    // Q[t][t']: wave t computes its row; lanes stride over c, fixed-order wave reduction
    {
        float q[4] = {0.f, 0.f, 0.f, 0.f};
        const float* ct = Ct + wave * a.E;
        for (int c = lane; c < a.E; c += 64) {
            const float v = ct[c];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = fmaf(v, Ct[u * a.E + c], q[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float s = wave_sum(q[u]);
            if (lane == 0) Q[wave * 4 + u] = s;
        }
    }
    __syncthreads();

    // ---- pixels of this cell: Y in [S I, S I + S) x X in [S J, S J + S), image coords y = Y - crop ----
    float bm[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) bm[u] = 0.f;
    double cos_sum = 0.0, cnt = 0.0;
    float* myA = Aw + wave * 4 * KP;
    for (int q = tid; q < S * S; q += 256) {         // S = 32: a wave covers 2 rows of the cell; S = 8: wave 0 holds the whole cell
        const int ty = q / S, tx = q % S;
        const int y = S * I + ty - a.crop, x = S * J + tx - a.crop;
        const bool inside = (y >= 0 && y < a.H && x >= 0 && x < a.W);
        long lbl = -1;
        float wt[4] = {0.f, 0.f, 0.f, 0.f};
        float aco = 0.f;
        if (inside) {
            const double fy1 = bil1d<S>(ty), fy0 = bil1d<S>(ty + S), fx1 = bil1d<S>(tx), fx0 = bil1d<S>(tx + S);
            wt[0] = (float)(fy0 * fx0); wt[1] = (float)(fy0 * fx1); wt[2] = (float)(fy1 * fx0); wt[3] = (float)(fy1 * fx1);
            float ss = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) ss = fmaf(wt[t] * wt[u], Q[t * 4 + u], ss);
            const float sn = sqrtf(ss);
            const size_t pix = ((size_t)b * a.H + y) * a.W + x;
            lbl = a.target ? a.target[pix] : -1;
            if (a.pred) {
                int best = 0;
                float bv = 0.f;
                for (int k = 0; k < a.K; ++k) {
                    float d = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) d = fmaf(wt[t], G[t * KP + k], d);
                    const float sim = d / (sn * en[k]);
                    if (k == 0 || sim > bv) { bv = sim; best = k; }
                }
                a.pred[pix] = best;
            }
            if (lbl >= 0) {
                const int kl = lbl < a.K ? (int)lbl : 0;
                float d = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) d = fmaf(wt[t], G[t * KP + kl], d);
                const float nt = ent[kl];
                const float cosv = d / (sn * nt);
                cos_sum += (double)cosv;
                cnt += 1.0;
                aco = 1.f / (sn * nt);
                const float bco = cosv / ss;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u) bm[t * 4 + u] = fmaf(wt[t] * wt[u], bco, bm[t * 4 + u]);
            }
        }
        // A[t][label] += w_t * aco, label by label in a fixed order (wave-uniform loop)
        unsigned long long todo = __ballot(lbl >= 0);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            const int kl = (int)__shfl((int)(lbl < a.K ? lbl : 0), src, 64);
            const bool mine = (lbl >= 0) && ((int)(lbl < a.K ? lbl : 0) == kl);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float s = wave_sum(mine ? wt[t] * aco : 0.f);
                if (lane == 0) myA[t * KP + kl] += s;
            }
            todo &= ~__ballot(mine);
        }
    }
    // ---- block reductions (fixed order) ----
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const float s = wave_sum(bm[u]);
        if (lane == 0) red[wave * 16 + u] = s;
    }
    cos_sum = wave_sum_d(cos_sum); cnt = wave_sum_d(cnt);
    if (lane == 0) { dred[wave * 2] = cos_sum; dred[wave * 2 + 1] = cnt; }
    __syncthreads();
    float* wc = a.ws_f + ws_cell_off(a.E, KP) + (size_t)blockIdx.x * ws_cell_stride(KP);
    for (int i = tid; i < 4 * KP; i += 256) wc[i] = (Aw[i] + Aw[4 * KP + i]) + (Aw[8 * KP + i] + Aw[12 * KP + i]);
    if (tid < 16) wc[4 * KP + tid] = (red[tid] + red[16 + tid]) + (red[32 + tid] + red[48 + tid]);
    if (tid == 0) {
        a.part[(size_t)blockIdx.x * 2] = (dred[0] + dred[2]) + (dred[4] + dred[6]);
        a.part[(size_t)blockIdx.x * 2 + 1] = (dred[1] + dred[3]) + (dred[5] + dred[7]);
    }
}

// ---- small cells (stride 8): per-POSITION tables instead of per-cell dot products -----------------------------------------------
// With 8x8 cells a block per cell would spend its time re-deriving G and Q (each coarse vector is a tap of 4 cells).  Both only
// depend on coarse positions:  D[pos][k] = C_pos . e_k,  N[pos][n] = C_pos . C_nbr for nbr in {self, E, S, SE, SW}; a cell's
// G[t][k] / Q[t][u] are lookups.  Same chains as fh_cell_kernel (ascending fmaf for D; 64 strided partial chains + xor butterfly
// for N), so the two paths -- and the CPU restatement -- agree bit for bit.
template <int KP>
__global__ __launch_bounds__(256) void fh_tables_kernel(FhArgs a, float* __restrict__ D, float* __restrict__ N) {
    extern __shared__ __attribute__((aligned(16))) float sm[];            // [4 waves][E]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long npos = (long)a.B * a.h * a.w;
    const long pos = (long)blockIdx.x * 4 + wave;
    const bool ok = pos < npos;
    const long p = ok ? pos : 0;
    const int b = (int)(p / (a.h * a.w)), r = (int)(p % (a.h * a.w));
    const int i = r / a.w, j = r % a.w;
    const float* cv = a.coarse + (size_t)p * a.ldc + a.c0;
    float* Cs = sm + wave * a.E;
    for (int c = lane; c < a.E; c += 64) Cs[c] = cv[c];
    __syncthreads();
    const float* embT = a.ws_f;
    for (int k = lane; ok && k < KP; k += 64) {
        float g = 0.f;
        for (int c = 0; c < a.E; ++c) g = fmaf(Cs[c], embT[(size_t)c * KP + k], g);
        D[(size_t)pos * KP + k] = g;
    }
    const int di[5] = {0, 0, 1, 1, 1}, dj[5] = {0, 1, 0, 1, -1};
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        const int ni = i + di[n], nj = j + dj[n];
        float q = 0.f;
        if (ni < a.h && nj >= 0 && nj < a.w) {
            const float* nv = a.coarse + (((size_t)b * a.h + ni) * a.w + nj) * a.ldc + a.c0;
            for (int c = lane; c < a.E; c += 64) q = fmaf(Cs[c], nv[c], q);
        }
        q = wave_sum(q);
        if (ok && lane == 0) N[(size_t)pos * 8 + n] = q;
    }
}

// one wave per cell (S * S <= 64 pixels), four cells per block; same per-pixel arithmetic and the same outputs as fh_cell_kernel
template <int KP, int S>
__global__ __launch_bounds__(256) void fh_cell_tab_kernel(FhArgs a, const float* __restrict__ D, const float* __restrict__ N) {
    static_assert(S * S <= 64, "one wave per cell");
    __shared__ float Gs[4][4 * KP];
    __shared__ float Qs[4][16];
    __shared__ float As[4][4 * KP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cells_w = a.w + 1, cells = (a.h + 1) * cells_w;
    const long ncell = (long)a.B * cells;
    const long cid = (long)blockIdx.x * 4 + wave;
    const bool ok = cid < ncell;
    const long cc = ok ? cid : 0;
    const int b = (int)(cc / cells), cell = (int)(cc % cells);
    const int I = cell / cells_w, J = cell % cells_w;
    const float* en = a.ws_f + (size_t)a.E * KP;
    const float* ent = en + KP;
    float* G = Gs[wave];
    float* Q = Qs[wave];
    float* myA = As[wave];
    long tp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ci = I - 1 + (t >> 1), cj = J - 1 + (t & 1);
        tp[t] = (ci >= 0 && ci < a.h && cj >= 0 && cj < a.w) ? ((long)b * a.h + ci) * a.w + cj : -1;
    }
    for (int k = lane; k < KP; k += 64) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            G[t * KP + k] = (tp[t] >= 0) ? D[(size_t)tp[t] * KP + k] : 0.f;
            myA[t * KP + k] = 0.f;
        }
    }
    if (lane < 16) {
        const int t = min(lane >> 2, lane & 3), u = max(lane >> 2, lane & 3);
        // (t, u) -> neighbour slot of the LOWER tap: self, E, S, SE | (1,2) SW, (1,3) S | (2,3) E
        const int slot = (t == u) ? 0 : (t == 0 ? u : (t == 1 ? (u == 2 ? 4 : 2) : 1));
        const long pt = (t == 0) ? tp[0] : (t == 1) ? tp[1] : (t == 2) ? tp[2] : tp[3];
        const long pu = (u == 0) ? tp[0] : (u == 1) ? tp[1] : (u == 2) ? tp[2] : tp[3];
        Q[lane] = (pt >= 0 && pu >= 0) ? N[(size_t)pt * 8 + slot] : 0.f;
    }
    __syncthreads();

    float bm[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) bm[u] = 0.f;
    double cos_sum = 0.0, cnt = 0.0;
    {
        const int q = lane;
        const int ty = q / S, tx = q % S;
        const int y = S * I + ty - a.crop, x = S * J + tx - a.crop;
        const bool inside = ok && q < S * S && (y >= 0 && y < a.H && x >= 0 && x < a.W);
        long lbl = -1;
        float wt[4] = {0.f, 0.f, 0.f, 0.f};
        float aco = 0.f;
        if (inside) {
            const double fy1 = bil1d<S>(ty), fy0 = bil1d<S>(ty + S), fx1 = bil1d<S>(tx), fx0 = bil1d<S>(tx + S);
            wt[0] = (float)(fy0 * fx0); wt[1] = (float)(fy0 * fx1); wt[2] = (float)(fy1 * fx0); wt[3] = (float)(fy1 * fx1);
            float ss = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) ss = fmaf(wt[t] * wt[u], Q[t * 4 + u], ss);
            const float sn = sqrtf(ss);
            const size_t pix = ((size_t)b * a.H + y) * a.W + x;
            lbl = a.target ? a.target[pix] : -1;
            if (a.pred) {
                int best = 0;
                float bv = 0.f;
                for (int k = 0; k < a.K; ++k) {
                    float d = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t) d = fmaf(wt[t], G[t * KP + k], d);
                    const float sim = d / (sn * en[k]);
                    if (k == 0 || sim > bv) { bv = sim; best = k; }
                }
                a.pred[pix] = best;
            }
            if (lbl >= 0) {
                const int kl = lbl < a.K ? (int)lbl : 0;
                float d = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) d = fmaf(wt[t], G[t * KP + kl], d);
                const float nt = ent[kl];
                const float cosv = d / (sn * nt);
                cos_sum += (double)cosv;
                cnt += 1.0;
                aco = 1.f / (sn * nt);
                const float bco = cosv / ss;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u) bm[t * 4 + u] = fmaf(wt[t] * wt[u], bco, bm[t * 4 + u]);
            }
        }
        unsigned long long todo = __ballot(lbl >= 0);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            const int kl = (int)__shfl((int)(lbl < a.K ? lbl : 0), src, 64);
            const bool mine = (lbl >= 0) && ((int)(lbl < a.K ? lbl : 0) == kl);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float s = wave_sum(mine ? wt[t] * aco : 0.f);
                if (lane == 0) myA[t * KP + kl] += s;
            }
            todo &= ~__ballot(mine);
        }
    }
    float bs[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) bs[u] = wave_sum(bm[u]);
    cos_sum = wave_sum_d(cos_sum); cnt = wave_sum_d(cnt);
    __syncthreads();
    if (!ok) return;
    float* wc = a.ws_f + ws_cell_off(a.E, KP) + (size_t)cid * ws_cell_stride(KP);
    for (int i = lane; i < 4 * KP; i += 64) wc[i] = myA[i];
    if (lane == 0) {
#pragma unroll
        for (int u = 0; u < 16; ++u) wc[4 * KP + u] = bs[u];
        a.part[(size_t)cid * 2] = cos_sum;
        a.part[(size_t)cid * 2 + 1] = cnt;
    }
}

// loss = mean_b (N_b - S_b)/N_b ; stats[b] = {S_b, N_b}.  One block per image sums its cells (fixed order: 256 strided double
// chains, wave butterflies, then the four wave totals), a single wave combines the images.
__global__ __launch_bounds__(256) void fh_image_sums_kernel(const double* __restrict__ part, int cells, float* __restrict__ stats,
                                                            double* __restrict__ sums) {
    __shared__ double red[4][2];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0, n = 0.0;
    for (int k = threadIdx.x; k < cells; k += 256) { s += part[((size_t)b * cells + k) * 2]; n += part[((size_t)b * cells + k) * 2 + 1]; }
    s = wave_sum_d(s); n = wave_sum_d(n);
    if (lane == 0) { red[wave][0] = s; red[wave][1] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double st = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]), nt = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        sums[2 * b] = st; sums[2 * b + 1] = nt;
        stats[2 * b] = (float)st; stats[2 * b + 1] = (float)nt;
    }
}

__global__ void fh_finalize_kernel(const double* __restrict__ sums, int B, float* __restrict__ loss) {
    if (threadIdx.x == 0) {
        double acc = 0.0;
        for (int b = 0; b < B; ++b) acc += (sums[2 * b + 1] - sums[2 * b]) / sums[2 * b + 1];
        loss[0] = (float)(acc / B);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fh_gather_kernel(const float* __restrict__ coarse, const float* __restrict__ embed,
                                                        const float* __restrict__ ws, const float* __restrict__ stats,
                                                        T* __restrict__ dcoarse, int B, int h, int w, int E, int ldc,
                                                        int c0, int K, int KP) {
    __shared__ float Al[4][256];     // A of the 4 cells (KP <= 256), row = the tap index this position has in that cell
    __shared__ float Bl[4][4];       // matching rows of Bm
    const int pos = blockIdx.x;
    const int b = pos / (h * w), r = pos % (h * w);
    const int i = r / w, j = r % w;
    const int cells_w = w + 1, cells = (h + 1) * cells_w;
    const float* cellbase = ws + ws_cell_off(E, KP);
    // cell u = (a, bb): this position is tap t = (a, bb) of cell (i + 1 - a, j + 1 - bb)
    for (int idx = threadIdx.x; idx < 4 * KP + 16; idx += 256) {
        if (idx < 4 * KP) {
            const int u = idx / KP, k = idx % KP;
            const int I = i + 1 - (u >> 1), J = j + 1 - (u & 1);
            const float* wc = cellbase + ((size_t)b * cells + I * cells_w + J) * ws_cell_stride(KP);
            Al[u][k] = wc[u * KP + k];
        } else {
            const int q = idx - 4 * KP, u = q >> 2, t2 = q & 3;
            const int I = i + 1 - (u >> 1), J = j + 1 - (u & 1);
            const float* wc = cellbase + ((size_t)b * cells + I * cells_w + J) * ws_cell_stride(KP);
            Bl[u][t2] = wc[4 * KP + u * 4 + t2];
        }
    }
    __syncthreads();
    // the class term is linear in A: sum the four cells' rows first, one pass over the K embeddings instead of four
    Al[0][threadIdx.x] = (threadIdx.x < KP) ? (Al[0][threadIdx.x] + Al[1][threadIdx.x]) + (Al[2][threadIdx.x] + Al[3][threadIdx.x]) : 0.f;
    __syncthreads();
    const float scale = 1.f / ((float)B * stats[2 * b + 1]);
    for (int c = threadIdx.x; c < E; c += 256) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(-Al[0][k], embed[(size_t)k * E + c], acc);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float bu = 0.f;
            const int I = i + 1 - (u >> 1), J = j + 1 - (u & 1);
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) {
                const int ci = I - 1 + (t2 >> 1), cj = J - 1 + (t2 & 1);
                if (ci >= 0 && ci < h && cj >= 0 && cj < w)
                    bu = fmaf(Bl[u][t2], coarse[(((size_t)b * h + ci) * w + cj) * ldc + c0 + c], bu);
            }
            acc += bu;
        }
        elem<T>::st(dcoarse + ((size_t)pos) * ldc + c0 + c, acc * scale);
    }
}

inline int kp_of(int K) { return K <= 24 ? 24 : (K <= 40 ? 40 : (K <= 64 ? 64 : (K + 63) / 64 * 64)); }      // K <= 256

}  // namespace

extern "C" size_t szn_fused_head_workspace_bytes(int B, int h, int w, int E, int K) {
    if (B <= 0 || h <= 0 || w <= 0 || E <= 0 || K <= 0 || K > 256) return 0;
    const int KP = kp_of(K);
    const size_t cells = (size_t)B * (h + 1) * (w + 1);
    size_t fl = ws_cell_off(E, KP) + cells * ws_cell_stride(KP);
    fl = (fl + 1) / 2 * 2;                                   // keep the double region 8-B aligned
    // + the per-position tables of the small-cell path (D [pos][KP], N [pos][8])
    return fl * sizeof(float) + (cells + B) * 2 * sizeof(double) + (size_t)B * h * w * (KP + 8) * sizeof(float);
}

static int fused_head_impl(int stride, int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop, int K,
                           const float* coarse, const float* embed, const int64_t* target, float* loss,
                           float* stats, int64_t* pred, int dcoarse_dtype, void* dcoarse, void* workspace,
                           szn_stream_t stream, bool prep);

extern "C" int szn_fused_head_strided(int stride, int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop, int K,
                                      const float* coarse, const float* embed, const int64_t* target, float* loss,
                                      float* stats, int64_t* pred, int dcoarse_dtype, void* dcoarse, void* workspace,
                                      szn_stream_t stream) {
    return fused_head_impl(stride, B, h, w, E, ldc, c0, H, W, crop, K, coarse, embed, target, loss, stats, pred, dcoarse_dtype, dcoarse, workspace,
                           stream, true);
}

// The class embeddings are constants of a training run (trainer_fcn.py:49-62 loads them once): their transpose and norms -- fh_prep_kernel, 23 us of
// a 2.5-8 ms step, a chain of dependent loads -- need not be rebuilt every step.  szn_fused_head_prepare writes them to the head of `workspace`
// once; szn_fused_head_prepared is szn_fused_head_strided without that launch, for a caller that keeps the workspace and re-prepares when the
// embeddings (or the workspace) change.  Same tables, same bits.
extern "C" int szn_fused_head_prepare(int E, int K, const float* embed, void* workspace, szn_stream_t stream) {
    if (!embed || !workspace || E <= 0 || K <= 0) SZN_FAIL(SZN_ERR_ARG, "fused_head_prepare: bad argument");
    if (K > 256) SZN_FAIL(SZN_ERR_UNSUPPORTED, "fused_head_prepare: K=%d > 256", K);
    if (((uintptr_t)workspace) & 15) SZN_FAIL(SZN_ERR_ARG, "fused_head_prepare: workspace must be 16-B aligned");
    const int KP = kp_of(K);
    hipLaunchKernelGGL(fh_prep_kernel, dim3(szn_div_up((long)E * KP, 256)), dim3(256), 0, (hipStream_t)stream, embed, (float*)workspace, E, K, KP);
    SZN_CHECK_LAUNCH("fh_prep_kernel");
    return SZN_OK;
}

extern "C" int szn_fused_head_prepared(int stride, int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop, int K,
                                       const float* coarse, const float* embed, const int64_t* target, float* loss,
                                       float* stats, int64_t* pred, int dcoarse_dtype, void* dcoarse, void* workspace,
                                       szn_stream_t stream) {
    return fused_head_impl(stride, B, h, w, E, ldc, c0, H, W, crop, K, coarse, embed, target, loss, stats, pred, dcoarse_dtype, dcoarse, workspace,
                           stream, false);
}

static int fused_head_impl(int stride, int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop, int K,
                           const float* coarse, const float* embed, const int64_t* target, float* loss,
                           float* stats, int64_t* pred, int dcoarse_dtype, void* dcoarse, void* workspace,
                           szn_stream_t stream, bool prep) {
    if (stride != 32 && stride != 8) SZN_FAIL(SZN_ERR_UNSUPPORTED, "fused_head: stride %d (32 and 8 are built)", stride);
    if (!coarse || !embed || !workspace || B <= 0 || h <= 0 || w <= 0 || E <= 0 || c0 < 0 || ldc < c0 + E || H <= 0 ||
        W <= 0 || crop < 0 || K <= 0)
        SZN_FAIL(SZN_ERR_ARG, "fused_head: bad argument");
    if (K > 256) SZN_FAIL(SZN_ERR_UNSUPPORTED, "fused_head: K=%d > 256", K);
    if (H + crop > stride * h + stride || W + crop > stride * w + stride)
        SZN_FAIL(SZN_ERR_ARG, "fused_head: crop window exceeds the deconv output");
    if ((target == nullptr) != (loss == nullptr) || (loss && !stats)) SZN_FAIL(SZN_ERR_ARG, "fused_head: target/loss/stats go together");
    if (dcoarse && !target) SZN_FAIL(SZN_ERR_ARG, "fused_head: dcoarse needs target");
    if (((uintptr_t)workspace) & 15) SZN_FAIL(SZN_ERR_ARG, "fused_head: workspace must be 16-B aligned");
    hipStream_t st = (hipStream_t)stream;
    const int KP = kp_of(K);
    const int cells = (h + 1) * (w + 1);
    float* ws_f = (float*)workspace;
    size_t fl = ws_cell_off(E, KP) + (size_t)B * cells * ws_cell_stride(KP);
    fl = (fl + 1) / 2 * 2;
    double* part = (double*)(ws_f + fl);
    double* sums = part + (size_t)B * cells * 2;             // per-image {sum cos, count}
    float* tabD = (float*)(sums + 2 * (size_t)B);
    float* tabN = tabD + (size_t)B * h * w * KP;
    if (prep) {
        hipLaunchKernelGGL(fh_prep_kernel, dim3(szn_div_up((long)E * KP, 256)), dim3(256), 0, st, embed, ws_f, E, K, KP);
        SZN_CHECK_LAUNCH("fh_prep_kernel");
    }
    FhArgs a;
    a.coarse = coarse; a.embed = embed; a.target = target; a.pred = pred; a.ws_f = ws_f; a.part = part;
    a.B = B; a.h = h; a.w = w; a.E = E; a.ldc = ldc; a.c0 = c0; a.H = H; a.W = W; a.crop = crop; a.K = K; a.KP = KP;
    // LDS floats: Ct 4E | G 4KP | Q 16 | Aw 16KP | red 64 | dred 8 doubles; 4E + 20KP + 80 must be even for the doubles
    size_t lfl = (size_t)4 * E + 20 * KP + 16 + 64;
    const size_t lds = lfl * sizeof(float) + 8 * sizeof(double);
    if (lds > 150 * 1024) SZN_FAIL(SZN_ERR_UNSUPPORTED, "fused_head: E=%d too large for LDS", E);
#define SZN_FH_LAUNCH(KPV)                                                                                           \
    do {                                                                                                             \
        if (stride == 8) {                                                                                           \
            hipLaunchKernelGGL(fh_tables_kernel<KPV>, dim3((unsigned)(((long)B * h * w + 3) / 4)), dim3(256),        \
                               (size_t)4 * E * sizeof(float), st, a, tabD, tabN);                                    \
            hipLaunchKernelGGL((fh_cell_tab_kernel<KPV, 8>), dim3((unsigned)(((long)B * cells + 3) / 4)), dim3(256), 0, st, a,  \
                               (const float*)tabD, (const float*)tabN);                                              \
        } else {                                                                                                     \
            auto kern = fh_cell_kernel<KPV, 32>;                                                                     \
            if (lds > 48 * 1024)                                                                                     \
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
            hipLaunchKernelGGL(kern, dim3(B * cells), dim3(256), lds, st, a);                                        \
        }                                                                                                            \
    } while (0)
    if (KP == 24) SZN_FH_LAUNCH(24);
    else if (KP == 40) SZN_FH_LAUNCH(40);
    else if (KP == 64) SZN_FH_LAUNCH(64);
    else if (KP == 128) SZN_FH_LAUNCH(128);
    else if (KP == 192) SZN_FH_LAUNCH(192);
    else SZN_FH_LAUNCH(256);
#undef SZN_FH_LAUNCH
    SZN_CHECK_LAUNCH("fh_cell_kernel");
    if (loss) {
        hipLaunchKernelGGL(fh_image_sums_kernel, dim3(B), dim3(256), 0, st, (const double*)part, cells, stats, sums);
        hipLaunchKernelGGL(fh_finalize_kernel, dim3(1), dim3(64), 0, st, (const double*)sums, B, loss);
        SZN_CHECK_LAUNCH("fh_finalize_kernel");
    }
    if (dcoarse) {
        if (dcoarse_dtype == SZN_F32)
            hipLaunchKernelGGL(fh_gather_kernel<float>, dim3(B * h * w), dim3(256), 0, st, coarse, embed, (const float*)ws_f,
                               (const float*)stats, (float*)dcoarse, B, h, w, E, ldc, c0, K, KP);
        else if (dcoarse_dtype == SZN_BF16)
            hipLaunchKernelGGL(fh_gather_kernel<bf16_raw>, dim3(B * h * w), dim3(256), 0, st, coarse, embed,
                               (const float*)ws_f, (const float*)stats, (bf16_raw*)dcoarse, B, h, w, E, ldc, c0, K, KP);
        else if (dcoarse_dtype == SZN_F16)
            hipLaunchKernelGGL(fh_gather_kernel<f16_raw>, dim3(B * h * w), dim3(256), 0, st, coarse, embed,
                               (const float*)ws_f, (const float*)stats, (f16_raw*)dcoarse, B, h, w, E, ldc, c0, K, KP);
        else
            SZN_FAIL(SZN_ERR_ARG, "fused_head: bad dcoarse_dtype %d", dcoarse_dtype);
        SZN_CHECK_LAUNCH("fh_gather_kernel");
    }
    return SZN_OK;
}

extern "C" int szn_fused_head(int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop, int K,
                              const float* coarse, const float* embed, const int64_t* target, float* loss, float* stats,
                              int64_t* pred, int dcoarse_dtype, void* dcoarse, void* workspace, szn_stream_t stream) {
    return szn_fused_head_strided(32, B, h, w, E, ldc, c0, H, W, crop, K, coarse, embed, target, loss, stats, pred, dcoarse_dtype,
                                  dcoarse, workspace, stream);
}
