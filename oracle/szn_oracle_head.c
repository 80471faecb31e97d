/* szn_oracle_head.c -- TEST INFRASTRUCTURE ONLY (the checker, never the product path).
 *
 * CPU restatement (plain C, fp32, compiled with -ffp-contract=off) of /root/reference/utils.py:
 *   cross_entropy2d  utils.py:19-48      mse_loss  utils.py:50-73      cosine_loss  utils.py:75-102
 *   _fast_hist       utils.py:104-119    infer_lbl utils.py:159-185
 *   infer_lbl_forced_unseen / infer_lbl_szn / stich_seen_unseen_with_mask  utils.py:188-205
 * Pinned against tests/golden/g4_*, g5_*, g6_* (captured from the reference).
 *
 * Arithmetic contract shared with the HIP kernels (csrc/szn_head.hip) so that the per-pixel argmax is
 * bit-exact between CPU and GPU: dot products and squared norms are fmaf chains over the channel index in
 * ascending order starting from 0.f, norms are sqrtf, similarities are IEEE divisions by (||s|| * ||e_k||).
 * Batched definition (the reference raises for n > 1): per-image loss, mean over images.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* mode 0 cosine: term = cos(score_px, t_px); mode 1 mse: term = sum_c (s-t)^2.
 * target_embed (B,E,H,W) may be NULL -> gather embed[K][E] by label (ignore pixels use row 0).
 * stats[b] = {sum of terms over valid px, #valid}.  Returns the loss.  dscore may be NULL.          */
static double embed_loss(int mode, int B, int E, int HW, int K, const float* score, const int64_t* target,
                         const float* embed, const float* tembed, float* stats, float* dscore) {
    double total = 0.0;
    for (int b = 0; b < B; ++b) {
        double s_sum = 0.0, n = 0.0;
        for (int p = 0; p < HW; ++p) {
            const int64_t lbl = target[(size_t)b * HW + p];
            if (lbl < 0) continue;
            const float* sp = score + (size_t)b * E * HW + p;
            const float* tp = tembed ? tembed + (size_t)b * E * HW + p : NULL;
            const float* er = embed ? embed + (size_t)(lbl < K ? lbl : 0) * E : NULL;
            float ss = 0.f, st = 0.f, tt = 0.f;
            for (int c = 0; c < E; ++c) {
                const float s = sp[(size_t)c * HW];
                const float t = tp ? tp[(size_t)c * HW] : er[c];
                if (mode == 0) { ss = fmaf(s, s, ss); st = fmaf(s, t, st); tt = fmaf(t, t, tt); }
                else { const float d = s - t; ss = fmaf(d, d, ss); }
            }
            s_sum += (mode == 0) ? (double)(st / (sqrtf(ss) * sqrtf(tt))) : (double)ss;
            n += 1.0;
        }
        stats[2 * b] = (float)s_sum;
        stats[2 * b + 1] = (float)n;
        total += (mode == 0) ? (n - s_sum) / n : s_sum / n;
    }
    if (dscore) {
        for (int b = 0; b < B; ++b) {
            const float g = 1.f / ((float)B * stats[2 * b + 1]);
            for (int p = 0; p < HW; ++p) {
                const int64_t lbl = target[(size_t)b * HW + p];
                const float* sp = score + (size_t)b * E * HW + p;
                float* dp = dscore + (size_t)b * E * HW + p;
                if (lbl < 0) { for (int c = 0; c < E; ++c) dp[(size_t)c * HW] = 0.f; continue; }
                const float* tp = tembed ? tembed + (size_t)b * E * HW + p : NULL;
                const float* er = embed ? embed + (size_t)(lbl < K ? lbl : 0) * E : NULL;
                if (mode == 0) {
                    float ss = 0.f, st = 0.f, tt = 0.f;
                    for (int c = 0; c < E; ++c) {
                        const float s = sp[(size_t)c * HW];
                        const float t = tp ? tp[(size_t)c * HW] : er[c];
                        ss = fmaf(s, s, ss); st = fmaf(s, t, st); tt = fmaf(t, t, tt);
                    }
                    const float ns = sqrtf(ss), nt = sqrtf(tt);
                    const float cosv = st / (ns * nt);
                    const float a = g / (ns * nt), bq = g * cosv / ss;
                    /* d/ds [ (N - sum cos)/N ] = -(t/(|s||t|) - cos * s/|s|^2) / N */
                    for (int c = 0; c < E; ++c) {
                        const float s = sp[(size_t)c * HW];
                        const float t = tp ? tp[(size_t)c * HW] : er[c];
                        dp[(size_t)c * HW] = bq * s - a * t;
                    }
                } else {
                    for (int c = 0; c < E; ++c) {
                        const float s = sp[(size_t)c * HW];
                        const float t = tp ? tp[(size_t)c * HW] : er[c];
                        dp[(size_t)c * HW] = 2.f * g * (s - t);
                    }
                }
            }
        }
    }
    return total / B;
}

double szo_cosine_loss(int B, int E, int HW, int K, const float* score, const int64_t* target, const float* embed,
                       const float* tembed, float* stats, float* dscore) {
    return embed_loss(0, B, E, HW, K, score, target, embed, tembed, stats, dscore);
}
double szo_mse_loss(int B, int E, int HW, int K, const float* score, const int64_t* target, const float* embed,
                    const float* tembed, float* stats, float* dscore) {
    return embed_loss(1, B, E, HW, K, score, target, embed, tembed, stats, dscore);
}

/* cross_entropy2d (utils.py:19-48): sum over all valid pixels of -weight[target] * log_softmax(score)[target] (weight may be
 * NULL); /N_valid (the pixel count, utils.py:47-48) if size_average.
 * pred = first channel argmax (score.data.max(1)[1]).                                                 */
double szo_ce2d(int B, int C, int HW, const float* score, const int64_t* target, const float* weight, int size_average,
                float* stats, float* dscore, int64_t* pred) {
    double total = 0.0, ntot = 0.0;
    for (int b = 0; b < B; ++b) {
        double s_sum = 0.0, n = 0.0;
        for (int p = 0; p < HW; ++p) {
            const float* sp = score + (size_t)b * C * HW + p;
            float mx = sp[0];
            int am = 0;
            for (int c = 1; c < C; ++c) if (sp[(size_t)c * HW] > mx) { mx = sp[(size_t)c * HW]; am = c; }
            if (pred) pred[(size_t)b * HW + p] = am;
            const int64_t lbl = target[(size_t)b * HW + p];
            if (lbl < 0 || lbl >= C) continue;
            float se = 0.f;
            for (int c = 0; c < C; ++c) se += expf(sp[(size_t)c * HW] - mx);
            float term = -(sp[(size_t)lbl * HW] - mx - logf(se));
            if (weight) term = weight[lbl] * term;          /* F.nll_loss(weight=, size_average=False): utils.py:46 */
            s_sum += (double)term;
            n += 1.0;
        }
        stats[2 * b] = (float)s_sum; stats[2 * b + 1] = (float)n;
        total += s_sum; ntot += n;
    }
    if (dscore) {
        const float g = size_average ? (float)(1.0 / ntot) : 1.f;
        for (int b = 0; b < B; ++b)
            for (int p = 0; p < HW; ++p) {
                const float* sp = score + (size_t)b * C * HW + p;
                float* dp = dscore + (size_t)b * C * HW + p;
                const int64_t lbl = target[(size_t)b * HW + p];
                if (lbl < 0 || lbl >= C) { for (int c = 0; c < C; ++c) dp[(size_t)c * HW] = 0.f; continue; }
                float mx = sp[0];
                for (int c = 1; c < C; ++c) mx = fmaxf(mx, sp[(size_t)c * HW]);
                float se = 0.f;
                for (int c = 0; c < C; ++c) se += expf(sp[(size_t)c * HW] - mx);
                const float gw = weight ? g * weight[lbl] : g;
                for (int c = 0; c < C; ++c)
                    dp[(size_t)c * HW] = gw * (expf(sp[(size_t)c * HW] - mx) / se - (c == lbl ? 1.f : 0.f));
            }
    }
    return size_average ? total / ntot : total;
}

/* infer_lbl family.  mode 0: all rows compete.  mode 1: seen-only / unseen-only matrices (other group's rows
 * zeroed -> similarity exactly 0, still competing), stitched by the seen-mask argmax (seenmask != NULL,
 * utils.py:197-198) or by the ground-truth label being unseen (utils.py:190-191).  First index wins ties.   */
/* class sets are word arrays: bit (k % 64) of word (k / 64) = class k, (K + 63) / 64 words, NULL = the empty set */
#define SZO_MAX_CLASSES 256
static int szo_in_set(const uint64_t* words, int K, int64_t k) {
    return words && k >= 0 && k < K && ((words[k >> 6] >> (k & 63)) & 1ull);
}
static int szo_set_nonempty(const uint64_t* words, int K) {
    if (!words) return 0;
    for (int i = 0; i < (K + 63) / 64; ++i) if (words[i]) return 1;
    return 0;
}

void szo_embed_argmax(int B, int E, int HW, int K, const float* score, const float* embed, int mode,
                      const uint64_t* unseen_words, const float* seenmask, const int64_t* target, int64_t* pred) {
    if (K > SZO_MAX_CLASSES) abort();
    float* en = (float*)malloc((size_t)K * sizeof(float));
    for (int k = 0; k < K; ++k) {
        float s = 0.f;
        for (int c = 0; c < E; ++c) s = fmaf(embed[(size_t)k * E + c], embed[(size_t)k * E + c], s);
        const float n = sqrtf(s);
        en[k] = (n == 0.f) ? 1.f : n;                        /* utils.py:175 */
    }
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < HW; ++p) {
            const float* sp = score + (size_t)b * E * HW + p;
            float acc[SZO_MAX_CLASSES];
            float ss = 0.f;
            for (int k = 0; k < K; ++k) acc[k] = 0.f;
            for (int c = 0; c < E; ++c) {
                const float s = sp[(size_t)c * HW];
                ss = fmaf(s, s, ss);
                for (int k = 0; k < K; ++k) acc[k] = fmaf(s, embed[(size_t)k * E + c], acc[k]);
            }
            const float sn = sqrtf(ss);
            int best = 0;
            if (mode == 0) {
                float bv = 0.f;
                for (int k = 0; k < K; ++k) {
                    const float sim = acc[k] / (sn * en[k]);
                    if (k == 0 || sim > bv) { bv = sim; best = k; }
                }
            } else {
                const float zero_sim = 0.f / (sn * 1.f);
                float bs = 0.f, bu = 0.f;
                int is = 0, iu = 0;
                for (int k = 0; k < K; ++k) {
                    const float sim = acc[k] / (sn * en[k]);
                    const int un = szo_in_set(unseen_words, K, k);
                    const float vs = un ? zero_sim : sim, vu = un ? sim : zero_sim;
                    if (k == 0 || vs > bs) { bs = vs; is = k; }
                    if (k == 0 || vu > bu) { bu = vu; iu = k; }
                }
                int take_unseen;
                if (seenmask) {
                    const float s0 = seenmask[((size_t)b * 2 + 0) * HW + p], s1 = seenmask[((size_t)b * 2 + 1) * HW + p];
                    take_unseen = !(s1 > s0);
                } else {
                    const int64_t t = target[(size_t)b * HW + p];
                    take_unseen = szo_in_set(unseen_words, K, t);
                }
                best = take_unseen ? iu : is;
            }
            pred[(size_t)b * HW + p] = best;
        }
    free(en);
}

/* hist[h][K][K], h = 0 all, 1 gt in seen, 2 gt in unseen (only h = 0 when the unseen set is empty) */
void szo_confusion_hist(long n, int K, const int64_t* lt, const int64_t* lp, const uint64_t* unseen_words, int64_t* hist) {
    const int split = szo_set_nonempty(unseen_words, K);
    for (long i = 0; i < n; ++i) {
        const int64_t t = lt[i], p = lp[i];
        if (t < 0 || t >= K || p < 0 || p >= K) continue;
        hist[t * K + p] += 1;
        if (split) hist[(size_t)(szo_in_set(unseen_words, K, t) ? 2 : 1) * K * K + t * K + p] += 1;
    }
}

/* ---------------------------------------------------------------------------------------------------------
 * szo_fused_head: CPU restatement of the fused-from-coarse head the training step runs
 * (csrc/szn_fused_head.hip): bilinear x32 upsample + crop (models.py:146-147), cosine loss (utils.py:75-102),
 * infer_lbl (utils.py:159-185), evaluated per 32x32 output cell from the four coarse vectors C_t that every
 * pixel of the cell blends:   s.e_k = sum_t w_t (C_t.e_k) = sum_t w_t G[t][k],   |s|^2 = sum_tu w_t w_u Q[t][u].
 *
 * Arithmetic contract for the class assignment (bit-exact with the kernel):
 *   G[t][k]  fmaf chain over the channel index, ascending, from 0.f
 *   Q[t][u]  64 strided partial chains (channel c goes to partial c % 64, ascending) combined by the
 *            xor-butterfly 32,16,8,4,2,1 (what a 64-lane wave reduction computes in lane 0)
 *   w_t      float( double(1-|ty-31.5|/32 ...) products ), ss = fmaf chain over (t,u) of (w_t*w_u) * Q[t][u],
 *            sim_k = (fmaf chain over t of w_t*G[t][k]) / (sqrtf(ss) * en_k), en_k = ||e_k|| with 0 -> 1, first index
 *            wins ties.
 * The loss / gradient reductions are done here in double (the kernel uses fixed-order float / double trees):
 * compared with tolerances, not bit-exactly.
 * dcoarse: (B,h,w,ldc) float, only channels [c0, c0+E) written (others left untouched), may be NULL.          */
static float wave_tree64(float* v) {
    for (int o = 32; o > 0; o >>= 1) {
        float n[64];
        for (int i = 0; i < 64; ++i) n[i] = v[i] + v[i ^ o];
        memcpy(v, n, sizeof(n));
    }
    return v[0];
}

static double fh_bil1d(int t, int S) { return 1.0 - fabs((double)t - ((double)S - 0.5)) / (double)S; }

/* S = the upsampling stride (kernel 2S): 32 = the reference's upscore (models.py:94,146-147); 8 = upscore8 of the public FCN8s
 * head (crop 31) -- same arithmetic over S x S cells.                                                                    */
double szo_fused_head_s(int S, int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop, int K, const float* coarse,
                        const float* embed, const int64_t* target, float* stats, int64_t* pred, float* dcoarse) {
    if (K > SZO_MAX_CLASSES) abort();
    float* en = (float*)malloc((size_t)K * sizeof(float));
    float* ent = (float*)malloc((size_t)K * sizeof(float));
    for (int k = 0; k < K; ++k) {
        float s = 0.f;
        for (int c = 0; c < E; ++c) s = fmaf(embed[(size_t)k * E + c], embed[(size_t)k * E + c], s);
        ent[k] = sqrtf(s);
        en[k] = (ent[k] == 0.f) ? 1.f : ent[k];
    }
    const int cells_w = w + 1, cells = (h + 1) * cells_w;
    double* cs = (double*)calloc((size_t)B * cells * 2, sizeof(double));          /* per cell: sum cos, count */
    double* dC = dcoarse ? (double*)calloc((size_t)B * cells * 4 * E, sizeof(double)) : NULL;   /* per cell, per tap */
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int b = 0; b < B; ++b)
        for (int cell = 0; cell < cells; ++cell) {
            const int I = cell / cells_w, J = cell % cells_w;
            float* Ct = (float*)calloc((size_t)4 * E, sizeof(float));
            float G[4][SZO_MAX_CLASSES], Q[16];
            for (int t = 0; t < 4; ++t) {
                const int ci = I - 1 + (t >> 1), cj = J - 1 + (t & 1);
                if (ci >= 0 && ci < h && cj >= 0 && cj < w)
                    for (int c = 0; c < E; ++c) Ct[t * E + c] = coarse[(((size_t)b * h + ci) * w + cj) * ldc + c0 + c];
            }
            for (int t = 0; t < 4; ++t)
                for (int k = 0; k < K; ++k) {
                    float g = 0.f;
                    for (int c = 0; c < E; ++c) g = fmaf(Ct[t * E + c], embed[(size_t)k * E + c], g);
                    G[t][k] = g;
                }
            for (int t = 0; t < 4; ++t)
                for (int u = 0; u < 4; ++u) {
                    float part[64];
                    for (int l = 0; l < 64; ++l) {
                        float q = 0.f;
                        for (int c = l; c < E; c += 64) q = fmaf(Ct[t * E + c], Ct[u * E + c], q);
                        part[l] = q;
                    }
                    Q[t * 4 + u] = wave_tree64(part);
                }
            double* dc = dC ? dC + ((size_t)b * cells + cell) * 4 * E : NULL;
            double csum = 0.0, cnt = 0.0;
            for (int ty = 0; ty < S; ++ty)
                for (int tx = 0; tx < S; ++tx) {
                    const int y = S * I + ty - crop, x = S * J + tx - crop;
                    if (y < 0 || y >= H || x < 0 || x >= W) continue;
                    const double fy1 = fh_bil1d(ty, S), fy0 = fh_bil1d(ty + S, S), fx1 = fh_bil1d(tx, S), fx0 = fh_bil1d(tx + S, S);
                    const float wt[4] = {(float)(fy0 * fx0), (float)(fy0 * fx1), (float)(fy1 * fx0), (float)(fy1 * fx1)};
                    float ss = 0.f;
                    for (int t = 0; t < 4; ++t)
                        for (int u = 0; u < 4; ++u) ss = fmaf(wt[t] * wt[u], Q[t * 4 + u], ss);
                    const float sn = sqrtf(ss);
                    const size_t pix = ((size_t)b * H + y) * W + x;
                    if (pred) {
                        int best = 0;
                        float bv = 0.f;
                        for (int k = 0; k < K; ++k) {
                            float d = 0.f;
                            for (int t = 0; t < 4; ++t) d = fmaf(wt[t], G[t][k], d);
                            const float sim = d / (sn * en[k]);
                            if (k == 0 || sim > bv) { bv = sim; best = k; }
                        }
                        pred[pix] = best;
                    }
                    const int64_t lbl = target ? target[pix] : -1;
                    if (lbl < 0) continue;
                    const int kl = lbl < K ? (int)lbl : 0;
                    float d = 0.f;
                    for (int t = 0; t < 4; ++t) d = fmaf(wt[t], G[t][kl], d);
                    const float nt = ent[kl];
                    const float cosv = d / (sn * nt);
                    csum += (double)cosv;
                    cnt += 1.0;
                    if (dc) {
                        /* d(-cos)/ds = cos * s/|s|^2 - e/(|s||e|), s = sum_t w_t C_t; dC_t += w_t * that */
                        const double aco = 1.0 / ((double)sn * (double)nt), bco = (double)cosv / (double)ss;
                        for (int c = 0; c < E; ++c) {
                            double s = 0.0;
                            for (int t = 0; t < 4; ++t) s += (double)wt[t] * (double)Ct[t * E + c];
                            const double gs = bco * s - aco * (double)embed[(size_t)kl * E + c];
                            for (int t = 0; t < 4; ++t) dc[t * E + c] += (double)wt[t] * gs;
                        }
                    }
                }
            cs[((size_t)b * cells + cell) * 2] = csum;
            cs[((size_t)b * cells + cell) * 2 + 1] = cnt;
            free(Ct);
        }
    double total = 0.0;
    for (int b = 0; b < B; ++b) {
        double s = 0.0, n = 0.0;
        for (int cell = 0; cell < cells; ++cell) { s += cs[((size_t)b * cells + cell) * 2]; n += cs[((size_t)b * cells + cell) * 2 + 1]; }
        if (stats) { stats[2 * b] = (float)s; stats[2 * b + 1] = (float)n; }
        total += (n - s) / n;
        if (dcoarse) {
            const double scale = 1.0 / ((double)B * n);
            for (int i = 0; i < h; ++i)
                for (int j = 0; j < w; ++j)
                    for (int c = 0; c < E; ++c) {
                        double acc = 0.0;
                        for (int u = 0; u < 4; ++u) {      /* position (i,j) is tap u of cell (i+1-(u>>1), j+1-(u&1)) */
                            const int I = i + 1 - (u >> 1), J = j + 1 - (u & 1);
                            acc += dC[(((size_t)b * cells + I * cells_w + J) * 4 + u) * E + c];
                        }
                        dcoarse[(((size_t)b * h + i) * w + j) * ldc + c0 + c] = (float)(acc * scale);
                    }
        }
    }
    free(en); free(ent); free(cs); free(dC);
    return target ? total / B : 0.0;
}

double szo_fused_head(int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop, int K, const float* coarse,
                      const float* embed, const int64_t* target, float* stats, int64_t* pred, float* dcoarse) {
    return szo_fused_head_s(32, B, h, w, E, ldc, c0, H, W, crop, K, coarse, embed, target, stats, pred, dcoarse);
}
