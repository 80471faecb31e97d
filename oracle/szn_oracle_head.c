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

/* cross_entropy2d: sum over all valid pixels of -log_softmax(score)[target]; /N_valid if size_average.
 * pred = first channel argmax (score.data.max(1)[1]).                                                 */
double szo_ce2d(int B, int C, int HW, const float* score, const int64_t* target, int size_average, float* stats,
                float* dscore, int64_t* pred) {
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
            s_sum += (double)(-(sp[(size_t)lbl * HW] - mx - logf(se)));
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
                for (int c = 0; c < C; ++c)
                    dp[(size_t)c * HW] = g * (expf(sp[(size_t)c * HW] - mx) / se - (c == lbl ? 1.f : 0.f));
            }
    }
    return size_average ? total / ntot : total;
}

/* infer_lbl family.  mode 0: all rows compete.  mode 1: seen-only / unseen-only matrices (other group's rows
 * zeroed -> similarity exactly 0, still competing), stitched by the seen-mask argmax (seenmask != NULL,
 * utils.py:197-198) or by the ground-truth label being unseen (utils.py:190-191).  First index wins ties.   */
void szo_embed_argmax(int B, int E, int HW, int K, const float* score, const float* embed, int mode,
                      uint64_t unseen_bits, const float* seenmask, const int64_t* target, int64_t* pred) {
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
            float acc[64];
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
                    const int un = (int)((unseen_bits >> k) & 1ull);
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
                    take_unseen = (t >= 0 && t < 64) && ((unseen_bits >> t) & 1ull);
                }
                best = take_unseen ? iu : is;
            }
            pred[(size_t)b * HW + p] = best;
        }
    free(en);
}

/* hist[h][K][K], h = 0 all, 1 gt in seen, 2 gt in unseen (only h = 0 when unseen_bits == 0) */
void szo_confusion_hist(long n, int K, const int64_t* lt, const int64_t* lp, uint64_t unseen_bits, int64_t* hist) {
    for (long i = 0; i < n; ++i) {
        const int64_t t = lt[i], p = lp[i];
        if (t < 0 || t >= K || p < 0 || p >= K) continue;
        hist[t * K + p] += 1;
        if (unseen_bits) hist[(size_t)(((unseen_bits >> t) & 1ull) ? 2 : 1) * K * K + t * K + p] += 1;
    }
}
