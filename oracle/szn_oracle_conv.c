/* szn_oracle_conv.c -- TEST INFRASTRUCTURE ONLY (the checker, never the product path).
 *
 * CPU restatement (plain C + OpenMP, fp32, NCHW / OIHW exactly like the reference's tensors) of the
 * torch ops that models.FCN32s.forward and its autograd backward invoke:
 *   nn.Conv2d stride 1          /root/reference/models.py:43-97 (layers), :116-149 (forward order)
 *   nn.ReLU                     models.py:44..90
 *   nn.MaxPool2d(2,2,ceil)      models.py:47,54,63,72,81
 *   nn.ConvTranspose2d(64,s32)  models.py:94,98,146-151 (dense weight; the crop models.py:147 is fused)
 * Pinned against tests/golden/g2_*, g3_*, g7_*, g8_* (captured from the reference by
 * tools/capture_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX4(b, c, h, w, C, H, W) ((((size_t)(b) * (C) + (c)) * (H) + (h)) * (W) + (w))

/* out[b][co][oh][ow] = bias[co] + sum_{ci,kh,kw} in[b][ci][oh+kh-pad][ow+kw-pad] * w[co][ci][kh][kw]
 * accumulation order per output element: ci, then kh, then kw ascending.                          */
void szo_conv2d_fwd(const float* in, const float* w, const float* bias, float* out, int B, int Ci, int Hi, int Wi,
                    int Co, int K, int pad, int relu) {
    const int Ho = Hi + 2 * pad - K + 1, Wo = Wi + 2 * pad - K + 1;
    const int CB = 4;
    const int ncb = (Co + CB - 1) / CB;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        for (int cb = 0; cb < ncb; ++cb) {
            const int co0 = cb * CB;
            const int nco = (Co - co0 < CB) ? Co - co0 : CB;
            float* acc = (float*)malloc((size_t)CB * Wo * sizeof(float));
            for (int oh = 0; oh < Ho; ++oh) {
                for (int j = 0; j < nco; ++j) {
                    const float bv = bias ? bias[co0 + j] : 0.f;
                    for (int ow = 0; ow < Wo; ++ow) acc[j * Wo + ow] = bv;
                }
                for (int ci = 0; ci < Ci; ++ci) {
                    for (int kh = 0; kh < K; ++kh) {
                        const int ih = oh + kh - pad;
                        if (ih < 0 || ih >= Hi) continue;
                        const float* irow = in + IDX4(b, ci, ih, 0, Ci, Hi, Wi);
                        for (int kw = 0; kw < K; ++kw) {
                            int lo = pad - kw; if (lo < 0) lo = 0;
                            int hi = Wi + pad - kw; if (hi > Wo) hi = Wo;
                            const float* ip = irow + (kw - pad);
                            for (int j = 0; j < nco; ++j) {
                                const float wv = w[(((size_t)(co0 + j) * Ci + ci) * K + kh) * K + kw];
                                float* a = acc + j * Wo;
                                for (int ow = lo; ow < hi; ++ow) a[ow] += wv * ip[ow];
                            }
                        }
                    }
                }
                for (int j = 0; j < nco; ++j) {
                    float* o = out + IDX4(b, co0 + j, oh, 0, Co, Ho, Wo);
                    for (int ow = 0; ow < Wo; ++ow) {
                        const float v = acc[j * Wo + ow];
                        o[ow] = (relu && v < 0.f) ? 0.f : v;
                    }
                }
            }
            free(acc);
        }
    }
}

/* din = conv(dout, flipped/transposed w) with pad' = K-1-pad (autograd of the op above wrt its input) */
void szo_conv2d_dgrad(const float* dout, const float* w, float* din, int B, int Ci, int Hi, int Wi, int Co, int K,
                      int pad) {
    const int Ho = Hi + 2 * pad - K + 1, Wo = Wi + 2 * pad - K + 1;
    float* wt = (float*)malloc((size_t)Ci * Co * K * K * sizeof(float));
    for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci)
            for (int kh = 0; kh < K; ++kh)
                for (int kw = 0; kw < K; ++kw)
                    wt[(((size_t)ci * Co + co) * K + (K - 1 - kh)) * K + (K - 1 - kw)] =
                        w[(((size_t)co * Ci + ci) * K + kh) * K + kw];
    szo_conv2d_fwd(dout, wt, NULL, din, B, Co, Ho, Wo, Ci, K, K - 1 - pad, 0);
    free(wt);
}

/* dw[co][ci][kh][kw] = sum_{b,oh,ow} dout[b][co][oh][ow] * in[b][ci][oh+kh-pad][ow+kw-pad]; db[co] = sum dout */
void szo_conv2d_wgrad(const float* in, const float* dout, float* dw, float* db, int B, int Ci, int Hi, int Wi, int Co,
                      int K, int pad) {
    const int Ho = Hi + 2 * pad - K + 1, Wo = Wi + 2 * pad - K + 1;
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
    for (int co = 0; co < Co; ++co) {
        for (int ci = 0; ci < Ci; ++ci) {
            for (int kh = 0; kh < K; ++kh) {
                for (int kw = 0; kw < K; ++kw) {
                    double s = 0.0;
                    int lo = pad - kw; if (lo < 0) lo = 0;
                    int hi = Wi + pad - kw; if (hi > Wo) hi = Wo;
                    for (int b = 0; b < B; ++b) {
                        for (int oh = 0; oh < Ho; ++oh) {
                            const int ih = oh + kh - pad;
                            if (ih < 0 || ih >= Hi) continue;
                            const float* dp = dout + IDX4(b, co, oh, 0, Co, Ho, Wo);
                            const float* ip = in + IDX4(b, ci, ih, 0, Ci, Hi, Wi) + (kw - pad);
                            float r = 0.f;
#pragma omp simd reduction(+ : r)
                            for (int ow = lo; ow < hi; ++ow) r += dp[ow] * ip[ow];
                            s += r;
                        }
                    }
                    dw[(((size_t)co * Ci + ci) * K + kh) * K + kw] = (float)s;
                }
            }
        }
    }
    if (db) {
#pragma omp parallel for
        for (int co = 0; co < Co; ++co) {
            double s = 0.0;
            for (int b = 0; b < B; ++b) {
                const float* dp = dout + IDX4(b, co, 0, 0, Co, Ho, Wo);
                for (size_t i = 0; i < (size_t)Ho * Wo; ++i) s += dp[i];
            }
            db[co] = (float)s;
        }
    }
}

/* MaxPool2d(2, stride 2, ceil_mode=True); idx = flat input index (ih*Wi+iw) of the FIRST maximum in scan order */
void szo_maxpool_fwd(const float* in, float* out, int32_t* idx, int B, int C, int Hi, int Wi) {
    const int Ho = (Hi + 1) / 2, Wo = (Wi + 1) / 2;
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int oh = 0; oh < Ho; ++oh)
                for (int ow = 0; ow < Wo; ++ow) {
                    float best = -INFINITY;
                    int bi = -1;
                    for (int dy = 0; dy < 2; ++dy)
                        for (int dx = 0; dx < 2; ++dx) {
                            const int ih = 2 * oh + dy, iw = 2 * ow + dx;
                            if (ih >= Hi || iw >= Wi) continue;
                            const float v = in[IDX4(b, c, ih, iw, C, Hi, Wi)];
                            if (bi < 0 || v > best) { best = v; bi = ih * Wi + iw; }
                        }
                    out[IDX4(b, c, oh, ow, C, Ho, Wo)] = best;
                    if (idx) idx[IDX4(b, c, oh, ow, C, Ho, Wo)] = bi;
                }
}

void szo_maxpool_bwd(const float* dout, const int32_t* idx, float* din, int B, int C, int Hi, int Wi) {
    const int Ho = (Hi + 1) / 2, Wo = (Wi + 1) / 2;
    memset(din, 0, (size_t)B * C * Hi * Wi * sizeof(float));
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int o = 0; o < Ho * Wo; ++o)
                din[((size_t)b * C + c) * Hi * Wi + idx[((size_t)b * C + c) * Ho * Wo + o]] +=
                    dout[((size_t)b * C + c) * Ho * Wo + o];
}

void szo_relu_bwd(const float* act, float* grad, size_t n) {
#pragma omp parallel for
    for (size_t i = 0; i < n; ++i)
        if (!(act[i] > 0.f)) grad[i] = 0.f;
}

/* out[b][co][y][x] = sum_{ci,i,j} in[b][ci][i][j] * wt[ci][co][y+crop-32i][x+crop-32j]   (kernel 64, stride 32)
 * diag != 0: wt is (C,64,64), one filter per channel (the fixed bilinear upscore, models.py:21-24)           */
void szo_deconv64s32_fwd(const float* in, const float* wt, float* out, int B, int Cin, int Cout, int h, int w, int H,
                         int W, int crop, int diag) {
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const int Y = y + crop, X = x + crop;
                    float acc = 0.f;
                    for (int ci = (diag ? co : 0); ci < (diag ? co + 1 : Cin); ++ci) {
                        const float* k = diag ? wt + (size_t)co * 4096 : wt + ((size_t)ci * Cout + co) * 4096;
                        for (int i = Y / 32 - 1; i <= Y / 32; ++i) {
                            if (i < 0 || i >= h) continue;
                            for (int j = X / 32 - 1; j <= X / 32; ++j) {
                                if (j < 0 || j >= w) continue;
                                acc += in[IDX4(b, ci, i, j, Cin, h, w)] * k[(Y - 32 * i) * 64 + (X - 32 * j)];
                            }
                        }
                    }
                    out[IDX4(b, co, y, x, Cout, H, W)] = acc;
                }
}

void szo_deconv64s32_dgrad(const float* dout, const float* wt, float* din, int B, int Cin, int Cout, int h, int w, int H,
                           int W, int crop, int diag) {
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int ci = 0; ci < Cin; ++ci)
            for (int i = 0; i < h; ++i)
                for (int j = 0; j < w; ++j) {
                    double acc = 0.0;
                    for (int co = (diag ? ci : 0); co < (diag ? ci + 1 : Cout); ++co) {
                        const float* k = diag ? wt + (size_t)ci * 4096 : wt + ((size_t)ci * Cout + co) * 4096;
                        for (int ky = 0; ky < 64; ++ky) {
                            const int y = 32 * i + ky - crop;
                            if (y < 0 || y >= H) continue;
                            for (int kx = 0; kx < 64; ++kx) {
                                const int x = 32 * j + kx - crop;
                                if (x < 0 || x >= W) continue;
                                acc += (double)dout[IDX4(b, co, y, x, Cout, H, W)] * k[ky * 64 + kx];
                            }
                        }
                    }
                    din[IDX4(b, ci, i, j, Cin, h, w)] = (float)acc;
                }
}

/* dense only: dwt[ci][co][ky][kx] = sum_{b,i,j} in[b][ci][i][j] * dout[b][co][32i+ky-crop][32j+kx-crop] */
void szo_deconv64s32_wgrad(const float* in, const float* dout, float* dwt, int B, int Cin, int Cout, int h, int w, int H,
                           int W, int crop) {
#pragma omp parallel for collapse(2)
    for (int ci = 0; ci < Cin; ++ci)
        for (int co = 0; co < Cout; ++co)
            for (int ky = 0; ky < 64; ++ky)
                for (int kx = 0; kx < 64; ++kx) {
                    double acc = 0.0;
                    for (int b = 0; b < B; ++b)
                        for (int i = 0; i < h; ++i) {
                            const int y = 32 * i + ky - crop;
                            if (y < 0 || y >= H) continue;
                            for (int j = 0; j < w; ++j) {
                                const int x = 32 * j + kx - crop;
                                if (x < 0 || x >= W) continue;
                                acc += (double)in[IDX4(b, ci, i, j, Cin, h, w)] * dout[IDX4(b, co, y, x, Cout, H, W)];
                            }
                        }
                    dwt[(((size_t)ci * Cout + co) * 64 + ky) * 64 + kx] = (float)acc;
                }
}
