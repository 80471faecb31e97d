// szn_cb.h -- tile bookkeeping of the constant-border hint (szn_conv_desc_t.cb_on), shared by conv3x3_regw (forward / dgrad: tiles whose
// output is one known value) and conv_wgrad_taps (weight gradient: tiles whose input patch is one known value per channel).
#pragma once
#include "szn_common.h"

namespace {

// ---- constant-border hint (szn_conv_desc_t.cb_on): which tiles have to run ---------------------------------------------------------
// A tile (TR rows x 16 columns of one image) can be skipped when it lies inside the rectangle in which the layers' zero padding is not
// felt (tile rows [fy0, fy1) x tile columns [fx0, fx1)) and outside the window of tiles the image can influence ([wy0, wy1) x
// [wx0, wx1), widened by one tile row: a tile of that extra row supplies the value every skipped pixel has).  The kept tiles of an
// image are numbered in dense order; decode() inverts that numbering with a few integer divisions (no list in memory: a list entry
// loaded per tile kept a scalar load outstanding across the MFMA phase and turned its counted LDS waits into full ones).
struct CbGeom {
    int on, tiles_y, tiles_x, fy0, fy1, fx0, fx1, wy0, wy1, wx0, wx1;
    int nF, nW, n0, n1, n2, n3, per_image;      // kept tiles per frame row / window row; cumulative counts of the five row bands
};

__host__ __device__ inline bool cb_skippable(const CbGeom& c, int ty, int tx) {
    if (ty < c.fy0 || ty >= c.fy1 || tx < c.fx0 || tx >= c.fx1) return false;
    return !(ty >= c.wy0 && ty < c.wy1 && tx >= c.wx0 && tx < c.wx1);
}
__host__ inline void cb_finish(CbGeom& c) {
    // window clipped to the padding-free rectangle (tiles outside it are kept anyway)
    c.wy0 = c.wy0 > c.fy0 ? c.wy0 : c.fy0; c.wy1 = c.wy1 < c.fy1 ? c.wy1 : c.fy1;
    c.wx0 = c.wx0 > c.fx0 ? c.wx0 : c.fx0; c.wx1 = c.wx1 < c.fx1 ? c.wx1 : c.fx1;
    if (c.wy1 < c.wy0) c.wy1 = c.wy0;
    if (c.wx1 < c.wx0) c.wx1 = c.wx0;
    c.nF = c.tiles_x - (c.fx1 - c.fx0);
    c.nW = c.nF + (c.wx1 - c.wx0);
    c.n0 = c.fy0 * c.tiles_x;
    c.n1 = c.n0 + (c.wy0 - c.fy0) * c.nF;
    c.n2 = c.n1 + (c.wy1 - c.wy0) * c.nW;
    c.n3 = c.n2 + (c.fy1 - c.wy1) * c.nF;
    c.per_image = c.n3 + (c.tiles_y - c.fy1) * c.tiles_x;
}
// index of a kept tile within its image -> (ty, tx)
__device__ __forceinline__ void cb_decode(const CbGeom& c, int v, int& ty, int& tx) {
    if (v < c.n0) { ty = v / c.tiles_x; tx = v - ty * c.tiles_x; return; }
    if (v >= c.n3) { const int u = v - c.n3; const int q = u / c.tiles_x; ty = c.fy1 + q; tx = u - q * c.tiles_x; return; }
    int k;
    if (v < c.n1) { const int u = v - c.n0; const int q = u / c.nF; ty = c.fy0 + q; k = u - q * c.nF; }
    else if (v >= c.n2) { const int u = v - c.n2; const int q = u / c.nF; ty = c.wy1 + q; k = u - q * c.nF; }
    else {
        const int u = v - c.n1; const int q = u / c.nW; ty = c.wy0 + q; k = u - q * c.nW;
        if (k < c.fx0) { tx = k; return; }
        k -= c.fx0;
        const int ow = c.wx1 - c.wx0;
        if (k < ow) { tx = c.wx0 + k; return; }
        tx = c.fx1 + (k - ow);
        return;
    }
    tx = k < c.fx0 ? k : k + (c.fx1 - c.fx0);
}

// rows / columns removed from a map (the constant band, models._band_cut): [ya, ye) and [ya2, ye2) per axis, ya <= ye <= ya2 <= ye2
struct BandCut {
    int ya, ye, ya2, ye2, xa, xe, xa2, xe2;
};
__host__ __device__ __forceinline__ int band_map(int v, int a, int e, int a2, int e2) {        // full index -> cropped index, -1 = removed
    return v < a ? v : (v < e ? -1 : (v < a2 ? v - (e - a) : (v < e2 ? -1 : v - (e - a) - (e2 - a2))));
}
__host__ __device__ __forceinline__ int band_unmap(int c, int a, int e, int a2, int e2) {      // cropped index -> full index
    return c < a ? c : (c < a2 - (e - a) ? c + (e - a) : c + (e - a) + (e2 - a2));
}

}  // namespace
