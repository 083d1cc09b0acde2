// cellstats.cpp -- CPU measurement tool (not product, not test): what K8's cell lists hold.
//
// Over the oracle's sorted tile lists of one view it evaluates, per (tile, list position, 4x4 cell):
//   cons  = the conservative hit K7 stores today (cell_hit(): box, then ellipse-band extent) cut at the cell's largest
//           n_contrib (what k_render_bwd_cells walks),
//   exact = some pixel of the cell takes a contribution from the entry (pos < n_contrib, !(power > 0), !(alpha < 1/255)),
// and counts contributing (pixel, entry) pairs, hits, 16-entry chunks under several packings, and histograms of hits
// per cell.  Uses the product's own host+device cull functions (ghr_device.h).  Built and driven by tools/cellstats/run.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../gaussianhaircut_amd/csrc/ghr_device.h"

extern "C" {

// out[0..]: see run.py.  hist_cons / hist_exact: 130 bins of hits per cell (0..128, 129 = more)
void cellstats(int W, int H, const uint32_t* ranges /*[T][2]*/, const uint32_t* point_list, const float* xy /*[P][2]*/,
               const float* conic_opacity /*[P][4]*/, const uint32_t* n_contrib /*[H*W]*/, double* out,
               uint64_t* hist_cons, uint64_t* hist_exact, uint64_t* hist_pix /*17 bins: pixels per exact hit*/,
               uint16_t* cell_hits /*[T][16] or NULL: stored-mask hits per cell*/)
{
    using namespace ghr;
    const int gx = (W + 15) / 16, gy = (H + 15) / 16, T = gx * gy;
    double pairs = 0, h_cons = 0, h_exact = 0, h_cons_uncut = 0;
    double ch_cons = 0, ch_exact = 0, ch_exact_band = 0, ch_exact_tile = 0, ch_cons_tile = 0, ch_cons_band = 0;
    double ch8_exact = 0;       // 8x4 regions (two cells), 8-entry chunks
    double cells_cons = 0, cells_exact = 0, one_chunk_cons = 0, one_chunk_exact = 0;
    double fp_margin = 0, fp_sat = 0;   // false positives: no pixel passes the alpha test / passes but all pixels are done
    double half_cons = 0, half_exact = 0;  // tail chunks at most half full
    double steps64_cell = 0, steps64_tile = 0;  // wave steps of 64 CONTRIBUTING pairs, packed per cell / per tile
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : pairs, h_cons, h_exact, h_cons_uncut, ch_cons, ch_exact, \
    ch_exact_band, ch_exact_tile, ch_cons_tile, ch_cons_band, ch8_exact, cells_cons, cells_exact, one_chunk_cons,      \
    one_chunk_exact, fp_margin, fp_sat, half_cons, half_exact, steps64_cell, steps64_tile)
    for (int tile = 0; tile < T; tile++) {
        const int tx = tile % gx, ty = tile / gx;
        const uint32_t beg = ranges[2 * tile], end = ranges[2 * tile + 1];
        if (end <= beg) continue;
        const uint32_t n = end - beg;
        uint32_t last[256], clast[16];
        for (int c = 0; c < 16; c++) clast[c] = 0;
        for (int p = 0; p < 256; p++) {
            const int px = tx * 16 + (p & 15), py = ty * 16 + (p >> 4);
            last[p] = (px < W && py < H) ? n_contrib[(size_t)py * W + px] : 0u;
            const int cell = ((p >> 4) >> 2) * 4 + ((p & 15) >> 2);
            if (last[p] > clast[cell]) clast[cell] = last[p];
        }
        uint32_t hc[16] = {0}, he[16] = {0};
        uint64_t pc[16] = {0};  // contributing pairs per cell
        uint32_t he8[8] = {0};
        for (uint32_t pos = 0; pos < n; pos++) {
            const uint32_t id = point_list[beg + pos];
            const f4 r0 = {xy[2 * id], xy[2 * id + 1], conic_opacity[4 * id], conic_opacity[4 * id + 1]};
            const f4 r1 = {conic_opacity[4 * id + 2], conic_opacity[4 * id + 3], 0.f, 0.f};
            const f4 bb = alpha_bbox(r0, r1), ep = ellipse_params(r0, r1);
            bool ex8[8] = {false, false, false, false, false, false, false, false};
            for (int cell = 0; cell < 16; cell++) {
                const int band = cell >> 2, g = cell & 3;
                const float X0 = (float)(tx * 16 + 4 * g), Y0 = (float)(ty * 16 + 4 * band);
                const bool cons_raw = cell_hit(bb, ep, r0, X0, Y0);
                const bool cons = cons_raw && pos < clast[cell];
                int npix = 0, nalpha = 0;
                for (int q = 0; q < 16; q++) {
                    const int lx = 4 * g + (q & 3), ly = 4 * band + (q >> 2);
                    const float pxf = (float)(tx * 16 + lx), pyf = (float)(ty * 16 + ly);
                    const float dx = r0.x - pxf, dy = r0.y - pyf;
                    const float power = -0.5f * (r0.z * dx * dx + r1.x * dy * dy) - r0.w * dx * dy;
                    const float alpha = fminf(0.99f, r1.y * expf(power));
                    const bool a_ok = !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
                    nalpha += a_ok;
                    if (a_ok && pos < last[ly * 16 + lx]) npix++;
                }
                if (cons_raw) h_cons_uncut++;
                if (cons) {
                    hc[cell]++;
                    if (npix == 0) { if (nalpha == 0) fp_margin++; else fp_sat++; }
                }
                if (npix > 0) {
                    he[cell]++;
                    pairs += npix;
                    pc[cell] += (uint64_t)npix;
                    ex8[band * 2 + (g >> 1)] = true;
#pragma omp atomic
                    hist_pix[npix]++;
                }
            }
            for (int r = 0; r < 8; r++) he8[r] += ex8[r];
        }
        if (cell_hits) for (int c = 0; c < 16; c++) cell_hits[(size_t)tile * 16 + c] = (uint16_t)(hc[c] > 65535u ? 65535u : hc[c]);
        uint32_t sc = 0, se = 0;
        for (int band = 0; band < 4; band++) {
            uint32_t bc = 0, be = 0;
            for (int g = 0; g < 4; g++) {
                const int cell = band * 4 + g;
                bc += hc[cell]; be += he[cell];
                h_cons += hc[cell]; h_exact += he[cell];
                ch_cons += (hc[cell] + 15) / 16; ch_exact += (he[cell] + 15) / 16;
                if (hc[cell]) { cells_cons++; if (hc[cell] <= 16) one_chunk_cons++; const uint32_t t = hc[cell] % 16; if (t && t <= 8) half_cons++; }
                if (he[cell]) { cells_exact++; if (he[cell] <= 16) one_chunk_exact++; const uint32_t t = he[cell] % 16; if (t && t <= 8) half_exact++; }
#pragma omp atomic
                hist_cons[hc[cell] > 128 ? 129 : hc[cell]]++;
#pragma omp atomic
                hist_exact[he[cell] > 128 ? 129 : he[cell]]++;
            }
            ch_cons_band += (bc + 15) / 16; ch_exact_band += (be + 15) / 16;
            sc += bc; se += be;
        }
        for (int r = 0; r < 8; r++) ch8_exact += (he8[r] + 7) / 8;
        { uint64_t tp = 0; for (int c = 0; c < 16; c++) { steps64_cell += (double)((pc[c] + 63) / 64); tp += pc[c]; } steps64_tile += (double)((tp + 63) / 64); }
        ch_cons_tile += (sc + 15) / 16; ch_exact_tile += (se + 15) / 16;
    }
    double* o = out;
    *o++ = pairs; *o++ = h_cons; *o++ = h_exact; *o++ = h_cons_uncut; *o++ = ch_cons; *o++ = ch_exact;
    *o++ = ch_cons_band; *o++ = ch_exact_band; *o++ = ch_cons_tile; *o++ = ch_exact_tile; *o++ = ch8_exact;
    *o++ = cells_cons; *o++ = cells_exact; *o++ = one_chunk_cons; *o++ = one_chunk_exact; *o++ = fp_margin; *o++ = fp_sat;
    *o++ = half_cons; *o++ = half_exact; *o++ = steps64_cell; *o++ = steps64_tile;
}
}
