/*
 * TEST INFRASTRUCTURE -- CPU binning oracle.
 *
 * The reference has no binning stage: rasterize_triangle visits every pixel of
 * a triangle's bounding box as soon as the triangle is submitted
 * (graphics.c:505-532, box from find_bounding_box, :355-364).  A tiled
 * rasteriser reproduces that order if every screen bin lists the triangles
 * whose box touches it in submission order.  This file states that rule in
 * integers only; the GPU's bin kernel must produce the same lists.
 *
 * Rule: input triangle t (submission index) owns fans[t] rasterised triangles
 * (the fan 0,k+1,k+2 of its clipped polygon, graphics.c:584-607) and the pixel
 * box boxes[t] = {x0 | x1 << 16, y0 | y1 << 16}, the union of the fan's
 * non-empty boxes.  Bin (bx, by) covers pixels [bx*edge, bx*edge + edge - 1] x
 * [by*edge, by*edge + edge - 1].  t enters a bin iff fans[t] > 0 and the two
 * closed integer rectangles intersect, and then contributes the entries
 * (t, 0) ... (t, fans[t] - 1) in that order.  Bins are numbered row-major,
 * triangles ascend inside a bin.
 */
#include "oracle.h"

size_t oracle_bin(const unsigned *boxes, const unsigned *fans, int num_tris, int width, int height, int edge,
                  unsigned *counts, unsigned *entries) {
    int bins_x = (width + edge - 1) / edge, bins_y = (height + edge - 1) / edge;
    size_t out = 0;
    int bx, by, t;
    unsigned k;
    for (by = 0; by < bins_y; by++) {
        for (bx = 0; bx < bins_x; bx++) {
            int rx0 = bx * edge, ry0 = by * edge, rx1 = rx0 + edge - 1, ry1 = ry0 + edge - 1;
            unsigned n = 0;
            for (t = 0; t < num_tris; t++) {
                int x0 = (int)(boxes[2 * t] & 0xffffu), x1 = (int)(boxes[2 * t] >> 16);
                int y0 = (int)(boxes[2 * t + 1] & 0xffffu), y1 = (int)(boxes[2 * t + 1] >> 16);
                if (fans[t] == 0 || x0 > x1 || y0 > y1) continue;
                if (x0 > rx1 || x1 < rx0 || y0 > ry1 || y1 < ry0) continue;
                for (k = 0; k < fans[t]; k++) {
                    if (entries) { entries[2 * out] = (unsigned)t; entries[2 * out + 1] = k; }
                    out++;
                    n++;
                }
            }
            counts[by * bins_x + bx] = n;
        }
    }
    return out;
}
