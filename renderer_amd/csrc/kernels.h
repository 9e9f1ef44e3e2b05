/*
 * The gfx950 kernels of the triangle pipeline.  A launch covers a batch of
 * passes (a pass = draws into one framebuffer; grid y or z = pass):
 *
 *   rz_vertex_kernel       vertex shaders incl. skeletal skinning, 1 thread / vertex; cull -> viewport -> bbox of
 *                          triangles that need no clipping
 *   rz_clip_kernel         homogeneous clipping -> fan for the rest, 1 wave / triangle
 *   rz_bin_count / _scan / _fill_kernel   ordered lists per 64x64-pixel bin
 *   rz_strip_kernel        passes without blending: coverage + depth, leaves depth and one
 *                          fragment reference per pixel; 1 wave / strip of 16x8-pixel blocks
 *   rz_sky_kernel, rz_shade_list_kernel   per-family fragment shading of those references
 *   rz_raster_list_kernel  passes with blending: coverage, depth and ordered in-kernel shading; 1 wave / 8x8 pixels,
 *                          over a list of tiles
 *   rz_blend_visibility_kernel   the same passes in large framebuffers: visibility first, ordered shading only
 *                          for the tiles where blended fragments pile up
 *   rz_clear_kernel, rz_*_to_texture(s)_kernel, rz_texture_*tile*_kernel   framebuffer / texture utilities
 *
 * All arithmetic that decides coverage, depth or colour follows the reference
 * expression order (renderer/core/graphics.c, cited inline).
 */
#ifndef RZ_KERNELS_H
#define RZ_KERNELS_H

#include <type_traits>
#include "shade.h"

/* largest d with first_tri[d] <= t */
RZ_DEV int find_draw(const uint32_t *first_tri, int num_draws, uint32_t t) {
    int lo = 0, hi = num_draws - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (first_tri[mid] <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
}

#include "kernels_geometry.h"
#include "kernels_binning.h"
#include "kernels_raster.h"
#include "kernels_shade.h"
#include "kernels_util.h"

#endif
