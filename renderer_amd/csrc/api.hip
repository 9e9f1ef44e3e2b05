/*
 * Host side of the MI355X triangle pipeline: the C ABI of include/mi355r.h.
 *
 * The reference API is immediate-mode and synchronous (graphics.c); here draw
 * calls are RECORDED -- the program's uniform block is snapshotted and its
 * host pointers (textures, joint matrices, IBL data) are translated to device
 * handles -- and executed in submission order at flush points:
 *   framebuffer_sync / framebuffer_read*, texture_from_{depth,color}buffer,
 *   a clear of a framebuffer that has draws pending, graphics_flush, releases.
 * A flush turns every maximal run of draws into one framebuffer into a "pass"
 * and launches vertex -> setup -> bin -> raster for it on the library's HIP
 * stream; nothing waits for the GPU except the calls that hand pixels back to
 * the host.  There is no CPU rendering path: without a GPU the first call
 * aborts with the HIP error.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <array>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <chrono>
#include <vector>

#include "../../include/mi355r.h"
#include "kernels.h"

static_assert(sizeof(rz_vertex) == 80, "vertex stream layout (mesh.h:8-15)");
static_assert(sizeof(rz_blinn_uniforms) == 344, "blinn uniform block (blinn_shader.h:23-44)");
static_assert(sizeof(rz_pbr_uniforms) == 440, "pbr uniform block (pbr_shader.h:39-76)");
static_assert(sizeof(rz_skybox_uniforms) == 72, "skybox uniform block (skybox_shader.h:16-19)");
static_assert(sizeof(rz_blinn_attribs) == 64, "blinn attribs");
static_assert(sizeof(TriRec) == 64, "TriRec is four float4");
static_assert(sizeof(BinEntry) == 80, "BinEntry is five float4");
static_assert(sizeof(DevTex) == 24, "DevTex");

/* The message goes to stderr and, when MI355R_FATAL_LOG names a file, to that file as well: a test
 * runner that captures stderr loses what it captured when the process aborts. */
[[noreturn]] static void fatal(const char *fmt, ...) {
    char text[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(text, sizeof text, fmt, ap);
    va_end(ap);
    fprintf(stderr, "mi355r: fatal: %s\n", text);
    fflush(stderr);
    if (const char *path = getenv("MI355R_FATAL_LOG")) {
        if (FILE *f = fopen(path, "a")) {
            fprintf(f, "mi355r: fatal: %s\n", text);
            fclose(f);
        }
    }
    abort();
}

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    fatal("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

/* The host side in reading order; one translation unit (see host_context.h). */
#include "host_context.h"
#include "host_resources.h"
#include "host_recorder.h"
#include "host_flush.h"
#include "host_readback.h"
#include "host_inspect.h"
