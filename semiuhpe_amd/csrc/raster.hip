// Mesh normals and z-buffer rasterisation of the demo renderer on gfx950.
//
// Replaces the reference's native C++ (Sim3DR/lib/rasterize_kernel.cpp): `_get_normal` (:163-221) and `_rasterize` (:225-297) with
// `is_point_in_tri` / `get_point_weight` (:27-85), behind Sim3DR/Sim3DR.py:8-29.  Both reference functions are sequential loops
// over the triangles whose results depend on that order (float accumulation per vertex; a pixel is re-blended every time a
// later triangle beats its depth), so the kernels are organised around the OWNER of each result and walk the triangles in index
// order, which makes them bit-identical to the reference at any alpha:
//   mesh_normals_kernel   a thread owns a vertex and adds the normals of its triangles in triangle order (CSR adjacency from the host);
//   raster_bbox_kernel    a thread owns a triangle: its clamped pixel bounding box (:253-261), once;
//   raster_tile_kernel    a workgroup owns a 16x16 pixel tile, a thread one pixel: the triangle list is swept in chunks of 256, the
//                         chunk's triangles that touch the tile are compacted IN ORDER into LDS (wave ballots + prefix counts) with
//                         their vertex data, and every pixel then runs the reference's inside test / barycentric weights / depth
//                         test / blend over that list with depth and colour in registers.
// Floating-point contraction is off in this file: the reference's x86-64 build has no fused multiply-adds, and divisions / square
// roots are hipcc's correctly rounded ones.
#include "common.h"

#pragma clang fp contract(off)

namespace suhpe {

__global__ __launch_bounds__(256) void mesh_normals_kernel(const float* __restrict__ vertices, const int* __restrict__ triangles,
                                                           const int* __restrict__ vt_start, const int* __restrict__ vt_tris, int nver,
                                                           float* __restrict__ normals) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nver) return;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    for (int q = vt_start[v]; q < vt_start[v + 1]; ++q) {
        const int i = vt_tris[q];
        const int a = triangles[3 * i], b = triangles[3 * i + 1], c = triangles[3 * i + 2];
        const float ax = vertices[3 * a], ay = vertices[3 * a + 1], az = vertices[3 * a + 2];
        const float v1x = vertices[3 * b] - ax, v1y = vertices[3 * b + 1] - ay, v1z = vertices[3 * b + 2] - az;
        const float v2x = vertices[3 * c] - ax, v2y = vertices[3 * c + 1] - ay, v2z = vertices[3 * c + 2] - az;
        nx += v1y * v2z - v1z * v2y;
        ny += v1z * v2x - v1x * v2z;
        nz += v1x * v2y - v1y * v2x;
    }
    const float det = sqrtf(nx * nx + ny * ny + nz * nz);       // (no zero guard, as rasterize_kernel.cpp:209-217)
    normals[3 * v] = nx / det;
    normals[3 * v + 1] = ny / det;
    normals[3 * v + 2] = nz / det;
}

__global__ __launch_bounds__(256) void raster_bbox_kernel(const float* __restrict__ vertices, const int* __restrict__ triangles, int ntri, int h,
                                                          int w, int4* __restrict__ bbox) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntri) return;
    const int a = triangles[3 * i], b = triangles[3 * i + 1], c = triangles[3 * i + 2];
    const float x0 = vertices[3 * a], y0 = vertices[3 * a + 1], x1 = vertices[3 * b], y1 = vertices[3 * b + 1], x2 = vertices[3 * c], y2 = vertices[3 * c + 1];
    int x_min = (int)floorf(fminf(x0, fminf(x1, x2))), x_max = (int)ceilf(fmaxf(x0, fmaxf(x1, x2)));
    int y_min = (int)floorf(fminf(y0, fminf(y1, y2))), y_max = (int)ceilf(fmaxf(y0, fmaxf(y1, y2)));
    x_min = x_min < 0 ? 0 : x_min;
    x_max = x_max > w - 1 ? w - 1 : x_max;
    y_min = y_min < 0 ? 0 : y_min;
    y_max = y_max > h - 1 ? h - 1 : y_max;
    bbox[i] = make_int4(x_min, x_max, y_min, y_max);
}

struct RasterTri {
    float x0, y0, z0, x1, y1, z1, x2, y2, z2;
    int a, b, c;
    int x_min, x_max, y_min, y_max;
};

constexpr int kRasterTile = 16;

__global__ __launch_bounds__(256) void raster_tile_kernel(unsigned char* __restrict__ image, const float* __restrict__ vertices,
                                                          const int* __restrict__ triangles, const float* __restrict__ colors,
                                                          float* __restrict__ depth_buffer, const int4* __restrict__ bbox, int ntri, int h, int w,
                                                          int c, float alpha, int reverse) {
    __shared__ RasterTri cand[256];
    __shared__ int wave_count[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tx0 = blockIdx.x * kRasterTile, ty0 = blockIdx.y * kRasterTile;
    const int x = tx0 + (t & 15), y = ty0 + (t >> 4);
    const bool inside_img = x < w && y < h;
    float depth = inside_img ? depth_buffer[y * w + x] : 0.f;
    unsigned char* px = image + ((int64_t)(reverse ? h - 1 - y : y) * w + x) * c;
    unsigned char col[4] = {0, 0, 0, 0};
    if (inside_img)
        for (int k = 0; k < c; ++k) col[k] = px[k];
    const float one_m_alpha = 1 - alpha, alpha255 = alpha * 255;
    const float fx = (float)x, fy = (float)y;
    for (int base = 0; base < ntri; base += 256) {
        const int i = base + t;
        bool hit = false;
        int4 bb = make_int4(0, -1, 0, -1);
        if (i < ntri) {
            bb = bbox[i];
            hit = bb.x <= bb.y && bb.z <= bb.w && bb.x <= tx0 + kRasterTile - 1 && bb.y >= tx0 && bb.z <= ty0 + kRasterTile - 1 && bb.w >= ty0;
        }
        // ordered compaction: position = hits of the earlier waves + hits of the earlier lanes of this wave
        const unsigned long long m = __ballot(hit);
        if (lane == 0) wave_count[wave] = __popcll(m);
        __syncthreads();
        int pos = __popcll(m & ((1ull << lane) - 1ull));
        int total = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < wave) pos += wave_count[q];
            total += wave_count[q];
        }
        if (hit) {
            const int a = triangles[3 * i], b = triangles[3 * i + 1], cc = triangles[3 * i + 2];
            RasterTri r;
            r.x0 = vertices[3 * a]; r.y0 = vertices[3 * a + 1]; r.z0 = vertices[3 * a + 2];
            r.x1 = vertices[3 * b]; r.y1 = vertices[3 * b + 1]; r.z1 = vertices[3 * b + 2];
            r.x2 = vertices[3 * cc]; r.y2 = vertices[3 * cc + 1]; r.z2 = vertices[3 * cc + 2];
            r.a = a; r.b = b; r.c = cc;
            r.x_min = bb.x; r.x_max = bb.y; r.y_min = bb.z; r.y_max = bb.w;
            cand[pos] = r;
        }
        __syncthreads();
        if (inside_img) {
            for (int q = 0; q < total; ++q) {
                const RasterTri& r = cand[q];
                if (x < r.x_min || x > r.x_max || y < r.y_min || y > r.y_max) continue;
                // is_point_in_tri / get_point_weight (rasterize_kernel.cpp:27-85): u along p2 - p0, v along p1 - p0
                const float v0x = r.x2 - r.x0, v0y = r.y2 - r.y0, v1x = r.x1 - r.x0, v1y = r.y1 - r.y0, v2x = fx - r.x0, v2y = fy - r.y0;
                const float dot00 = v0x * v0x + v0y * v0y, dot01 = v0x * v1x + v0y * v1y, dot02 = v0x * v2x + v0y * v2y;
                const float dot11 = v1x * v1x + v1y * v1y, dot12 = v1x * v2x + v1y * v2y;
                const float den = dot00 * dot11 - dot01 * dot01;
                const float inv = den == 0 ? 0 : 1 / den;
                const float u = (dot11 * dot02 - dot01 * dot12) * inv, v = (dot00 * dot12 - dot01 * dot02) * inv;
                if (!((u >= 0) && (v >= 0) && (u + v < 1))) continue;
                const float w0 = 1 - u - v, w1 = v, w2 = u;
                const float p_depth = w0 * r.z0 + w1 * r.z1 + w2 * r.z2;
                if (p_depth > depth) {
                    for (int k = 0; k < c; ++k) {
                        const float p_color = w0 * colors[c * r.a + k] + w1 * colors[c * r.b + k] + w2 * colors[c * r.c + k];
                        col[k] = (unsigned char)(one_m_alpha * col[k] + alpha255 * p_color);
                    }
                    depth = p_depth;
                }
            }
        }
        __syncthreads();                                        // the list is rebuilt by the next chunk
    }
    if (inside_img) {
        depth_buffer[y * w + x] = depth;
        for (int k = 0; k < c; ++k) px[k] = col[k];
    }
}

}  // namespace suhpe

using namespace suhpe;

extern "C" {

int suhpe_mesh_normals(const float* vertices, const int* triangles, const int* vt_start, const int* vt_tris, int nver, int ntri,
                       float* normals, suhpe_stream_t stream) {
    SUHPE_REQUIRE(vertices && triangles && vt_start && vt_tris && normals && nver > 0 && ntri > 0, "suhpe_mesh_normals: bad arguments");
    hipLaunchKernelGGL(mesh_normals_kernel, dim3((unsigned)ceil_div(nver, 256)), dim3(256), 0, as_stream(stream), vertices, triangles, vt_start, vt_tris,
                       nver, normals);
    return check_launch("suhpe_mesh_normals");
}

int64_t suhpe_rasterize_workspace_bytes(int ntri) { return ntri > 0 ? (int64_t)ntri * 16 : 0; }

int suhpe_rasterize(unsigned char* image, const float* vertices, const int* triangles, const float* colors, float* depth_buffer, int ntri,
                    int h, int w, int c, float alpha, int reverse, void* workspace, int64_t workspace_bytes, suhpe_stream_t stream) {
    SUHPE_REQUIRE(image && vertices && triangles && colors && depth_buffer && workspace, "suhpe_rasterize: null pointer");
    SUHPE_REQUIRE(ntri > 0 && h > 0 && w > 0 && c >= 1 && c <= 4 && (int64_t)h * w * c < (1LL << 31), "suhpe_rasterize: bad sizes (1 <= channels <= 4)");
    if (workspace_bytes < suhpe_rasterize_workspace_bytes(ntri)) {
        set_error("suhpe_rasterize: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)suhpe_rasterize_workspace_bytes(ntri));
        return SUHPE_EWORKSPACE;
    }
    int4* bbox = static_cast<int4*>(workspace);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(raster_bbox_kernel, dim3((unsigned)ceil_div(ntri, 256)), dim3(256), 0, st, vertices, triangles, ntri, h, w, bbox);
    hipLaunchKernelGGL(raster_tile_kernel, dim3((unsigned)ceil_div(w, kRasterTile), (unsigned)ceil_div(h, kRasterTile)), dim3(256), 0, st, image, vertices,
                       triangles, colors, depth_buffer, bbox, ntri, h, w, c, alpha, reverse);
    return check_launch("suhpe_rasterize");
}

}  // extern "C"
