/* oracle/sim3dr_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C restatement of the two native functions the reference's demo renderer calls:
 *   sim3dr_get_normal   <- _get_normal          /root/reference/Sim3DR/lib/rasterize_kernel.cpp:163-221
 *   sim3dr_rasterize    <- _rasterize           /root/reference/Sim3DR/lib/rasterize_kernel.cpp:225-297
 *                          is_point_in_tri      :27-53, get_point_weight :55-85
 * Pinned by the reference itself: oracle/Makefile compiles that file, where it lies, into oracle/_ref/libsim3dr_ref.so and
 * tests/test_oracle_sim3dr.py compares the two bit for bit (and against tests/golden/sim3dr.npz on machines without the reference).
 * Built without floating-point contraction (gcc -ffp-contract=off): the reference's x86-64 build has no fused multiply-adds.
 */
#include <math.h>

/* per-vertex normals: un-normalised triangle normals accumulated over each vertex's triangles IN TRIANGLE ORDER, then normalised
 * (no guard against a zero sum, as the reference: rasterize_kernel.cpp:209-217) */
void sim3dr_get_normal(float *ver_normal, const float *vertices, const int *triangles, int nver, int ntri) {
    for (int i = 0; i < ntri; i++) {                                    /* :171-188 fused with :191-201 (same order per vertex) */
        const int a = triangles[3 * i], b = triangles[3 * i + 1], c = triangles[3 * i + 2];
        const float v1x = vertices[3 * b] - vertices[3 * a], v1y = vertices[3 * b + 1] - vertices[3 * a + 1], v1z = vertices[3 * b + 2] - vertices[3 * a + 2];
        const float v2x = vertices[3 * c] - vertices[3 * a], v2y = vertices[3 * c + 1] - vertices[3 * a + 1], v2z = vertices[3 * c + 2] - vertices[3 * a + 2];
        const float n[3] = {v1y * v2z - v1z * v2y, v1z * v2x - v1x * v2z, v1x * v2y - v1y * v2x};
        for (int j = 0; j < 3; j++) {
            ver_normal[3 * a + j] += n[j];
            ver_normal[3 * b + j] += n[j];
            ver_normal[3 * c + j] += n[j];
        }
    }
    for (int i = 0; i < nver; ++i) {                                    /* :204-218 */
        const float nx = ver_normal[3 * i], ny = ver_normal[3 * i + 1], nz = ver_normal[3 * i + 2];
        const float det = sqrtf(nx * nx + ny * ny + nz * nz);
        ver_normal[3 * i] = nx / det;
        ver_normal[3 * i + 1] = ny / det;
        ver_normal[3 * i + 2] = nz / det;
    }
}

/* barycentric coordinates of pixel p in triangle (p0, p1, p2): u along p2 - p0, v along p1 - p0 (:27-53 and :55-85 compute the same
 * u, v); returns 1 when the pixel is inside (u >= 0, v >= 0, u + v < 1) */
static int bary(float px, float py, float x0, float y0, float x1, float y1, float x2, float y2, float *u_, float *v_) {
    const float v0x = x2 - x0, v0y = y2 - y0, v1x = x1 - x0, v1y = y1 - y0, v2x = px - x0, v2y = py - y0;
    const float dot00 = v0x * v0x + v0y * v0y, dot01 = v0x * v1x + v0y * v1y, dot02 = v0x * v2x + v0y * v2y;
    const float dot11 = v1x * v1x + v1y * v1y, dot12 = v1x * v2x + v1y * v2y;
    const float den = dot00 * dot11 - dot01 * dot01;
    const float inv = den == 0 ? 0 : 1 / den;
    const float u = (dot11 * dot02 - dot01 * dot12) * inv, v = (dot00 * dot12 - dot01 * dot02) * inv;
    *u_ = u;
    *v_ = v;
    return (u >= 0) && (v >= 0) && (u + v < 1);
}

/* z-buffer rasterisation with per-vertex colours, triangles in index order; a pixel is (re)painted whenever a triangle's
 * interpolated depth is STRICTLY larger than the buffer's, blended with what the image holds at that moment (:225-297) */
void sim3dr_rasterize(unsigned char *image, const float *vertices, const int *triangles, const float *colors, float *depth_buffer,
                      int ntri, int h, int w, int c, float alpha, int reverse) {
    for (int i = 0; i < ntri; i++) {
        const int a = triangles[3 * i], b = triangles[3 * i + 1], cc = triangles[3 * i + 2];
        const float x0 = vertices[3 * a], y0 = vertices[3 * a + 1], z0 = vertices[3 * a + 2];
        const float x1 = vertices[3 * b], y1 = vertices[3 * b + 1], z1 = vertices[3 * b + 2];
        const float x2 = vertices[3 * cc], y2 = vertices[3 * cc + 1], z2 = vertices[3 * cc + 2];
        int x_min = (int)floorf(fminf(x0, fminf(x1, x2))), x_max = (int)ceilf(fmaxf(x0, fmaxf(x1, x2)));      /* :253-257 */
        int y_min = (int)floorf(fminf(y0, fminf(y1, y2))), y_max = (int)ceilf(fmaxf(y0, fmaxf(y1, y2)));
        if (x_min < 0) x_min = 0;
        if (x_max > w - 1) x_max = w - 1;
        if (y_min < 0) y_min = 0;
        if (y_max > h - 1) y_max = h - 1;
        if (x_max < x_min || y_max < y_min) continue;
        for (int y = y_min; y <= y_max; y++)
            for (int x = x_min; x <= x_max; x++) {
                float u, v;
                if (!bary((float)x, (float)y, x0, y0, x1, y1, x2, y2, &u, &v)) continue;
                const float w0 = 1 - u - v, w1 = v, w2 = u;                                                    /* :81-84 */
                const float p_depth = w0 * z0 + w1 * z1 + w2 * z2;
                if (p_depth > depth_buffer[y * w + x]) {
                    unsigned char *px = image + ((reverse ? h - 1 - y : y) * w + x) * c;
                    for (int k = 0; k < c; k++) {
                        const float p_color = w0 * colors[c * a + k] + w1 * colors[c * b + k] + w2 * colors[c * cc + k];
                        px[k] = (unsigned char)((1 - alpha) * px[k] + alpha * 255 * p_color);                 /* :278-286 */
                    }
                    depth_buffer[y * w + x] = p_depth;
                }
            }
    }
}
