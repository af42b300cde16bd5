// API-edge layout change (NCHW batch from the loader -> pixel-major NHWC) fused with the strong-view
// CutOut / CutMix (src/augments.py:12-35, 67-94).  The reference runs 2*N*3 slice-assign kernels plus a full
// gather copy of the batch and 12*N scalar D2H syncs; here one pass reads the pristine source once and writes the
// augmented destination once (out of place, so the CutMix source is never a half-written image).
//   final(i,y,x) = inside any CutMix box of i ? CO(perm[i],y,x) : CO(i,y,x)
//   CO(m,y,x)    = inside any CutOut box of m ? 0 : src(m,y,x)
#include "common.h"

namespace suhpe {

__device__ __forceinline__ bool in_boxes(const int32_t* __restrict__ boxes, int img, int holes, int y, int x) {
    if (!boxes) return false;
    bool hit = false;
    for (int j = 0; j < holes; ++j) {
        const int32_t* b = boxes + ((int64_t)img * holes + j) * 4;   // x0,y0,x1,y1
        hit |= (x >= b[0]) & (y >= b[1]) & (x < b[2]) & (y < b[3]);
    }
    return hit;
}

// src_nhwc / dst_nhwc select the element order of each side
__global__ __launch_bounds__(256) void strong_aug_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int c, int h,
                                                         int w, int src_nhwc, int dst_nhwc, const int32_t* __restrict__ cutout,
                                                         const int32_t* __restrict__ cutmix, const int32_t* __restrict__ perm,
                                                         int holes) {
    const int64_t total = (int64_t)n * h * w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)((i / w) % h), img = (int)(i / ((int64_t)w * h));
        int m = img;
        if (cutmix && in_boxes(cutmix, img, holes, y, x)) m = perm[img];
        const bool zero = in_boxes(cutout, m, holes, y, x);
        for (int ch = 0; ch < c; ++ch) {
            const int64_t so = src_nhwc ? (((int64_t)m * h + y) * w + x) * c + ch : (((int64_t)m * c + ch) * h + y) * w + x;
            const int64_t dof = dst_nhwc ? (((int64_t)img * h + y) * w + x) * c + ch : (((int64_t)img * c + ch) * h + y) * w + x;
            dst[dof] = zero ? 0.f : src[so];
        }
    }
}

}  // namespace suhpe

using namespace suhpe;

extern "C" {

int suhpe_strong_aug(const float* src, float* dst, int n, int c, int h, int w, int src_nhwc, int dst_nhwc,
                     const int32_t* cutout_boxes, const int32_t* cutmix_boxes, const int32_t* perm, int holes,
                     suhpe_stream_t stream) {
    SUHPE_REQUIRE(src && dst && src != dst && n > 0 && c > 0 && h > 0 && w > 0, "suhpe_strong_aug: bad arguments (src and dst must differ)");
    SUHPE_REQUIRE(!cutmix_boxes || perm, "suhpe_strong_aug: cutmix boxes need perm");
    SUHPE_REQUIRE(holes >= 0, "suhpe_strong_aug: negative holes");
    hipLaunchKernelGGL(strong_aug_kernel, dim3(grid_for((int64_t)n * h * w, 256)), dim3(256), 0, as_stream(stream), src, dst, n, c, h, w,
                       src_nhwc, dst_nhwc, cutout_boxes, cutmix_boxes, perm, holes);
    return check_launch("suhpe_strong_aug");
}

int suhpe_nchw_to_nhwc(const float* src, int n, int c, int h, int w, float* dst, suhpe_stream_t stream) {
    return suhpe_strong_aug(src, dst, n, c, h, w, 0, 1, nullptr, nullptr, nullptr, 0, stream);
}

}  // extern "C"
