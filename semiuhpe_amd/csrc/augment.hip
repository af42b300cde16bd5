// Device-side image augmentation of the data pipeline: the per-sample PIL / torchvision chain of
// src/datasets/dataset_300WLP.py:103-161 (`process_annotated_image`; dataset_WildHead.py:76-124 is the same chain) for a
// batch of decoded 224x224 RGB crops already in HBM:
//   flip (FLIP_LEFT_RIGHT) -> 5 % ImageFilter.BLUR -> weak RandomResizedCrop(0.8..1.25) ............ -> ToTensor, Normalize
//                                                  -> rotate(+-30 deg, NEAREST, expand) + centre crop
//                                                     -> strong RandomResizedCrop(0.6..1.5) ............ -> ToTensor, Normalize
// The random decisions (flip, blur, crop boxes, angle) are drawn on the host in the reference's RNG order and arrive as
// small per-image arrays; every pixel operation is integer arithmetic that reproduces Pillow bit for bit
// (libImaging Filter.c 5x5, Geometry.c affine_fixed 16.16 stepping, Resample.c 22-bit fixed-point triangle filter with a
// uint8 intermediate between the horizontal and the vertical pass), so the device output equals the CPU pipeline exactly.
// Output: fp32 (N, 224, 224, 3) pixel-major, the layout the backbone kernels consume (no NCHW detour).
// HBM-bound: ~150 KB read + 600 KB written per view.
#include "common.h"

namespace suhpe {

constexpr int kImg = 224;                     // the reference's crop size
constexpr int kPix = kImg * kImg;
constexpr int kResBits = 32 - 8 - 2;          // Resample.c PRECISION_BITS
constexpr int kTaps = 3;                      // triangle filter, reduction factor <= 1: at most 3 taps

// A thread owns four consecutive pixels of a row (12 bytes = three aligned dwords); block = 56 pixel groups x 4 rows,
// grid = (56 row quads, images).  The flip is folded into the source index of whoever reads the original image, the blur
// (5 % of the images) is a pre-pass that writes the flipped + blurred image to `base` for those images only, and the
// rotation is folded into the horizontal resize pass of the strong view (its taps fetch rotated pixels), so a view costs
// two kernels: horizontal resize (uint8 intermediate, as Pillow) and vertical resize + ToTensor + Normalize.
constexpr int kGroups = kImg / 4;             // 56 pixel groups per row

struct AugSrc {                               // where an image's (flipped, blurred) pixels come from
    const uint8_t* p;
    bool flip;
};
__device__ __forceinline__ AugSrc aug_source(const uint8_t* src, const uint8_t* base, const int32_t* flags, int img) {
    const int f = flags[img];
    AugSrc s;
    s.p = ((f & 2) ? base : src) + (int64_t)img * kPix * 3;     // blurred images were written to `base` already flipped
    s.flip = (f & 1) && !(f & 2);
    return s;
}
__device__ __forceinline__ const uint8_t* aug_pixel(const AugSrc& s, int y, int x) { return s.p + (y * kImg + (s.flip ? kImg - 1 - x : x)) * 3; }

__device__ __forceinline__ void store12(uint8_t* dst, const uint8_t* v) {       // dst is 4-byte aligned (x multiple of 4)
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = (uint32_t)v[4 * k] | ((uint32_t)v[4 * k + 1] << 8) | ((uint32_t)v[4 * k + 2] << 16) | ((uint32_t)v[4 * k + 3] << 24);
}

// flags: bit 0 flip, bit 1 blur.  Pre-pass for the blurred images only: base = BLUR(flip(src)) (the reference flips first);
// ring of the 5x5 neighbourhood (16 ones) / 16 rounded, the two border rows / columns copied.
__global__ __launch_bounds__(224) void aug_blur_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ flags,
                                                       uint8_t* __restrict__ base) {
    const int img = blockIdx.y, f = flags[img];
    if (!(f & 2)) return;
    const int y = blockIdx.x * 4 + threadIdx.y, x0 = threadIdx.x * 4;
    const bool flip = f & 1;
    const uint8_t* s = src + (int64_t)img * kPix * 3;
    auto at = [&](int yy, int xx, int c) { return (int)s[(yy * kImg + (flip ? kImg - 1 - xx : xx)) * 3 + c]; };
    uint8_t v[12];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int x = x0 + q;
        const bool inner = y >= 2 && y < kImg - 2 && x >= 2 && x < kImg - 2;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int sum = at(y, x, c);
            if (inner) {
                sum = 8;
#pragma unroll
                for (int k = -2; k <= 2; ++k) sum += at(y - 2, x + k, c) + at(y + 2, x + k, c);
#pragma unroll
                for (int k = -1; k <= 1; ++k) sum += at(y + k, x - 2, c) + at(y + k, x + 2, c);
                sum >>= 4;
            }
            v[q * 3 + c] = (uint8_t)sum;
        }
    }
    store12(base + ((int64_t)img * kPix + y * kImg + x0) * 3, v);
}

// tables[size][out][5] = (first input index, count, k0, k1, k2) of Resample.c for resizing `size` -> 224 (size <= 224).
// Horizontal pass over the crop box (i, j, h, w): tmp[img][r][x] for r < h.  With coef != NULL the source pixel (Y, X) of
// the (flipped, blurred) image is read through the rotation: coef per image = a0 a1 a2 a3 a4 a5 (16.16 fixed point,
// libImaging affine_fixed) x0 y0 (origin of the centre crop in the expanded canvas);
// rotated(X, Y) = in((a2 + a0 (X+x0) + a1 (Y+y0)) >> 16, (a5 + a3 (X+x0) + a4 (Y+y0)) >> 16), black outside.
template <bool ROT>
__global__ __launch_bounds__(224) void aug_resize_h_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ base,
                                                           const int32_t* __restrict__ flags, const int32_t* __restrict__ box,
                                                           const int64_t* __restrict__ coef, const int32_t* __restrict__ tables,
                                                           uint8_t* __restrict__ tmp) {
    const int img = blockIdx.y, r = blockIdx.x * 4 + threadIdx.y, x0 = threadIdx.x * 4;
    const int i = box[img * 4], j = box[img * 4 + 1], h = box[img * 4 + 2], w = box[img * 4 + 3];
    if (r >= h) return;
    const AugSrc s = aug_source(src, base, flags, img);
    int64_t a[8];
    if (ROT) {
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = coef[(int64_t)img * 8 + k];
    }
    uint8_t v[12];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int32_t* t = tables + ((int64_t)w * kImg + x0 + q) * 5;
        int acc[3] = {1 << (kResBits - 1), 1 << (kResBits - 1), 1 << (kResBits - 1)};
        for (int k = 0; k < t[1]; ++k) {
            const int Y = i + r, X = j + t[0] + k;
            int p0 = 0, p1 = 0, p2 = 0;
            if (ROT) {
                const int64_t XX = X + a[6], YY = Y + a[7];
                const int64_t xin = (a[2] + a[0] * XX + a[1] * YY) >> 16, yin = (a[5] + a[3] * XX + a[4] * YY) >> 16;
                if (xin >= 0 && xin < kImg && yin >= 0 && yin < kImg) { const uint8_t* px = aug_pixel(s, (int)yin, (int)xin); p0 = px[0]; p1 = px[1]; p2 = px[2]; }
            } else {
                const uint8_t* px = aug_pixel(s, Y, X);
                p0 = px[0]; p1 = px[1]; p2 = px[2];
            }
            acc[0] += p0 * t[2 + k]; acc[1] += p1 * t[2 + k]; acc[2] += p2 * t[2 + k];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { const int o = acc[c] >> kResBits; v[q * 3 + c] = (uint8_t)(o < 0 ? 0 : (o > 255 ? 255 : o)); }
    }
    store12(tmp + ((int64_t)img * kPix + r * kImg + x0) * 3, v);
}

// vertical pass + ToTensor + Normalize: out[img][y][x][c] = (u8 / 255 - mean[c]) / std[c]   (fp32, as torchvision computes it)
__global__ __launch_bounds__(224) void aug_resize_v_norm_kernel(const uint8_t* __restrict__ tmp, const int32_t* __restrict__ box,
                                                                const int32_t* __restrict__ tables, float* __restrict__ out,
                                                                uint8_t* __restrict__ out_u8) {
    const int img = blockIdx.y, y = blockIdx.x * 4 + threadIdx.y, x0 = threadIdx.x * 4;
    const int h = box[img * 4 + 2];
    const int32_t* t = tables + ((int64_t)h * kImg + y) * 5;
    const uint8_t* s = tmp + ((int64_t)img * kPix + t[0] * kImg + x0) * 3;
    int acc[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = 1 << (kResBits - 1);
    for (int k = 0; k < t[1]; ++k) {
        const uint32_t* row = reinterpret_cast<const uint32_t*>(s + k * kImg * 3);     // 12 bytes, 4-byte aligned
        const int kk = t[2 + k];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const uint32_t u = row[d];
#pragma unroll
            for (int bsel = 0; bsel < 4; ++bsel) acc[d * 4 + bsel] += (int)((u >> (8 * bsel)) & 0xff) * kk;
        }
    }
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    float o[12];
    uint8_t v[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        int q = acc[e] >> kResBits;
        q = q < 0 ? 0 : (q > 255 ? 255 : q);
        v[e] = (uint8_t)q;
        o[e] = ((float)q / 255.0f - mean[e % 3]) / stdv[e % 3];
    }
    float4* dst = reinterpret_cast<float4*>(out + ((int64_t)img * kPix + y * kImg + x0) * 3);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    dst[2] = make_float4(o[8], o[9], o[10], o[11]);
    if (out_u8) store12(out_u8 + ((int64_t)img * kPix + y * kImg + x0) * 3, v);
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for the triangle filter, in0 = 0, in1 = size, out = 224.
// Host code (x86-64, no FMA contraction): the doubles are the ones Pillow computes.
static void resample_table(int size, int32_t* t) {
    const double scale = (double)size / (double)kImg;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale, ss = 1.0 / filterscale;
    for (int xx = 0; xx < kImg; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > size) xmax = size;
        xmax -= xmin;
        double w[kTaps] = {0.0, 0.0, 0.0}, ww = 0.0;
        for (int x = 0; x < xmax && x < kTaps; ++x) {
            double v = (x + xmin - center + 0.5) * ss;
            if (v < 0.0) v = -v;
            w[x] = v < 1.0 ? 1.0 - v : 0.0;
            ww += w[x];
        }
        int32_t* o = t + xx * 5;
        o[0] = xmin;
        o[1] = xmax < kTaps ? xmax : kTaps;
        for (int x = 0; x < kTaps; ++x) {
            const double k = (x < xmax && ww != 0.0) ? w[x] / ww : 0.0;
            o[2 + x] = k < 0 ? (int32_t)(-0.5 + k * (1 << kResBits)) : (int32_t)(0.5 + k * (1 << kResBits));
        }
    }
}

}  // namespace suhpe

using namespace suhpe;

extern "C" {

int64_t suhpe_augment_tables_bytes(void) { return (int64_t)(kImg + 1) * kImg * 5 * sizeof(int32_t); }

int suhpe_augment_build_tables(int32_t* host_tables) {
    SUHPE_REQUIRE(host_tables, "suhpe_augment_build_tables: null pointer");
    for (int i = 0; i < kImg * 5; ++i) host_tables[i] = 0;                       // size 0: unused
    for (int size = 1; size <= kImg; ++size) resample_table(size, host_tables + (int64_t)size * kImg * 5);
    return SUHPE_OK;
}

int64_t suhpe_augment_workspace_bytes(int n) { return (int64_t)2 * n * kPix * 3; }

int suhpe_augment_views(const uint8_t* src, int n, const int32_t* flags, const int32_t* weak_box, const int64_t* rot_coef,
                        const int32_t* strong_box, const int32_t* tables, void* workspace, float* weak_out, float* strong_out,
                        uint8_t* weak_u8, uint8_t* strong_u8, suhpe_stream_t stream) {
    SUHPE_REQUIRE(src && flags && weak_box && tables && workspace && weak_out && n > 0, "suhpe_augment_views: null pointer or empty batch");
    SUHPE_REQUIRE(!strong_box == !strong_out, "suhpe_augment_views: strong_box and strong_out go together");
    uint8_t* base = static_cast<uint8_t*>(workspace);       // flipped + blurred copies of the blurred images
    uint8_t* tmp = base + (int64_t)n * kPix * 3;            // horizontal-pass intermediate (uint8, as Pillow)
    const dim3 grid(kImg / 4, n), block(kGroups, 4);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(aug_blur_kernel, grid, block, 0, st, src, flags, base);
    hipLaunchKernelGGL(aug_resize_h_kernel<false>, grid, block, 0, st, src, base, flags, weak_box, nullptr, tables, tmp);
    hipLaunchKernelGGL(aug_resize_v_norm_kernel, grid, block, 0, st, tmp, weak_box, tables, weak_out, weak_u8);
    if (strong_out) {
        if (rot_coef) hipLaunchKernelGGL(aug_resize_h_kernel<true>, grid, block, 0, st, src, base, flags, strong_box, rot_coef, tables, tmp);
        else hipLaunchKernelGGL(aug_resize_h_kernel<false>, grid, block, 0, st, src, base, flags, strong_box, nullptr, tables, tmp);
        hipLaunchKernelGGL(aug_resize_v_norm_kernel, grid, block, 0, st, tmp, strong_box, tables, strong_out, strong_u8);
    }
    return check_launch("suhpe_augment_views");
}

}  // extern "C"
