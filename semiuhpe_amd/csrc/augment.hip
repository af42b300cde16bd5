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
// HBM-bound: ~150 KB read + 600 KB written per view; one thread per output pixel (3 channels), images on blockIdx.y.
#include "common.h"

namespace suhpe {

constexpr int kImg = 224;                     // the reference's crop size
constexpr int kPix = kImg * kImg;
constexpr int kResBits = 32 - 8 - 2;          // Resample.c PRECISION_BITS
constexpr int kTaps = 3;                      // triangle filter, reduction factor <= 1: at most 3 taps

// flags: bit 0 flip, bit 1 blur.  The blur runs on the flipped image (the reference flips first).
__global__ __launch_bounds__(256) void aug_prep_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ flags,
                                                       uint8_t* __restrict__ dst) {
    const int p = blockIdx.x * 256 + threadIdx.x, img = blockIdx.y;
    if (p >= kPix) return;
    const int y = p / kImg, x = p - y * kImg;
    const int f = flags[img];
    const bool flip = f & 1, blur = f & 2;
    const uint8_t* s = src + (int64_t)img * kPix * 3;
    auto at = [&](int yy, int xx, int c) { return (int)s[(yy * kImg + (flip ? kImg - 1 - xx : xx)) * 3 + c]; };
    uint8_t* d = dst + ((int64_t)img * kPix + p) * 3;
    if (blur && y >= 2 && y < kImg - 2 && x >= 2 && x < kImg - 2) {
        // ring of the 5x5 neighbourhood (16 ones) / 16, rounded: floor(sum/16 + 0.5)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int sum = 0;
#pragma unroll
            for (int k = -2; k <= 2; ++k) sum += at(y - 2, x + k, c) + at(y + 2, x + k, c);
#pragma unroll
            for (int k = -1; k <= 1; ++k) sum += at(y + k, x - 2, c) + at(y + k, x + 2, c);
            d[c] = (uint8_t)((sum + 8) >> 4);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) d[c] = (uint8_t)at(y, x, c);
    }
}

// coef per image: a0 a1 a2 a3 a4 a5 (16.16 fixed point, libImaging affine_fixed) x0 y0 (origin of the centre crop in the
// expanded canvas): out(x, y) = in(((a2 + a0 X + a1 Y) >> 16), ((a5 + a3 X + a4 Y) >> 16)), X = x + x0, Y = y + y0; black outside
__global__ __launch_bounds__(256) void aug_rotate_kernel(const uint8_t* __restrict__ src, const int64_t* __restrict__ coef,
                                                         uint8_t* __restrict__ dst) {
    const int p = blockIdx.x * 256 + threadIdx.x, img = blockIdx.y;
    if (p >= kPix) return;
    const int y = p / kImg, x = p - y * kImg;
    const int64_t* a = coef + (int64_t)img * 8;
    const int64_t X = x + a[6], Y = y + a[7];
    const int64_t xin = (a[2] + a[0] * X + a[1] * Y) >> 16, yin = (a[5] + a[3] * X + a[4] * Y) >> 16;
    const bool ok = xin >= 0 && xin < kImg && yin >= 0 && yin < kImg;
    const uint8_t* s = src + ((int64_t)img * kPix + (ok ? yin * kImg + xin : 0)) * 3;
    uint8_t* d = dst + ((int64_t)img * kPix + p) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) d[c] = ok ? s[c] : (uint8_t)0;
}

// tables[size][out][5] = (first input index, count, k0, k1, k2) of Resample.c for resizing `size` -> 224 (size <= 224)
// horizontal pass over the crop box (i, j, h, w): tmp[img][r][x] for r < h
__global__ __launch_bounds__(256) void aug_resize_h_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ box,
                                                           const int32_t* __restrict__ tables, uint8_t* __restrict__ tmp) {
    const int p = blockIdx.x * 256 + threadIdx.x, img = blockIdx.y;
    if (p >= kPix) return;
    const int r = p / kImg, x = p - r * kImg;
    const int i = box[img * 4], j = box[img * 4 + 1], h = box[img * 4 + 2], w = box[img * 4 + 3];
    if (r >= h) return;
    const int32_t* t = tables + ((int64_t)w * kImg + x) * 5;
    const uint8_t* s = src + ((int64_t)img * kPix + (i + r) * kImg + j + t[0]) * 3;
    uint8_t* d = tmp + ((int64_t)img * kPix + p) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int acc = 1 << (kResBits - 1);
        for (int k = 0; k < t[1]; ++k) acc += (int)s[k * 3 + c] * t[2 + k];
        acc >>= kResBits;
        d[c] = (uint8_t)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
    }
}

// vertical pass + ToTensor + Normalize: out[img][y][x][c] = (u8 / 255 - mean[c]) / std[c]   (fp32, as torchvision computes it)
__global__ __launch_bounds__(256) void aug_resize_v_norm_kernel(const uint8_t* __restrict__ tmp, const int32_t* __restrict__ box,
                                                                const int32_t* __restrict__ tables, float* __restrict__ out,
                                                                uint8_t* __restrict__ out_u8) {
    const int p = blockIdx.x * 256 + threadIdx.x, img = blockIdx.y;
    if (p >= kPix) return;
    const int y = p / kImg, x = p - y * kImg;
    const int h = box[img * 4 + 2];
    const int32_t* t = tables + ((int64_t)h * kImg + y) * 5;
    const uint8_t* s = tmp + ((int64_t)img * kPix + t[0] * kImg + x) * 3;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int acc = 1 << (kResBits - 1);
        for (int k = 0; k < t[1]; ++k) acc += (int)s[k * kImg * 3 + c] * t[2 + k];
        acc >>= kResBits;
        acc = acc < 0 ? 0 : (acc > 255 ? 255 : acc);
        const float v = (float)acc / 255.0f;
        out[((int64_t)img * kPix + p) * 3 + c] = (v - mean[c]) / stdv[c];
        if (out_u8) out_u8[((int64_t)img * kPix + p) * 3 + c] = (uint8_t)acc;
    }
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

int64_t suhpe_augment_workspace_bytes(int n) { return (int64_t)3 * n * kPix * 3; }

int suhpe_augment_views(const uint8_t* src, int n, const int32_t* flags, const int32_t* weak_box, const int64_t* rot_coef,
                        const int32_t* strong_box, const int32_t* tables, void* workspace, float* weak_out, float* strong_out,
                        uint8_t* weak_u8, uint8_t* strong_u8, suhpe_stream_t stream) {
    SUHPE_REQUIRE(src && flags && weak_box && tables && workspace && weak_out && n > 0, "suhpe_augment_views: null pointer or empty batch");
    SUHPE_REQUIRE(!strong_box == !strong_out, "suhpe_augment_views: strong_box and strong_out go together");
    uint8_t* base = static_cast<uint8_t*>(workspace);
    uint8_t* rot = base + (int64_t)n * kPix * 3;
    uint8_t* tmp = rot + (int64_t)n * kPix * 3;
    const dim3 grid((kPix + 255) / 256, n), block(256);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(aug_prep_kernel, grid, block, 0, st, src, flags, base);
    hipLaunchKernelGGL(aug_resize_h_kernel, grid, block, 0, st, base, weak_box, tables, tmp);
    hipLaunchKernelGGL(aug_resize_v_norm_kernel, grid, block, 0, st, tmp, weak_box, tables, weak_out, weak_u8);
    if (strong_out) {
        const uint8_t* sbase = base;
        if (rot_coef) {
            hipLaunchKernelGGL(aug_rotate_kernel, grid, block, 0, st, base, rot_coef, rot);
            sbase = rot;
        }
        hipLaunchKernelGGL(aug_resize_h_kernel, grid, block, 0, st, sbase, strong_box, tables, tmp);
        hipLaunchKernelGGL(aug_resize_v_norm_kernel, grid, block, 0, st, tmp, strong_box, tables, strong_out, strong_u8);
    }
    return check_launch("suhpe_augment_views");
}

}  // extern "C"
