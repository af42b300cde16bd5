// Direct 3x3 convolution of a 3-channel (RGB) image: the stems of the backbones (EfficientNetV2-S 3 -> 24, TinyViT 3 -> 48, RepVGG-B1g2
// 3 -> 64, all stride 2; torchvision efficientnet_v2_s features[0], src/tiny_vit.py:57-70, src/repvgg.py:171).  As an implicit GEMM the
// layer has K = 27 and N = 24 .. 64: the 64x64 MFMA tile with scalar loaders ran it at 10 TF/s, far below its HBM time (96 MB in,
// 193 MB out for 160 images).  Here a thread owns one output pixel and all its channels: 27 input values in registers, the filter
// as [tap][cout] in LDS (every lane reads the same 16 bytes: a broadcast), COUT fused multiply-adds per tap.  Workgroup = 256
// consecutive output pixels = one "row tile" of the BatchNorm statistics records (same (S, Q, k, n) format and segment rule as the
// GEMM epilogue, shift k = the tile's first row).
template <int COUT, bool STATS>
__global__ __launch_bounds__(256) void rgb_conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ y, int H, int W, int OH, int OW, int stride, int pad, int64_t M,
                                                          float* __restrict__ stats, float* __restrict__ stats_extra, int64_t seg_row) {
    __shared__ __attribute__((aligned(16))) float wl[27 * COUT];
    __shared__ float red[4 * COUT * 2];
    __shared__ float kk[COUT];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 27 * COUT; i += 256) { const int tap = i / COUT, co = i - tap * COUT; wl[i] = w[co * 27 + tap]; }     // [cout][r][s][ci] -> [tap][cout]
    __syncthreads();
    const int64_t m0 = (int64_t)blockIdx.x * 256, m = m0 + t;
    const bool valid = m < M;
    const int64_t mm = valid ? m : M - 1;
    const int n = (int)(mm / ((int64_t)OH * OW));
    const int rem = (int)(mm - (int64_t)n * OH * OW);
    const int oh = rem / OW, ow = rem - oh * OW;
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = bias ? bias[c] : 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int ih = oh * stride - pad + r;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int iw = ow * stride - pad + s;
            const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
            const float* px = x + (((int64_t)n * H + (ok ? ih : 0)) * W + (ok ? iw : 0)) * 3;
            float xv[3] = {px[0], px[1], px[2]};
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float v = ok ? xv[ci] : 0.f;
                const float4* wr = reinterpret_cast<const float4*>(&wl[((r * 3 + s) * 3 + ci) * COUT]);
#pragma unroll
                for (int q = 0; q < COUT / 4; ++q) {
                    const float4 wv = wr[q];
                    acc[q * 4 + 0] += v * wv.x; acc[q * 4 + 1] += v * wv.y; acc[q * 4 + 2] += v * wv.z; acc[q * 4 + 3] += v * wv.w;
                }
            }
        }
    }
    if (valid) {
        float4* dst = reinterpret_cast<float4*>(y + m * COUT);
#pragma unroll
        for (int q = 0; q < COUT / 4; ++q) dst[q] = make_float4(acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]);
    }
    if (!STATS) return;
    if (t == 0) {
#pragma unroll
        for (int c = 0; c < COUT; ++c) kk[c] = acc[c];                   // row m0 (always valid)
    }
    __syncthreads();
    auto emit = [&](int64_t lo, int64_t hi, float4* dst0) __attribute__((always_inline)) {      // tile-local rows [lo, hi)
        const bool in = valid && t >= lo && t < hi;
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
            const float v = in ? acc[c] - kk[c] : 0.f;
            float a = v, q = v * v;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); q += __shfl_xor(q, o, 64); }
            if (lane == 0) { red[(wave * COUT + c) * 2] = a; red[(wave * COUT + c) * 2 + 1] = q; }
        }
        __syncthreads();
        if (t < COUT && hi > lo) {
            float a = 0.f, q = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) { a += red[(wv * COUT + t) * 2]; q += red[(wv * COUT + t) * 2 + 1]; }
            dst0[t] = make_float4(a, q, kk[t], (float)(hi - lo));
        }
        __syncthreads();
    };
    const int64_t nrows = M - m0 < 256 ? M - m0 : 256;
    float4* rec = reinterpret_cast<float4*>(stats) + (int64_t)blockIdx.x * COUT;
    const int64_t cut = seg_row - m0;
    if (stats_extra && cut > 0 && cut < nrows) {
        emit(0, cut, rec);
        emit(cut, nrows, reinterpret_cast<float4*>(stats_extra));
    } else {
        emit(0, nrows, rec);
    }
}

constexpr int kStemRows = 256;
static const bool kStemDirect = !(getenv("SUHPE_STEM_DIRECT") && atoi(getenv("SUHPE_STEM_DIRECT")) == 0);     // SUHPE_STEM_DIRECT=0: the implicit-GEMM path
// 3 -> 24 (EfficientNetV2-S) and 3 -> 48 (TinyViT).  RepVGG's 3 -> 64 stem stays on the GEMM: the kernel handles it (tests/test_gpu_kernels.py
// runs COUT = 64 through launch_stem's instantiation), but RepVGG's reference-golden gradient check (4 images of 64x64, BatchNorm over 64
// samples in layer3) sits on ReLU-mask flips: with a different -- equally accurate -- rounding of the stem one sampled gradient moves by
// 5 % (tools/debug/stem_ab2.py: forward activations agree to 1e-7 .. 3e-5), past that test's bar.
static bool stem_ok(const suhpe_conv_t* d) {
    static const bool all = getenv("SUHPE_STEM_DIRECT") && atoi(getenv("SUHPE_STEM_DIRECT")) == 2;       // 2: the 64-channel stem too
    return kStemDirect && d && d->groups == 1 && d->cin == 3 && d->r == 3 && d->s == 3 && (d->cout == 24 || d->cout == 48 || (all && d->cout == 64)) &&
           (int64_t)d->n * d->h * d->w * 3 < ((int64_t)1 << 31);
}
static void launch_stem(const suhpe_conv_t* d, const float* x, const float* w, const float* bias, float* y, float* stats, float* stats_extra,
                        int64_t seg_row, hipStream_t st) {
    const int64_t M = (int64_t)d->n * d->ho * d->wo;
    const dim3 grid((unsigned)ceil_div(M, kStemRows)), block(256);
#define STEM_LAUNCH(CO)                                                                                                              \
    do {                                                                                                                             \
        if (stats) hipLaunchKernelGGL((rgb_conv3x3_kernel<CO, true>), grid, block, 0, st, x, w, bias, y, d->h, d->w, d->ho, d->wo, d->stride, d->pad, M, stats, stats_extra, seg_row); \
        else hipLaunchKernelGGL((rgb_conv3x3_kernel<CO, false>), grid, block, 0, st, x, w, bias, y, d->h, d->w, d->ho, d->wo, d->stride, d->pad, M, stats, stats_extra, seg_row);    \
    } while (0)
    if (d->cout == 24) STEM_LAUNCH(24);
    else if (d->cout == 48) STEM_LAUNCH(48);
    else STEM_LAUNCH(64);
#undef STEM_LAUNCH
}
