// Laboratory for the depthwise 3x3 / stride 1 kernels on the small feature maps of the MBConv stages (14x14, 7x7), round 3:
// the sliding-window kernel of dwconv.hip keeps ONE column step of loads in flight per thread (a dependent chain of W steps, 6 loads each)
// and reaches 2.6-3.7 TB/s; here a workgroup stages whole images of a 64-channel chunk in LDS (zero ring included), with the next
// group's loads in flight in registers while the current one is computed from LDS.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto/dw_lab.hip -o tools/proto/bin/dw_lab && tools/proto/bin/dw_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// reference: one thread per output float4
__global__ void dw_ref(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int N, int H, int W, int C) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, cv = C / 4;
    if (i >= (int64_t)N * H * W * cv) return;
    const int vc = (int)(i % cv);
    const int64_t p = i / cv;
    const int ow = (int)(p % W), oh = (int)((p / W) % H), n = (int)(p / ((int64_t)W * H));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < 3; ++r)
        for (int s = 0; s < 3; ++s) {
            const int ih = oh - 1 + r, iw = ow - 1 + s;
            if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
            const float4 v = *reinterpret_cast<const float4*>(x + (((int64_t)n * H + ih) * W + iw) * C + vc * 4);
            const float xv[4] = {v.x, v.y, v.z, v.w};
            for (int q = 0; q < 4; ++q) acc[q] += xv[q] * w[(vc * 4 + q) * 9 + r * 3 + s];
        }
    *reinterpret_cast<float4*>(y + p * C + vc * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

__device__ __forceinline__ void fma4(float4& a, const float4& x, const float4& w) { a.x += x.x * w.x; a.y += x.y * w.y; a.z += x.z * w.z; a.w += x.w * w.w; }

// LPT: interior float4s per thread and group (>= ceil(imgs*H*W / 16))
template <int LPT>
__global__ __launch_bounds__(256) void dw_tile(const float* __restrict__ src, const float* __restrict__ w, float* __restrict__ dst, int N, int H, int W,
                                               int C, int imgs, int ngroups) {
    extern __shared__ __attribute__((aligned(16))) float tile[];          // [imgs][H+2][W+2][64]
    const int t = threadIdx.x, vc = t & 15, slot = t >> 4;
    const int c0 = blockIdx.y * 64 + vc * 4;
    const bool cok = c0 < C;
    const int HP = H + 2, WP = W + 2, npix = imgs * H * W;
    // zero the whole tile once: the ring stays zero, the interior is overwritten by every group
    for (int i = t; i < imgs * HP * WP * 16; i += 256) reinterpret_cast<float4*>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        wv[k] = cok ? make_float4(w[(c0 + 0) * 9 + k], w[(c0 + 1) * 9 + k], w[(c0 + 2) * 9 + k], w[(c0 + 3) * 9 + k]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // this thread's interior pixels: flat index q = slot + 16 k over (img, h, w) -> LDS float offset / global pixel offset inside a group
    int loff[LPT], goff[LPT];
#pragma unroll
    for (int k = 0; k < LPT; ++k) {
        const int q = slot + 16 * k;
        if (q < npix) {
            const int img = q / (H * W), rem = q - img * H * W, h = rem / W, x = rem - h * W;
            loff[k] = ((img * HP + h + 1) * WP + x + 1) * 64 + vc * 4;
            goff[k] = q;
        } else { loff[k] = -1; goff[k] = 0; }
    }
    float4 pf[LPT];
    auto issue = [&](int g) __attribute__((always_inline)) {
        const int64_t n0 = (int64_t)g * imgs;
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int64_t pix = n0 * H * W + goff[k];
            const bool ok = cok && loff[k] >= 0 && pix < (int64_t)N * H * W;
            pf[k] = ok ? *reinterpret_cast<const float4*>(src + pix * C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int g = blockIdx.x;
    if (g < ngroups) issue(g);
    __syncthreads();
    const int img = slot / H, row = slot - img * H;
    const bool crow = slot < imgs * H;
    for (; g < ngroups; g += gridDim.x) {
#pragma unroll
        for (int k = 0; k < LPT; ++k)
            if (loff[k] >= 0) *reinterpret_cast<float4*>(&tile[loff[k]]) = pf[k];
        __syncthreads();
        if (g + (int)gridDim.x < ngroups) issue(g + gridDim.x);
        const int64_t n = (int64_t)g * imgs + img;
        if (crow && cok && n < N) {
            const float* base = &tile[((img * HP + row) * WP) * 64 + vc * 4];      // padded (row, col 0) of the window's top row
            float4 a[3], b[3], c[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                a[r] = *reinterpret_cast<const float4*>(base + (r * WP + 0) * 64);
                b[r] = *reinterpret_cast<const float4*>(base + (r * WP + 1) * 64);
            }
            float* out = dst + ((n * H + row) * W) * C + c0;
            for (int ow = 0; ow < W; ++ow) {
#pragma unroll
                for (int r = 0; r < 3; ++r) c[r] = *reinterpret_cast<const float4*>(base + (r * WP + ow + 2) * 64);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < 3; ++r) { fma4(acc, a[r], wv[r * 3]); fma4(acc, b[r], wv[r * 3 + 1]); fma4(acc, c[r], wv[r * 3 + 2]); }
                *reinterpret_cast<float4*>(out + (int64_t)ow * C) = acc;
#pragma unroll
                for (int r = 0; r < 3; ++r) { a[r] = b[r]; b[r] = c[r]; }
            }
        }
        __syncthreads();
    }
}

static float* g_flush;
static size_t g_flush_bytes = 1ull << 30;
__global__ void flush_read(const float4* __restrict__ p, size_t n4, float* sink) {
    float a = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 12345.678f) *sink = a;
}
__global__ void producer(float4* __restrict__ x, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float f = (float)((i * 2654435761u) >> 8 & 0xffff) * (1.f / 32768.f) - 1.f;
        x[i] = make_float4(f, -f, 0.5f * f, f + 0.25f);
    }
}
static float* g_x; static size_t g_n;
template <class F> static double time_warm(F launch, int reps = 7) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        flush_read<<<2048, 256>>>(reinterpret_cast<const float4*>(g_flush), g_flush_bytes / 16, g_flush);
        producer<<<2048, 256>>>(reinterpret_cast<float4*>(g_x), g_n / 4);
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2] * 1e3;
}

template <int LPT> static void run_tile(const char* label, const float* x, const float* w, float* y, const float* yref, int N, int H, int W, int C, int imgs, int gx) {
    const int ngroups = (N + imgs - 1) / imgs;
    if (gx > ngroups) gx = ngroups;
    const size_t lds = (size_t)imgs * (H + 2) * (W + 2) * 64 * 4;
    auto kern = dw_tile<LPT>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t n = (size_t)N * H * W * C;
    CK(hipMemset(y, 0, n * 4));
    const double us = time_warm([&] { kern<<<dim3(gx, (C + 63) / 64), 256, lds>>>(x, w, y, N, H, W, C, imgs, ngroups); });
    std::vector<float> a(n), b(n);
    CK(hipMemcpy(a.data(), y, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), yref, n * 4, hipMemcpyDeviceToHost));
    double md = 0;
    for (size_t i = 0; i < n; ++i) md = std::max(md, (double)fabsf(a[i] - b[i]));
    printf("  %-40s gx %4d lds %6zu B %8.1f us %7.0f GB/s  max|diff| %.2e\n", label, gx, lds, us, 2.0 * n * 4 / us / 1e3, md);
}

int main() {
    CK(hipMalloc(&g_flush, g_flush_bytes)); CK(hipMemset(g_flush, 0, g_flush_bytes));
    const int shapes[][4] = {{160, 14, 14, 960}, {160, 14, 14, 512}, {160, 7, 7, 1536}, {128, 14, 14, 960}};
    for (auto& s : shapes) {
        const int N = s[0], H = s[1], W = s[2], C = s[3];
        const size_t n = (size_t)N * H * W * C;
        float *x, *y, *yref, *w;
        CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&yref, n * 4)); CK(hipMalloc(&w, C * 9 * 4));
        std::vector<float> hw(C * 9);
        for (int i = 0; i < C * 9; ++i) hw[i] = (float)((i * 40503u) & 0xff) / 256.f - 0.5f;
        CK(hipMemcpy(w, hw.data(), C * 36, hipMemcpyHostToDevice));
        g_x = x; g_n = n;
        producer<<<2048, 256>>>(reinterpret_cast<float4*>(x), n / 4);
        dw_ref<<<(unsigned)((n / 4 + 255) / 256), 256>>>(x, w, yref, N, H, W, C);
        CK(hipDeviceSynchronize());
        printf("== %d x %dx%d x %d (%.1f MB per tensor)\n", N, H, W, C, n * 4 / 1e6);
        if (H == 14) {
            for (int gx : {160, 80, 40, 34, 17}) run_tile<13>("tile, 1 image per group", x, w, y, yref, N, H, W, C, 1, gx);
        } else {
            for (int gx : {80, 40, 22, 11}) run_tile<7>("tile, 2 images per group", x, w, y, yref, N, H, W, C, 2, gx);
        }
        CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(yref)); CK(hipFree(w));
    }
    return 0;
}
