// Squeeze-excitation MLP as two micro-kernels per direction (torchvision SqueezeExcitation inside MBConv,
// src/networks.py:303): per sample  p(c) -> a1 = W1 p + b1 -> h = SiLU(a1) -> a2 = W2 h + b2 -> s = sigmoid(a2).
// The tensors are tiny (c <= 1536, squeeze <= 64, n <= a few hundred) so a tiled GEMM is pure launch latency
// (the unfused form was 12 launches of ~20-36 us per SE block); here one workgroup owns one sample and the weight
// gradients are one pass over the n samples with 2*squeeze accumulators per thread.
#include "common.h"

namespace suhpe {

constexpr int kMaxSq = 64;

__device__ __forceinline__ float silu_f(float u) { return u / (1.f + expf(-u)); }
__device__ __forceinline__ float silu_g(float u) { const float s = 1.f / (1.f + expf(-u)); return s * (1.f + u * (1.f - s)); }

// grid = n samples.  Writes a1 (n,csq) and s (n,c).  Latency-bound: every loop keeps several independent loads in flight.
__global__ __launch_bounds__(256) void se_mlp_fwd_kernel(const float* __restrict__ p, const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2, int c, int csq,
                                                         float* __restrict__ a1, float* __restrict__ s) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* pl = sm;            // [c]
    float* hl = sm + c;        // [csq]
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t * 4; i < c; i += 1024) *reinterpret_cast<float4*>(pl + i) = *reinterpret_cast<const float4*>(p + (int64_t)n * c + i);
    __syncthreads();
    // fc1: a wavefront takes 4 squeeze outputs at a time (4 independent row streams)
    for (int j0 = wave * 4; j0 < csq; j0 += 16) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int i = lane * 4; i < c; i += 256) {
            const float4 pv = *reinterpret_cast<const float4*>(pl + i);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = *reinterpret_cast<const float4*>(w1 + (int64_t)(j0 + q) * c + i);
                acc[q] += w.x * pv.x + w.y * pv.y + w.z * pv.z + w.w * pv.w;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = wave_sum(acc[q]);
        if (lane < 4) {
            const float a = acc[lane] + b1[j0 + lane];
            a1[(int64_t)n * csq + j0 + lane] = a;
            hl[j0 + lane] = silu_f(a);
        }
    }
    __syncthreads();
    // fc2: a thread owns up to 8 output channels (t, t+256, ...), all rows streamed together
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int cc = t + 256 * k; acc[k] = cc < c ? b2[cc] : 0.f; }
#pragma unroll 4
    for (int j = 0; j < csq; j += 4) {              // unrolled: 32 independent 16-byte loads in flight per thread
        const float4 hv = *reinterpret_cast<const float4*>(hl + j);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cc = t + 256 * k;
            if (cc < c) {
                const float4 w = *reinterpret_cast<const float4*>(w2 + (int64_t)cc * csq + j);
                acc[k] += w.x * hv.x + w.y * hv.y + w.z * hv.z + w.w * hv.w;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int cc = t + 256 * k; if (cc < c) s[(int64_t)n * c + cc] = 1.f / (1.f + expf(-acc[k])); }
}

// grid = n samples.  ds (n,c) -> da2 (n,c), da1 (n,csq), h (n,csq), dp (n,c)
__global__ __launch_bounds__(256) void se_mlp_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ s, const float* __restrict__ a1,
                                                         const float* __restrict__ w1, const float* __restrict__ w2, int c, int csq,
                                                         float* __restrict__ da2, float* __restrict__ da1, float* __restrict__ hbuf,
                                                         float* __restrict__ dp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* gl = sm;                     // da2 [c]
    float* red = sm + c;                // [4][kMaxSq] partial dh
    float* dl = red + 4 * kMaxSq;       // da1 [kMaxSq]
    const int n = blockIdx.x, t = threadIdx.x;
    for (int cc = t; cc < c; cc += 256) {
        const float sv = s[(int64_t)n * c + cc];
        const float g = ds[(int64_t)n * c + cc] * sv * (1.f - sv);
        gl[cc] = g;
        da2[(int64_t)n * c + cc] = g;
    }
    __syncthreads();
    // dh[j] = sum_c w2[c][j] * da2[c]: thread = (j, quarter of the channels); rows of w2 are read coalesced across j
    {
        const int j = t & 63, q = t >> 6;
        const int cq = (c + 3) / 4, c0 = q * cq, c1 = (c0 + cq < c) ? c0 + cq : c;
        float acc = 0.f;
        if (j < csq) {
#pragma unroll 16
            for (int cc = c0; cc < c1; ++cc) acc += w2[(int64_t)cc * csq + j] * gl[cc];
        }
        red[q * kMaxSq + j] = acc;
    }
    __syncthreads();
    if (t < csq) {
        const float a = a1[(int64_t)n * csq + t];
        const float g = (red[t] + red[kMaxSq + t] + red[2 * kMaxSq + t] + red[3 * kMaxSq + t]) * silu_g(a);
        dl[t] = g;
        da1[(int64_t)n * csq + t] = g;
        hbuf[(int64_t)n * csq + t] = silu_f(a);
    }
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll 8
    for (int j = 0; j < csq; ++j) {                 // unrolled: up to 64 independent loads in flight per thread
        const float g = dl[j];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int cc = t + 256 * k;
            if (cc < c) acc[k] += w1[(int64_t)j * c + cc] * g;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int cc = t + 256 * k; if (cc < c) dp[(int64_t)n * c + cc] = acc[k]; }
}

// Weight gradients, one thread per (channel, squeeze output):
//   dW2[c][j] += sum_n da2[n][c] h[n][j];  dW1[j][c] += sum_n da1[n][j] p[n][c];  db2[c] += sum_n da2[n][c];  db1[j] += sum_n da1[n][j]
// block = 64 channels x 4 squeeze outputs; grid (c/64, csq/4).  All loads of the n-loop are independent (unrolled).
__global__ __launch_bounds__(256) void se_wgrad_kernel(const float* __restrict__ p, const float* __restrict__ hbuf, const float* __restrict__ da1,
                                                       const float* __restrict__ da2, int n, int c, int csq, float* __restrict__ dw1,
                                                       float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2) {
    const int cc = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (cc >= c || j >= csq) return;
    float a2 = 0.f, a1v = 0.f, s2 = 0.f, s1 = 0.f;
#pragma unroll 8
    for (int i = 0; i < n; ++i) {
        const float g2 = da2[(int64_t)i * c + cc], pv = p[(int64_t)i * c + cc];
        const float hv = hbuf[(int64_t)i * csq + j], g1 = da1[(int64_t)i * csq + j];
        a2 += g2 * hv;
        a1v += g1 * pv;
        s2 += g2;
        s1 += g1;
    }
    dw2[(int64_t)cc * csq + j] += a2;
    dw1[(int64_t)j * c + cc] += a1v;
    if (j == 0) db2[cc] += s2;
    if (cc == 0) db1[j] += s1;
}

}  // namespace suhpe

using namespace suhpe;

extern "C" {

int suhpe_se_mlp_fwd(const float* p, const float* w1, const float* b1, const float* w2, const float* b2, int n, int c, int csq, float* a1,
                     float* s, suhpe_stream_t stream) {
    SUHPE_REQUIRE(p && w1 && b1 && w2 && b2 && a1 && s && n > 0, "suhpe_se_mlp_fwd: null pointer or empty batch");
    SUHPE_REQUIRE(c % 4 == 0 && csq % 4 == 0 && csq <= kMaxSq && c <= 8192, "suhpe_se_mlp_fwd: needs c %% 4 == 0, squeeze %% 4 == 0, squeeze <= %d", kMaxSq);
    SUHPE_REQUIRE(c <= 2048, "suhpe_se_mlp_fwd: c <= 2048");
    hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3(n), dim3(256), (size_t)(c + csq) * sizeof(float), as_stream(stream), p, w1, b1, w2, b2, c, csq, a1, s);
    return check_launch("suhpe_se_mlp_fwd");
}

int suhpe_se_mlp_bwd(const float* ds, const float* s, const float* a1, const float* p, const float* w1, const float* w2, int n, int c, int csq,
                     float* dp, float* dw1, float* db1, float* dw2, float* db2, void* workspace, suhpe_stream_t stream) {
    SUHPE_REQUIRE(ds && s && a1 && p && w1 && w2 && dp && dw1 && db1 && dw2 && db2 && workspace && n > 0, "suhpe_se_mlp_bwd: null pointer");
    SUHPE_REQUIRE(c % 4 == 0 && csq % 4 == 0 && csq <= kMaxSq && c <= 2048, "suhpe_se_mlp_bwd: unsupported sizes");
    float* da2 = static_cast<float*>(workspace);            // (n,c)
    float* da1 = da2 + (int64_t)n * c;                      // (n,csq)
    float* hb = da1 + (int64_t)n * csq;                     // (n,csq)
    hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3(n), dim3(256), (size_t)(c + 5 * kMaxSq) * sizeof(float), as_stream(stream), ds, s, a1, w1, w2, c, csq,
                       da2, da1, hb, dp);
    hipLaunchKernelGGL(se_wgrad_kernel, dim3((c + 63) / 64, (csq + 3) / 4), dim3(256), 0, as_stream(stream), p, hb, da1, da2, n, c, csq, dw1, db1, dw2,
                       db2);
    return check_launch("suhpe_se_mlp_bwd");
}

}  // extern "C"
