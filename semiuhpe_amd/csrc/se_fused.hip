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
    for (int j = 0; j < csq; j += 4) {
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
    float* dl = sm;                 // da1 [kMaxSq]
    float* red = sm + kMaxSq;       // [4 waves][kMaxSq] partial dh
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // dh[j] = sum_c w2[c][j] * da2[c]: each thread folds its channels into csq register partials (rows of w2 are contiguous)
    float part[kMaxSq];
#pragma unroll
    for (int j = 0; j < kMaxSq; ++j) part[j] = 0.f;
    for (int cc = t; cc < c; cc += 256) {
        const float sv = s[(int64_t)n * c + cc];
        const float g = ds[(int64_t)n * c + cc] * sv * (1.f - sv);
        da2[(int64_t)n * c + cc] = g;
#pragma unroll
        for (int j = 0; j < kMaxSq; j += 4) {
            if (j < csq) {
                const float4 w = *reinterpret_cast<const float4*>(w2 + (int64_t)cc * csq + j);
                part[j] += w.x * g; part[j + 1] += w.y * g; part[j + 2] += w.z * g; part[j + 3] += w.w * g;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kMaxSq; ++j) {
        if (j < csq) {
            const float v = wave_sum(part[j]);
            if (lane == 0) red[wave * kMaxSq + j] = v;
        }
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
    for (int j = 0; j < csq; ++j) {
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

// Weight gradients.  A block owns 64 channels; thread = (channel, quarter of the squeeze outputs); one pass over n.
//   dW2[c][j] += sum_n da2[n][c] h[n][j];  dW1[j][c] += sum_n da1[n][j] p[n][c];  db2[c] += sum_n da2[n][c];  db1[j] += sum_n da1[n][j]
__global__ __launch_bounds__(256) void se_wgrad_kernel(const float* __restrict__ p, const float* __restrict__ hbuf, const float* __restrict__ da1,
                                                       const float* __restrict__ da2, int n, int c, int csq, float* __restrict__ dw1,
                                                       float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2) {
    constexpr int JQ = kMaxSq / 4;                               // up to 16 squeeze outputs per thread
    __shared__ float hl[32 * kMaxSq], gl[32 * kMaxSq];           // h and da1 of a chunk of 32 samples
    const int t = threadIdx.x, cl = t & 63, q = t >> 6;
    const int cc = blockIdx.x * 64 + cl;
    const bool ok = cc < c;
    const int jn = (csq + 3) / 4;                                 // squeeze outputs per quarter
    const int jbase = q * jn;
    float acc1[JQ], acc2[JQ], sb2 = 0.f, sb1 = 0.f;
#pragma unroll
    for (int j = 0; j < JQ; ++j) { acc1[j] = 0.f; acc2[j] = 0.f; }
    for (int n0 = 0; n0 < n; n0 += 32) {
        const int nn = n - n0 < 32 ? n - n0 : 32;
        __syncthreads();
        for (int e = t; e < nn * csq; e += 256) { hl[e] = hbuf[(int64_t)n0 * csq + e]; gl[e] = da1[(int64_t)n0 * csq + e]; }
        __syncthreads();
        if (blockIdx.x == 0 && t < csq)
            for (int i = 0; i < nn; ++i) sb1 += gl[i * csq + t];
        if (ok) {
            for (int i = 0; i < nn; ++i) {
                const float g2 = da2[(int64_t)(n0 + i) * c + cc], pv = p[(int64_t)(n0 + i) * c + cc];
                sb2 += g2;
#pragma unroll
                for (int j = 0; j < JQ; ++j) {
                    if (j < jn && jbase + j < csq) { acc2[j] += g2 * hl[i * csq + jbase + j]; acc1[j] += gl[i * csq + jbase + j] * pv; }
                }
            }
        }
    }
    if (ok) {
#pragma unroll
        for (int j = 0; j < JQ; ++j) {
            if (j < jn && jbase + j < csq) { dw2[(int64_t)cc * csq + jbase + j] += acc2[j]; dw1[(int64_t)(jbase + j) * c + cc] += acc1[j]; }
        }
        if (q == 0) db2[cc] += sb2;
    }
    if (blockIdx.x == 0 && t < csq) db1[t] += sb1;
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
    hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3(n), dim3(256), (size_t)(5 * kMaxSq) * sizeof(float), as_stream(stream), ds, s, a1, w1, w2, c, csq,
                       da2, da1, hb, dp);
    hipLaunchKernelGGL(se_wgrad_kernel, dim3((c + 63) / 64), dim3(256), 0, as_stream(stream), p, hb, da1, da2, n, c, csq, dw1, db1, dw2, db2);
    return check_launch("suhpe_se_mlp_bwd");
}

}  // extern "C"
