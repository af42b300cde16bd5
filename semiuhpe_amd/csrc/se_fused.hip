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

// grid = n samples.  Writes a1 (n,csq) and s (n,c).
__global__ __launch_bounds__(256) void se_mlp_fwd_kernel(const float* __restrict__ p, const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2, int c, int csq,
                                                         float* __restrict__ a1, float* __restrict__ s) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* pl = sm;            // [c]
    float* hl = sm + c;        // [csq]
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < c; i += 256) pl[i] = p[(int64_t)n * c + i];
    __syncthreads();
    for (int j = wave; j < csq; j += 4) {
        float acc = 0.f;
        for (int i = lane * 4; i < c; i += 256) {
            const float4 w = *reinterpret_cast<const float4*>(w1 + (int64_t)j * c + i);
            acc += w.x * pl[i] + w.y * pl[i + 1] + w.z * pl[i + 2] + w.w * pl[i + 3];
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            const float a = acc + b1[j];
            a1[(int64_t)n * csq + j] = a;
            hl[j] = silu_f(a);
        }
    }
    __syncthreads();
    for (int cc = t; cc < c; cc += 256) {
        float acc = b2[cc];
        for (int j = 0; j < csq; j += 4) {
            const float4 w = *reinterpret_cast<const float4*>(w2 + (int64_t)cc * csq + j);
            acc += w.x * hl[j] + w.y * hl[j + 1] + w.z * hl[j + 2] + w.w * hl[j + 3];
        }
        s[(int64_t)n * c + cc] = 1.f / (1.f + expf(-acc));
    }
}

// grid = n samples.  ds (n,c) -> da2 (n,c), da1 (n,csq), dp (n,c)
__global__ __launch_bounds__(256) void se_mlp_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ s, const float* __restrict__ a1,
                                                         const float* __restrict__ w1, const float* __restrict__ w2, int c, int csq,
                                                         float* __restrict__ da2, float* __restrict__ da1, float* __restrict__ dp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* gl = sm;            // da2 [c]
    float* dl = sm + c;        // da1 [csq]
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < c; i += 256) {
        const float sv = s[(int64_t)n * c + i];
        const float g = ds[(int64_t)n * c + i] * sv * (1.f - sv);
        gl[i] = g;
        da2[(int64_t)n * c + i] = g;
    }
    __syncthreads();
    for (int j = wave; j < csq; j += 4) {
        float acc = 0.f;
        for (int i = lane; i < c; i += 64) acc += w2[(int64_t)i * csq + j] * gl[i];
        acc = wave_sum(acc);
        if (lane == 0) {
            const float g = acc * silu_g(a1[(int64_t)n * csq + j]);
            dl[j] = g;
            da1[(int64_t)n * csq + j] = g;
        }
    }
    __syncthreads();
    for (int cc = t; cc < c; cc += 256) {
        float acc = 0.f;
        for (int j = 0; j < csq; ++j) acc += w1[(int64_t)j * c + cc] * dl[j];
        dp[(int64_t)n * c + cc] = acc;
    }
}

// grid over channels (256 per block).  dW2[c][j] += sum_n da2[n][c] h[n][j];  dW1[j][c] += sum_n da1[n][j] p[n][c];
// db2[c] += sum_n da2[n][c];  db1[j] += sum_n da1[n][j] (block 0 only).
__global__ __launch_bounds__(256) void se_wgrad_kernel(const float* __restrict__ p, const float* __restrict__ a1, const float* __restrict__ da1,
                                                       const float* __restrict__ da2, int n, int c, int csq, float* __restrict__ dw1,
                                                       float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2) {
    __shared__ float hl[32 * kMaxSq], gl[32 * kMaxSq];          // h and da1 of a chunk of 32 samples
    const int t = threadIdx.x, cc = blockIdx.x * 256 + t;
    const bool ok = cc < c;
    float acc1[kMaxSq], acc2[kMaxSq], sb2 = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxSq; ++j) { acc1[j] = 0.f; acc2[j] = 0.f; }
    float sb1 = 0.f;                                              // thread j < csq of block 0 sums db1[j]
    for (int n0 = 0; n0 < n; n0 += 32) {
        const int nn = n - n0 < 32 ? n - n0 : 32;
        __syncthreads();
        for (int e = t; e < nn * csq; e += 256) {
            const float a = a1[(int64_t)n0 * csq + e];
            hl[e] = silu_f(a);
            gl[e] = da1[(int64_t)n0 * csq + e];
        }
        __syncthreads();
        if (blockIdx.x == 0 && t < csq)
            for (int i = 0; i < nn; ++i) sb1 += gl[i * csq + t];
        if (ok) {
            for (int i = 0; i < nn; ++i) {
                const float g2 = da2[(int64_t)(n0 + i) * c + cc], pv = p[(int64_t)(n0 + i) * c + cc];
                sb2 += g2;
#pragma unroll
                for (int j = 0; j < kMaxSq; ++j) {
                    if (j < csq) { acc2[j] += g2 * hl[i * csq + j]; acc1[j] += gl[i * csq + j] * pv; }
                }
            }
        }
    }
    if (ok) {
#pragma unroll
        for (int j = 0; j < kMaxSq; ++j) {
            if (j < csq) { dw2[(int64_t)cc * csq + j] += acc2[j]; dw1[(int64_t)j * c + cc] += acc1[j]; }
        }
        db2[cc] += sb2;
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
    hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3(n), dim3(256), (size_t)(c + csq) * sizeof(float), as_stream(stream), p, w1, b1, w2, b2, c, csq, a1, s);
    return check_launch("suhpe_se_mlp_fwd");
}

int suhpe_se_mlp_bwd(const float* ds, const float* s, const float* a1, const float* p, const float* w1, const float* w2, int n, int c, int csq,
                     float* dp, float* dw1, float* db1, float* dw2, float* db2, void* workspace, suhpe_stream_t stream) {
    SUHPE_REQUIRE(ds && s && a1 && p && w1 && w2 && dp && dw1 && db1 && dw2 && db2 && workspace && n > 0, "suhpe_se_mlp_bwd: null pointer");
    SUHPE_REQUIRE(c % 4 == 0 && csq % 4 == 0 && csq <= kMaxSq && c <= 8192, "suhpe_se_mlp_bwd: unsupported sizes");
    float* da2 = static_cast<float*>(workspace);            // (n,c)
    float* da1 = da2 + (int64_t)n * c;                      // (n,csq)
    hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3(n), dim3(256), (size_t)(c + csq) * sizeof(float), as_stream(stream), ds, s, a1, w1, w2, c, csq, da2,
                       da1, dp);
    hipLaunchKernelGGL(se_wgrad_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), p, a1, da1, da2, n, c, csq, dw1, db1, dw2, db2);
    return check_launch("suhpe_se_mlp_bwd");
}

}  // extern "C"
