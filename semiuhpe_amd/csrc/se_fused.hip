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

constexpr int kSeThreads = 1024;     // 16 wavefronts: the kernels are chains of dependent L2-latency rounds, so a sample's work is spread wide
constexpr int kSeWaves = kSeThreads / 64;

// grid = n samples.  Writes a1 (n,csq) and s (n,c).  Latency-bound: every loop keeps several independent loads in flight.
__global__ __launch_bounds__(kSeThreads) void se_mlp_fwd_kernel(const float* __restrict__ p, const float* __restrict__ w1,
                                                                const float* __restrict__ b1, const float* __restrict__ w2,
                                                                const float* __restrict__ b2, int c, int csq, float* __restrict__ a1,
                                                                float* __restrict__ s) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* pl = sm;            // [c]
    float* hl = sm + c;        // [csq]
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t * 4; i < c; i += kSeThreads * 4) *reinterpret_cast<float4*>(pl + i) = *reinterpret_cast<const float4*>(p + (int64_t)n * c + i);
    __syncthreads();
    // fc1: a wavefront takes 4 squeeze outputs at a time (4 independent row streams); 16 waves cover 64 outputs in one round
    for (int j0 = wave * 4; j0 < csq; j0 += kSeWaves * 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i0 = lane * 4; i0 < c; i0 += 1024) {       // 16 independent 16-byte loads in flight (clamped, not branched)
            float4 wv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256 < c ? i0 + u * 256 : c - 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) wv[u][q] = *reinterpret_cast<const float4*>(w1 + (int64_t)(j0 + q) * c + i);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = i0 + u * 256 < c;
                const float4 pv = *reinterpret_cast<const float4*>(pl + (ok ? i0 + u * 256 : 0));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float d = wv[u][q].x * pv.x + wv[u][q].y * pv.y + wv[u][q].z * pv.z + wv[u][q].w * pv.w;
                    acc[q] += ok ? d : 0.f;
                }
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
    // fc2: 16 lanes share one output channel: together they read its whole w2 row (<= 64 floats) as one coalesced 256-byte
    // request and reduce with four xor-shuffles (a lane-per-row layout fetches a full cache line per 16 useful bytes).
    // Out-of-range rows / columns read a clamped address and are dropped afterwards: no branch around the loads.
    {
        const int sub = t & 15, grp = t >> 4;               // 64 row slots per pass
        const bool okj = sub * 4 < csq;
        const int jq = okj ? sub * 4 : csq - 4;
        float4 hv = *reinterpret_cast<const float4*>(hl + jq);
        if (!okj) hv = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r0 = grp; r0 < c; r0 += 64 * 8) {
            float4 wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = r0 + u * 64 < c ? r0 + u * 64 : c - 1;
                wv[u] = *reinterpret_cast<const float4*>(w2 + (int64_t)row * csq + jq);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float d = wv[u].x * hv.x + wv[u].y * hv.y + wv[u].z * hv.z + wv[u].w * hv.w;
                d += __shfl_xor(d, 8, 64);
                d += __shfl_xor(d, 4, 64);
                d += __shfl_xor(d, 2, 64);
                d += __shfl_xor(d, 1, 64);
                const int row = r0 + u * 64;
                if (sub == 0 && row < c) s[(int64_t)n * c + row] = 1.f / (1.f + expf(-(d + b2[row])));
            }
        }
    }
}

// grid = n samples.  ds (n,c) -> da2 (n,c), da1 (n,csq), h (n,csq), dp (n,c)
__global__ __launch_bounds__(kSeThreads) void se_mlp_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ s,
                                                                const float* __restrict__ a1, const float* __restrict__ w1,
                                                                const float* __restrict__ w2, int c, int csq, float* __restrict__ da2,
                                                                float* __restrict__ da1, float* __restrict__ hbuf, float* __restrict__ dp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* gl = sm;                          // da2 [c]
    float* red = sm + c;                     // [kSeWaves][kMaxSq] partial dh
    float* dl = red + kSeWaves * kMaxSq;     // da1 [kMaxSq]
    const int n = blockIdx.x, t = threadIdx.x;
    for (int cc = t; cc < c; cc += kSeThreads) {
        const float sv = s[(int64_t)n * c + cc];
        const float g = ds[(int64_t)n * c + cc] * sv * (1.f - sv);
        gl[cc] = g;
        da2[(int64_t)n * c + cc] = g;
    }
    __syncthreads();
    // dh[j] = sum_c w2[c][j] * da2[c]: lane = j, wavefront = one sixteenth of the channels; rows of w2 are read coalesced
    // across j.  Loads go to a register block first (32 in flight), clamped instead of branched.
    {
        const int j = t & 63, q = t >> 6;
        const int jj = j < csq ? j : csq - 1;
        const int cq = (c + kSeWaves - 1) / kSeWaves, c0 = q * cq, c1 = (c0 + cq < c) ? c0 + cq : c;
        float acc0 = 0.f, acc1 = 0.f;
        for (int base = c0; base < c1; base += 32) {
            float wv[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int cc = base + u < c1 ? base + u : c1 - 1;
                wv[u] = w2[(int64_t)cc * csq + jj];
            }
#pragma unroll
            for (int u = 0; u < 32; u += 2) {
                acc0 += base + u < c1 ? wv[u] * gl[base + u] : 0.f;
                acc1 += base + u + 1 < c1 ? wv[u + 1] * gl[base + u + 1] : 0.f;
            }
        }
        red[q * kMaxSq + j] = acc0 + acc1;
    }
    __syncthreads();
    if (t < csq) {
        const float a = a1[(int64_t)n * csq + t];
        float r = 0.f;
#pragma unroll
        for (int q = 0; q < kSeWaves; ++q) r += red[q * kMaxSq + t];
        const float g = r * silu_g(a);
        dl[t] = g;
        da1[(int64_t)n * csq + t] = g;
        hbuf[(int64_t)n * csq + t] = silu_f(a);
    }
    __syncthreads();
    float acc[2] = {0.f, 0.f};
    int ccl[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) { const int cc = t + kSeThreads * k; ccl[k] = cc < c ? cc : c - 1; }
    for (int j0 = 0; j0 < csq; j0 += 16) {          // 32 independent loads in flight per thread
        float wv[2][16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int j = j0 + u < csq ? j0 + u : csq - 1;
#pragma unroll
            for (int k = 0; k < 2; ++k) wv[k][u] = w1[(int64_t)j * c + ccl[k]];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float g = j0 + u < csq ? dl[j0 + u] : 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) acc[k] += wv[k][u] * g;
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) { const int cc = t + kSeThreads * k; if (cc < c) dp[(int64_t)n * c + cc] = acc[k]; }
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
    hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3(n), dim3(kSeThreads), (size_t)(c + csq) * sizeof(float), as_stream(stream), p, w1, b1, w2, b2, c, csq, a1, s);
    return check_launch("suhpe_se_mlp_fwd");
}

int suhpe_se_mlp_bwd(const float* ds, const float* s, const float* a1, const float* p, const float* w1, const float* w2, int n, int c, int csq,
                     float* dp, float* dw1, float* db1, float* dw2, float* db2, void* workspace, suhpe_stream_t stream) {
    SUHPE_REQUIRE(ds && s && a1 && p && w1 && w2 && dp && dw1 && db1 && dw2 && db2 && workspace && n > 0, "suhpe_se_mlp_bwd: null pointer");
    SUHPE_REQUIRE(c % 4 == 0 && csq % 4 == 0 && csq <= kMaxSq && c <= 2048, "suhpe_se_mlp_bwd: unsupported sizes");
    float* da2 = static_cast<float*>(workspace);            // (n,c)
    float* da1 = da2 + (int64_t)n * c;                      // (n,csq)
    float* hb = da1 + (int64_t)n * csq;                     // (n,csq)
    hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3(n), dim3(kSeThreads), (size_t)(c + (kSeWaves + 1) * kMaxSq) * sizeof(float), as_stream(stream), ds, s, a1, w1, w2, c, csq,
                       da2, da1, hb, dp);
    hipLaunchKernelGGL(se_wgrad_kernel, dim3((c + 63) / 64, (csq + 3) / 4), dim3(256), 0, as_stream(stream), p, hb, da1, da2, n, c, csq, dw1, db1, dw2,
                       db2);
    return check_launch("suhpe_se_mlp_bwd");
}

}  // extern "C"
