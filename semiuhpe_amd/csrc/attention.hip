// TinyViT windowed multi-head attention with gathered relative-position bias on gfx950 (src/tiny_vit.py:262-286):
//     P = softmax(q k^T * scale + bias_table[h, bias_idx[i,j]]),   out = P v
// One workgroup (4 wavefronts) per (window, head); N = ws*ws tokens (49 or 196), head_dim = 32.
// Tokens are addressed in place in the (B, H, W, C) pixel-major tensor — the window partition / reverse copies of the
// reference (tiny_vit.py:356-366) never materialise: row(b, wy, wx, iy, ix) = (b*H + wy*ws+iy)*W + wx*ws+ix.
// K and V of the window live in LDS with a 33-float row pitch (lane = key reads K[j][c]: bank (j+c)%32, conflict
// free; lane = channel reads V[j][c]: consecutive banks).  A wavefront owns one query row at a time: scores with
// lane = key, softmax by wave shuffles, P*V with lane = (channel, key half).  The attention matmuls are 5% of
// TinyViT's MACs (0.212 of 4.25 GMAC), so this first version stays on the VALU; probabilities are saved for backward.
#include "common.h"

namespace suhpe {

constexpr int DH = 32;          // head dim (TinyViT-21M: 192/6 = 384/12 = 576/18 = 32)
constexpr int PITCH = DH + 1;   // LDS row pitch in floats
constexpr int MAXN = 196;       // largest window (14x14)

struct AttnGeom {
    int B, H, W, ws, heads, noff;
    int nwy, nwx, N;
    float scale;
};

__device__ __forceinline__ int64_t token_row(const AttnGeom& g, int win, int t) {
    const int per = g.nwy * g.nwx;
    const int b = win / per, r = win - b * per;
    const int wy = r / g.nwx, wx = r - wy * g.nwx;
    const int iy = t / g.ws, ix = t - iy * g.ws;
    return ((int64_t)b * g.H + wy * g.ws + iy) * g.W + wx * g.ws + ix;
}

// cooperative load of one 32-float slice per token: dst[t*PITCH + c] = src[row(t)*ld + off + c]
__device__ __forceinline__ void load_slice(const AttnGeom& g, int win, const float* __restrict__ src, int ld, int off, float* dst) {
    for (int e = threadIdx.x; e < g.N * (DH / 4); e += blockDim.x) {
        const int t = e / (DH / 4), c4 = (e - t * (DH / 4)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + token_row(g, win, t) * ld + off + c4);
        float* d = dst + t * PITCH + c4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

__global__ __launch_bounds__(256) void window_attn_fwd_kernel(const AttnGeom g, const float* __restrict__ qkv,
                                                              const float* __restrict__ bias_table, const int32_t* __restrict__ bias_idx,
                                                              float* __restrict__ out, float* __restrict__ probs) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                       // [N][PITCH]
    float* Vs = Ks + g.N * PITCH;           // [N][PITCH]
    float* qb = Vs + g.N * PITCH;           // [4][DH]
    float* pb = qb + 4 * DH;                // [4][MAXN]
    const int win = blockIdx.x / g.heads, h = blockIdx.x - win * g.heads;
    const int ld = g.heads * 3 * DH, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    load_slice(g, win, qkv, ld, h * 3 * DH + DH, Ks);
    load_slice(g, win, qkv, ld, h * 3 * DH + 2 * DH, Vs);
    __syncthreads();
    float* q = qb + wave * DH;
    float* p = pb + wave * MAXN;
    const float* bt = bias_table + (int64_t)h * g.noff;
    for (int it = 0; it < (g.N + 3) / 4; ++it) {           // uniform trip count: the LDS hand-offs use block barriers
        const int i = it * 4 + wave;
        const bool active = i < g.N;
        const int64_t row = active ? token_row(g, win, i) : 0;
        if (active && lane < DH) q[lane] = qkv[row * ld + h * 3 * DH + lane];
        __syncthreads();
        float s[4];
        float m = -3.0e38f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = jj * 64 + lane;
            s[jj] = -3.0e38f;
            if (active && j < g.N) {
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < DH; ++c) a += q[c] * Ks[j * PITCH + c];
                s[jj] = a * g.scale + bt[bias_idx[i * g.N + j]];
                m = fmaxf(m, s[jj]);
            }
        }
        m = wave_max(m);
        float sum = 0.f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = jj * 64 + lane;
            s[jj] = (active && j < g.N) ? expf(s[jj] - m) : 0.f;
            sum += s[jj];
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = jj * 64 + lane;
            if (active && j < g.N) {
                const float pv = s[jj] * inv;
                p[j] = pv;
                if (probs) probs[((int64_t)blockIdx.x * g.N + i) * g.N + j] = pv;
            }
        }
        __syncthreads();
        // out[i][c] = sum_j p[j] V[j][c]; lane -> (c, half of the keys)
        const int c = lane & 31, half = lane >> 5;
        const int j0 = half * ((g.N + 1) / 2), j1 = half ? g.N : (g.N + 1) / 2;
        float acc = 0.f;
        for (int j = j0; j < j1; ++j) acc += p[j] * Vs[j * PITCH + c];
        acc += __shfl_xor(acc, 32, 64);
        if (active && lane < DH) out[row * (g.heads * DH) + h * DH + c] = acc;
        __syncthreads();
    }
}

// Backward.  Phase A (wave per query row): dP = dO V^T, dS = P*(dP - sum(P*dP)), dQ = scale dS K; dS is written to the
// scratch matrix.  Phase B (wave per key): dK = scale dS^T Q, dV = P^T dO from the columns of dS / P.
__global__ __launch_bounds__(256) void window_attn_bwd_kernel(const AttnGeom g, const float* __restrict__ qkv,
                                                              const float* __restrict__ probs, const float* __restrict__ dout,
                                                              float* __restrict__ dS, float* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;
    float* Ks = Qs + g.N * PITCH;
    float* Vs = Ks + g.N * PITCH;
    float* Os = Vs + g.N * PITCH;           // dO
    float* b0 = Os + g.N * PITCH;           // [4][MAXN]
    float* b1 = b0 + 4 * MAXN;              // [4][MAXN]
    const int win = blockIdx.x / g.heads, h = blockIdx.x - win * g.heads;
    const int ld = g.heads * 3 * DH, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    load_slice(g, win, qkv, ld, h * 3 * DH, Qs);
    load_slice(g, win, qkv, ld, h * 3 * DH + DH, Ks);
    load_slice(g, win, qkv, ld, h * 3 * DH + 2 * DH, Vs);
    load_slice(g, win, dout, g.heads * DH, h * DH, Os);
    __syncthreads();
    float* ds = b0 + wave * MAXN;
    const float* P = probs + (int64_t)blockIdx.x * g.N * g.N;
    float* S = dS + (int64_t)blockIdx.x * g.N * g.N;
    const int c = lane & 31, half = lane >> 5;
    for (int it = 0; it < (g.N + 3) / 4; ++it) {
        const int i = it * 4 + wave;
        const bool active = i < g.N;
        float pv[4], dp[4], D = 0.f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = jj * 64 + lane;
            pv[jj] = 0.f; dp[jj] = 0.f;
            if (active && j < g.N) {
                pv[jj] = P[i * g.N + j];
                float a = 0.f;
#pragma unroll
                for (int cc = 0; cc < DH; ++cc) a += Os[i * PITCH + cc] * Vs[j * PITCH + cc];
                dp[jj] = a;
                D += pv[jj] * a;
            }
        }
        D = wave_sum(D);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = jj * 64 + lane;
            if (active && j < g.N) {
                const float v = pv[jj] * (dp[jj] - D);
                ds[j] = v;
                S[i * g.N + j] = v;
            }
        }
        __syncthreads();
        const int j0 = half * ((g.N + 1) / 2), j1 = half ? g.N : (g.N + 1) / 2;
        float acc = 0.f;
        for (int j = j0; j < j1; ++j) acc += ds[j] * Ks[j * PITCH + c];
        acc += __shfl_xor(acc, 32, 64);
        if (active && lane < DH) dqkv[token_row(g, win, i) * ld + h * 3 * DH + c] = acc * g.scale;
        __syncthreads();
    }
    __syncthreads();                        // every dS row of this (window, head) is in memory (same CU, barrier-fenced)
    float* cs = b0 + wave * MAXN;           // column j of dS
    float* cp = b1 + wave * MAXN;           // column j of P
    for (int it = 0; it < (g.N + 3) / 4; ++it) {
        const int j = it * 4 + wave;
        const bool active = j < g.N;
        if (active)
            for (int i = lane; i < g.N; i += 64) { cs[i] = S[i * g.N + j]; cp[i] = P[i * g.N + j]; }
        __syncthreads();
        const int i0 = half * ((g.N + 1) / 2), i1 = half ? g.N : (g.N + 1) / 2;
        float dk = 0.f, dv = 0.f;
        for (int i = i0; i < i1; ++i) { dk += cs[i] * Qs[i * PITCH + c]; dv += cp[i] * Os[i * PITCH + c]; }
        dk += __shfl_xor(dk, 32, 64);
        dv += __shfl_xor(dv, 32, 64);
        if (active && lane < DH) {
            const int64_t row = token_row(g, win, j);
            dqkv[row * ld + h * 3 * DH + DH + c] = dk * g.scale;
            dqkv[row * ld + h * 3 * DH + 2 * DH + c] = dv;
        }
        __syncthreads();
    }
}

// dbias_table[h][bias_idx[i][j]] += sum over windows of dS[w][h][i][j]
__global__ __launch_bounds__(256) void window_attn_dbias_kernel(const float* __restrict__ dS, const int32_t* __restrict__ bias_idx, int nwin,
                                                                int heads, int N, int noff, float* __restrict__ dbias) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // (h, i*N + j)
    if (e >= (int64_t)heads * N * N) return;
    const int h = (int)(e / ((int64_t)N * N)), ij = (int)(e - (int64_t)h * N * N);
    float s = 0.f;
    for (int w = 0; w < nwin; ++w) s += dS[((int64_t)w * heads + h) * N * N + ij];
    atomicAdd(&dbias[h * noff + bias_idx[ij]], s);
}

static int attn_geom(AttnGeom* g, int B, int H, int W, int ws, int heads, int dh, int noff, float scale, const char* who) {
    if (dh != DH || ws <= 0 || H % ws || W % ws || ws * ws > MAXN || B <= 0 || heads <= 0) {
        set_error("%s: needs head_dim 32, window dividing the feature map and window area <= %d (got dh=%d ws=%d H=%d W=%d)", who, MAXN, dh, ws, H, W);
        return SUHPE_EINVAL;
    }
    g->B = B; g->H = H; g->W = W; g->ws = ws; g->heads = heads; g->noff = noff;
    g->nwy = H / ws; g->nwx = W / ws; g->N = ws * ws; g->scale = scale;
    return SUHPE_OK;
}

}  // namespace suhpe

using namespace suhpe;

extern "C" {

int suhpe_window_attn_fwd(const float* qkv, const float* bias_table, const int32_t* bias_idx, int B, int H, int W, int ws, int heads, int dh,
                          int noff, float scale, float* out, float* probs, suhpe_stream_t stream) {
    AttnGeom g;
    int rc = attn_geom(&g, B, H, W, ws, heads, dh, noff, scale, "suhpe_window_attn_fwd");
    if (rc) return rc;
    SUHPE_REQUIRE(qkv && bias_table && bias_idx && out, "suhpe_window_attn_fwd: null pointer");
    const size_t lds = (size_t)(2 * g.N * PITCH + 4 * DH + 4 * MAXN) * sizeof(float);
    hipLaunchKernelGGL(window_attn_fwd_kernel, dim3(B * g.nwy * g.nwx * heads), dim3(256), lds, as_stream(stream), g, qkv, bias_table, bias_idx,
                       out, probs);
    return check_launch("suhpe_window_attn_fwd");
}

int64_t suhpe_window_attn_probs_elems(int B, int H, int W, int ws, int heads) {
    if (ws <= 0) return 0;
    return (int64_t)B * (H / ws) * (W / ws) * heads * ws * ws * ws * ws;
}

int suhpe_window_attn_bwd(const float* qkv, const float* probs, const float* dout, const int32_t* bias_idx, int B, int H, int W, int ws,
                          int heads, int dh, int noff, float scale, float* dqkv, float* dbias_table, void* workspace, suhpe_stream_t stream) {
    AttnGeom g;
    int rc = attn_geom(&g, B, H, W, ws, heads, dh, noff, scale, "suhpe_window_attn_bwd");
    if (rc) return rc;
    SUHPE_REQUIRE(qkv && probs && dout && bias_idx && dqkv && workspace, "suhpe_window_attn_bwd: null pointer");
    const size_t lds = (size_t)(4 * g.N * PITCH + 8 * MAXN) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    float* dS = static_cast<float*>(workspace);
    const int nwin = B * g.nwy * g.nwx;
    hipLaunchKernelGGL(window_attn_bwd_kernel, dim3(nwin * heads), dim3(256), lds, as_stream(stream), g, qkv, probs, dout, dS, dqkv);
    if (dbias_table) {
        const int64_t total = (int64_t)heads * g.N * g.N;
        hipLaunchKernelGGL(window_attn_dbias_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, as_stream(stream), dS, bias_idx, nwin, heads,
                           g.N, noff, dbias_table);
    }
    return check_launch("suhpe_window_attn_bwd");
}

}  // extern "C"
