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
    const int ld = g.heads * 3 * DH, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
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
            s[jj] = (active && j < g.N) ? __builtin_amdgcn_exp2f((s[jj] - m) * 1.4426950408889634f) : 0.f;
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

// ---- forward on the matrix cores.  A wavefront owns 32 query rows: S = Q K^T for the whole window stays in registers
// (NP/32 tiles of 32x32, `v_mfma_f32_32x32x2_f32`), the softmax is exact (row max / sum by xor-shuffles over the 32 lanes
// that share a row in the MFMA C layout), and each probability tile goes through a wave-private LDS tile to become the A
// operand of P V.  K rows sit in LDS as [key][32+4] (B operand of Q K^T, one ds_read_b128 per 4 MFMAs), V transposed as
// [d][NP+4] (B operand of P V).  NP = window tokens padded to a multiple of 32 (64 or 224); padded keys get probability 0.
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int KP = DH + 4;      // row pitch of the K tile and of the probability staging tile (conflict-free b128 reads)
constexpr int PP = 40;          // row pitch of the staging tile: the two lane halves (rows +4) land 32 banks apart

// Dense additive bias per head, padded to the tile grid: dense[h][i][j] = table[h][idx[i][j]] inside the window, -3e38 (the
// softmax mask) in the padding, so that the score epilogue is one coalesced unconditional load per element.
__global__ __launch_bounds__(256) void window_attn_dense_bias_kernel(const float* __restrict__ table, const int32_t* __restrict__ bias_idx, int heads,
                                                                      int N, int NP, int noff, float* __restrict__ dense) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)heads * NP * NP) return;
    const int h = (int)(e / (NP * NP)), ij = (int)(e - (int64_t)h * NP * NP), i = ij / NP, j = ij - i * NP;
    dense[e] = (i < N && j < N) ? table[h * noff + bias_idx[i * N + j]] : -3.0e38f;
}

template <int NP>
__global__ __launch_bounds__(NP == 64 ? 128 : 256, 2) void window_attn_fwd_mfma(const AttnGeom g, const float* __restrict__ qkv,
                                                                            const float* __restrict__ dense_bias, float* __restrict__ out,
                                                                            float* __restrict__ probs) {
    constexpr int QB = NP / 32;                         // query / key tiles
    constexpr int NW = NP == 64 ? 2 : 4;                // wavefronts (eight, one per query block of a 196-token window, were measured: 228 vs 221 us --
                                                        // the forward is bound by its probability stores, not by the imbalance of 7 blocks over 4 wavefronts)
    constexpr int VP = NP + 4;                          // row pitch of V^T
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                                   // [NP][KP]
    float* Vt = Ks + NP * KP;                           // [DH][VP]
    float* Pst = Vt + DH * VP;                          // [NW][32][PP]
    const int win = blockIdx.x / g.heads, hd = blockIdx.x - win * g.heads;
    const int ld = g.heads * 3 * DH, t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const int r32 = lane & 31, h = lane >> 5;
    // every qkv read is a buffer load whose offset is pushed out of range for tokens beyond the window (-> zeros): no
    // branches around the loads, so they are all in flight together
    const BufRsrc rq = make_rsrc(qkv, (int64_t)g.B * g.H * g.W * ld * 4);
    // K rows and V^T; tokens beyond the window are zero
#pragma unroll
    for (int e = t; e < NP * (DH / 4); e += NW * 64) {
        const int tok = e / (DH / 4), c4 = (e - tok * (DH / 4)) * 4;
        float4 kv, vv;
        const unsigned off = tok < g.N ? (unsigned)((token_row(g, win, tok) * ld + hd * 3 * DH + DH + c4) * 4) : kOob;
        buf_load<4>(rq, off, &kv.x);
        buf_load<4>(rq, off + (tok < g.N ? DH * 4u : 0u), &vv.x);
        *reinterpret_cast<float4*>(&Ks[tok * KP + c4]) = kv;
        Vt[(c4 + 0) * VP + tok] = vv.x; Vt[(c4 + 1) * VP + tok] = vv.y; Vt[(c4 + 2) * VP + tok] = vv.z; Vt[(c4 + 3) * VP + tok] = vv.w;
    }
    __syncthreads();
    const float* bd = dense_bias + (int64_t)hd * NP * NP;
    float* pst = Pst + wave * 32 * PP;
    for (int qb = wave; qb < QB; qb += NW) {
        // A operand of Q K^T straight from global memory: lane (row r32, half h) holds k = 8*kg + 4*h .. +3
        const int qrow = qb * 32 + r32;
        float4 qf[DH / 8];
        {
            const unsigned qoff = qrow < g.N ? (unsigned)((token_row(g, win, qrow) * ld + hd * 3 * DH + h * 4) * 4) : kOob;
#pragma unroll
            for (int kg = 0; kg < DH / 8; ++kg) buf_load<4>(rq, qoff + (qrow < g.N ? kg * 32u : 0u), &qf[kg].x);
        }
        floatx16 sc[QB];
#pragma unroll
        for (int kt = 0; kt < QB; ++kt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sc[kt][e] = 0.f;
#pragma unroll
            for (int kg = 0; kg < DH / 8; ++kg) {
                const float4 kf = *reinterpret_cast<const float4*>(&Ks[(kt * 32 + r32) * KP + kg * 8 + h * 4]);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[kg].x, kf.x, sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[kg].y, kf.y, sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[kg].z, kf.z, sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[kg].w, kf.w, sc[kt], 0, 0, 0);
            }
        }
        // C layout: column (key) = kt*32 + r32, row (query) = qb*32 + (e&3) + 8*(e>>2) + 4*h.  Scale, bias, mask, row max.
        float rmax[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) rmax[e] = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < QB; ++kt) {
            const int j = kt * 32 + r32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const float v = sc[kt][e] * g.scale + bd[i * NP + j];         // padding: -3e38 swallows the (finite) score
                sc[kt][e] = v;
                rmax[e] = fmaxf(rmax[e], v);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) rmax[e] = fmaxf(rmax[e], __shfl_xor(rmax[e], m, 64));
        }
        float rsum[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) rsum[e] = 0.f;
#pragma unroll
        for (int kt = 0; kt < QB; ++kt) {
            const int j = kt * 32 + r32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pv = j < g.N ? __builtin_amdgcn_exp2f((sc[kt][e] - rmax[e]) * 1.4426950408889634f) : 0.f;      // (v_exp_f32: 1 ulp; libm's expf is ~10 VALU instructions per probability)  rows beyond N: exp(0) = 1, never stored
                sc[kt][e] = pv;
                rsum[e] += pv;
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) rsum[e] += __shfl_xor(rsum[e], m, 64);
            rsum[e] = 1.f / rsum[e];
        }
        // P V: tile by tile through the wave's staging tile
        floatx16 oacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[e] = 0.f;
#pragma unroll
        for (int kt = 0; kt < QB; ++kt) {
            const int j = kt * 32 + r32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = (e & 3) + 8 * (e >> 2) + 4 * h;
                const float pv = sc[kt][e] * rsum[e];
                pst[rl * PP + r32] = pv;
                const int i = qb * 32 + rl;
                if (probs && i < g.N && j < g.N) probs[((int64_t)blockIdx.x * g.N + i) * g.N + j] = pv;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0): the wave's own LDS writes have landed (one wave owns the tile)
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                const float4 pf = *reinterpret_cast<const float4*>(&pst[r32 * PP + kg * 8 + h * 4]);
                const float4 vf = *reinterpret_cast<const float4*>(&Vt[r32 * VP + kt * 32 + kg * 8 + h * 4]);
                oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.x, vf.x, oacc, 0, 0, 0);
                oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.y, vf.y, oacc, 0, 0, 0);
                oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.z, vf.z, oacc, 0, 0, 0);
                oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.w, vf.w, oacc, 0, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);         // reads done before the next tile overwrites the staging tile
        }
        // O in C layout: column = channel r32, row = query
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (i < g.N) out[token_row(g, win, i) * (g.heads * DH) + hd * DH + r32] = oacc[e];
        }
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
    const int ld = g.heads * 3 * DH, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
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

// ---- backward on the matrix cores (probabilities P from the forward pass).
//   dP = dO V^T,  delta_i = sum_j P_ij dP_ij,  dS = P o (dP - delta),  dQ = scale dS K,  dK = scale dS^T Q,  dV = P^T dO
// One wavefront per 32-row block in both phases (QB blocks: 2 wavefronts for 64-token windows, 7 of 8 for 196-token ones), two wavefronts
// per SIMD on the large windows.  Round 3: the first version kept all dP and P tiles of a query block in registers (256 VGPRs + 153
// AGPRs: ONE workgroup of four wavefronts per CU, every P load used right behind its issue) and ran at 29 TF/s; now
// Phase A, a wavefront per 32 query rows, in TWO sweeps over the key tiles: sweep 1 accumulates delta alone (dP tile, P tile, multiply-add:
//   nothing kept), sweep 2 recomputes each dP tile (16 MFMAs), forms dS, writes it to the scratch matrix (for the bias gradient) and feeds
//   it, through a wave-private LDS tile, to dQ += dS K.  The P tile of the NEXT key tile is requested before the current tile's MFMAs.
// Phase B, a wavefront per 32 keys: dP / dS tiles recomputed per query block; dK, dV accumulate in the "TN" form (reduction index = query =
//   row index of both operands): dS^T from the LDS tile one value per lane, P^T from registers (the P tile is turned through the same LDS
//   tile before dS overwrites it -- one staging tile per wavefront instead of two).
// LDS: V, dO and (K, then Q in the same buffer) as [token][32+4]; one staging tile per wavefront; delta.
template <int NP>
__global__ __launch_bounds__(NP == 64 ? 128 : 512, 2) void window_attn_bwd_mfma(const AttnGeom g, const float* __restrict__ qkv,
                                                                               const float* __restrict__ probs, const float* __restrict__ dout,
                                                                               float* __restrict__ dS_out, float* __restrict__ dqkv) {
    constexpr int QB = NP / 32;
    constexpr int NW = NP == 64 ? 2 : 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* KQ = smem;                                   // [NP][KP]: K during phase A, Q during phase B
    float* Vs = KQ + NP * KP;                           // [NP][KP]
    float* Os = Vs + NP * KP;                           // [NP][KP]  dO
    float* stS = Os + NP * KP;                          // [NW][32][PP]
    float* delta = stS + NW * 32 * PP;                  // [NP]
    const int win = blockIdx.x / g.heads, hd = blockIdx.x - win * g.heads;
    const int ld = g.heads * 3 * DH, lo = g.heads * DH, t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const int r32 = lane & 31, h = lane >> 5;
    const float* P = probs + (int64_t)blockIdx.x * g.N * g.N;
    float* S = dS_out + (int64_t)blockIdx.x * g.N * g.N;
    // buffer loads with the offset pushed out of range where the element does not exist (-> 0): no branches around loads
    const BufRsrc rq = make_rsrc(qkv, (int64_t)g.B * g.H * g.W * ld * 4);
    const BufRsrc ro = make_rsrc(dout, (int64_t)g.B * g.H * g.W * lo * 4);
    const BufRsrc rp = make_rsrc(P, (int64_t)g.N * g.N * 4);
    auto load_rows = [&](BufRsrc src, int ldsrc, int off, float* dst) __attribute__((always_inline)) {
        for (int e = t; e < NP * (DH / 4); e += NW * 64) {
            const int tok = e / (DH / 4), c4 = (e - tok * (DH / 4)) * 4;
            float4 v;
            buf_load<4>(src, tok < g.N ? (unsigned)((token_row(g, win, tok) * ldsrc + off + c4) * 4) : kOob, &v.x);
            *reinterpret_cast<float4*>(&dst[tok * KP + c4]) = v;
        }
    };
    load_rows(rq, ld, hd * 3 * DH + DH, KQ);            // K
    load_rows(rq, ld, hd * 3 * DH + 2 * DH, Vs);
    load_rows(ro, lo, hd * DH, Os);
    __syncthreads();
    float* sS = stS + wave * 32 * PP;
    // the P tile (query block qb, key tile kt) in the MFMA C layout: element e of this lane = P[qb*32 + (e&3) + 8(e>>2) + 4h][kt*32 + r32]
    auto load_p = [&](int qb, int kt, float (&pv)[16]) __attribute__((always_inline)) {
        const int j = kt * 32 + r32;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            buf_load<1>(rp, ((i < g.N) & (j < g.N) & (kt < QB) & (qb < QB)) ? (unsigned)((i * g.N + j) * 4) : kOob, &pv[e]);
        }
    };
    // dP tile = dO[query block] V[key tile]^T (16 MFMAs, operands from LDS)
    auto dp_tile = [&](const float4 (&of)[DH / 8], int kt) __attribute__((always_inline)) -> floatx16 {
        floatx16 dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) dp[e] = 0.f;
#pragma unroll
        for (int kg = 0; kg < DH / 8; ++kg) {
            const float4 vf = *reinterpret_cast<const float4*>(&Vs[(kt * 32 + r32) * KP + kg * 8 + h * 4]);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(of[kg].x, vf.x, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(of[kg].y, vf.y, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(of[kg].z, vf.z, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(of[kg].w, vf.w, dp, 0, 0, 0);
        }
        return dp;
    };

    // ---------------- phase A: query blocks
    for (int qb = wave; qb < QB; qb += NW) {
        float4 of[DH / 8];
#pragma unroll
        for (int kg = 0; kg < DH / 8; ++kg) of[kg] = *reinterpret_cast<const float4*>(&Os[(qb * 32 + r32) * KP + kg * 8 + h * 4]);
        float dl[16], pa[16], pb[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) dl[e] = 0.f;
        // sweep 1: delta
        load_p(qb, 0, pa);
        for (int kt = 0; kt < QB; kt += 2) {
            load_p(qb, kt + 1, pb);
            floatx16 dp = dp_tile(of, kt);
#pragma unroll
            for (int e = 0; e < 16; ++e) dl[e] += pa[e] * dp[e];
            if (kt + 1 >= QB) break;
            load_p(qb, kt + 2, pa);
            dp = dp_tile(of, kt + 1);
#pragma unroll
            for (int e = 0; e < 16; ++e) dl[e] += pb[e] * dp[e];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) dl[e] += __shfl_xor(dl[e], m, 64);
            if (r32 == 0) delta[qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] = dl[e];
        }
        // sweep 2: dS tiles and dQ
        floatx16 dq;
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[e] = 0.f;
        auto ds_dq = [&](int kt, const float (&pv)[16]) __attribute__((always_inline)) {
            const floatx16 dp = dp_tile(of, kt);
            const int j = kt * 32 + r32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = (e & 3) + 8 * (e >> 2) + 4 * h, i = qb * 32 + rl;
                const float ds = pv[e] * (dp[e] - dl[e]);
                sS[rl * PP + r32] = ds;
                if (i < g.N && j < g.N) S[i * g.N + j] = ds;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                const float4 sf = *reinterpret_cast<const float4*>(&sS[r32 * PP + kg * 8 + h * 4]);
                const int key = kt * 32 + kg * 8 + h * 4;
                dq = __builtin_amdgcn_mfma_f32_32x32x2f32(sf.x, KQ[(key + 0) * KP + r32], dq, 0, 0, 0);
                dq = __builtin_amdgcn_mfma_f32_32x32x2f32(sf.y, KQ[(key + 1) * KP + r32], dq, 0, 0, 0);
                dq = __builtin_amdgcn_mfma_f32_32x32x2f32(sf.z, KQ[(key + 2) * KP + r32], dq, 0, 0, 0);
                dq = __builtin_amdgcn_mfma_f32_32x32x2f32(sf.w, KQ[(key + 3) * KP + r32], dq, 0, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
        };
        load_p(qb, 0, pa);
        for (int kt = 0; kt < QB; kt += 2) {
            load_p(qb, kt + 1, pb);
            ds_dq(kt, pa);
            if (kt + 1 >= QB) break;
            load_p(qb, kt + 2, pa);
            ds_dq(kt + 1, pb);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = qb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (i < g.N) dqkv[token_row(g, win, i) * ld + hd * 3 * DH + r32] = dq[e] * g.scale;
        }
    }
    __syncthreads();                                    // every wave is done with K; delta is complete
    load_rows(rq, ld, hd * 3 * DH, KQ);                 // Q
    __syncthreads();

    // ---------------- phase B: key tiles
    for (int kt = wave; kt < QB; kt += NW) {
        floatx16 dk, dv;
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
        float pa[16], pb[16];
        auto dk_dv = [&](int qb, const float (&pv)[16]) __attribute__((always_inline)) {
            float4 of[DH / 8];
#pragma unroll
            for (int kg = 0; kg < DH / 8; ++kg) of[kg] = *reinterpret_cast<const float4*>(&Os[(qb * 32 + r32) * KP + kg * 8 + h * 4]);
            const floatx16 dp = dp_tile(of, kt);
            // turn the P tile into the A-operand layout of P^T dO: lane (key r32, half h) takes P[query 2 kk + h][key] for kk = 0..15
            // (reading the tile a second time from L1 / L2 in that layout instead was measured slower: 650 vs 611 us)
#pragma unroll
            for (int e = 0; e < 16; ++e) sS[((e & 3) + 8 * (e >> 2) + 4 * h) * PP + r32] = pv[e];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            float pt[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) pt[kk] = sS[(2 * kk + h) * PP + r32];
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int rl = (e & 3) + 8 * (e >> 2) + 4 * h;
                sS[rl * PP + r32] = pv[e] * (dp[e] - delta[qb * 32 + rl]);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int q = 2 * kk + h;               // query inside the block = reduction index
                dk = __builtin_amdgcn_mfma_f32_32x32x2f32(sS[q * PP + r32], KQ[(qb * 32 + q) * KP + r32], dk, 0, 0, 0);
                dv = __builtin_amdgcn_mfma_f32_32x32x2f32(pt[kk], Os[(qb * 32 + q) * KP + r32], dv, 0, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
        };
        load_p(0, kt, pa);
        for (int qb = 0; qb < QB; qb += 2) {
            load_p(qb + 1, kt, pb);
            dk_dv(qb, pa);
            if (qb + 1 >= QB) break;
            load_p(qb + 2, kt, pa);
            dk_dv(qb + 1, pb);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (key < g.N) {
                const int64_t row = token_row(g, win, key);
                dqkv[row * ld + hd * 3 * DH + DH + r32] = dk[e] * g.scale;
                dqkv[row * ld + hd * 3 * DH + 2 * DH + r32] = dv[e];
            }
        }
    }
}

// ---- Round 5: the recomputing form.  The forward keeps two floats per query row (the row max of the scaled, biased scores and the reciprocal of
// the row's exp sum) instead of the N x N probabilities; the backward recomputes every probability tile from Q K^T and those two numbers with
// the forward's own instruction sequence, and takes delta_i = sum_j P_ij dP_ij from the identity delta_i = dO_i . O_i (O = the forward's
// output, which the proj GEMM keeps for its filter gradient anyway): the 3 x N x N probability reads of the old backward, its delta sweep and the
// forward's N x N stores (784-byte rows, the bound of the old forward) are gone.
// No LDS staging between the chained products either: whichever product feeds the next one is computed TRANSPOSED, so that its MFMA C layout
// (lane r32 = column, register e = row (e&3) + 8(e>>2) + 4h) IS the next product's A operand (lane r32 = row, step kk of the half h = reduction
// index) once the reduction index of step kk is read as (kk&3) + 8(kk>>2) + 4h -- the other operand comes from LDS one value per lane at exactly
// that row.  With the query on the lane (forward, backward phase A) the softmax statistics are per-lane scalars: two cross-half shuffles per
// query block where the old kernels ran 2 x 16 x 5 xor-shuffle steps.
//   forward / phase A (a wavefront per 32 queries):  S^T = K Q^T, dP^T = V dO^T (A operand: K / V rows of the key tile from LDS, b128; B: the
//       query block's Q / dO rows in registers)  ->  P^T, dS^T in registers = A operands of O = P V and dQ = dS K (B: V / K from LDS).
//   phase B (a wavefront per 32 keys):  S = Q K^T, dP = dO V^T (A: Q / dO rows of the query block from LDS; B: the key tile's K / V rows in
//       registers)  ->  P, dS with the key on the lane = A operands of dV = P^T dO and dK = dS^T Q (B: dO / Q from LDS); dS goes to the scratch
//       matrix for the bias gradient from here (rows of 32 consecutive keys: coalesced).
// The additive bias comes from two tile-ordered expansions of the table (one per lane assignment), so that a lane's 16 values of a tile are four
// coalesced 16-byte loads: tiledA[h][qb][kt][g][lane (r32 = query, half)][4 keys 8g + 4 half ..], tiledB[h][qb][kt][g][lane (r32 = key)][4 queries].
constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ int c_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

__global__ __launch_bounds__(256) void window_attn_tiled_bias_kernel(const float* __restrict__ table, const int32_t* __restrict__ bias_idx, int heads,
                                                                      int N, int QB, int noff, float* __restrict__ tiledA, float* __restrict__ tiledB) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)heads * QB * QB * 1024) return;
    const int q = (int)(e & 3), lane = (int)((e >> 2) & 63), gq = (int)((e >> 8) & 3);
    const int tile = (int)(e >> 10), kt = tile % QB, qb = (tile / QB) % QB, hd = tile / (QB * QB);
    const int r32 = lane & 31, hh = lane >> 5, sub = 8 * gq + 4 * hh + q;
    {
        const int i = qb * 32 + r32, j = kt * 32 + sub;
        tiledA[e] = (i < N && j < N) ? table[hd * noff + bias_idx[i * N + j]] : -3.0e38f;
    }
    {
        const int i = qb * 32 + sub, j = kt * 32 + r32;
        tiledB[e] = (i < N && j < N) ? table[hd * noff + bias_idx[i * N + j]] : -3.0e38f;
    }
}

// tok[t] = row of window token t in the (B*H*W, C) tensors, -1 beyond the window: computed once per workgroup (three integer divisions per
// token; the first version redid them for every loaded slice and every stored row: a tenth of the kernel's instructions)
template <int NP>
__device__ __forceinline__ void attn_token_table(const AttnGeom& g, int win, int* tok) {
    for (int t = threadIdx.x; t < NP; t += blockDim.x) tok[t] = t < g.N ? (int)token_row(g, win, t) : -1;
}
__device__ __forceinline__ unsigned attn_off(int row, int ld, int col) { return row >= 0 ? (unsigned)((row * ld + col) * 4) : kOob; }
// rows of one 32-float slice per token into LDS, [token][KP]; tokens beyond the window are zero
template <int NP, int THREADS>
__device__ __forceinline__ void attn_load_rows(const int* tok, BufRsrc src, int ldsrc, int off, float* dst) {
    constexpr int IT = NP * (DH / 4) / THREADS;         // 7 (224 tokens, 256 threads) or 4 (64 tokens, 128 threads): exact
    static_assert(IT * THREADS == NP * (DH / 4), "attn_load_rows: the slices must divide over the threads");
    // all row numbers, then all loads, then all LDS writes: written as one loop, the compiler kept `tok` (LDS) and `dst` (LDS) ordered against
    // each other and every slice went load -> wait -> write before the next row number was read
    int rows[IT];
    float4 v[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) rows[k] = tok[(threadIdx.x + k * THREADS) >> 3];
#pragma unroll
    for (int k = 0; k < IT; ++k) buf_load<4>(src, attn_off(rows[k], ldsrc, off + ((threadIdx.x + k * THREADS) & 7) * 4), &v[k].x);
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int e = threadIdx.x + k * THREADS;
        *reinterpret_cast<float4*>(&dst[(e >> 3) * KP + (e & 7) * 4]) = v[k];
    }
}
// store the 16 rows (c_row(e, h) of the 32-row block `blk`) x column r32 of a C-layout tile, scaled, to rows tok[] of a (rows, ld) tensor:
// buffer stores whose offset is pushed out of range beyond the window (no branch per row)
__device__ __forceinline__ void attn_store_tile(const floatx16& c, float mul, const int* tok, int blk, int h, BufRsrc dst, int ld, int col) {
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const int4 r4 = *reinterpret_cast<const int4*>(&tok[blk * 32 + 8 * gq + 4 * h]);
        const int rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(c[4 * gq + q] * mul), dst, attn_off(rr[q], ld, col), 0, 0);
    }
}
// the A / B fragment of one row held in registers: lane (row r32, half h) takes channels 8 kg + 4 h .. + 3 of its row
__device__ __forceinline__ void attn_row_frag(BufRsrc src, int row, int ld, int col, float4 (&f)[DH / 8]) {
#pragma unroll
    for (int kg = 0; kg < DH / 8; ++kg) buf_load<4>(src, attn_off(row, ld, col + kg * 8), &f[kg].x);
}
// rows(LDS)[32 x 32] * frag^T: tile[row of lds][lane's row]  (A = the LDS rows, one b128 read per four MFMAs; B = the register fragment)
__device__ __forceinline__ floatx16 attn_tile(const float* __restrict__ lds_rows, int r32, int h, const float4 (&f)[DH / 8]) {
    floatx16 c;
#pragma unroll
    for (int e = 0; e < 16; ++e) c[e] = 0.f;
#pragma unroll
    for (int kg = 0; kg < DH / 8; ++kg) {
        const float4 a = *reinterpret_cast<const float4*>(&lds_rows[r32 * KP + kg * 8 + h * 4]);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, f[kg].x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, f[kg].y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, f[kg].z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, f[kg].w, c, 0, 0, 0);
    }
    return c;
}
// acc += a^T-chained product: A = the C-layout registers of the previous product, B = rows of `lds_rows` at the C layout's row order
__device__ __forceinline__ void attn_chain(const float (&a)[16], const float* __restrict__ lds_rows, int r32, int h, floatx16& acc) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], lds_rows[c_row(kk, h) * KP + r32], acc, 0, 0, 0);
}
__device__ __forceinline__ void attn_bias_tile(const float* __restrict__ tiled, int tile, int lane, float (&b)[16]) {
    const float4* p = reinterpret_cast<const float4*>(tiled) + (int64_t)tile * 256 + lane;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const float4 v = p[gq * 64];
        b[4 * gq + 0] = v.x; b[4 * gq + 1] = v.y; b[4 * gq + 2] = v.z; b[4 * gq + 3] = v.w;
    }
}

template <int NP>
__global__ __launch_bounds__(NP == 64 ? 128 : 256, 2) void window_attn_fwd_rs(const AttnGeom g, const float* __restrict__ qkv,
                                                                          const float* __restrict__ tiledA, float* __restrict__ out,
                                                                          float2* __restrict__ rowstats) {
    constexpr int QB = NP / 32, NW = NP == 64 ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                                   // [NP][KP]
    float* Vs = Ks + NP * KP;                           // [NP][KP]
    int* tok = reinterpret_cast<int*>(Vs + NP * KP);    // [NP]
    const int win = blockIdx.x / g.heads, hd = blockIdx.x - win * g.heads;
    const int ld = g.heads * 3 * DH, lo = g.heads * DH, t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const int r32 = lane & 31, h = lane >> 5;
    const BufRsrc rq = make_rsrc(qkv, (int64_t)g.B * g.H * g.W * ld * 4);
    const BufRsrc rout = make_rsrc(out, (int64_t)g.B * g.H * g.W * lo * 4);
    attn_token_table<NP>(g, win, tok);
    __syncthreads();
    attn_load_rows<NP, NW * 64>(tok, rq, ld, hd * 3 * DH + DH, Ks);
    attn_load_rows<NP, NW * 64>(tok, rq, ld, hd * 3 * DH + 2 * DH, Vs);
    __syncthreads();
    for (int qb = wave; qb < QB; qb += NW) {
        const int qrow = qb * 32 + r32;
        float4 qf[DH / 8];
        attn_row_frag(rq, tok[qrow], ld, hd * 3 * DH + h * 4, qf);
        const int tile0 = (hd * QB + qb) * QB;
        float sc[QB][16], bt[16];
        float lmax = -3.0e38f;
        attn_bias_tile(tiledA, tile0, lane, bt);
#pragma unroll
        for (int kt = 0; kt < QB; ++kt) {
            const floatx16 st = attn_tile(Ks + kt * 32 * KP, r32, h, qf);         // [key][query = lane]
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = st[e] * g.scale + bt[e];                          // padding: -3e38 swallows the (finite) score
                sc[kt][e] = v;
                lmax = fmaxf(lmax, v);
            }
            // the next tile's bias is requested behind this tile's use of the registers and lands under the next tile's MFMAs (the fence keeps
            // the compiler from requesting all seven tiles up front: 112 more live registers, spills)
            asm volatile("" ::: "memory");
            if (kt + 1 < QB) attn_bias_tile(tiledA, tile0 + kt + 1, lane, bt);
        }
        lmax = fmaxf(lmax, __shfl_xor(lmax, 32, 64));
        float lsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < QB; ++kt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pv = __builtin_amdgcn_exp2f((sc[kt][e] - lmax) * kLog2e);      // keys beyond the window: (-3e38 - max) log2(e) = -inf -> 0
                sc[kt][e] = pv;
                lsum += pv;
            }
        }
        lsum += __shfl_xor(lsum, 32, 64);
        const float rinv = 1.f / lsum;
        floatx16 oacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[e] = 0.f;
#pragma unroll
        for (int kt = 0; kt < QB; ++kt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sc[kt][e] *= rinv;
            attn_chain(sc[kt], Vs + kt * 32 * KP, r32, h, oacc);
        }
        attn_store_tile(oacc, 1.f, tok, qb, h, rout, lo, hd * DH + r32);         // O in C layout: column = channel r32, row = query
        if (rowstats && h == 0) rowstats[(int64_t)blockIdx.x * NP + qrow] = make_float2(lmax, qrow < g.N ? rinv : 0.f);
    }
}

template <int NP>
__global__ __launch_bounds__(NP == 64 ? 128 : 256, 2) void window_attn_bwd_rs(const AttnGeom g, const float* __restrict__ qkv,
                                                                          const float* __restrict__ fwd_out, const float2* __restrict__ rowstats,
                                                                          const float* __restrict__ dout, const float* __restrict__ tiledA,
                                                                          const float* __restrict__ tiledB, float* __restrict__ dS_out,
                                                                          float* __restrict__ dqkv) {
    constexpr int QB = NP / 32, NW = NP == 64 ? 2 : 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* A0 = smem;                                   // [NP][KP]: K in phase A, Q in phase B
    float* A1 = A0 + NP * KP;                           // [NP][KP]: V in phase A, dO in phase B
    float* st_m = A1 + NP * KP;                         // [NP] row max, 1 / row sum (0 beyond the window), delta
    float* st_r = st_m + NP;
    float* st_d = st_r + NP;
    int* tok = reinterpret_cast<int*>(st_d + NP);       // [NP]
    const int win = blockIdx.x / g.heads, hd = blockIdx.x - win * g.heads;
    const int ld = g.heads * 3 * DH, lo = g.heads * DH, t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const int r32 = lane & 31, h = lane >> 5;
    float* S = dS_out + (int64_t)blockIdx.x * g.N * g.N;
    const BufRsrc rq = make_rsrc(qkv, (int64_t)g.B * g.H * g.W * ld * 4);
    const BufRsrc ro = make_rsrc(dout, (int64_t)g.B * g.H * g.W * lo * 4);
    const BufRsrc rx = make_rsrc(fwd_out, (int64_t)g.B * g.H * g.W * lo * 4);
    const BufRsrc rs = make_rsrc(S, (int64_t)g.N * g.N * 4);
    const BufRsrc rd = make_rsrc(dqkv, (int64_t)g.B * g.H * g.W * ld * 4);
    attn_token_table<NP>(g, win, tok);
    __syncthreads();
    attn_load_rows<NP, NW * 64>(tok, rq, ld, hd * 3 * DH + DH, A0);               // K
    attn_load_rows<NP, NW * 64>(tok, rq, ld, hd * 3 * DH + 2 * DH, A1);           // V
    __syncthreads();
    // ---------------- phase A: query blocks, the query on the lane
    for (int qb = wave; qb < QB; qb += NW) {
        const int qrow = qb * 32 + r32;
        const bool valid = qrow < g.N;
        const int trow = tok[qrow];
        float4 qf[DH / 8], of[DH / 8], xf[DH / 8];
        attn_row_frag(rq, trow, ld, hd * 3 * DH + h * 4, qf);
        attn_row_frag(ro, trow, lo, hd * DH + h * 4, of);
        attn_row_frag(rx, trow, lo, hd * DH + h * 4, xf);
        const float2 ms = rowstats[(int64_t)blockIdx.x * NP + qrow];
        const float rmax = ms.x, rinv = valid ? ms.y : 0.f;
        float delta = 0.f;
#pragma unroll
        for (int kg = 0; kg < DH / 8; ++kg) delta += of[kg].x * xf[kg].x + of[kg].y * xf[kg].y + of[kg].z * xf[kg].z + of[kg].w * xf[kg].w;
        delta += __shfl_xor(delta, 32, 64);
        if (h == 0) { st_m[qrow] = rmax; st_r[qrow] = rinv; st_d[qrow] = delta; }
        const int tile0 = (hd * QB + qb) * QB;
        floatx16 dq;
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[e] = 0.f;
        float bt[16];
        attn_bias_tile(tiledA, tile0, lane, bt);
#pragma unroll 1
        for (int kt = 0; kt < QB; ++kt) {
            const floatx16 st = attn_tile(A0 + kt * 32 * KP, r32, h, qf);         // S^T  [key][query = lane]
            const floatx16 dp = attn_tile(A1 + kt * 32 * KP, r32, h, of);         // dP^T
            float ds[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = st[e] * g.scale + bt[e];
                const float pv = __builtin_amdgcn_exp2f((v - rmax) * kLog2e) * rinv;      // 0 beyond the window: the bias mask (keys), rinv (queries)
                ds[e] = pv * (dp[e] - delta);
            }
            asm volatile("" ::: "memory");
            if (kt + 1 < QB) attn_bias_tile(tiledA, tile0 + kt + 1, lane, bt);     // lands under the 16 MFMAs below and the next tile's 32
            attn_chain(ds, A0 + kt * 32 * KP, r32, h, dq);                        // dQ += dS K
        }
        attn_store_tile(dq, g.scale, tok, qb, h, rd, ld, hd * 3 * DH + r32);
    }
    __syncthreads();                                    // every wave is done with K and V; the row statistics are complete
    attn_load_rows<NP, NW * 64>(tok, rq, ld, hd * 3 * DH, A0);                    // Q
    attn_load_rows<NP, NW * 64>(tok, ro, lo, hd * DH, A1);                        // dO
    __syncthreads();
    // ---------------- phase B: key tiles, the key on the lane
    for (int kt = wave; kt < QB; kt += NW) {
        const int key = kt * 32 + r32;
        const bool valid = key < g.N;
        float4 kf[DH / 8], vf[DH / 8];
        attn_row_frag(rq, tok[key], ld, hd * 3 * DH + DH + h * 4, kf);
        attn_row_frag(rq, tok[key], ld, hd * 3 * DH + 2 * DH + h * 4, vf);
        floatx16 dk, dv;
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
        // the bias tile of the NEXT query block is requested at the top of a block, ahead of this block's 16 dS stores: the wait for it (in-order
        // counter) then never covers a store.  (Requested behind the stores, every block waited for its stores to be acknowledged: the first
        // version of this kernel ran no faster than the probability-keeping one, matrix pipes 49 % busy, 45 % of the wave cycles in waits.)
        float ba[16], bb[16];
        auto block = [&](int qb, const float (&bcur)[16], float (&bnext)[16]) __attribute__((always_inline)) {
            if (qb + 1 < QB) attn_bias_tile(tiledB, (hd * QB + qb + 1) * QB + kt, lane, bnext);
            const floatx16 st = attn_tile(A0 + qb * 32 * KP, r32, h, kf);         // S   [query][key = lane]
            const floatx16 dp = attn_tile(A1 + qb * 32 * KP, r32, h, vf);         // dP
            float pv[16], ds[16];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 m4 = *reinterpret_cast<const float4*>(&st_m[qb * 32 + 8 * gq + 4 * h]);
                const float4 r4 = *reinterpret_cast<const float4*>(&st_r[qb * 32 + 8 * gq + 4 * h]);
                const float4 d4 = *reinterpret_cast<const float4*>(&st_d[qb * 32 + 8 * gq + 4 * h]);
                const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = 4 * gq + q;
                    const float v = st[e] * g.scale + bcur[e];
                    const float p1 = __builtin_amdgcn_exp2f((v - mm[q]) * kLog2e) * rr[q];      // 0 beyond the window: the bias mask (keys), rr (queries)
                    pv[e] = p1;
                    ds[e] = p1 * (dp[e] - dd[q]);
                    const int i = qb * 32 + 8 * gq + 4 * h + q;
                    // (a buffer store whose offset is pushed out of range beyond the window: no branch per element)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ds[e]), rs, (i < g.N && valid) ? (unsigned)((i * g.N + key) * 4) : kOob, 0, 0);
                }
            }
            attn_chain(ds, A0 + qb * 32 * KP, r32, h, dk);                        // dK += dS^T Q
            attn_chain(pv, A1 + qb * 32 * KP, r32, h, dv);                        // dV += P^T dO
        };
        attn_bias_tile(tiledB, hd * QB * QB + kt, lane, ba);
#pragma unroll 1
        for (int qb = 0; qb < QB; qb += 2) {
            block(qb, ba, bb);
            if (qb + 1 >= QB) break;
            block(qb + 1, bb, ba);
        }
        attn_store_tile(dk, g.scale, tok, kt, h, rd, ld, hd * 3 * DH + DH + r32);
        attn_store_tile(dv, 1.f, tok, kt, h, rd, ld, hd * 3 * DH + 2 * DH + r32);
    }
}

// dbias_table[h][bias_idx[i][j]] += sum over windows of dS[w][h][i][j]
// (blockIdx.y selects a chunk of windows: more loads in flight, and the chunks' partial sums meet in the atomic)
__global__ __launch_bounds__(256) void window_attn_dbias_kernel(const float* __restrict__ dS, const int32_t* __restrict__ bias_idx, int nwin,
                                                                int heads, int N, int noff, float* __restrict__ dbias) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // (h, i*N + j)
    if (e >= (int64_t)heads * N * N) return;
    const int h = (int)(e / ((int64_t)N * N)), ij = (int)(e - (int64_t)h * N * N);
    const int per = (nwin + gridDim.y - 1) / gridDim.y, w0 = blockIdx.y * per, w1 = min(nwin, w0 + per);
    float s = 0.f;
#pragma unroll 8
    for (int w = w0; w < w1; ++w) s += dS[((int64_t)w * heads + h) * N * N + ij];
    if (w1 > w0) atomicAdd(&dbias[h * noff + bias_idx[ij]], s);
}

// ---- the same gradient without atomics (bit-reproducible): stage 1 sums dS over chunks of windows in a fixed order,
//   part[chunk][h][ij] = sum_{w in chunk} dS[w][h][ij];
// stage 2 gives a thread one table entry (h, offset) and adds, in a fixed order, the (i, j) pairs that use that offset (their
// list comes from the inverse of bias_idx, built once per attention module on the host) over all chunks.
__global__ __launch_bounds__(256) void window_attn_dbias_part_kernel(const float* __restrict__ dS, int nwin, int heads, int N, float* __restrict__ part) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, per_head = (int64_t)N * N, total = heads * per_head;
    if (e >= total) return;
    const int per = (nwin + gridDim.y - 1) / gridDim.y, w0 = blockIdx.y * per, w1 = min(nwin, w0 + per);
    float s = 0.f;
#pragma unroll 8
    for (int w = w0; w < w1; ++w) s += dS[(int64_t)w * total + e];
    part[(int64_t)blockIdx.y * total + e] = s;
}
// one wavefront per table entry (h, offset): lane l adds the pairs l, l + 64, ... of the entry's list over all chunks, then the 64 lane
// sums meet in a fixed xor-shuffle tree (a single thread per entry walking 196 pairs x 8 chunks took ~0.7 ms per layer)
__global__ __launch_bounds__(256) void window_attn_dbias_gather_kernel(const float* __restrict__ part, int chunks, int heads, int N, int noff,
                                                                      const int32_t* __restrict__ inv_ptr, const int32_t* __restrict__ inv_list,
                                                                      float* __restrict__ dbias) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= heads * noff) return;
    const int h = e / noff, off = e - h * noff;
    const int64_t per_head = (int64_t)N * N, total = heads * per_head;
    float s = 0.f;
    for (int q = inv_ptr[off] + lane; q < inv_ptr[off + 1]; q += 64) {
        const int64_t at = h * per_head + inv_list[q];
        for (int c = 0; c < chunks; ++c) s += part[c * total + at];
    }
    s = wave_sum(s);
    if (lane == 0) dbias[e] += s;
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

int64_t suhpe_window_attn_fwd_workspace_bytes(int ws, int heads) {
    const int n = ws * ws, np = n <= 64 ? 64 : 224;
    return (n <= 224 && heads > 0) ? (int64_t)heads * np * np * (int64_t)sizeof(float) : 0;
}

int suhpe_window_attn_fwd(const float* qkv, const float* bias_table, const int32_t* bias_idx, int B, int H, int W, int ws, int heads, int dh,
                          int noff, float scale, float* out, float* probs, void* workspace, suhpe_stream_t stream) {
    AttnGeom g;
    int rc = attn_geom(&g, B, H, W, ws, heads, dh, noff, scale, "suhpe_window_attn_fwd");
    if (rc) return rc;
    SUHPE_REQUIRE(qkv && bias_table && bias_idx && out, "suhpe_window_attn_fwd: null pointer");
    const dim3 grid(B * g.nwy * g.nwx * heads);
    float* dense = static_cast<float*>(workspace);
    if (g.N <= 224) {
        SUHPE_REQUIRE(dense, "suhpe_window_attn_fwd: workspace of suhpe_window_attn_fwd_workspace_bytes() is required");
        SUHPE_REQUIRE((int64_t)B * H * W * heads * 3 * DH * 4 < 0x7fffffffLL, "suhpe_window_attn_fwd: qkv must be smaller than 2 GiB");
        const int np = g.N <= 64 ? 64 : 224;
        hipLaunchKernelGGL(window_attn_dense_bias_kernel, dim3((unsigned)ceil_div((int64_t)heads * np * np, 256)), dim3(256), 0, as_stream(stream),
                           bias_table, bias_idx, heads, g.N, np, noff, dense);
    }
    if (g.N <= 64) {
        const size_t lds = (size_t)(64 * KP + DH * (64 + 4) + 2 * 32 * PP) * sizeof(float);
        hipLaunchKernelGGL(window_attn_fwd_mfma<64>, grid, dim3(128), lds, as_stream(stream), g, qkv, dense, out, probs);
    } else if (g.N <= 224) {
        const size_t lds = (size_t)(224 * KP + DH * (224 + 4) + 4 * 32 * PP) * sizeof(float);
        // (per-device attribute: set before every launch, no process-wide flag)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_fwd_mfma<224>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(window_attn_fwd_mfma<224>, grid, dim3(256), lds, as_stream(stream), g, qkv, dense, out, probs);
    } else {
        const size_t lds = (size_t)(2 * g.N * PITCH + 4 * DH + 4 * MAXN) * sizeof(float);
        hipLaunchKernelGGL(window_attn_fwd_kernel, grid, dim3(256), lds, as_stream(stream), g, qkv, bias_table, bias_idx, out, probs);
    }
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
    SUHPE_REQUIRE((int64_t)B * H * W * heads * 3 * DH * 4 < 0x7fffffffLL, "suhpe_window_attn_bwd: qkv must be smaller than 2 GiB");
    const size_t lds = (size_t)(4 * g.N * PITCH + 8 * MAXN) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   // per device
    float* dS = static_cast<float*>(workspace);
    const int nwin = B * g.nwy * g.nwx;
    if (g.N <= 64) {
        const size_t l2 = (size_t)(3 * 64 * KP + 2 * 32 * PP + 64) * sizeof(float);
        hipLaunchKernelGGL(window_attn_bwd_mfma<64>, dim3(nwin * heads), dim3(128), l2, as_stream(stream), g, qkv, probs, dout, dS, dqkv);
    } else if (g.N <= 224) {
        const size_t l2 = (size_t)(3 * 224 * KP + 8 * 32 * PP + 224) * sizeof(float);
        // (the attribute is per device: set before every launch, no process-wide flag to go stale on a second GPU)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_bwd_mfma<224>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(window_attn_bwd_mfma<224>, dim3(nwin * heads), dim3(512), l2, as_stream(stream), g, qkv, probs, dout, dS, dqkv);
    } else {
        hipLaunchKernelGGL(window_attn_bwd_kernel, dim3(nwin * heads), dim3(256), lds, as_stream(stream), g, qkv, probs, dout, dS, dqkv);
    }
    if (dbias_table) {
        const int64_t total = (int64_t)heads * g.N * g.N;
        const int chunks = nwin >= 64 ? 8 : (nwin >= 8 ? 2 : 1);
        hipLaunchKernelGGL(window_attn_dbias_kernel, dim3((unsigned)ceil_div(total, 256), chunks), dim3(256), 0, as_stream(stream), dS, bias_idx, nwin,
                           heads, g.N, noff, dbias_table);
    }
    return check_launch("suhpe_window_attn_bwd");
}

// ---- the recomputing form (row statistics instead of probabilities)
int64_t suhpe_window_attn_rs_workspace_bytes(int ws, int heads) {
    const int n = ws * ws, np = n <= 64 ? 64 : 224;
    return (n <= 224 && heads > 0) ? (int64_t)2 * heads * np * np * (int64_t)sizeof(float) : 0;
}

int64_t suhpe_window_attn_rowstats_elems(int B, int H, int W, int ws, int heads) {
    if (ws <= 0 || ws * ws > 224) return 0;
    const int np = ws * ws <= 64 ? 64 : 224;
    return (int64_t)B * (H / ws) * (W / ws) * heads * np * 2;
}

static void attn_tiled_bias(const AttnGeom& g, const float* bias_table, const int32_t* bias_idx, int heads, int noff, float* tiled, hipStream_t st) {
    const int np = g.N <= 64 ? 64 : 224, qb = np / 32;
    const int64_t total = (int64_t)heads * np * np;
    hipLaunchKernelGGL(window_attn_tiled_bias_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, bias_table, bias_idx, heads, g.N, qb, noff,
                       tiled, tiled + total);
}

int suhpe_window_attn_fwd_rs(const float* qkv, const float* bias_table, const int32_t* bias_idx, int B, int H, int W, int ws, int heads, int dh,
                             int noff, float scale, float* out, float* rowstats, void* workspace, suhpe_stream_t stream) {
    AttnGeom g;
    int rc = attn_geom(&g, B, H, W, ws, heads, dh, noff, scale, "suhpe_window_attn_fwd_rs");
    if (rc) return rc;
    SUHPE_REQUIRE(qkv && bias_table && bias_idx && out && workspace, "suhpe_window_attn_fwd_rs: null pointer");
    SUHPE_REQUIRE((int64_t)B * H * W * heads * 3 * DH * 4 < 0x7fffffffLL, "suhpe_window_attn_fwd_rs: qkv must be smaller than 2 GiB");
    float* tiled = static_cast<float*>(workspace);
    attn_tiled_bias(g, bias_table, bias_idx, heads, noff, tiled, as_stream(stream));
    const dim3 grid(B * g.nwy * g.nwx * heads);
    if (g.N <= 64) {
        hipLaunchKernelGGL(window_attn_fwd_rs<64>, grid, dim3(128), (size_t)(2 * 64 * KP + 64) * sizeof(float), as_stream(stream), g, qkv, tiled, out,
                           reinterpret_cast<float2*>(rowstats));
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_fwd_rs<224>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   // per device
        hipLaunchKernelGGL(window_attn_fwd_rs<224>, grid, dim3(256), (size_t)(2 * 224 * KP + 224) * sizeof(float), as_stream(stream), g, qkv, tiled, out,
                           reinterpret_cast<float2*>(rowstats));
    }
    return check_launch("suhpe_window_attn_fwd_rs");
}

int suhpe_window_attn_bwd_rs(const float* qkv, const float* fwd_out, const float* rowstats, const float* dout, const float* bias_table,
                             const int32_t* bias_idx, int B, int H, int W, int ws, int heads, int dh, int noff, float scale, float* dqkv,
                             float* dbias_table, void* dS_workspace, void* bias_workspace, suhpe_stream_t stream) {
    AttnGeom g;
    int rc = attn_geom(&g, B, H, W, ws, heads, dh, noff, scale, "suhpe_window_attn_bwd_rs");
    if (rc) return rc;
    SUHPE_REQUIRE(qkv && fwd_out && rowstats && dout && bias_table && bias_idx && dqkv && dS_workspace && bias_workspace,
                  "suhpe_window_attn_bwd_rs: null pointer");
    SUHPE_REQUIRE((int64_t)B * H * W * heads * 3 * DH * 4 < 0x7fffffffLL, "suhpe_window_attn_bwd_rs: qkv must be smaller than 2 GiB");
    float* tiled = static_cast<float*>(bias_workspace);
    float* dS = static_cast<float*>(dS_workspace);
    attn_tiled_bias(g, bias_table, bias_idx, heads, noff, tiled, as_stream(stream));
    const int np = g.N <= 64 ? 64 : 224, nwin = B * g.nwy * g.nwx;
    const float* tiledB = tiled + (int64_t)heads * np * np;
    const float2* rs = reinterpret_cast<const float2*>(rowstats);
    if (g.N <= 64) {
        hipLaunchKernelGGL(window_attn_bwd_rs<64>, dim3(nwin * heads), dim3(128), (size_t)(2 * 64 * KP + 4 * 64) * sizeof(float), as_stream(stream), g, qkv,
                           fwd_out, rs, dout, tiled, tiledB, dS, dqkv);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_bwd_rs<224>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   // per device
        hipLaunchKernelGGL(window_attn_bwd_rs<224>, dim3(nwin * heads), dim3(256), (size_t)(2 * 224 * KP + 4 * 224) * sizeof(float), as_stream(stream), g,
                           qkv, fwd_out, rs, dout, tiled, tiledB, dS, dqkv);
    }
    if (dbias_table) {
        const int64_t total = (int64_t)heads * g.N * g.N;
        const int chunks = nwin >= 64 ? 8 : (nwin >= 8 ? 2 : 1);
        hipLaunchKernelGGL(window_attn_dbias_kernel, dim3((unsigned)ceil_div(total, 256), chunks), dim3(256), 0, as_stream(stream), dS, bias_idx, nwin,
                           heads, g.N, noff, dbias_table);
    }
    return check_launch("suhpe_window_attn_bwd_rs");
}

int64_t suhpe_window_attn_dbias_workspace_bytes(int ws, int heads) { return (int64_t)8 * heads * ws * ws * ws * ws * (int64_t)sizeof(float); }

int suhpe_window_attn_dbias(const float* dS, const int32_t* inv_ptr, const int32_t* inv_list, int B, int H, int W, int ws, int heads, int noff,
                            float* dbias_table, void* workspace, suhpe_stream_t stream) {
    SUHPE_REQUIRE(dS && inv_ptr && inv_list && dbias_table && workspace && ws > 0 && B > 0 && heads > 0 && noff > 0 && H % ws == 0 && W % ws == 0,
                  "suhpe_window_attn_dbias: bad arguments");
    const int N = ws * ws, nwin = B * (H / ws) * (W / ws);
    const int chunks = nwin >= 64 ? 8 : (nwin >= 8 ? 2 : 1);
    const int64_t total = (int64_t)heads * N * N;
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(window_attn_dbias_part_kernel, dim3((unsigned)ceil_div(total, 256), chunks), dim3(256), 0, as_stream(stream), dS, nwin, heads, N, part);
    hipLaunchKernelGGL(window_attn_dbias_gather_kernel, dim3((heads * noff + 3) / 4), dim3(256), 0, as_stream(stream), part, chunks, heads, N, noff,
                       inv_ptr, inv_list, dbias_table);
    return check_launch("suhpe_window_attn_dbias");
}

}  // extern "C"
