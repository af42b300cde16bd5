// Prototype: fp32 GEMM C[M][N] = A[M][K] * B[N][K]^T on the bf16 matrix cores by operand splitting.
// Every fp32 operand is written as x = h + m + l with h, m, l bf16 (8 significant bits each, round-to-nearest at each step:
// |m| <= 2^-8 |x|, |l| <= 2^-16 |x|, and the three-way sum restores x to <= 2^-24 |x|); a product a*b is the sum of the six
// largest of the nine cross terms (hh, hm, mh, hl, lh, mm), each EXACT in the fp32 accumulator of v_mfma_f32_32x32x16_bf16; the
// dropped terms (ml, lm, ll) are <= 2^-24 |a||b| each.  NPROD = 9 keeps all nine.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/proto/split_gemm.hip -L semiuhpe_amd/csrc -lsuhpe -o /tmp/split_gemm
// Measures against the library's fp32-MFMA kernel (suhpe_conv2d_fwd on the same 1x1-conv shape) and a double reference.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <type_traits>
#include "suhpe.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ void split2(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = pk_bf16(ra, rb);
    l = pk_bf16(ra - __uint_as_float(m << 16), rb - __uint_as_float(m & 0xffff0000u));
}

constexpr int BK = 32, PITCH = 80;      // bytes per LDS row (64 of data + 16 of padding)

template <int NPROD, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128) ? 2 : 3) void split_gemm_nt(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    constexpr int PLA = BM * PITCH, PLB = BN * PITCH, TM = WM / 32, TN = WN / 32, WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) uint8_t lds[3 * PLA + 3 * PLB];
    uint8_t* la = lds;
    uint8_t* lb = lds + 3 * PLA;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int r32 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kq = t & 7, lr = t >> 3;
    float4 ra[BM / 32], rb[BN / 32];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < BM / 32; ++i) {
            const int row = lr + 32 * i;
            ra[i] = (m0 + row < M) ? *reinterpret_cast<const float4*>(A + (int64_t)(m0 + row) * K + k0 + kq * 4) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            const int row = lr + 32 * i;
            rb[i] = (n0 + row < N) ? *reinterpret_cast<const float4*>(B + (int64_t)(n0 + row) * K + k0 + kq * 4) : make_float4(0, 0, 0, 0);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < BM / 32; ++i) {
            const int row = lr + 32 * i;
            uint32_t h0, m0_, l0, h1, m1, l1;
            split2(ra[i].x, ra[i].y, h0, m0_, l0);
            split2(ra[i].z, ra[i].w, h1, m1, l1);
            *reinterpret_cast<uint2*>(la + 0 * PLA + row * PITCH + kq * 8) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(la + 1 * PLA + row * PITCH + kq * 8) = make_uint2(m0_, m1);
            *reinterpret_cast<uint2*>(la + 2 * PLA + row * PITCH + kq * 8) = make_uint2(l0, l1);
        }
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            const int row = lr + 32 * i;
            uint32_t h0, m0_, l0, h1, m1, l1;
            split2(rb[i].x, rb[i].y, h0, m0_, l0);
            split2(rb[i].z, rb[i].w, h1, m1, l1);
            *reinterpret_cast<uint2*>(lb + 0 * PLB + row * PITCH + kq * 8) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(lb + 1 * PLB + row * PITCH + kq * 8) = make_uint2(m0_, m1);
            *reinterpret_cast<uint2*>(lb + 2 * PLB + row * PITCH + kq * 8) = make_uint2(l0, l1);
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i][p] = *reinterpret_cast<const bf16x8*>(la + p * PLA + (wm * WM + i * 32 + r32) * PITCH + s * 32 + h * 16);
#pragma unroll
                for (int i = 0; i < TN; ++i) fb[i][p] = *reinterpret_cast<const bf16x8*>(lb + p * PLB + (wn * WN + i * 32 + r32) * PITCH + s * 32 + h * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
                    if (NPROD == 9) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][2], c, 0, 0, 0);
                    }
                    if (NPROD >= 6) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
                    }
                    if (NPROD >= 3) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
    }
    // C/D layout: col = lane & 31 (the B operand's row = n), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (the A operand's row = m)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int n = n0 + wn * WN + j * 32 + r32;
                if (m < M && n < N) C[(int64_t)m * N + n] = acc[i][j][e];
            }
}


// Double-buffered variant: the split + LDS store of slab kt+1 is interleaved with the MFMAs of slab kt (one barrier per slab);
// two register stages keep the global loads two slabs ahead.  Dynamic LDS: 2 buffers x (3 planes x (BM + BN) rows x PITCH).
template <int NPROD, int BM, int BN, int WM, int WN, int SCHED>
__global__ __launch_bounds__(256, 1) void split_gemm_nt_db(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    constexpr int PLA = BM * PITCH, PLB = BN * PITCH, TM = WM / 32, TN = WN / 32, WAVES_N = BN / WN, BUF = 3 * PLA + 3 * PLB;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int r32 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kq = t & 7, lr = t >> 3;
    constexpr int AR = BM / 32, BR = BN / 32;
    float4 st[2][AR + BR];
    auto gload = [&](int k0, int sg) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int row = lr + 32 * i;
            st[sg][i] = (m0 + row < M) ? *reinterpret_cast<const float4*>(A + (int64_t)(m0 + row) * K + k0 + kq * 4) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int row = lr + 32 * i;
            st[sg][AR + i] = (n0 + row < N) ? *reinterpret_cast<const float4*>(B + (int64_t)(n0 + row) * K + k0 + kq * 4) : make_float4(0, 0, 0, 0);
        }
    };
    auto store_unit = [&](int u, int sg, uint8_t* buf) __attribute__((always_inline)) {      // u in [0, AR + BR)
        const float4 v = st[sg][u];
        const int row = lr + 32 * (u < AR ? u : u - AR);
        uint8_t* base = buf + (u < AR ? 0 : 3 * PLA) + row * PITCH + kq * 8;
        const int pl = u < AR ? PLA : PLB;
        uint32_t h0, m0_, l0, h1, m1, l1;
        split2(v.x, v.y, h0, m0_, l0);
        split2(v.z, v.w, h1, m1, l1);
        *reinterpret_cast<uint2*>(base) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(base + pl) = make_uint2(m0_, m1);
        *reinterpret_cast<uint2*>(base + 2 * pl) = make_uint2(l0, l1);
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nk = K / BK;
    gload(0, 0);
#pragma unroll
    for (int u = 0; u < AR + BR; ++u) store_unit(u, 0, lds);
    if (nk > 1) gload(BK, 1);
    __syncthreads();
    auto body = [&](int kt, auto cur_c) __attribute__((always_inline)) {
        constexpr int cur = decltype(cur_c)::value;             // LDS buffer / register stage parity of slab kt
        uint8_t* lc = lds + cur * BUF;
        uint8_t* ln = lds + (cur ^ 1) * BUF;
        const bool more = kt + 1 < nk;
        if (kt + 2 < nk) gload((kt + 2) * BK, cur);             // stage `cur` is free: slab kt is in LDS
        int unit = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i][p] = *reinterpret_cast<const bf16x8*>(lc + p * PLA + (wm * WM + i * 32 + r32) * PITCH + s * 32 + h * 16);
#pragma unroll
                for (int i = 0; i < TN; ++i) fb[i][p] = *reinterpret_cast<const bf16x8*>(lc + 3 * PLA + p * PLB + (wn * WN + i * 32 + r32) * PITCH + s * 32 + h * 16);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
                    if (NPROD >= 6) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                    // one split unit of the NEXT slab per accumulator tile (AR + BR units over 2 * TM * TN tiles)
                    constexpr int per = (AR + BR + 2 * TM * TN - 1) / (2 * TM * TN);
#pragma unroll
                    for (int q = 0; q < per; ++q) {
                        if (more && unit < AR + BR) store_unit(unit, cur ^ 1, ln);
                        ++unit;
                    }
                }
        }
        if (SCHED) {
            // pin the interleaving: per MFMA, a handful of VALU and at most one LDS access
#pragma unroll
            for (int q = 0; q < 2 * TM * TN * NPROD; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, SCHED, 0);   // VALU
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // 1 DS write
            }
        }
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        body(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < nk) body(kt + 1, std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int n = n0 + wn * WN + j * 32 + r32;
                if (m < M && n < N) C[(int64_t)m * N + n] = acc[i][j][e];
            }
}


// B (the filter) pre-split into three bf16 planes [3][N][K] by a pass of its own (weights change once per optimizer step);
// the kernel splits only the activation operand.  Any number of waves.
__global__ void presplit_kernel(const float* __restrict__ B, uint16_t* __restrict__ out, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    uint32_t h, m, l;
    split2(B[i], B[i + 1], h, m, l);
    *reinterpret_cast<uint32_t*>(out + i) = h;
    *reinterpret_cast<uint32_t*>(out + n + i) = m;
    *reinterpret_cast<uint32_t*>(out + 2 * n + i) = l;
}

template <int NPROD, int BM, int BN, int WM, int WN, int MINW, int DIAG = 0>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, MINW) void split_gemm_nt_pre(const float* __restrict__ A, const uint16_t* __restrict__ Bp, float* __restrict__ C, int M, int N, int K) {
    constexpr int PLA = BM * PITCH, PLB = BN * PITCH, TM = WM / 32, TN = WN / 32, WAVES_N = BN / WN, NT = (BM / WM) * (BN / WN) * 64;
    __shared__ __attribute__((aligned(16))) uint8_t lds[3 * PLA + 3 * PLB];
    uint8_t* la = lds;
    uint8_t* lb = lds + 3 * PLA;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int r32 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kq = t & 7, lr = t >> 3;
    constexpr int RS = NT / 8, AR = BM / RS;                   // A: float4 per thread per slab
    constexpr int BCH = (3 * BN * 4 + NT - 1) / NT;            // B: 16-byte chunks (8 bf16) per thread per slab: chunk = (plane, row, k-octet)
    static_assert(BM % RS == 0, "loader shapes");
    float4 ra[AR];
    uint4 rb[BCH];
    const int64_t nk_elems = (int64_t)N * K;
    auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int row = lr + RS * i;
            ra[i] = (m0 + row < M) ? *reinterpret_cast<const float4*>(A + (int64_t)(m0 + row) * K + k0 + kq * 4) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int c = t + NT * i, oct = c & 3, row = (c >> 2) % BN, pl = c / (4 * BN);
            rb[i] = (c < 3 * BN * 4 && n0 + row < N) ? *reinterpret_cast<const uint4*>(Bp + pl * nk_elems + (int64_t)(n0 + row) * K + k0 + oct * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int row = lr + RS * i;
            uint32_t h0, m0_, l0, h1, m1, l1;
            if (DIAG == 1) {                // no split arithmetic: raw bits (wrong values, timing only)
                h0 = __float_as_uint(ra[i].x); m0_ = __float_as_uint(ra[i].y); l0 = h0 ^ m0_; h1 = __float_as_uint(ra[i].z); m1 = __float_as_uint(ra[i].w); l1 = h1 ^ m1;
            } else {
                split2(ra[i].x, ra[i].y, h0, m0_, l0);
                split2(ra[i].z, ra[i].w, h1, m1, l1);
            }
            *reinterpret_cast<uint2*>(la + 0 * PLA + row * PITCH + kq * 8) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(la + 1 * PLA + row * PITCH + kq * 8) = make_uint2(m0_, m1);
            *reinterpret_cast<uint2*>(la + 2 * PLA + row * PITCH + kq * 8) = make_uint2(l0, l1);
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int c = t + NT * i, oct = c & 3, row = (c >> 2) % BN, pl = c / (4 * BN);
            if (c < 3 * BN * 4) *reinterpret_cast<uint4*>(lb + pl * PLB + row * PITCH + oct * 16) = rb[i];
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    gload(0);
    bf16x8 fa[TM][3], fb[TN][3];
    for (int k0 = 0; k0 < K; k0 += BK) {
        if (DIAG < 3 || k0 == 0) {
            __syncthreads();
            lstore();
            __syncthreads();
        }
        if (k0 + BK < K && (DIAG < 2 || k0 == 0)) gload(k0 + BK);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (DIAG < 4 || k0 == 0) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[i][p] = *reinterpret_cast<const bf16x8*>(la + p * PLA + (wm * WM + i * 32 + r32) * PITCH + s * 32 + h * 16);
#pragma unroll
                    for (int i = 0; i < TN; ++i) fb[i][p] = *reinterpret_cast<const bf16x8*>(lb + p * PLB + (wn * WN + i * 32 + r32) * PITCH + s * 32 + h * 16);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
                    if (NPROD >= 6) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int n = n0 + wn * WN + j * 32 + r32;
                if (m < M && n < N) C[(int64_t)m * N + n] = acc[i][j][e];
            }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static double err_of(const std::vector<float>& c, const std::vector<float>& a, const std::vector<float>& b, int M, int N, int K, double* mean) {
    double worst = 0, sum = 0;
    uint64_t s = 12345;
    const int samples = 4096;
    for (int i = 0; i < samples; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const int m = (int)((s >> 33) % M);
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const int n = (int)((s >> 33) % N);
        double ref = 0, mag = 0;
        for (int k = 0; k < K; ++k) { const double p = (double)a[(size_t)m * K + k] * b[(size_t)n * K + k]; ref += p; mag += std::fabs(p); }
        const double e = std::fabs(c[(size_t)m * N + n] - ref) / mag;
        worst = e > worst ? e : worst;
        sum += e;
    }
    *mean = sum / samples;
    return worst;
}

int main(int argc, char** argv) {
    struct Shape { int n, hw, K, N; } shapes[] = {{160, 14, 160, 960}, {160, 14, 960, 160}, {160, 7, 1536, 256}, {160, 7, 256, 1536}, {160, 14, 960, 960}, {160, 28, 256, 256}, {160, 56, 192, 48}};
    for (const Shape& sh : shapes) {
        const int M = sh.n * sh.hw * sh.hw, N = sh.N, K = sh.K;
        std::vector<float> a((size_t)M * K), b((size_t)N * K), c((size_t)M * N);
        uint64_t s = 777;
        auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
        for (auto& v : a) v = rnd() * 1.7f + 0.1f;
        for (auto& v : b) v = rnd() * 0.3f;
        float *dA, *dB, *dC;
        CK(hipMalloc(&dA, a.size() * 4)); CK(hipMalloc(&dB, b.size() * 4)); CK(hipMalloc(&dC, c.size() * 4));
        CK(hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        uint16_t* dBp;
        CK(hipMalloc(&dBp, b.size() * 2 * 3));
        hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)((b.size() / 2 + 255) / 256)), dim3(256), 0, 0, dB, dBp, (int64_t)b.size());
        const double flop = 2.0 * M * N * K;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto run = [&](int which) {
            if (which == 0) {
                suhpe_conv_t d = {sh.n, sh.hw, sh.hw, K, sh.hw, sh.hw, N, 1, 1, 1, 0, 1};
                if (suhpe_conv2d_fwd(&d, dA, dB, nullptr, dC, nullptr)) { printf("lib error: %s\n", suhpe_last_error()); exit(1); }
            } else {
#define GRID(BM_, BN_) dim3((N + BN_ - 1) / BN_, (M + BM_ - 1) / BM_)
                if (which == 3) hipLaunchKernelGGL((split_gemm_nt<3, 128, 128, 64, 64>), GRID(128, 128), dim3(256), 0, 0, dA, dB, dC, M, N, K);
                if (which == 6) hipLaunchKernelGGL((split_gemm_nt<6, 128, 128, 64, 64>), GRID(128, 128), dim3(256), 0, 0, dA, dB, dC, M, N, K);
                if (which == 9) hipLaunchKernelGGL((split_gemm_nt<9, 128, 128, 64, 64>), GRID(128, 128), dim3(256), 0, 0, dA, dB, dC, M, N, K);
                if (which == 61) hipLaunchKernelGGL((split_gemm_nt<6, 64, 64, 32, 32>), GRID(64, 64), dim3(256), 0, 0, dA, dB, dC, M, N, K);
                if (which == 62) hipLaunchKernelGGL((split_gemm_nt<6, 128, 64, 64, 32>), GRID(128, 64), dim3(256), 0, 0, dA, dB, dC, M, N, K);
                if (which == 63) hipLaunchKernelGGL((split_gemm_nt<6, 64, 128, 32, 64>), GRID(64, 128), dim3(256), 0, 0, dA, dB, dC, M, N, K);
                if (which == 64) hipLaunchKernelGGL((split_gemm_nt<6, 128, 64, 32, 64>), GRID(128, 64), dim3(256), 0, 0, dA, dB, dC, M, N, K);
#define DB(W, BM_, BN_, WM_, WN_, SC)                                                                                              \
                if (which == W) {                                                                                                  \
                    auto kern = split_gemm_nt_db<6, BM_, BN_, WM_, WN_, SC>;                                                       \
                    const int lds_bytes = 2 * 3 * (BM_ + BN_) * PITCH;                                                             \
                    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));             \
                    hipLaunchKernelGGL(kern, GRID(BM_, BN_), dim3(256), lds_bytes, 0, dA, dB, dC, M, N, K);                        \
                }
#define PRE(W, BM_, BN_, WM_, WN_, MW) if (which == W) hipLaunchKernelGGL((split_gemm_nt_pre<6, BM_, BN_, WM_, WN_, MW>), GRID(BM_, BN_), dim3((BM_ / WM_) * (BN_ / WN_) * 64), 0, 0, dA, dBp, dC, M, N, K);
                PRE(80, 128, 128, 64, 64, 2) PRE(81, 128, 128, 64, 32, 2) PRE(82, 128, 128, 32, 64, 2) PRE(83, 128, 64, 64, 32, 3) PRE(84, 128, 64, 32, 32, 2) PRE(85, 64, 64, 32, 32, 5) PRE(86, 256, 128, 64, 64, 1) PRE(87, 64, 128, 32, 64, 3)
#define PRED(W, DG) if (which == W) hipLaunchKernelGGL((split_gemm_nt_pre<6, 128, 128, 64, 64, 2, DG>), GRID(128, 128), dim3(256), 0, 0, dA, dBp, dC, M, N, K);
                PRED(91, 1) PRED(92, 2) PRED(93, 3) PRED(94, 4)
                DB(70, 128, 128, 64, 64, 0) DB(71, 128, 128, 64, 64, 4) DB(72, 128, 128, 64, 64, 5) DB(73, 128, 64, 64, 32, 0) DB(74, 128, 64, 64, 32, 5)
            }
        };
        printf("M=%d N=%d K=%d:\n", M, N, K);
        for (int which : {0, 3, 6, 9, 80, 86, 91, 92, 93, 94}) {
            CK(hipMemset(dC, 0, c.size() * 4));
            for (int i = 0; i < 5; ++i) run(which);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            const int iters = 30;
            for (int i = 0; i < iters; ++i) run(which);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
            double mean;
            const double worst = err_of(c, a, b, M, N, K, &mean);
            printf("    [%-22s] %7.1f us %4.0f TF/s-eq err max %.2e mean %.2e\n", which == 0 ? "fp32-mfma lib" : which == 3 ? "bf16x3 128x128/64x64" : which == 6 ? "bf16x6 128x128/64x64" : which == 9 ? "bf16x9 128x128/64x64" : which == 61 ? "bf16x6 64x64/32x32" : which == 62 ? "bf16x6 128x64/64x32" : which == 63 ? "bf16x6 64x128/32x64" : which == 64 ? "bf16x6 128x64/32x64" : which == 91 ? "diag1 no split VALU" : which == 92 ? "diag2 +no global loads" : which == 93 ? "diag3 +no LDS stores/barriers" : which == 94 ? "diag4 +no LDS reads (MFMA only)" : which == 80 ? "pre 128x128/64x64 4w" : which == 81 ? "pre 128x128/64x32 8w" : which == 82 ? "pre 128x128/32x64 8w" : which == 83 ? "pre 128x64/64x32 4w" : which == 84 ? "pre 128x64/32x32 8w" : which == 85 ? "pre 64x64/32x32 4w" : which == 86 ? "pre 256x128/64x64 8w" : which == 87 ? "pre 64x128/32x64 4w" : which == 70 ? "x6 db 128x128" : which == 71 ? "x6 db 128x128 sched4" : which == 72 ? "x6 db 128x128 sched5" : which == 73 ? "x6 db 128x64" : "x6 db 128x64 sched5", ms / iters * 1e3,
                   flop / (ms / iters * 1e-3) / 1e12, worst, mean);
        }
        printf("\n");
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    return 0;
}
