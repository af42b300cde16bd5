// GEMM laboratory (round 3): prototypes of the operand-resident fp32-MFMA GEMM, timed against the library's tiles on the pointwise
// shapes of the C2 step.      out[M][N] = A[M][K] * F[N][K]^T   (A = pixels x channels-in, F = filter rows, both K-contiguous)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/proto/gemm_lab.hip -L semiuhpe_amd/csrc -lsuhpe -Wl,-rpath,'$ORIGIN/../../../semiuhpe_amd/csrc' -o tools/proto/bin/gemm_lab
//   tools/proto/bin/gemm_lab [M N K]...
//
// Kernel R ("resident"): a workgroup stages its BM x K block of A in LDS ONCE (padded rows, conflict-free ds_read_b128) and every
// wavefront then works alone: it walks 32-column stripes of the output, loads the stripe's filter rows straight from global memory
// in MFMA-operand shape (lane (j, h) takes F[n0+j][8g+4h .. +3]: one 16-byte load feeds four v_mfma_f32_32x32x2_f32) through a
// two-slab register ring, multiplies against the resident block (BM/32 accumulator tiles per wave) and stores.  No barrier after the
// staging one: the waves drift apart on their own, so one wave's epilogue stores run under its SIMD partner's MFMAs.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#include "suhpe.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
using BufRsrc = __amdgpu_buffer_rsrc_t;
constexpr unsigned kOob = 0x80000000u;

__device__ __forceinline__ BufRsrc make_rsrc(const float* base, int64_t bytes) {
    const uint64_t addr = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr), hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
    const int nb = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffff ? 0x7fffffff : bytes));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(BufRsrc r, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

static unsigned long long* g_dbg = nullptr;
static int g_dbg_blocks = 0, g_dbg_nw = 0;

struct RParams {
    const float* a; const float* f; float* out;
    int M, N, K;
    int lda, ldf, ldo;
    int Kp;                 // K rounded up to 32
    unsigned long long* dbg;   // optional per-wave time stamps (ABL & 32): [block][wave][24]
    int col_ranges;         // the N/32 stripes are cut into this many contiguous ranges; workgroup = (row tile, range)
    int stripes_per_range;
};

// ABL (timing-only ablations): bit 0 = no output stores, bit 1 = filter rows loaded once, bit 2 = A fragments read from LDS once
template <int BM, int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 64, NW / 4) void gemm_resident(const RParams p) {
    constexpr int TM = BM / 32, NT = NW * 64;
    extern __shared__ __attribute__((aligned(16))) float As[];           // [BM][Kp + 4]
    const int pitch = p.Kp + 4;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    // workgroups that share a row tile (the column ranges) are consecutive multiples of 8 apart -> one XCD
    const int b = blockIdx.x;
    const int jx = b >> 3;
    const int range = jx % p.col_ranges;
    const int row_tile = (jx / p.col_ranges) * 8 + (b & 7);
    const int m0 = row_tile * BM;
    if (m0 >= p.M) return;
    if (ABL & 8) return;                                   // launch overhead only
    const BufRsrc ar = make_rsrc(p.a, (int64_t)p.M * p.lda * 4);
    const BufRsrc fr = make_rsrc(p.f, (int64_t)p.N * p.ldf * 4);

    // ---- stage the A block: 8 threads cover one 128-byte row chunk
    {
        const int lcol = (t & 7) * 4, lrow = t >> 3;
        constexpr int RSTEP = NT / 8;
        for (int kc = 0; kc < p.Kp; kc += 32) {
#pragma unroll
            for (int i = 0; i < (BM + RSTEP - 1) / RSTEP; ++i) {
                const int row = lrow + i * RSTEP;
                if (BM % RSTEP != 0 && row >= BM) continue;
                const bool ok = (m0 + row < p.M) & (kc + lcol < p.K);
                const float4 v = buf_load4(ar, ok ? ((unsigned)(m0 + row) * (unsigned)p.lda + (unsigned)(kc + lcol)) * 4u : kOob);
                *reinterpret_cast<float4*>(&As[row * pitch + kc + lcol]) = v;
            }
        }
    }
    __syncthreads();
    if (ABL & 16) return;                                  // launch + staging only

    const int nslab = p.Kp / 32;
    const int st0 = range * p.stripes_per_range;
    int nst = (p.N + 31) / 32 - st0;
    if (nst > p.stripes_per_range) nst = p.stripes_per_range;
    const int mine = nst > wave ? (nst - wave + NW - 1) / NW : 0;         // stripes wave, wave+NW, ...
    if (mine == 0) return;
    const int total = mine * nslab;

    float4 bq[2][4];
    int l_it = 0, l_st = st0 + wave, l_slab = 0;
    auto loadB = [&](auto buf) __attribute__((always_inline)) {
        if (l_it >= total) return;
        if ((ABL & 2) && l_it >= 2) { ++l_it; if (++l_slab == nslab) { l_slab = 0; l_st += NW; } return; }
        const int col = l_st * 32 + r32;
        const unsigned base = (unsigned)col * (unsigned)p.ldf;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int k = l_slab * 32 + g * 8 + h * 4;
            const bool ok = (col < p.N) & (k < p.K);
            bq[decltype(buf)::value][g] = buf_load4(fr, ok ? (base + (unsigned)k) * 4u : kOob);
        }
        ++l_it;
        if (++l_slab == nslab) { l_slab = 0; l_st += NW; }
    };

    floatx16 acc[TM];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    };
    zero_acc();
    int c_it = 0, c_st = st0 + wave, c_slab = 0;
    const float* arow = As + r32 * pitch + h * 4;
    auto compute = [&](auto buf) __attribute__((always_inline)) {
        const float* ap = arow + ((ABL & 4) ? 0 : c_slab * 32);
        float4 af[2][TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const float4*>(ap + i * 32 * pitch);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cur = g & 1;
            if (g + 1 < 4) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[cur ^ 1][i] = *reinterpret_cast<const float4*>(ap + i * 32 * pitch + (g + 1) * 8);
            }
            const float4 bv = bq[decltype(buf)::value][g];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].x, bv.x, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].y, bv.y, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].z, bv.z, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].w, bv.w, acc[i], 0, 0, 0);
            }
        }
    };
    auto epilogue = [&]() __attribute__((always_inline)) {
        const int col = c_st * 32 + r32;
        if (col < p.N) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if ((ABL & 1) ? (row < p.M && acc[i][e] == 12345.678f) : (row < p.M)) p.out[(int64_t)row * p.ldo + col] = acc[i][e];
                }
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    auto step = [&](auto cur, auto nxt) __attribute__((always_inline)) -> bool {
        loadB(nxt);
        compute(cur);
        ++c_it;
        if (++c_slab == nslab) {
            epilogue();
            zero_acc();
            c_slab = 0;
            c_st += NW;
        }
        return c_it < total;
    };
    loadB(S0{});
    while (step(S0{}, S1{}) && step(S1{}, S0{})) {}
}


// ---- kernel R2: the same idea with the code generation under control.  A wave's unit of work is (32-column stripe, TMU row blocks of
// the resident tile); units go round-robin to the waves, so with 2 * units-per-stripe * stripes divisible by 4 every SIMD's pair of
// waves carries the same MFMA count.  Output through buffer stores (one lane offset + a scalar row offset per store: no 64-bit
// address per store kept live across the main loop), A fragments of k-group g+1 read under the MFMAs of g.
__device__ __forceinline__ void buf_store1(float v, BufRsrc r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

template <int BM, int TMU, int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 64, NW / 4) void gemm_resident2(const RParams p) {
    constexpr int NT = NW * 64, RU = BM / (32 * TMU);
    extern __shared__ __attribute__((aligned(16))) float As[];           // [BM][Kp + 4]
    const int pitch = p.Kp + 4;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const int jx = b >> 3;
    const int range = jx % p.col_ranges;
    const int row_tile = (jx / p.col_ranges) * 8 + (b & 7);
    const int m0 = row_tile * BM;
    if (m0 >= p.M) return;
    const BufRsrc ar = make_rsrc(p.a, (int64_t)p.M * p.lda * 4);
    const BufRsrc fr = make_rsrc(p.f, (int64_t)p.N * p.ldf * 4);
    const BufRsrc orr = make_rsrc(p.out, (int64_t)p.M * p.ldo * 4);
    {
        const int lcol = (t & 7) * 4, lrow = t >> 3;
        constexpr int RSTEP = NT / 8;
        for (int kc = 0; kc < p.Kp; kc += 32) {
#pragma unroll
            for (int i = 0; i < (BM + RSTEP - 1) / RSTEP; ++i) {
                const int row = lrow + i * RSTEP;
                if (BM % RSTEP != 0 && row >= BM) continue;
                const bool ok = (m0 + row < p.M) & (kc + lcol < p.K);
                const float4 v = buf_load4(ar, ok ? ((unsigned)(m0 + row) * (unsigned)p.lda + (unsigned)(kc + lcol)) * 4u : kOob);
                *reinterpret_cast<float4*>(&As[row * pitch + kc + lcol]) = v;
            }
        }
    }
    __syncthreads();

    const int nslab = p.Kp / 32;
    const int st0 = range * p.stripes_per_range;
    int nst = (p.N + 31) / 32 - st0;
    if (nst > p.stripes_per_range) nst = p.stripes_per_range;
    const int nunits = nst * RU;
    const int mine = nunits > wave ? (nunits - wave + NW - 1) / NW : 0;
    if (mine == 0) return;
    const int total = mine * nslab;

    float4 bq0[4], bq1[4];
    int l_u = wave, l_slab = 0;
    auto loadB = [&](float4 (&bq)[4]) __attribute__((always_inline)) {
        const int col = (st0 + l_u / RU) * 32 + r32;
        const unsigned base = (unsigned)col * (unsigned)p.ldf + (unsigned)(l_slab * 32 + h * 4);
        const bool okc = col < p.N;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bool ok = okc & (l_slab * 32 + g * 8 + h * 4 < p.K);
            bq[g] = buf_load4(fr, ok ? (base + (unsigned)(g * 8)) * 4u : kOob);
        }
        if (++l_slab == nslab) { l_slab = 0; l_u += NW; }
    };

    floatx16 acc[TMU];
#pragma unroll
    for (int i = 0; i < TMU; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    int c_u = wave, c_slab = 0;
    const float* arow = As + r32 * pitch + h * 4;
    auto compute = [&](const float4 (&bq)[4]) __attribute__((always_inline)) {
        const float* ap = arow + (c_u % RU) * (TMU * 32) * pitch + c_slab * 32;
        float4 af[2][TMU];
#pragma unroll
        for (int i = 0; i < TMU; ++i) af[0][i] = *reinterpret_cast<const float4*>(ap + i * 32 * pitch);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cur = g & 1;
            if (g + 1 < 4) {
#pragma unroll
                for (int i = 0; i < TMU; ++i) af[cur ^ 1][i] = *reinterpret_cast<const float4*>(ap + i * 32 * pitch + (g + 1) * 8);
            }
            const float4 bv = bq[g];
#pragma unroll
            for (int i = 0; i < TMU; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].x, bv.x, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TMU; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].y, bv.y, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TMU; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].z, bv.z, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TMU; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].w, bv.w, acc[i], 0, 0, 0);
        }
        if (++c_slab == nslab) {
            // epilogue of the unit: C layout col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 h
            const int col = (st0 + c_u / RU) * 32 + r32;
            const int rowl = m0 + (c_u % RU) * (TMU * 32) + 4 * h;              // this lane's first row
            const unsigned voff = col < p.N ? ((unsigned)rowl * (unsigned)p.ldo + (unsigned)col) * 4u : kOob;
            const int lim = p.M - rowl;                                            // rows ro < lim are valid
            const unsigned ldo4 = (unsigned)p.ldo * 4u;
#pragma unroll
            for (int i = 0; i < TMU; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ro = i * 32 + (e & 3) + 8 * (e >> 2);
                    const bool ok = (ABL & 1) ? (ro < lim && acc[i][e] == 12345.678f) : (ro < lim);
                    buf_store1(acc[i][e], orr, ok ? voff : kOob, (unsigned)ro * ldo4);
                }
#pragma unroll
            for (int i = 0; i < TMU; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
            c_slab = 0;
            c_u += NW;
        }
    };
    loadB(bq0);
    for (int it = 0; it < total; it += 2) {
        if (it + 1 < total) loadB(bq1);
        compute(bq0);
        if (it + 1 >= total) break;
        if (it + 2 < total) loadB(bq0);
        compute(bq1);
    }
}

template <int BM, int TMU, int NW, int ABL = 0> static void launch_resident2(const float* a, const float* f, float* out, int M, int N, int K, int col_ranges) {
    RParams p;
    p.a = a; p.f = f; p.out = out; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldf = K; p.ldo = N;
    p.Kp = (K + 31) / 32 * 32;
    const int stripes = (N + 31) / 32;
    p.col_ranges = col_ranges;
    p.stripes_per_range = (stripes + col_ranges - 1) / col_ranges;
    const int row_tiles8 = ((M + BM - 1) / BM + 7) / 8 * 8;
    const size_t lds = (size_t)BM * (p.Kp + 4) * 4;
    static bool attr = false;
    if (!attr) { CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_resident2<BM, TMU, NW, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    hipLaunchKernelGGL((gemm_resident2<BM, TMU, NW, ABL>), dim3(row_tiles8 * col_ranges), dim3(NW * 64), lds, 0, p);
}


// ---- kernel R3: R2 with every vector-memory operation of a step issued at the step's START, behind a wait for everything older:
//   step = [s_waitcnt vmcnt(0)] [stores of the previous unit, from a saved copy of its accumulators] [filter loads of the next step] [MFMAs]
// vmcnt counts loads and stores together, in order: a wait for filter rows that were requested before a batch of stores but is
// placed after it waits for the stores too (and the compiler merges control-flow paths to the smallest count).  Here whatever a
// wait covers is at least one step (32 * TMU MFMAs) old.
template <int BM, int TMU, int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 64, NW / 4) void gemm_resident3(const RParams p) {
    constexpr int NT = NW * 64, RU = BM / (32 * TMU);
    extern __shared__ __attribute__((aligned(16))) float As[];           // [BM][Kp + 4]
    const int pitch = p.Kp + 4;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const int jx = b >> 3;
    const int range = jx % p.col_ranges;
    const int row_tile = (jx / p.col_ranges) * 8 + (b & 7);
    const int m0 = row_tile * BM;
    if (m0 >= p.M) return;
    unsigned long long* dbg = nullptr;
    int dbg_n = 0;
    if (ABL & 32) {
        dbg = p.dbg + ((size_t)blockIdx.x * NW + wave) * 24;
        if (lane == 0) { dbg[0] = __builtin_amdgcn_s_memrealtime(); dbg[22] = __builtin_amdgcn_s_memtime(); dbg[1] = __builtin_amdgcn_s_getreg((16 << 11) | (0 << 6) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 20) << 32); }
        dbg_n = 3;
    }
    const BufRsrc ar = make_rsrc(p.a, (int64_t)p.M * p.lda * 4);
    const BufRsrc fr = make_rsrc(p.f, (int64_t)p.N * p.ldf * 4);
    const BufRsrc orr = make_rsrc(p.out, (int64_t)p.M * p.ldo * 4);
    {
        const int lcol = (t & 7) * 4, lrow = t >> 3;
        constexpr int RSTEP = NT / 8;
        for (int kc = 0; kc < p.Kp; kc += 32) {
#pragma unroll
            for (int i = 0; i < (BM + RSTEP - 1) / RSTEP; ++i) {
                const int row = lrow + i * RSTEP;
                if (BM % RSTEP != 0 && row >= BM) continue;
                const bool ok = (m0 + row < p.M) & (kc + lcol < p.K);
                const float4 v = buf_load4(ar, ok ? ((unsigned)(m0 + row) * (unsigned)p.lda + (unsigned)(kc + lcol)) * 4u : kOob);
                *reinterpret_cast<float4*>(&As[row * pitch + kc + lcol]) = v;
            }
        }
    }
    __syncthreads();
    if ((ABL & 32) && lane == 0) dbg[2] = __builtin_amdgcn_s_memtime();

    const int nslab = p.Kp / 32;
    const int st0 = range * p.stripes_per_range;
    int nst = (p.N + 31) / 32 - st0;
    if (nst > p.stripes_per_range) nst = p.stripes_per_range;
    const int nunits = nst * RU;
    const int mine = nunits > wave ? (nunits - wave + NW - 1) / NW : 0;
    if (mine == 0) return;
    const int total = mine * nslab;

    float4 bq0[4], bq1[4];
    int l_it = 0, l_u = wave, l_slab = 0;
    auto loadB = [&](float4 (&bq)[4]) __attribute__((always_inline)) {       // always issued; past the end every lane is out of range
        const int col = (st0 + l_u / RU) * 32 + r32;
        const unsigned base = (unsigned)col * (unsigned)p.ldf + (unsigned)(l_slab * 32 + h * 4);
        const bool okc = (col < p.N) & (l_it < total);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bool ok = okc & (l_slab * 32 + g * 8 + h * 4 < p.K);
            bq[g] = buf_load4(fr, ok ? (base + (unsigned)(g * 8)) * 4u : kOob);
        }
        ++l_it;
        if (++l_slab == nslab) { l_slab = 0; l_u += NW; }
    };

    floatx16 acc[TMU], sav[TMU];
    int c_u = wave, c_slab = 0, s_u = -1;                                       // s_u: unit whose results wait in sav (-1: none)
    const float* arow = As + r32 * pitch + h * 4;
    auto flush_saved = [&]() __attribute__((always_inline)) {
        const int col = (st0 + s_u / RU) * 32 + r32;
        const int rowl = m0 + (s_u % RU) * (TMU * 32) + 4 * h;
        const unsigned voff = col < p.N ? ((unsigned)rowl * (unsigned)p.ldo + (unsigned)col) * 4u : kOob;
        const int lim = p.M - rowl;
        const unsigned ldo4 = (unsigned)p.ldo * 4u;
#pragma unroll
        for (int i = 0; i < TMU; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ro = i * 32 + (e & 3) + 8 * (e >> 2);
                const bool ok = (ABL & 1) ? (ro < lim && sav[i][e] == 12345.678f) : (ro < lim);
                buf_store1(sav[i][e], orr, ok ? voff : kOob, (unsigned)ro * ldo4);
            }
        s_u = -1;
    };
    auto step = [&](const float4 (&bq)[4], float4 (&bqn)[4]) __attribute__((always_inline)) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                                     // vmcnt(0): this step's filter rows (and older stores)
        if (s_u >= 0) flush_saved();
        loadB(bqn);
        __builtin_amdgcn_sched_barrier(0);                                     // the loads stay here, a whole step ahead of their use
        const float* ap = arow + (c_u % RU) * (TMU * 32) * pitch + c_slab * 32;
        float4 af[2][TMU];
#pragma unroll
        for (int i = 0; i < TMU; ++i) af[0][i] = *reinterpret_cast<const float4*>(ap + i * 32 * pitch);
        const bool first = c_slab == 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cur = g & 1;
            if (g + 1 < 4) {
#pragma unroll
                for (int i = 0; i < TMU; ++i) af[cur ^ 1][i] = *reinterpret_cast<const float4*>(ap + i * 32 * pitch + (g + 1) * 8);
            }
            const float4 bv = bq[g];
            if (g == 0) {
#pragma unroll
                for (int i = 0; i < TMU; ++i) {
                    floatx16 c = acc[i];
                    if (first) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) c[e] = 0.f;
                    }
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].x, bv.x, c, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < TMU; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].x, bv.x, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TMU; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].y, bv.y, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TMU; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].z, bv.z, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TMU; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].w, bv.w, acc[i], 0, 0, 0);
            if (g + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, TMU, 0);      // fragments of k-group g+1 first ...
            __builtin_amdgcn_sched_group_barrier(0x8, 4 * TMU, 0);                   // ... then the MFMAs of k-group g
        }
        __builtin_amdgcn_sched_barrier(0);
        if (++c_slab == nslab) {
#pragma unroll
            for (int i = 0; i < TMU; ++i) sav[i] = acc[i];
            s_u = c_u;
            c_slab = 0;
            c_u += NW;
            if ((ABL & 32) && lane == 0 && dbg_n < 21) dbg[dbg_n] = __builtin_amdgcn_s_memtime();
            ++dbg_n;
        }
    };
    loadB(bq0);
    for (int it = 0; it < total; it += 2) {
        step(bq0, bq1);
        if (it + 1 >= total) break;
        step(bq1, bq0);
    }
    if (s_u >= 0) flush_saved();
    if ((ABL & 32) && lane == 0) { __builtin_amdgcn_s_waitcnt(0x0F70); dbg[21] = __builtin_amdgcn_s_memtime(); dbg[23] = __builtin_amdgcn_s_memrealtime(); }
}

template <int BM, int TMU, int NW, int ABL = 0> static void launch_resident3(const float* a, const float* f, float* out, int M, int N, int K, int col_ranges) {
    RParams p;
    p.a = a; p.f = f; p.out = out; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldf = K; p.ldo = N;
    p.Kp = (K + 31) / 32 * 32;
    const int stripes = (N + 31) / 32;
    p.col_ranges = col_ranges;
    p.stripes_per_range = (stripes + col_ranges - 1) / col_ranges;
    const int row_tiles8 = ((M + BM - 1) / BM + 7) / 8 * 8;
    const size_t lds = (size_t)BM * (p.Kp + 4) * 4;
    static bool attr = false;
    if (!attr) { CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_resident3<BM, TMU, NW, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    p.dbg = g_dbg;
    hipLaunchKernelGGL((gemm_resident3<BM, TMU, NW, ABL>), dim3(row_tiles8 * col_ranges), dim3(NW * 64), lds, 0, p);
    g_dbg_blocks = row_tiles8 * col_ranges; g_dbg_nw = NW;
}

// ---- reference: one thread per output, k-ascending fmaf chain
__global__ void ref_gemm(const float* __restrict__ a, const float* __restrict__ f, float* __restrict__ out, int M, int N, int K) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    const int m = (int)(i / N), n = (int)(i % N);
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (double)a[(int64_t)m * K + k] * (double)f[(int64_t)n * K + k];
    out[i] = (float)s;
}
__global__ void fill_kernel(float* p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void maxdiff_kernel(const float* a, const float* b, int64_t n, float* out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(a[i] - b[i]));
    atomicMax(reinterpret_cast<int*>(out), __float_as_int(m));
}

static float* g_flush = nullptr;
static const int64_t kFlushN = (int64_t)1 << 26;
static const char* g_name = "";
template <class F> static double time_launch(F&& fn, int reps, bool flush) {
    if (getenv("LAB_ONLY") && !strstr(g_name, getenv("LAB_ONLY"))) return -1.0;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<float> ts;
    fn();
    CHECK(hipDeviceSynchronize());
    for (int r = 0; r < reps; ++r) {
        if (flush) hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, g_flush, kFlushN, 0.f);
        CHECK(hipEventRecord(e0, 0));
        fn();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return ts[ts.size() / 2] * 1e-3;
}

template <int BM, int NW, int ABL = 0> static void launch_resident(const float* a, const float* f, float* out, int M, int N, int K, int col_ranges) {
    RParams p;
    p.a = a; p.f = f; p.out = out; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldf = K; p.ldo = N;
    p.Kp = (K + 31) / 32 * 32;
    const int stripes = (N + 31) / 32;
    p.col_ranges = col_ranges;
    p.stripes_per_range = (stripes + col_ranges - 1) / col_ranges;
    const int row_tiles8 = ((M + BM - 1) / BM + 7) / 8 * 8;
    const size_t lds = (size_t)BM * (p.Kp + 4) * 4;
    static bool attr = false;
    if (!attr) { CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_resident<BM, NW, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
    hipLaunchKernelGGL((gemm_resident<BM, NW, ABL>), dim3(row_tiles8 * col_ranges), dim3(NW * 64), lds, 0, p);
}

int main(int argc, char** argv) {
    std::vector<std::array<int, 3>> shapes;
    for (int i = 1; i + 2 < argc; i += 3) shapes.push_back({atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2])});
    if (shapes.empty()) shapes = {{31360, 960, 160}, {25088, 960, 160}, {7840, 1536, 256}, {6272, 1536, 256}, {31360, 512, 128}, {125440, 256, 64}};
    CHECK(hipMalloc(&g_flush, kFlushN * 4));
    CHECK(hipMalloc(&g_dbg, (size_t)8192 * 16 * 24 * 8));
    CHECK(hipMemset(g_dbg, 0, (size_t)8192 * 16 * 24 * 8));
    const bool flush = !getenv("LAB_NOFLUSH");
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        float *a, *f, *out, *ref, *d;
        CHECK(hipMalloc(&a, (size_t)M * K * 4)); CHECK(hipMalloc(&f, (size_t)N * K * 4));
        CHECK(hipMalloc(&out, (size_t)M * N * 4)); CHECK(hipMalloc(&ref, (size_t)M * N * 4)); CHECK(hipMalloc(&d, 4));
        std::vector<float> ha((size_t)M * K), hf((size_t)N * K);
        uint32_t st = 12345u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto& v : ha) v = rnd();
        for (auto& v : hf) v = rnd();
        CHECK(hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(f, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(ref_gemm, dim3((unsigned)(((int64_t)M * N + 255) / 256)), dim3(256), 0, 0, a, f, ref, M, N, K);
        CHECK(hipDeviceSynchronize());
        const double fl = 2.0 * M * N * K;
        const char* only = getenv("LAB_ONLY");
        auto report = [&](const char* name, double tsec) {
            if (tsec < 0) return;
            CHECK(hipMemset(d, 0, 4));
            hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, 0, out, ref, (int64_t)M * N, d);
            float md = 0.f;
            CHECK(hipMemcpy(&md, d, 4, hipMemcpyDeviceToHost));
            printf("M=%6d N=%5d K=%5d  %-28s %8.1f us  %6.1f TF/s   maxdiff %.2e\n", M, N, K, name, tsec * 1e6, fl / tsec / 1e12, md);
            fflush(stdout);
            CHECK(hipMemset(out, 0xff, (size_t)M * N * 4));
        };
        // library tiles (1x1 conv descriptor: n = 1, h = M, w = 1)
        suhpe_conv_t cd;
        cd.n = 1; cd.h = M; cd.w = 1; cd.cin = K; cd.ho = M; cd.wo = 1; cd.cout = N; cd.r = 1; cd.s = 1; cd.stride = 1; cd.pad = 0; cd.groups = 1;
        for (int tile : {1, 2, 4}) {
            suhpe_conv2d_set_tile(&cd, 0, tile);
            g_name = "lib";
            const double tl = time_launch([&] { suhpe_conv2d_fwd(&cd, a, f, nullptr, out, nullptr); }, 9, flush);
            char nm[64]; snprintf(nm, sizeof nm, "lib %s", suhpe_conv2d_tile_name(0, tile) + 15);
            report(nm, tl);
        }
        const int cr = getenv("LAB_RANGES") ? atoi(getenv("LAB_RANGES")) : 1;
        const bool fits128 = (size_t)128 * ((K + 31) / 32 * 32 + 4) * 4 <= 160 * 1024;
        if (fits128) {
            g_name = "R<128,8>", report("R<128,8>", time_launch([&] { launch_resident<128, 8>(a, f, out, M, N, K, cr); }, 9, flush));
            g_name = "R2<128,4,8>", report("R2<128,4,8>", time_launch([&] { launch_resident2<128, 4, 8>(a, f, out, M, N, K, cr); }, 9, flush));
            g_name = "R2<128,2,8>", report("R2<128,2,8>", time_launch([&] { launch_resident2<128, 2, 8>(a, f, out, M, N, K, cr); }, 9, flush));
            g_name = "R2<128,2,8> no stores", report("R2<128,2,8> no stores", time_launch([&] { launch_resident2<128, 2, 8, 1>(a, f, out, M, N, K, cr); }, 9, flush));
            g_name = "R2<128,1,8>", report("R2<128,1,8>", time_launch([&] { launch_resident2<128, 1, 8>(a, f, out, M, N, K, cr); }, 9, flush));
        }
        if (fits128) {
            g_name = "R3<128,4,8>", report("R3<128,4,8>", time_launch([&] { launch_resident3<128, 4, 8>(a, f, out, M, N, K, cr); }, 9, flush));
            g_name = "R3<128,2,8>", report("R3<128,2,8>", time_launch([&] { launch_resident3<128, 2, 8>(a, f, out, M, N, K, cr); }, 9, flush));
            g_name = "R3<128,2,8> no stores", report("R3<128,2,8> no stores", time_launch([&] { launch_resident3<128, 2, 8, 1>(a, f, out, M, N, K, cr); }, 9, flush));
        }
        if (fits128) {
            g_name = "R3<128,2,12>", report("R3<128,2,12>", time_launch([&] { launch_resident3<128, 2, 12>(a, f, out, M, N, K, cr); }, 9, flush));
            g_name = "R3<128,2,16>", report("R3<128,2,16>", time_launch([&] { launch_resident3<128, 2, 16>(a, f, out, M, N, K, cr); }, 9, flush));
            g_name = "R3<128,1,16>", report("R3<128,1,16>", time_launch([&] { launch_resident3<128, 1, 16>(a, f, out, M, N, K, cr); }, 9, flush));
            g_name = "R3<128,4,12>", report("R3<128,4,12>", time_launch([&] { launch_resident3<128, 4, 12>(a, f, out, M, N, K, cr); }, 9, flush));
            g_name = "R2<128,2,12>", report("R2<128,2,12>", time_launch([&] { launch_resident2<128, 2, 12>(a, f, out, M, N, K, cr); }, 9, flush));
            g_name = "R2<128,2,16>", report("R2<128,2,16>", time_launch([&] { launch_resident2<128, 2, 16>(a, f, out, M, N, K, cr); }, 9, flush));
        }
        g_name = "R3<64,2,12>", report("R3<64,2,12>", time_launch([&] { launch_resident3<64, 2, 12>(a, f, out, M, N, K, cr); }, 9, flush));
        g_name = "R3<64,1,16>", report("R3<64,1,16>", time_launch([&] { launch_resident3<64, 1, 16>(a, f, out, M, N, K, cr); }, 9, flush));
        g_name = "R3<64,2,8>", report("R3<64,2,8>", time_launch([&] { launch_resident3<64, 2, 8>(a, f, out, M, N, K, cr); }, 9, flush));
        g_name = "R3<64,2,4>", report("R3<64,2,4>", time_launch([&] { launch_resident3<64, 2, 4>(a, f, out, M, N, K, cr); }, 9, flush));
        if (getenv("LAB_STAMPS")) {
            CHECK(hipMemset(g_dbg, 0, (size_t)8192 * 16 * 24 * 8));
            if (atoi(getenv("LAB_STAMPS")) == 128) launch_resident3<128, 2, 8, 32>(a, f, out, M, N, K, cr); else launch_resident3<64, 2, 8, 32>(a, f, out, M, N, K, cr);
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> hd((size_t)g_dbg_blocks * g_dbg_nw * 24);
            CHECK(hipMemcpy(hd.data(), g_dbg, hd.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull;
            for (size_t i = 0; i < hd.size(); i += 24) if (hd[i] && hd[i] < t0) t0 = hd[i];
            printf("# stamps: block wave hwid(hex) xcc | start(10ns since first wave) staged(cycles since wave start) | unit ends (cycles) | end(cycles) end(10ns)\n");
            for (int bq = 0; bq < g_dbg_blocks; ++bq)
                for (int w = 0; w < g_dbg_nw; ++w) {
                    const unsigned long long* d = &hd[((size_t)bq * g_dbg_nw + w) * 24];
                    if (!d[0]) continue;
                    printf("S %4d %d %05llx %llu |", bq, w, d[1] & 0xfffff, d[1] >> 32);
                    printf(" %llu %llu |", d[0] - t0, d[2] - d[22]);
                    for (int k = 3; k < 21; ++k) if (d[k]) printf(" %llu", d[k] - d[22]);
                    printf(" | %llu %llu\n", d[21] - d[22], d[23] - t0);
                }
        }
        g_name = "R2<64,2,8>", report("R2<64,2,8>", time_launch([&] { launch_resident2<64, 2, 8>(a, f, out, M, N, K, cr); }, 9, flush));
        g_name = "R2<64,2,4>", report("R2<64,2,4>", time_launch([&] { launch_resident2<64, 2, 4>(a, f, out, M, N, K, cr); }, 9, flush));
        g_name = "R2<64,1,8>", report("R2<64,1,8>", time_launch([&] { launch_resident2<64, 1, 8>(a, f, out, M, N, K, cr); }, 9, flush));
        CHECK(hipFree(a)); CHECK(hipFree(f)); CHECK(hipFree(out)); CHECK(hipFree(ref)); CHECK(hipFree(d));
    }
    return 0;
}
