// Pointwise-GEMM laboratory (round 5): a persistent fp32-MFMA GEMM whose BOTH operands travel global -> LDS by LDS-DMA through a ring of
// k-slabs, timed against the library's tiles on the short-reduction pointwise shapes of the C2 step.
//     out[M][N] = A[M][K] * F[N][K]^T        (A = pixels x channels-in, F = filter rows, both K-contiguous; K % 32 == 0)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/proto/pw_lab.hip -L semiuhpe_amd/csrc -lsuhpe -Wl,-rpath,'$ORIGIN/../../../semiuhpe_amd/csrc' -o tools/proto/bin/pw_lab
//   tools/proto/bin/pw_lab [M N K]...
//
// Structure (what conv_wino.hip's pipeline looks like for a plain GEMM):
//   * one workgroup per CU, persistent over 128 x 128 output tiles (XCD-aware order: the column tiles of a row tile consecutively on one XCD);
//   * the (tile, k-slab) pairs of a workgroup form ONE stream; slab = 32 k of both operands = 16 KB + 16 KB, copied by
//     global_load_lds_dwordx4 into a ring of STAGES slabs, STAGES - 1 slabs ahead of the MFMAs, across tile boundaries;
//   * LDS rows are 128 bytes, unpadded; the 16-byte chunk c of row r sits at chunk position c ^ ((r >> 1) & 7) (the DMA lane that fills a position
//     fetches the chunk that belongs there), which makes every ds_read_b128 of the MFMA fragments conflict-free;
//   * ONE barrier per slab, in the MIDDLE of the slab's MFMAs: it publishes slab k + 1 (each wavefront has waited for its own DMA pieces) and frees
//     slab k - 1's stage for the next DMA; fragment reads run one k-group ahead, also across the slab boundary, so the matrix pipe never waits
//     for a barrier with nothing to do;
//   * no vector-memory operation is visible to hipcc: DMA, stores and waits are inline asm with hand-counted vmcnt (see conv_wino.hip's notes).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
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
typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0x80000000u;

struct PwParams {
    const float* a; const float* f; float* y;
    int M, N, K;
    int lda, ldf, ldy;          // row strides in floats
    int nk;                     // K / 32
    int col_tiles, row_tiles;   // 128-wide tiles
    int total_vb;               // virtual tile count = col_tiles * (row_tiles rounded up to 8)
};

constexpr int PW_BM = 128, PW_BN = 128, PW_BK = 32;
constexpr int PW_STAGE_FLOATS = (PW_BM + PW_BN) * PW_BK;        // 8192 floats = 32 KB

// ABL (timing experiments, wrong results): 1 = no output stores, 2 = no DMA (operands stale), 4 = no MFMAs, 8 = no barrier
template <int TM, int TN, int STAGES, int ABL = 0, bool SWAP = false>
__global__ __launch_bounds__((4 / TM) * (4 / TN) * 64) void pw_dma_kernel(const PwParams p) {
    constexpr int WAVES_N = 4 / TN, NW = (4 / TM) * (4 / TN);
    constexpr int PPW = 32 / NW;                                  // DMA pieces (1 KiB) per wavefront and slab
    constexpr int NST = (SWAP ? 4 : 16) * TM * TN;                // output stores per wavefront and tile
    static_assert(STAGES >= 3 && STAGES <= 5, "ring depth");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int gstride = gridDim.x;

    auto tile_of = [&](int vb, int* m0, int* n0) __attribute__((always_inline)) {
        const int jx = vb >> 3;
        *n0 = (jx % p.col_tiles) * PW_BN;
        *m0 = ((jx / p.col_tiles) * 8 + (vb & 7)) * PW_BM;
    };

    // ---- loader cursor over the stream of (tile, slab) pairs; per tile each wavefront owns PPW pieces: q < 16 -> rows 8 q .. 8 q + 7 of A, else of F
    int l_vb = blockIdx.x, l_kt = 0, l_stage = 0;
    unsigned voff[PPW];
    auto setup_loader = [&]() __attribute__((always_inline)) {
        int m0, n0;
        tile_of(l_vb, &m0, &n0);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave * PPW + i;                        // (uniform)
            const int rl = 8 * (q & 15) + (lane >> 3);           // row inside the operand tile
            const int chunk = (lane & 7) ^ ((rl >> 1) & 7);      // the chunk that belongs at this lane's position
            if (q < 16) {
                int r = m0 + rl; r = r < p.M ? r : p.M - 1;
                voff[i] = ((unsigned)r * (unsigned)p.lda + 4u * chunk) * 4u;
            } else {
                int r = n0 + rl; r = r < p.N ? r : p.N - 1;
                voff[i] = ((unsigned)r * (unsigned)p.ldf + 4u * chunk) * 4u;
            }
        }
    };
    auto issue_slab = [&]() __attribute__((always_inline)) {     // the loader cursor's slab -> stage l_stage
        if (!(ABL & 2)) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int q = wave * PPW + i;
                const float* base = (q < 16 ? p.a : p.f) + l_kt * PW_BK;
                const unsigned lds_dst = __builtin_amdgcn_readfirstlane(
                    (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(smem + l_stage * PW_STAGE_FLOATS + q * 256));
                unsigned keep;
                asm volatile(
                    "s_mov_b32 %0, m0\n\t"
                    "s_mov_b32 m0, %2\n\t"
                    "s_nop 2\n\t"
                    "global_load_lds_dwordx4 %1, %3\n\t"
                    "s_mov_b32 m0, %0"
                    : "=&s"(keep) : "v"(voff[i]), "s"(lds_dst), "s"(base) : "memory");
            }
        }
        l_stage = l_stage == STAGES - 1 ? 0 : l_stage + 1;
        if (++l_kt == p.nk) {
            l_kt = 0;
            l_vb += gstride;
            if (l_vb < p.total_vb) setup_loader();
        }
    };

    // ---- fragment addresses: lane (r32 = lane & 31, h = lane >> 5) reads row r32 of its 32-row block, k-group kg -> logical chunk 2 kg + h
    const int r32 = lane & 31, h = lane >> 5;
    int a_rd[TM], b_rd[TN];                                       // float offsets inside a stage for kg = 0; kg adds XOR (kg << 3) floats
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + r32;
        a_rd[i] = row * 32 + ((h ^ ((row >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 32 + r32;
        b_rd[j] = PW_BM * 32 + row * 32 + ((h ^ ((row >> 1) & 7)) << 2);
    }
    float4 af[2][TM], bf[2][TN];
    auto read_frags = [&](int stage, int kg, int set) __attribute__((always_inline)) {
        const float* S = smem + stage * PW_STAGE_FLOATS;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const float4*>(&S[a_rd[i] ^ (kg << 3)]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[set][j] = *reinterpret_cast<const float4*>(&S[b_rd[j] ^ (kg << 3)]);
    };
    floatx16 acc[TM][TN];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    auto mma = [&](int set) __attribute__((always_inline)) {
        if (ABL & 4) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (SWAP) {            // D[n][m]: a lane ends with 4 x 4 consecutive output channels of ONE pixel -> 16-byte stores
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[set][j].x, af[set][i].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[set][j].y, af[set][i].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[set][j].z, af[set][i].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[set][j].w, af[set][i].w, acc[i][j], 0, 0, 0);
                    continue;
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i].x, bf[set][j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i].y, bf[set][j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i].z, bf[set][j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i].w, bf[set][j].w, acc[i][j], 0, 0, 0);
            }
    };
    // buffer descriptor of the output as four scalars: a lane that must not store (row >= M, column >= N) gets an out-of-range offset and the hardware
    // drops its store -- EVERY wavefront issues exactly NST store instructions per tile, which the hand-counted waits of the slab loop rely on
    u32x4w yrs;
    {
        const uint64_t addr = reinterpret_cast<uint64_t>(p.y);
        const int64_t nb = (int64_t)p.M * p.ldy * 4;
        yrs.x = __builtin_amdgcn_readfirstlane((uint32_t)addr);
        yrs.y = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32) & 0xffffu);
        yrs.z = __builtin_amdgcn_readfirstlane((uint32_t)(nb > 0x7fffffff ? 0x7fffffff : nb));
        yrs.w = 0x00020000u;
    }
    auto epilogue = [&](int vb) __attribute__((always_inline)) {
        int m0, n0;
        tile_of(vb, &m0, &n0);
        // (the stores are asm statements: hipcc's hazard recognizer does not pad an MFMA result -> vector-memory read for them; 20 wait states cover
        //  the 16-pass MFMA that wrote the last accumulator)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
        if constexpr (SWAP) {
            typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = m0 + (wm * TM + i) * 32 + r32;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col0 = n0 + (wn * TN + j) * 32 + 4 * h;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int col = col0 + 8 * q;
                        const bool ok = row < p.M && col < p.N && !(ABL & 1);
                        const unsigned o = ok ? ((unsigned)row * (unsigned)p.ldy + (unsigned)col) * 4u : kOob;
                        const f4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        asm volatile("buffer_store_dwordx4 %1, %0, %2, 0 offen\n\ts_nop 1" : : "v"(o), "v"(v), "s"(yrs) : "memory");
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + r32;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row0 = m0 + (wm * TM + i) * 32 + 4 * h;
                unsigned off = ((unsigned)row0 * (unsigned)p.ldy + (unsigned)col) * 4u;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = row0 + (e & 3) + 8 * (e >> 2);
                    const bool ok = row < p.M && col < p.N && !(ABL & 1);
                    const unsigned o = ok ? off + (unsigned)(((e & 3) + 8 * (e >> 2)) * p.ldy) * 4u : kOob;
                    const float v = acc[i][j][e];
                    asm volatile("buffer_store_dword %1, %0, %2, 0 offen" : : "v"(o), "v"(v), "s"(yrs) : "memory");
                }
            }
        }
    };

    if ((int)blockIdx.x >= p.total_vb) return;
    // my stream
    const int my_tiles = (p.total_vb - (int)blockIdx.x + gstride - 1) / gstride;
    const int n_slabs = my_tiles * p.nk;
    setup_loader();
    int issued = 0;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (issued < n_slabs) { issue_slab(); ++issued; }
    // slab 0 landed: everything but the slabs issued after it
    if (n_slabs >= STAGES - 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"((STAGES - 2) * PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    zero_acc();
    int c_vb = blockIdx.x, c_kt = 0, c_stage = 0;
    // bit b: a tile ended (its NST stores were issued) at the end of slab g - 1 - b.  Slab g + 1's DMA was issued at mid-slab g + 2 - STAGES, so the
    // stores of tiles that ended at slabs g + 2 - STAGES .. g - 1 are younger than it: bits 0 .. STAGES - 3
    unsigned ended = 0;
    constexpr unsigned ENDED_MASK = (1u << (STAGES - 2)) - 1u;
    read_frags(0, 0, 0);
    for (int g = 0; g < n_slabs; ++g) {
        const int n_stage = c_stage == STAGES - 1 ? 0 : c_stage + 1;
        read_frags(c_stage, 1, 1);
        mma(0);
        read_frags(c_stage, 2, 0);
        mma(1);
        // ---- mid-slab: publish slab g + 1, free slab g - 1's stage, issue slab g + STAGES - 1
        if (g + 1 < n_slabs) {
            // operations younger than slab g + 1's DMA: the slabs issued after it, and the stores of a tile that ended at slab g - 1 or g - 2
            if (issued - (g + 2) == STAGES - 3) {
                if (ended & ENDED_MASK) asm volatile("s_waitcnt vmcnt(%0)" : : "n"((STAGES - 3) * PPW + NST > 63 ? 63 : (STAGES - 3) * PPW + NST) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" : : "n"((STAGES - 3) * PPW) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if (!(ABL & 8)) __builtin_amdgcn_s_barrier();
        if (issued < n_slabs) { issue_slab(); ++issued; }
        read_frags(c_stage, 3, 1);
        mma(0);
        const bool last = c_kt == p.nk - 1;
        if (g + 1 < n_slabs) read_frags(n_stage, 0, 0);
        mma(1);
        ended <<= 1;
        if (last) {
            epilogue(c_vb);
            zero_acc();
            c_kt = 0;
            c_vb += gstride;
            ended |= 1u;
        } else {
            ++c_kt;
        }
        c_stage = n_stage;
    }
}


// ------------------------------------------------------------------------------------------------ v2: nothing happens in bursts
// The v1 ablations (profiles/r05_pw_lab.txt): output stores 10 %, DMA issue 7 %, barrier 3 % of the launch -- every wavefront of the workgroup runs the
// same program between the same barriers, so all of them issue their DMA pieces together (right behind the barrier) and all of them store their
// accumulators together (tile end), and for that long no wavefront of the CU feeds the matrix pipe.  v2 keeps the ring and the mid-slab barrier and
// spreads both: ONE vector-memory instruction between two MFMAs, never more -- the DMA pieces of a slab behind the first MFMAs after the barrier, the
// stores of a finished tile (kept in a second register set) one per MFMA over the whole next slab.  STAGES = 4.
template <int TM, int TN, int ABL = 0>
__global__ __launch_bounds__((4 / TM) * (4 / TN) * 64) void pw_dma2_kernel(const PwParams p) {
    constexpr int STAGES = 4;
    constexpr int WAVES_N = 4 / TN, NW = (4 / TM) * (4 / TN);
    constexpr int PPW = 32 / NW;                                  // DMA pieces per wavefront and slab
    constexpr int SLOTS = 4 * TM * TN;                            // MFMAs per k-group = stores of a held tile per k-group (NST / 4)
    static_assert(PPW <= SLOTS, "one DMA piece per MFMA slot");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int gstride = gridDim.x;
    auto tile_of = [&](int vb, int* m0, int* n0) __attribute__((always_inline)) {
        const int jx = vb >> 3;
        *n0 = (jx % p.col_tiles) * PW_BN;
        *m0 = ((jx / p.col_tiles) * 8 + (vb & 7)) * PW_BM;
    };
    int l_vb = blockIdx.x, l_kt = 0, l_stage = 0;
    unsigned voff[PPW];
    auto setup_loader = [&]() __attribute__((always_inline)) {
        int m0, n0;
        tile_of(l_vb, &m0, &n0);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave * PPW + i;
            const int rl = 8 * (q & 15) + (lane >> 3);
            const int chunk = (lane & 7) ^ ((rl >> 1) & 7);
            if (q < 16) {
                int r = m0 + rl; r = r < p.M ? r : p.M - 1;
                voff[i] = ((unsigned)r * (unsigned)p.lda + 4u * chunk) * 4u;
            } else {
                int r = n0 + rl; r = r < p.N ? r : p.N - 1;
                voff[i] = ((unsigned)r * (unsigned)p.ldf + 4u * chunk) * 4u;
            }
        }
    };
    auto issue_piece = [&](auto idx) __attribute__((always_inline)) {        // piece idx of the loader cursor's slab
        constexpr int i = decltype(idx)::value;
        if (ABL & 2) return;
        const int q = wave * PPW + i;
        const float* base = (q < 16 ? p.a : p.f) + l_kt * PW_BK;
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane(
            (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(smem + l_stage * PW_STAGE_FLOATS + q * 256));
        unsigned keep;
        const unsigned vo = voff[i];                             // (a generic lambda captures a variable only through an ordinary use, not through an asm operand)
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 2\n\t"
            "global_load_lds_dwordx4 %1, %3\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "v"(vo), "s"(lds_dst), "s"(base) : "memory");
    };
    auto advance_loader = [&]() __attribute__((always_inline)) {
        l_stage = l_stage == STAGES - 1 ? 0 : l_stage + 1;
        if (++l_kt == p.nk) {
            l_kt = 0;
            l_vb += gstride;
            if (l_vb < p.total_vb) setup_loader();
        }
    };
    const int r32 = lane & 31, h = lane >> 5;
    int a_rd[TM], b_rd[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + r32;
        a_rd[i] = row * 32 + ((h ^ ((row >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 32 + r32;
        b_rd[j] = PW_BM * 32 + row * 32 + ((h ^ ((row >> 1) & 7)) << 2);
    }
    float4 af[2][TM], bf[2][TN];
    auto read_frags = [&](int stage, int kg, int set) __attribute__((always_inline)) {
        const float* S = smem + stage * PW_STAGE_FLOATS;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const float4*>(&S[a_rd[i] ^ (kg << 3)]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[set][j] = *reinterpret_cast<const float4*>(&S[b_rd[j] ^ (kg << 3)]);
    };
    floatx16 acc[TM][TN], hold[TM][TN];
    u32x4w yrs;
    {
        const uint64_t addr = reinterpret_cast<uint64_t>(p.y);
        const int64_t nb = (int64_t)p.M * p.ldy * 4;
        yrs.x = __builtin_amdgcn_readfirstlane((uint32_t)addr);
        yrs.y = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32) & 0xffffu);
        yrs.z = __builtin_amdgcn_readfirstlane((uint32_t)(nb > 0x7fffffff ? 0x7fffffff : nb));
        yrs.w = 0x00020000u;
    }
    // the held tile: byte offset of this lane's element of column block j (register 0 of row block 0), or out of range for a column beyond N; a row
    // beyond M lands beyond the descriptor's extent by itself (offset >= M * ldy * 4), so a store costs one scalar multiply and one vector add
    unsigned h_off[TN];
    const int ldy4 = p.ldy * 4;
    auto store_one = [&](auto t_) __attribute__((always_inline)) {            // store number t of the held tile
        constexpr int t = decltype(t_)::value;
        constexpr int e = t % 16, blk = t / 16, j = blk % TN, i = blk / TN;
        const unsigned o = h_off[j] + (unsigned)((i * 32 + (e & 3) + 8 * (e >> 2)) * ldy4);
        const float v = hold[i][j][e];
        const u32x4w yr = yrs;
        asm volatile("buffer_store_dword %1, %0, %2, 0 offen" : : "v"(o), "v"(v), "s"(yr) : "memory");
    };
    // one k-group: SLOTS MFMAs on fragment set `set`; behind MFMA m: DMA piece m (DMA phase) and store Q * SLOTS + m of the held tile (STORES)
    auto kgroup = [&](auto set_, auto q_, auto dma_, auto stores_) __attribute__((always_inline)) {
        constexpr int set = decltype(set_)::value, Q = decltype(q_)::value;
        constexpr bool DMA = decltype(dma_)::value, STORES = decltype(stores_)::value;
        auto slot = [&](auto m_) __attribute__((always_inline)) {
            constexpr int m = decltype(m_)::value;
            if constexpr (DMA && m < PPW) issue_piece(std::integral_constant<int, m>{});
            if constexpr (STORES) store_one(std::integral_constant<int, Q * SLOTS + m>{});
            if constexpr ((DMA && m < PPW) || STORES) __builtin_amdgcn_sched_barrier(0);
        };
        auto four = [&](auto i_, auto j_) __attribute__((always_inline)) {
            constexpr int i = decltype(i_)::value, j = decltype(j_)::value, m0 = (i * TN + j) * 4;
            if (!(ABL & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i].x, bf[set][j].x, acc[i][j], 0, 0, 0);
            slot(std::integral_constant<int, m0>{});
            if (!(ABL & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i].y, bf[set][j].y, acc[i][j], 0, 0, 0);
            slot(std::integral_constant<int, m0 + 1>{});
            if (!(ABL & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i].z, bf[set][j].z, acc[i][j], 0, 0, 0);
            slot(std::integral_constant<int, m0 + 2>{});
            if (!(ABL & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i].w, bf[set][j].w, acc[i][j], 0, 0, 0);
            slot(std::integral_constant<int, m0 + 3>{});
        };
        four(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        if constexpr (TN == 2) four(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        if constexpr (TM == 2) {
            four(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            if constexpr (TN == 2) four(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        }
    };
    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;

    if ((int)blockIdx.x >= p.total_vb) return;
    const int my_tiles = (p.total_vb - (int)blockIdx.x + gstride - 1) / gstride;
    const int n_slabs = my_tiles * p.nk;
    setup_loader();
    int issued = 0;
    auto issue_all = [&]() __attribute__((always_inline)) {       // (prologue only: a whole slab at once)
        issue_piece(C0{});
        if constexpr (PPW > 1) issue_piece(C1{});
        if constexpr (PPW > 2) { issue_piece(C2{}); issue_piece(C3{}); }
        if constexpr (PPW > 4) { issue_piece(std::integral_constant<int, 4>{}); issue_piece(std::integral_constant<int, 5>{}); issue_piece(std::integral_constant<int, 6>{}); issue_piece(std::integral_constant<int, 7>{}); }
        advance_loader();
    };
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (issued < n_slabs) { issue_all(); ++issued; }
    if (n_slabs >= STAGES - 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"((STAGES - 2) * PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int c_vb = blockIdx.x, c_kt = 0, c_stage = 0;
    // sflags bit b: slab g - b is a store slab (it carries the stores of the tile that ended at slab g - b - 1)
    unsigned sflags = 0;
    read_frags(0, 0, 0);
    auto slab = [&](int g, auto stores_) __attribute__((always_inline)) {
        constexpr bool ST = decltype(stores_)::value;
        using STt = std::integral_constant<bool, ST>;
        const int n_stage = c_stage == STAGES - 1 ? 0 : c_stage + 1;
        read_frags(c_stage, 1, 1);
        kgroup(C0{}, C0{}, std::false_type{}, STt{});
        read_frags(c_stage, 2, 0);
        kgroup(C1{}, C1{}, std::false_type{}, STt{});
        if (g + 1 < n_slabs) {
            // younger than the last DMA piece of slab g + 1 (issued in k-group 2 of slab g - 2): at least the k-group-3 stores of slab g - 2, all stores of
            // slab g - 1, the first two k-groups' stores of slab g, and the DMA of slab g + 2; a smaller count only waits longer
            if (issued - (g + 2) == STAGES - 3) {
                const int nst = SLOTS * (int)(((sflags >> 2) & 1u) + 4u * ((sflags >> 1) & 1u) + 2u * (sflags & 1u));
                if (nst >= 4 * SLOTS) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PPW + 4 * SLOTS > 63 ? 63 : PPW + 4 * SLOTS) : "memory");
                else if (nst >= 2 * SLOTS) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PPW + 2 * SLOTS > 63 ? 63 : PPW + 2 * SLOTS) : "memory");
                else if (nst >= SLOTS) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PPW + SLOTS) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PPW) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if (!(ABL & 8)) __builtin_amdgcn_s_barrier();
        read_frags(c_stage, 3, 1);
        if (issued < n_slabs) {
            kgroup(C0{}, C2{}, std::true_type{}, STt{});
            advance_loader();
            ++issued;
        } else {
            kgroup(C0{}, C2{}, std::false_type{}, STt{});
        }
        if (g + 1 < n_slabs) read_frags(n_stage, 0, 0);
        kgroup(C1{}, C3{}, std::false_type{}, STt{});
        c_stage = n_stage;
    };
    for (int g = 0; g < n_slabs; ++g) {
        const bool last = c_kt == p.nk - 1;
        if (sflags & 1u) slab(g, std::true_type{}); else slab(g, std::false_type{});
        sflags <<= 1;
        if (last) {
            // the finished tile moves to the second register set; its stores ride in the next slab (or right here behind the last slab of the stream)
            int m0, n0;
            tile_of(c_vb, &m0, &n0);
            const int h_row0 = m0 + wm * TM * 32 + 4 * h;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + (wn * TN + j) * 32 + r32;
                h_off[j] = (col < p.N && !(ABL & 1)) ? ((unsigned)h_row0 * (unsigned)p.ldy + (unsigned)col) * 4u : kOob;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    hold[i][j] = acc[i][j];
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
                }
            c_kt = 0;
            c_vb += gstride;
            sflags |= 1u;
            if (g + 1 == n_slabs) {                              // end of the stream: flush
                auto flush = [&](auto t0_) __attribute__((always_inline)) {
                    constexpr int t0 = decltype(t0_)::value;
                    store_one(std::integral_constant<int, t0>{}); store_one(std::integral_constant<int, t0 + 1>{});
                    store_one(std::integral_constant<int, t0 + 2>{}); store_one(std::integral_constant<int, t0 + 3>{});
                };
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
                flush(std::integral_constant<int, 0>{}); flush(std::integral_constant<int, 4>{}); flush(std::integral_constant<int, 8>{}); flush(std::integral_constant<int, 12>{});
                if constexpr (TM * TN >= 2) { flush(std::integral_constant<int, 16>{}); flush(std::integral_constant<int, 20>{}); flush(std::integral_constant<int, 24>{}); flush(std::integral_constant<int, 28>{}); }
                if constexpr (TM * TN == 4) {
                    flush(std::integral_constant<int, 32>{}); flush(std::integral_constant<int, 36>{}); flush(std::integral_constant<int, 40>{}); flush(std::integral_constant<int, 44>{});
                    flush(std::integral_constant<int, 48>{}); flush(std::integral_constant<int, 52>{}); flush(std::integral_constant<int, 56>{}); flush(std::integral_constant<int, 60>{});
                }
            }
        } else {
            ++c_kt;
        }
    }
}

// ------------------------------------------------------------------------------------------------ harness
__global__ void ref_gemm(const float* __restrict__ a, const float* __restrict__ f, float* __restrict__ o, int M, int N, int K) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int m = (int)(idx / N), n = (int)(idx % N);
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (double)a[(int64_t)m * K + k] * (double)f[(int64_t)n * K + k];
    o[idx] = (float)s;
}
__global__ void maxdiff_kernel(const float* __restrict__ x, const float* __restrict__ r, int64_t n, float* out) {
    float md = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = fabsf(x[i] - r[i]);
        md = fmaxf(md, d != d ? 1e30f : d);
    }
    atomicMax(reinterpret_cast<int*>(out), __float_as_int(md));
}
__global__ void flush_kernel(float* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

template <typename F> static double time_launch(F fn, int iters, float* flush) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    fn(); CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int i = 0; i < iters; ++i) {
        hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(256), 0, 0, flush, (int64_t)1 << 26);       // 256 MiB: cold L2 / Infinity Cache
        CHECK(hipEventRecord(e0));
        fn();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms * 1e-3);
    }
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return best;
}

template <int TM, int TN, int STAGES, int ABL = 0, bool SWAP = false> static void launch_pw(const float* a, const float* f, float* y, int M, int N, int K, int grid) {
    PwParams p;
    p.a = a; p.f = f; p.y = y; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldf = K; p.ldy = N; p.nk = K / 32;
    p.col_tiles = (N + PW_BN - 1) / PW_BN; p.row_tiles = (M + PW_BM - 1) / PW_BM;
    p.total_vb = p.col_tiles * ((p.row_tiles + 7) / 8 * 8);
    auto kern = pw_dma_kernel<TM, TN, STAGES, ABL, SWAP>;
    const int lds = STAGES * PW_STAGE_FLOATS * 4;
    static bool set = false;
    if (!set) { CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); set = true; }
    constexpr int NT = (4 / TM) * (4 / TN) * 64;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, p);
}

template <int TM, int TN, int ABL = 0> static void launch_pw2(const float* a, const float* f, float* y, int M, int N, int K, int grid) {
    PwParams p;
    p.a = a; p.f = f; p.y = y; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldf = K; p.ldy = N; p.nk = K / 32;
    p.col_tiles = (N + PW_BN - 1) / PW_BN; p.row_tiles = (M + PW_BM - 1) / PW_BM;
    p.total_vb = p.col_tiles * ((p.row_tiles + 7) / 8 * 8);
    auto kern = pw_dma2_kernel<TM, TN, ABL>;
    const int lds = 4 * PW_STAGE_FLOATS * 4;
    static bool set = false;
    if (!set) { CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); set = true; }
    constexpr int NT = (4 / TM) * (4 / TN) * 64;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, p);
}

int main(int argc, char** argv) {
    std::vector<std::array<int, 3>> shapes;
    for (int i = 1; i + 2 < argc; i += 3) shapes.push_back({atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2])});
    if (shapes.empty()) shapes = {{31360, 960, 160}, {7840, 1536, 256}, {31360, 512, 128}, {31360, 160, 960}, {7840, 256, 1536}, {1000, 200, 64}};
    float* flush; CHECK(hipMalloc(&flush, (size_t)1 << 28));
    const int grid = getenv("LAB_GRID") ? atoi(getenv("LAB_GRID")) : 256;
    for (auto s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        if (K % 32) { printf("K %% 32 != 0: skipped\n"); continue; }
        float *a, *f, *out, *ref, *d;
        CHECK(hipMalloc(&a, (size_t)M * K * 4)); CHECK(hipMalloc(&f, (size_t)N * K * 4)); CHECK(hipMalloc(&out, (size_t)M * N * 4));
        CHECK(hipMalloc(&ref, (size_t)M * N * 4)); CHECK(hipMalloc(&d, 4));
        std::vector<float> ha((size_t)M * K), hf((size_t)N * K);
        uint32_t st = 12345u + M;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.f - 1.f; };
        for (auto& v : ha) v = rnd();
        for (auto& v : hf) v = rnd();
        CHECK(hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(f, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(ref_gemm, dim3((unsigned)(((int64_t)M * N + 255) / 256)), dim3(256), 0, 0, a, f, ref, M, N, K);
        CHECK(hipDeviceSynchronize());
        const double fl = 2.0 * M * N * K;
        auto report = [&](const char* name, double tsec) {
            CHECK(hipMemset(d, 0, 4));
            hipLaunchKernelGGL(maxdiff_kernel, dim3(1024), dim3(256), 0, 0, out, ref, (int64_t)M * N, d);
            float md = 0.f;
            CHECK(hipMemcpy(&md, d, 4, hipMemcpyDeviceToHost));
            printf("M=%6d N=%5d K=%5d  %-30s %8.1f us  %6.1f TF/s   maxdiff %.2e\n", M, N, K, name, tsec * 1e6, fl / tsec / 1e12, md);
            fflush(stdout);
            CHECK(hipMemset(out, 0xff, (size_t)M * N * 4));
        };
        if (!getenv("LAB_NOLIB")) {
            suhpe_conv_t cd;
            cd.n = 1; cd.h = M; cd.w = 1; cd.cin = K; cd.ho = M; cd.wo = 1; cd.cout = N; cd.r = 1; cd.s = 1; cd.stride = 1; cd.pad = 0; cd.groups = 1;
            const int nt = suhpe_conv2d_tile_count(0);
            for (int tile = 0; tile < nt; ++tile) {
                const char* tn = suhpe_conv2d_tile_name(0, tile);
                if (strstr(tn, "3x3") || strstr(tn, "wino") || strstr(tn, "256,") || strstr(tn, "128,128")) continue;
                if (suhpe_conv2d_set_tile(&cd, 0, tile) != 0) continue;
                const double tl = time_launch([&] { suhpe_conv2d_fwd(&cd, a, f, nullptr, out, nullptr); }, 7, flush);
                char nm[80]; snprintf(nm, sizeof nm, "lib %s", tn);
                report(nm, tl);
            }
            suhpe_conv2d_set_tile(&cd, 0, -1);
        }
        report("pw<1,1,4> 16 waves", time_launch([&] { launch_pw<1, 1, 4>(a, f, out, M, N, K, grid); }, 7, flush));
        report("pw<2,1,4> 8 waves", time_launch([&] { launch_pw<2, 1, 4>(a, f, out, M, N, K, grid); }, 7, flush));
        report("pw<1,2,4> 8 waves", time_launch([&] { launch_pw<1, 2, 4>(a, f, out, M, N, K, grid); }, 7, flush));
        report("pw<2,2,4> 4 waves", time_launch([&] { launch_pw<2, 2, 4>(a, f, out, M, N, K, grid); }, 7, flush));
        report("pw2<1,1> spread", time_launch([&] { launch_pw2<1, 1>(a, f, out, M, N, K, grid); }, 7, flush));
        report("pw2<2,1> spread", time_launch([&] { launch_pw2<2, 1>(a, f, out, M, N, K, grid); }, 7, flush));
        report("pw2<1,2> spread", time_launch([&] { launch_pw2<1, 2>(a, f, out, M, N, K, grid); }, 7, flush));
        report("pw2<2,2> spread", time_launch([&] { launch_pw2<2, 2>(a, f, out, M, N, K, grid); }, 7, flush));
        if (getenv("LAB_ABL")) {
            report("pw2<1,1> no stores", time_launch([&] { launch_pw2<1, 1, 1>(a, f, out, M, N, K, grid); }, 7, flush));
            report("pw2<1,1> no dma", time_launch([&] { launch_pw2<1, 1, 2>(a, f, out, M, N, K, grid); }, 7, flush));
            report("pw2<1,1> no barrier", time_launch([&] { launch_pw2<1, 1, 8>(a, f, out, M, N, K, grid); }, 7, flush));
            report("pw2<1,1> no st/dma/barrier", time_launch([&] { launch_pw2<1, 1, 11>(a, f, out, M, N, K, grid); }, 7, flush));
        }
        report("pw<1,1,4> swap x4 stores", time_launch([&] { launch_pw<1, 1, 4, 0, true>(a, f, out, M, N, K, grid); }, 7, flush));
        report("pw<2,1,4> swap x4 stores", time_launch([&] { launch_pw<2, 1, 4, 0, true>(a, f, out, M, N, K, grid); }, 7, flush));
        report("pw<1,2,4> swap x4 stores", time_launch([&] { launch_pw<1, 2, 4, 0, true>(a, f, out, M, N, K, grid); }, 7, flush));
        report("pw<2,2,4> swap x4 stores", time_launch([&] { launch_pw<2, 2, 4, 0, true>(a, f, out, M, N, K, grid); }, 7, flush));
        if (getenv("LAB_ABL")) {
            report("pw<2,1,4> no stores", time_launch([&] { launch_pw<2, 1, 4, 1>(a, f, out, M, N, K, grid); }, 7, flush));
            report("pw<2,1,4> no dma", time_launch([&] { launch_pw<2, 1, 4, 2>(a, f, out, M, N, K, grid); }, 7, flush));
            report("pw<2,1,4> no mfma", time_launch([&] { launch_pw<2, 1, 4, 4>(a, f, out, M, N, K, grid); }, 7, flush));
            report("pw<2,1,4> no barrier", time_launch([&] { launch_pw<2, 1, 4, 8>(a, f, out, M, N, K, grid); }, 7, flush));
            report("pw<2,1,4> no st/dma/barrier", time_launch([&] { launch_pw<2, 1, 4, 11>(a, f, out, M, N, K, grid); }, 7, flush));
            report("pw<1,1,4> no st/dma/barrier", time_launch([&] { launch_pw<1, 1, 4, 11>(a, f, out, M, N, K, grid); }, 7, flush));
            report("pw<2,1,4> no st/dma", time_launch([&] { launch_pw<2, 1, 4, 3>(a, f, out, M, N, K, grid); }, 7, flush));
        }
        CHECK(hipFree(a)); CHECK(hipFree(f)); CHECK(hipFree(out)); CHECK(hipFree(ref)); CHECK(hipFree(d));
    }
    return 0;
}
