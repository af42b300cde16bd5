// Dense / grouped convolution and Linear on gfx950 as implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD — gfx950 has no TF32/xf32).
//
//   igemm_nt : out[m][j] = sum_k G[m][k] * F[j][k]      forward (G = gathered input pixels, F = filter rows)
//                                                      and dgrad (G = gathered dy, F = transposed filter)
//   igemm_tn : dW[i][k'] = sum_m dY[m][i] * G[m][k']    wgrad, split over m into slabs + fixed-order reduce
//
// Data layout: activations pixel-major NHWC, so the reduction index k = (r,s,ci) is contiguous per pixel for
// the gathered operand and per output channel for the filter [cout][r][s][ci]: both tiles are staged through
// registers into LDS as [row][32+4 floats] (16-byte vectors, padded rows -> conflict-free ds_read_b128), and a
// lane feeds 4 consecutive MFMAs from one 16-byte LDS read (lane half h takes k = 8g+4h+{0..3}).
// Workgroup = 256 threads = 4 wavefronts; each wave owns a (TM x TN) grid of 32x32 accumulator tiles.
//
// Replaces the cuDNN / cuBLAS calls behind nn.Conv2d / nn.Linear in src/networks.py:229-332,407-441,
// src/repvgg.py:28-32,67-76 and src/tiny_vit.py:28-50,84-151 (forward, input gradient, filter gradient).
#include "common.h"
#include "conv_params.h"
#include <cstdlib>
#ifndef SUHPE_STAGES
#define SUHPE_STAGES 2     // register stages of the persistent kernel (3 = one more slab of prefetch distance; experiment)
#endif
#ifndef SUHPE_PROBE
#define SUHPE_PROBE 0      // tools/micro/probe_gemm.sh builds instrumented copies (1-4) to split the persistent kernel's time
#endif
#include <cstring>
#include <mutex>
#include <type_traits>
#include <unordered_map>

namespace suhpe {

__device__ __forceinline__ bool gather_coord(const GatherGeom& g, int o, int tap, int extent, int* out) {
    if (g.mode == 0) {
        const int v = o * g.stride - g.pad + tap;
        *out = v;
        return v >= 0 && v < extent;
    }
    const int t = o + g.pad - tap;
    if (t < 0) return false;
    if (g.stride == 1) { *out = t; return t < extent; }
    const int q = t / g.stride;
    *out = q;
    return (q * g.stride == t) && q < extent;
}

// Workgroups that start together on a CU finish together: their epilogues (no MFMA work) and the prologues of the workgroups
// that replace them coincide, and the matrix pipes idle for that whole interval every generation.  The first generation is
// therefore spread out once: wave 0 of a workgroup sleeps (its wave slot on the SIMD) * quanta * 4096 cycles before the first
// barrier (its first loads are already in flight); later generations inherit the offsets, because a workgroup starts when one ends.
__device__ __forceinline__ void stagger_wait(int quanta, int first_gen) {
    if (quanta > 0 && (int)blockIdx.x < first_gen && threadIdx.x < 64) {
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 7u;      // HW_REG_HW_ID.WAVE_ID
        for (unsigned i = 0; i < slot * (unsigned)quanta; ++i) __builtin_amdgcn_s_sleep(64);
    }
}

template <int VEC> __device__ __forceinline__ void vload_lds(const float* p, float* o) {
    if constexpr (VEC == 4) { const float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    else o[0] = p[0];
}

// ---------------------------------------------------------------------------------------------- igemm_nt
// MODE 0: generic gather (any stride, dgrad with stride > 1: per-tap divisibility test)
// MODE 1: pointwise (1x1, stride 1, no padding): the gathered row IS row m
// MODE 2: affine taps, coordinate = o*gm + sgn*tap + off (forward with any stride, dgrad with stride 1)
// In every mode the tap of a thread's k advances incrementally (no division in the k loop), addresses are 32-bit
// element offsets from a uniform base, and out-of-range taps are predicated (offset 0, value zeroed): no branches.
template <int BM, int BN, int WM, int WN, int VEC, int MODE, bool AFF = false, int XF = 0, bool ACT2 = false>
// The second launch bound (minimum waves per SIMD) caps the register budget: left alone the compiler spends 88-148 VGPRs on
// prefetch depth, which leaves 2 workgroups (or 1, for the wgrad kernel) per CU; at <= 80 VGPRs three 8-wave workgroups are
// resident and the MFMA pipe sees 6 waves per SIMD.
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64,
                             (VEC == 1 || BM == 256) ? 1 : (XF ? (XF == 2 ? 3 : 4) : ((WM == 64 && WN == 64) ? 2 : ((WM == 64 && (BM / WM) * (BN / WN) == 4) ? 3 : (((BM / WM) * (BN / WN) > 8 || BN == 32 || BN == 128 || MODE == 0) ? 5 : 6)))))
void igemm_nt_kernel(const NtParams p) {
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;          // threads: 4 to 10 wavefronts (tile widths are multiples of 32)
    static_assert(NT >= 256 && NT <= 1024 && NT % 32 == 0, "4 to 16 waves per workgroup");
    __shared__ __attribute__((aligned(16))) float As[BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDK];

    const GatherGeom& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane((int)(t >> 6));      // (uniform: wave-level index arithmetic on the scalar ALU)
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int grp = blockIdx.z;
    // XCD-aware tile order: workgroups b and b+8 share an XCD (round-robin dispatch, speed only).  Column tiles of one
    // row tile reuse the same gathered rows, so they are made consecutive on ONE XCD: xcd = b%8, j = b/8,
    // col tile = j % col_tiles, row tile = (j / col_tiles)*8 + xcd.  Row tiles are padded to a multiple of 8 by the host.
    // Convolutions with a spatial footprint (every mode but the pointwise one) also share input rows BETWEEN neighbouring
    // row tiles (the halo of the taps), so there each XCD takes one contiguous run of row tiles instead of every eighth.
    const int b = blockIdx.x;
    const int jx = b >> 3;
    const int col_tile = jx % p.col_tiles;
    int row_tile = (jx / p.col_tiles) * 8 + (b & 7);
    if (MODE != 1) {
        const int row_tiles = (int)((g.M + BM - 1) / BM), q = row_tiles >> 3, rem = row_tiles & 7, xcd = b & 7, jl = jx / p.col_tiles;
        if (jl >= q + (xcd < rem ? 1 : 0)) return;      // padding workgroup
        row_tile = xcd * q + (xcd < rem ? xcd : rem) + jl;
    }
    const int64_t m0 = (int64_t)row_tile * BM;
    if (m0 >= g.M) return;                               // padding workgroup (whole block exits before any barrier)
    const int n0 = col_tile * BN;
    const BufRsrc xr = make_rsrc(p.x + (int64_t)grp * g.Cg, p.x_bytes - (int64_t)grp * g.Cg * 4);
    const BufRsrc fr = make_rsrc(p.f + (int64_t)grp * p.Ng * p.ldf, p.f_bytes - (int64_t)grp * p.Ng * p.ldf * 4);

    // ---- loader geometry.  VEC=4: thread -> (vector column j = t&7, rows (t>>3)+32i).  VEC=1: (k = t&31, rows (t>>5)+8i)
    constexpr int ROW_STEP = VEC == 4 ? NT / 8 : NT / 32;
    constexpr int A_ROWS = (BM + ROW_STEP - 1) / ROW_STEP;     // the last pass may run past the tile: those rows are predicated
    constexpr int B_ROWS = (BN + ROW_STEP - 1) / ROW_STEP;
    const int lcol = VEC == 4 ? (t & 7) * 4 : (t & 31);
    const int lrow = VEC == 4 ? (t >> 3) : (t >> 5);

    // per-row constants: a_pix = pixel index of tap (0,0) (MODE 2) / of the row itself (MODE 1) / n*GH*GW (MODE 0);
    // a_h/a_w = o*gm (MODE 2) or o (MODE 0).  a_pix < 0 marks rows beyond M.
    int a_pix[A_ROWS], a_h[A_ROWS], a_w[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int64_t m = m0 + lrow + i * ROW_STEP;
        a_h[i] = 0; a_w[i] = 0;
        if (m < g.M && lrow + i * ROW_STEP < BM) {
            if (MODE == 1) {
                a_pix[i] = (int)m;
            } else {
                const unsigned n = fast_div((unsigned)m, g.div_hw_mul, g.div_hw_shr);
                const unsigned rem = (unsigned)m - n * (unsigned)(g.OH * g.OW);
                const unsigned oh = fast_div(rem, g.div_w_mul, g.div_w_shr);
                const unsigned ow = rem - oh * (unsigned)g.OW;
                if (MODE == 2 || MODE == 3) {
                    a_h[i] = (int)oh * g.gm; a_w[i] = (int)ow * g.gm;
                    a_pix[i] = (int)n * g.GH * g.GW + a_h[i] * g.GW + a_w[i];
                } else {
                    a_h[i] = (int)oh; a_w[i] = (int)ow;
                    a_pix[i] = (int)n * g.GH * g.GW;
                }
            }
        } else {
            a_pix[i] = -1;
        }
    }
    // ---- input transform (XF): the scale / shift vectors of a slab's k columns ride with the slab's loads through the register
    // stages (tiny, cache-resident arrays); a tile that straddles the segment boundary loads both statistics sets; the per-sample
    // scale (XF == 2) is one more vector per row
    int xf_tile_seg = 0;                                     // 0 / 1: every row of the tile in that segment, 2: the tile straddles the boundary
    bool xf_row_seg[XF ? A_ROWS : 1];
    int xf_row_n[XF == 2 ? A_ROWS : 1];
    BufRsrc xf_sr = xr, xf_hr = xr, xf_rr = xr;
    if (XF) {
        if (p.a_seg_row > 0) xf_tile_seg = m0 + BM <= p.a_seg_row ? 0 : (m0 >= p.a_seg_row ? 1 : 2);
        const int64_t seg_bytes = (int64_t)(p.a_seg_row > 0 ? 2 : 1) * g.K * 4;
        xf_sr = make_rsrc(p.a_scale, seg_bytes);
        xf_hr = make_rsrc(p.a_shift, seg_bytes);
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const int64_t m = m0 + lrow + i * ROW_STEP;
            xf_row_seg[i] = p.a_seg_row > 0 && m >= p.a_seg_row;
            if (XF == 2) xf_row_n[i] = (int)fast_div((unsigned)(m < g.M ? m : g.M - 1), p.a_div_mul, p.a_div_shr);
        }
        if (XF == 2) xf_rr = make_rsrc(p.a_rowscale, (int64_t)((g.M + p.a_hw - 1) / p.a_hw) * g.K * 4);
    }
    float xf_sc0[2][VEC], xf_sh0[2][VEC], xf_sc1[2][VEC], xf_sh1[2][VEC];
    float xf_rs[2][XF == 2 ? A_ROWS : 1][VEC];

    // this thread's position in the reduction: k = tap*Cg + ci, tap = (r, s); advanced by BK per tile
    int kk = lcol, ci = lcol, tr = 0, ts = 0;
    if (MODE != 1 && g.R * g.S > 1) { const int tap = kk / g.Cg; ci = kk - tap * g.Cg; tr = tap / g.S; ts = tap - tr * g.S; }
    int b_off[B_ROWS];
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
        const int col = n0 + lrow + i * ROW_STEP;
        b_off[i] = (col < p.Ng && lrow + i * ROW_STEP < BN) ? col * p.ldf : -1;
    }

    // two register stages: the loads of tile kt+2 are issued while tile kt is multiplied, so a workgroup covers the global
    // latency with two MFMA phases of its own instead of relying on co-resident workgroups
    float a_stage[2][A_ROWS][VEC], b_stage[2][B_ROWS][VEC];

    auto load_tiles = [&](auto stage) {
        auto& a_reg = a_stage[decltype(stage)::value];
        auto& b_reg = b_stage[decltype(stage)::value];
        const bool kvalid = kk < g.K;
        int dh = 0, dw = 0, dpix = 0;
        if (MODE == 2) { dh = g.sgn * tr + g.off; dw = g.sgn * ts + g.off; dpix = dh * g.GW + dw; }
        if (MODE == 3) { dh = g.off_h - tr; dw = g.off_w - ts; dpix = dh * g.GW + dw; }
        // position of this k inside a filter row (differs from kk only when the taps are a sub-lattice, MODE 3)
        const int fk = MODE == 3 ? ((g.r0 + g.om * tr) * g.Sfull + g.s0 + g.om * ts) * g.Cg + ci : kk;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            bool ok = kvalid & (a_pix[i] >= 0);                  // bitwise: no short-circuit branches around the address math
            int pix = a_pix[i];
            if (MODE == 2 || MODE == 3) {
                ok = ok & ((unsigned)(a_h[i] + dh) < (unsigned)g.GH) & ((unsigned)(a_w[i] + dw) < (unsigned)g.GW);
                pix += dpix;
            } else if (MODE == 0) {
                int gh = 0, gw = 0;
                ok = ok && gather_coord(g, a_h[i], tr, g.GH, &gh);
                ok = ok && gather_coord(g, a_w[i], ts, g.GW, &gw);
                pix += gh * g.GW + gw;
            }
            buf_load<VEC>(xr, ok ? ((unsigned)pix * (unsigned)g.ldg + (unsigned)ci) * 4u : kOob, a_reg[i]);
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) {
            const bool ok = kvalid & (b_off[i] >= 0);
            buf_load<VEC>(fr, ok ? ((unsigned)b_off[i] + (unsigned)fk) * 4u : kOob, b_reg[i]);
        }
        if (XF) {
            constexpr int st = decltype(stage)::value;
            const unsigned k4 = (unsigned)kk * 4u, seg0 = xf_tile_seg == 1 ? (unsigned)g.K * 4u : 0u;
            buf_load<VEC>(xf_sr, kvalid ? seg0 + k4 : kOob, xf_sc0[st]);
            buf_load<VEC>(xf_hr, kvalid ? seg0 + k4 : kOob, xf_sh0[st]);
            if (xf_tile_seg == 2) {                          // (workgroup-uniform: one tile per launch at most)
                buf_load<VEC>(xf_sr, kvalid ? (unsigned)g.K * 4u + k4 : kOob, xf_sc1[st]);
                buf_load<VEC>(xf_hr, kvalid ? (unsigned)g.K * 4u + k4 : kOob, xf_sh1[st]);
            }
            if (XF == 2) {                                   // per-sample scale of this slab's k columns, one vector per row
#pragma unroll
                for (int i = 0; i < A_ROWS; ++i) buf_load<VEC>(xf_rr, kvalid ? ((unsigned)xf_row_n[i] * (unsigned)g.K + (unsigned)kk) * 4u : kOob, xf_rs[st][i]);
            }
        }
        // advance to the next tile
        kk += BK;
        ci += BK;
        if (MODE != 1) {
            while (ci >= g.Cg && g.R * g.S > 1) { ci -= g.Cg; if (++ts == g.S) { ts = 0; ++tr; } }
        }
    };
    // the input transform of a staged tile, in registers.  It is issued BEFORE the MFMAs of the previous tile (the tile's loads
    // are one slab old by then), so its VALU work fills the issue slots between that wave's dependent MFMAs instead of
    // lengthening the store phase between the two barriers
    auto xform_tiles = [&](auto stage) __attribute__((always_inline)) {
        if (!XF) return;
        auto& a_reg = a_stage[decltype(stage)::value];
        constexpr int st = decltype(stage)::value;
        // rows beyond M / columns beyond K carry act(shift) instead of 0: harmless, those rows are never stored and the
        // filter's columns beyond K are zero
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const bool q = xf_tile_seg == 2 && xf_row_seg[i];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float u = xf_act(a_reg[i][v] * (q ? xf_sc1[st][v] : xf_sc0[st][v]) + (q ? xf_sh1[st][v] : xf_sh0[st][v]), p.a_act);
                if (XF == 2) u *= xf_rs[st][i][v];
                a_reg[i][v] = u;
            }
        }
    };
    auto store_tiles = [&](auto stage, int kt_store = 0) {
        auto& a_reg = a_stage[decltype(stage)::value];
        auto& b_reg = b_stage[decltype(stage)::value];
        (void)kt_store;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            if (BM % ROW_STEP != 0 && lrow + i * ROW_STEP >= BM) continue;
            float* dst = &As[(lrow + i * ROW_STEP) * LDK + lcol];
            if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst) = make_float4(a_reg[i][0], a_reg[i][1], a_reg[i][2], a_reg[i][3]);
            else dst[0] = a_reg[i][0];
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) {
            if (lrow + i * ROW_STEP >= BN) continue;
            float* dst = &Bs[(lrow + i * ROW_STEP) * LDK + lcol];
            if constexpr (VEC == 4) *reinterpret_cast<float4*>(dst) = make_float4(b_reg[i][0], b_reg[i][1], b_reg[i][2], b_reg[i][3]);
            else dst[0] = b_reg[i][0];
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int r32 = lane & 31, h = lane >> 5;
    const int nk = (g.K + BK - 1) / BK;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    auto mma_tile = [&]() {
        // fragments of k-group kg+1 are read from LDS before the MFMAs of kg issue (two register sets)
        float4 af[2][TM], bf[2][TN];
        auto read_frags = [&](int kg, int set) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const float4*>(&As[(wm * WM + i * 32 + r32) * LDK + kg * 8 + h * 4]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[set][j] = *reinterpret_cast<const float4*>(&Bs[(wn * WN + j * 32 + r32) * LDK + kg * 8 + h * 4]);
        };
        read_frags(0, 0);
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            const int cur = kg & 1;
            if (kg + 1 < BK / 8) read_frags(kg + 1, cur ^ 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].x, bf[cur][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].y, bf[cur][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].z, bf[cur][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].w, bf[cur][j].w, acc[i][j], 0, 0, 0);
                }
        }
    };
    // tile j lives in register stage j & 1
    auto step = [&](int kt, auto cur, auto nxt) {
        if (kt + 2 < nk) load_tiles(cur);             // stage `cur` held tile kt, already in LDS: refill it with tile kt+2
        if (kt + 1 < nk) xform_tiles(nxt);            // (XF) tile kt+1 in registers, under this tile's MFMAs
        mma_tile();
        __syncthreads();
        if (kt + 1 < nk) {
            store_tiles(nxt, kt + 1);                 // tile kt+1, loaded one iteration ago
            __syncthreads();
        }
    };
    load_tiles(S0{});
    stagger_wait(p.stagger, p.stagger_first);
    xform_tiles(S0{});
    store_tiles(S0{}, 0);
    __syncthreads();
    if (nk > 1) load_tiles(S1{});
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, S0{}, S1{});
        if (kt + 1 < nk) step(kt + 1, S1{}, S0{});
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* __restrict__ y = p.y + (int64_t)grp * p.Ng;
    // two compile-time copies of the store loop (plain / folded affine+activation+skip) behind one uniform branch, so the
    // plain path carries no per-element tests
    // res_mode 5 (ACT2 instantiations): per column block the two segments' (rstd, -mean rstd, scale, shift) and this lane's running sums
    float bn5[ACT2 ? TN : 1][2][4], sum5[ACT2 ? TN : 1][4];
    if constexpr (ACT2) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + r32;
            const bool on = p.res_mode == 5 && col < p.Ng;
#pragma unroll
            for (int sset = 0; sset < 2; ++sset) {
                const int ci = (sset && p.seg_row > 0 ? p.Ng : 0) + (on ? col : 0);
                const float rs = on ? p.bn_rstd[ci] : 0.f, mu = on ? p.bn_mean[ci] : 0.f;
                bn5[j][sset][0] = rs; bn5[j][sset][1] = -mu * rs;
                bn5[j][sset][2] = on ? p.scale[ci] : 0.f; bn5[j][sset][3] = on ? p.shift[ci] : 0.f;
            }
            sum5[j][0] = sum5[j][1] = sum5[j][2] = sum5[j][3] = 0.f;
        }
    }
    auto store_out = [&](auto aff) __attribute__((always_inline)) {
        constexpr bool AFF_ = decltype(aff)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + r32;
            if (col >= p.Ng) continue;
            const float bv = (!AFF_ && p.bias) ? p.bias[grp * p.Ng + col] : 0.f;
            const float sc = AFF_ ? p.scale[grp * p.Ng + col] : 1.f;
            const float sh = AFF_ ? p.shift[grp * p.Ng + col] : 0.f;
            // fp32 MFMAs run on the vector ALU (tools/micro/coissue_probe.hip), so this loop is paid in matrix throughput: the element's offset is
            // ONE 64-bit product per 32-row block plus a scalar multiple of the row pitch (the first version multiplied 64-bit row x pitch per
            // element: ~25 vector instructions each), and a tile whose rows all exist -- all but the last -- runs the copy without row tests
            auto rows_of = [&](auto full_t) __attribute__((always_inline)) {
                constexpr bool FULL = decltype(full_t)::value;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int64_t row0 = m0 + wm * WM + i * 32 + 4 * h;
                    const int64_t off0 = row0 * p.ldy;          // (MODE != 3: output row = GEMM row)
                    auto out_off = [&](int e, int64_t* row) __attribute__((always_inline)) -> int64_t {
                        const int lr = (e & 3) + 8 * (e >> 2);
                        *row = row0 + lr;
                        if (MODE != 3) return off0 + (int64_t)lr * p.ldy;
                        // sub-grid row (n,a,b) -> pixel (n, a*om+ph, b*om+pw) of the full output
                        const unsigned n = fast_div((unsigned)*row, g.div_hw_mul, g.div_hw_shr);
                        const unsigned rem = (unsigned)*row - n * (unsigned)(g.OH * g.OW);
                        const unsigned a = fast_div(rem, g.div_w_mul, g.div_w_shr);
                        const unsigned bb = rem - a * (unsigned)g.OW;
                        return (((int64_t)n * g.YH + a * g.om + g.ph) * g.YW + bb * g.om + g.pw) * p.ldy;
                    };
                    // the residual operand of the block (skip gradient, pre-activation, BatchNorm input): every load is issued BEFORE the first
                    // store -- residual and y may alias as far as the compiler knows, so a load inside the store loop waits for the store ahead of it
                    float rv[16];
                    if (p.res_mode) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            int64_t row;
                            const int64_t off = out_off(e, &row);
                            rv[e] = (FULL || row < g.M) ? p.residual[off + grp * p.Ng + col] : 0.f;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        int64_t row;
                        const int64_t off = out_off(e, &row);
                        if (!FULL && row >= g.M) continue;
                        if constexpr (ACT2) {
                            float v = acc[i][j][e] + bv;
                            if (p.res_mode == 5) {
                                const bool second = p.seg_row > 0 && row >= p.seg_row;
                                const float xv = rv[e];
                                const float xh = xv * (second ? bn5[j][1][0] : bn5[j][0][0]) + (second ? bn5[j][1][1] : bn5[j][0][1]);
                                v *= act_grad(xv * (second ? bn5[j][1][2] : bn5[j][0][2]) + (second ? bn5[j][1][3] : bn5[j][0][3]), p.act);
                                sum5[j][second ? 2 : 0] += v;
                                sum5[j][second ? 3 : 1] += v * xh;
                            }
                            if (p.res_mode == 4) v *= act_grad(rv[e], p.act);
                            y[off + col] = v;
                            if (p.y2) p.y2[off + grp * p.Ng + col] = act_apply(v, p.act);
                        } else if constexpr (!AFF_) {
                            // res_mode 3: plain skip-connection add (input gradient + the gradient that bypassed the block; forward: x + keep * f(x))
                            float v = acc[i][j][e] + bv;
                            if (p.o_rowscale) v *= p.o_rowscale[fast_div((unsigned)row, p.o_div_mul, p.o_div_shr)];
                            y[off + col] = v + (p.res_mode == 3 ? rv[e] : 0.f);
                        } else {
                            float v = acc[i][j][e] * sc + sh;
                            if (p.res_mode == 1) v += rv[e];
                            v = act_apply(v, p.act);
                            if (p.res_mode == 2) v += rv[e];
                            y[off + col] = v;
                        }
                    }
                }
            };
            if (m0 + BM <= g.M) rows_of(std::true_type{}); else rows_of(std::false_type{});
        }
    };
    store_out(std::integral_constant<bool, AFF>{});       // AFF: a separate instantiation, so the plain kernel carries no affine code

    if constexpr (ACT2) {
        if (p.res_mode == 5) {                              // BatchNorm backward sums of the tile: lanes -> waves (LDS) -> one float2 record per column and plane
            __syncthreads();
            float* red = As;                                // [BM/WM][BN][4]
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) sum5[j][q] += __shfl_xor(sum5[j][q], 32, 64);
                if (h == 0) {
                    const int idx = wm * BN + wn * WN + j * 32 + r32;
#pragma unroll
                    for (int q = 0; q < 4; ++q) red[idx * 4 + q] = sum5[j][q];
                }
            }
            __syncthreads();
            const int row_tiles = (int)((g.M + BM - 1) / BM);
            float2* rec = reinterpret_cast<float2*>(p.stats);
            for (int idx = t; idx < BN; idx += NT) {
                const int col = n0 + idx;
                if (col >= p.Ng) continue;
                float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < BM / WM; ++w)
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[q] += red[(w * BN + idx) * 4 + q];
                rec[(int64_t)row_tile * p.Ng + col] = make_float2(a[0], a[1]);
                if (p.seg_row > 0) rec[((int64_t)row_tiles + row_tile) * p.Ng + col] = make_float2(a[2], a[3]);
            }
            return;
        }
    }

    // ---- fused BatchNorm statistics.  Each tile emits (S, Q, k, n) per column: sums of (y-k) and (y-k)^2 over its n valid
    // rows, shifted by k = the tile's first-row value so the variance never forms from two large cancelling sums.
    if (p.stats) {
        __syncthreads();                                    // every wave is past its last LDS tile read
        float* red = As;                                    // [BM/WM][BN][2]
        float* kk = As + (BM / WM) * BN * 2;                // [BN]
        if (wm == 0 && h == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) kk[wn * WN + j * 32 + r32] = acc[0][j][0];     // row m0 (always valid)
        }
        __syncthreads();
        // rows [lo, hi) of the tile (tile-local) -> one record per column at dst0[col]
        auto emit = [&](int64_t lo, int64_t hi, float4* dst0) __attribute__((always_inline)) {
            const int64_t wlo = lo - wm * WM, whi = hi - wm * WM;          // in this wave's rows
            // (fp32 MFMAs run on the vector ALU -- tools/micro/coissue_probe.hip -- so the epilogue's vector instructions are paid in matrix
            //  throughput: a wavefront whose rows all lie inside [lo, hi), i.e. nearly every one, skips the per-element range tests)
            const bool full = wlo <= 0 && whi >= WM;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float k = kk[wn * WN + j * 32 + r32];
                float a = 0.f, q = 0.f;
                if (full) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) { const float v = acc[i][j][e] - k; a += v; q += v * v; }
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int lr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                            const float v = (lr >= wlo && lr < whi) ? acc[i][j][e] - k : 0.f;
                            a += v;
                            q += v * v;
                        }
                }
                a += __shfl_xor(a, 32, 64);
                q += __shfl_xor(q, 32, 64);
                if (h == 0) {
                    const int idx = wm * BN + wn * WN + j * 32 + r32;
                    red[idx * 2] = a;
                    red[idx * 2 + 1] = q;
                }
            }
            __syncthreads();
            for (int idx = t; idx < BN; idx += NT) {
                const int col = n0 + idx;
                if (col >= p.Ng) continue;
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int w = 0; w < BM / WM; ++w) { a += red[(w * BN + idx) * 2]; q += red[(w * BN + idx) * 2 + 1]; }
                dst0[grp * p.Ng + col] = make_float4(a, q, kk[idx], (float)(hi - lo));
            }
            __syncthreads();
        };
        const int64_t nrows = g.M - m0 < BM ? g.M - m0 : BM;
        float4* rec = reinterpret_cast<float4*>(p.stats) + (int64_t)row_tile * p.stat_cout;
        const int64_t cut = p.seg_row - m0;                                 // segment boundary inside this tile?
        if (p.stats_extra && cut > 0 && cut < nrows) {
            emit(0, cut, rec);
            emit(cut, nrows, reinterpret_cast<float4*>(p.stats_extra));
        } else {
            emit(0, nrows, rec);
        }
    }
}

// ---------------------------------------------------------------------------------------------- gemm_nt_persistent
// Pointwise (1x1, stride 1) convolutions are plain GEMMs with a short reduction (K = 24..1536, often 5-8 slabs of 32):
// with one tile per workgroup most of a workgroup's life is its prologue (first loads) and epilogue (stores), and since
// all resident workgroups start together their MFMA phases coincide too: the matrix pipe measured 50 % busy.  Here a
// workgroup is persistent: it walks tiles vb = blockIdx.x, +gridDim.x, ... and streams k-slabs through the two register
// stages ACROSS tile boundaries, so the loads of the next tile's first slabs are in flight during the current tile's
// last MFMAs and its stores, and the pipe only idles for the stores themselves.
template <int BM, int BN, int WM, int WN, bool AFF>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, (BM * BN >= 128 * 128) ? 4 : 5) void gemm_nt_persistent(const NtParams p, int total_vb) {
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    __shared__ __attribute__((aligned(16))) float As[BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDK];

    const GatherGeom& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane((int)(t >> 6));      // (uniform: wave-level index arithmetic on the scalar ALU)
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int grp = blockIdx.z;
    const BufRsrc xr = make_rsrc(p.x + (int64_t)grp * g.Cg, p.x_bytes - (int64_t)grp * g.Cg * 4);
    const BufRsrc fr = make_rsrc(p.f + (int64_t)grp * p.Ng * p.ldf, p.f_bytes - (int64_t)grp * p.Ng * p.ldf * 4);
    constexpr int ROW_STEP = NT / 8;
    constexpr int A_ROWS = (BM + ROW_STEP - 1) / ROW_STEP;
    constexpr int B_ROWS = (BN + ROW_STEP - 1) / ROW_STEP;
    const int lcol = (t & 7) * 4, lrow = t >> 3;
    const int nk = (g.K + BK - 1) / BK;
    const int M = (int)g.M;

    // tile cursors (same XCD-aware order as igemm_nt_kernel; gridDim.x is a multiple of 8, so a workgroup stays on its XCD's
    // tile sequence).  Padding tiles (row tile beyond M) load nothing and store nothing.  Plain scalars, no aggregate: the
    // cursors must live in registers.
    const int gstride = gridDim.x;
    int l_vb = blockIdx.x, l_kt = 0, l_m0, l_n0;              // loader: runs two slabs ahead of the MFMAs
    int c_vb = blockIdx.x, c_kt = 0, c_m0, c_n0, c_rt;        // compute
    {
        const int jx = l_vb >> 3;
        const int ct = jx % p.col_tiles, rt = (jx / p.col_tiles) * 8 + (l_vb & 7);
        l_m0 = c_m0 = rt * BM; l_n0 = c_n0 = ct * BN; c_rt = rt;
    }
    if (c_vb >= total_vb) return;

    float a_stage[SUHPE_STAGES][A_ROWS][4], b_stage[SUHPE_STAGES][B_ROWS][4];
    auto load_slab = [&](auto stage) __attribute__((always_inline)) {
        if (l_vb >= total_vb) return;
#if SUHPE_PROBE >= 1 && SUHPE_PROBE <= 3
        if (++l_kt == nk) { l_kt = 0; l_vb += gstride; }
        return;
#endif
        auto& a_reg = a_stage[decltype(stage)::value];
        auto& b_reg = b_stage[decltype(stage)::value];
        const int kk = l_kt * BK + lcol;
        const bool kvalid = kk < g.K;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const int row = lrow + i * ROW_STEP, m = l_m0 + row;
            const bool ok = kvalid & (BM % ROW_STEP == 0 || row < BM) & (m < M);
            buf_load<4>(xr, ok ? ((unsigned)m * (unsigned)g.ldg + (unsigned)kk) * 4u : kOob, a_reg[i]);
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) {
            const int row = lrow + i * ROW_STEP, col = l_n0 + row;
            const bool ok = kvalid & (BN % ROW_STEP == 0 || row < BN) & (col < p.Ng);
            buf_load<4>(fr, ok ? ((unsigned)col * (unsigned)p.ldf + (unsigned)kk) * 4u : kOob, b_reg[i]);
        }
        if (++l_kt == nk) {
            l_kt = 0;
            l_vb += gstride;
            const int jx = l_vb >> 3;
            const int ct = jx % p.col_tiles;
            l_m0 = ((jx / p.col_tiles) * 8 + (l_vb & 7)) * BM;
            l_n0 = ct * BN;
        }
    };
    auto store_slab = [&](auto stage) __attribute__((always_inline)) {
#if SUHPE_PROBE >= 2 && SUHPE_PROBE <= 3
        return;
#endif
        auto& a_reg = a_stage[decltype(stage)::value];
        auto& b_reg = b_stage[decltype(stage)::value];
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            if (BM % ROW_STEP != 0 && lrow + i * ROW_STEP >= BM) continue;
            *reinterpret_cast<float4*>(&As[(lrow + i * ROW_STEP) * LDK + lcol]) = make_float4(a_reg[i][0], a_reg[i][1], a_reg[i][2], a_reg[i][3]);
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) {
            if (BN % ROW_STEP != 0 && lrow + i * ROW_STEP >= BN) continue;
            *reinterpret_cast<float4*>(&Bs[(lrow + i * ROW_STEP) * LDK + lcol]) = make_float4(b_reg[i][0], b_reg[i][1], b_reg[i][2], b_reg[i][3]);
        }
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
    zero_acc();
    const int r32 = lane & 31, h = lane >> 5;
    auto mma_tile = [&]() __attribute__((always_inline)) {
        // fragments of k-group kg+1 are read from LDS before the MFMAs of kg issue (two register sets)
        float4 af[2][TM], bf[2][TN];
        auto read_frags = [&](int kg, int set) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const float4*>(&As[(wm * WM + i * 32 + r32) * LDK + kg * 8 + h * 4]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[set][j] = *reinterpret_cast<const float4*>(&Bs[(wn * WN + j * 32 + r32) * LDK + kg * 8 + h * 4]);
        };
        read_frags(0, 0);
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            const int cur = kg & 1;
            if (kg + 1 < BK / 8) read_frags(kg + 1, cur ^ 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].x, bf[cur][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].y, bf[cur][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].z, bf[cur][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].w, bf[cur][j].w, acc[i][j], 0, 0, 0);
                }
        }
    };
    // epilogue of the tile under the compute cursor (same arithmetic as igemm_nt_kernel's)
    auto epilogue = [&]() __attribute__((always_inline)) {
        const int m0 = c_m0, n0 = c_n0;
        float* __restrict__ y = p.y + (int64_t)grp * p.Ng;
        auto store_out = [&](auto aff) __attribute__((always_inline)) {
            constexpr bool AFF_ = decltype(aff)::value;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * WN + j * 32 + r32;
                if (col >= p.Ng) continue;
                const float bv = (!AFF_ && p.bias) ? p.bias[grp * p.Ng + col] : 0.f;
                const float sc = AFF_ ? p.scale[grp * p.Ng + col] : 1.f;
                const float sh = AFF_ ? p.shift[grp * p.Ng + col] : 0.f;
                // (as in igemm_nt_kernel: one 64-bit row product per 32-row block, scalar multiples of the row pitch per element, no row tests in a full tile)
                auto rows_of = [&](auto full_t) __attribute__((always_inline)) {
                    constexpr bool FULL = decltype(full_t)::value;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int row0 = m0 + wm * WM + i * 32 + 4 * h;
                        const int64_t off0 = (int64_t)row0 * p.ldy;
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int lr = (e & 3) + 8 * (e >> 2);
                            if (!FULL && row0 + lr >= M) continue;
                            const int64_t off = off0 + (int64_t)lr * p.ldy;
                            if constexpr (!AFF_) {
#if SUHPE_PROBE == 4
                                if (acc[i][j][e] == 12345.678f)
#endif
                                y[off + col] = acc[i][j][e] + bv + (p.res_mode == 3 ? p.residual[off + grp * p.Ng + col] : 0.f);
                            } else {
                                float v = acc[i][j][e] * sc + sh;
                                const float rv = p.res_mode ? p.residual[off + grp * p.Ng + col] : 0.f;
                                if (p.res_mode == 1) v += rv;
                                v = act_apply(v, p.act);
                                if (p.res_mode == 2) v += rv;
                                y[off + col] = v;
                            }
                        }
                    }
                };
                if (m0 + BM <= M) rows_of(std::true_type{}); else rows_of(std::false_type{});
            }
        };
        store_out(std::integral_constant<bool, AFF>{});
        if (p.stats) {                                          // BatchNorm statistics records, as in igemm_nt_kernel
            float* red = As;                                    // [BM/WM][BN][2]   (every wave is past its last LDS read: barrier above)
            float* kk = As + (BM / WM) * BN * 2;                // [BN]
            if (wm == 0 && h == 0) {
#pragma unroll
                for (int j = 0; j < TN; ++j) kk[wn * WN + j * 32 + r32] = acc[0][j][0];
            }
            __syncthreads();
            auto emit = [&](int lo, int hi, float4* dst0) __attribute__((always_inline)) {
                const int wlo = lo - wm * WM, whi = hi - wm * WM;
                const bool full = wlo <= 0 && whi >= WM;        // (as in igemm_nt_kernel: no range tests where every row counts)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float k = kk[wn * WN + j * 32 + r32];
                    float a = 0.f, q = 0.f;
                    if (full) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int e = 0; e < 16; ++e) { const float v = acc[i][j][e] - k; a += v; q += v * v; }
                    } else {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int lr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                                const float v = (lr >= wlo && lr < whi) ? acc[i][j][e] - k : 0.f;
                                a += v;
                                q += v * v;
                            }
                    }
                    a += __shfl_xor(a, 32, 64);
                    q += __shfl_xor(q, 32, 64);
                    if (h == 0) {
                        const int idx = wm * BN + wn * WN + j * 32 + r32;
                        red[idx * 2] = a;
                        red[idx * 2 + 1] = q;
                    }
                }
                __syncthreads();
                for (int idx = t; idx < BN; idx += NT) {
                    const int col = n0 + idx;
                    if (col >= p.Ng) continue;
                    float a = 0.f, q = 0.f;
#pragma unroll
                    for (int w = 0; w < BM / WM; ++w) { a += red[(w * BN + idx) * 2]; q += red[(w * BN + idx) * 2 + 1]; }
                    if (hi > lo) dst0[grp * p.Ng + col] = make_float4(a, q, kk[idx], (float)(hi - lo));
                }
                __syncthreads();                                // the scratch is tile storage again from here on
            };
            const int nrows = M - m0 < BM ? M - m0 : BM;
            float4* rec = reinterpret_cast<float4*>(p.stats) + (int64_t)c_rt * p.stat_cout;
            const int64_t cut = p.seg_row - m0;
            if (p.stats_extra && cut > 0 && cut < nrows) {
                emit(0, (int)cut, rec);
                emit((int)cut, nrows, reinterpret_cast<float4*>(p.stats_extra));
            } else {
                emit(0, nrows > 0 ? nrows : 0, rec);
            }
        }
    };

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // slab s lives in register stage s & 1; returns false after the workgroup's last slab
    auto step = [&](auto cur, auto nxt) __attribute__((always_inline)) -> bool {
        load_slab(cur);                               // slab s+2 (possibly the next tile's) refills the stage slab s came from
        mma_tile();
#if SUHPE_PROBE != 3
        __syncthreads();
#endif
        if (++c_kt == nk) {
            epilogue();
            zero_acc();
            c_kt = 0;
            c_vb += gstride;
            const int jx = c_vb >> 3;
            const int ct = jx % p.col_tiles;
            c_rt = (jx / p.col_tiles) * 8 + (c_vb & 7);
            c_m0 = c_rt * BM;
            c_n0 = ct * BN;
        }
        if (c_vb >= total_vb) return false;
        store_slab(nxt);                              // slab s+1, loaded one step ago
#if SUHPE_PROBE != 3
        __syncthreads();
#endif
        return true;
    };
    load_slab(S0{});
    store_slab(S0{});
    __syncthreads();
    load_slab(S1{});
#if SUHPE_STAGES == 3
    using S2 = std::integral_constant<int, 2>;
    load_slab(S2{});
    while (step(S0{}, S1{}) && step(S1{}, S2{}) && step(S2{}, S0{})) {}
#else
    while (step(S0{}, S1{}) && step(S1{}, S0{})) {}
#endif
}

// ---------------------------------------------------------------------------------------------- igemm_tn (wgrad)
struct TnParams {
    GatherGeom g;        // gather of x (mode 0); rows m = (n, oh, ow) of dy
    const float* dy;     // [M][lddy]
    const float* x;
    float* slab;         // [splits][groups*Ng][K]
    int Ng;              // output channels per group (rows of dW per group)
    int lddy;
    int64_t dy_bytes, x_bytes;  // tensor extents (buffer descriptors)
    int64_t m_per_split; // multiple of BK
    int tiles_m, tiles_n; // output tiles along cout and along k'
    // optional transform of x while it is staged (pointwise convs, XF instantiations), see NtParams: the filter gradient of a
    // conv whose input act(BN(x)) * rowscale was never materialised
    const float* x_scale = nullptr;
    const float* x_shift = nullptr;
    const float* x_rowscale = nullptr;
    int x_act = 0, x_hw = 1;
    int64_t x_seg_row = 0;
    unsigned x_div_mul = 0, x_div_shr = 0;
};

template <int BM, int BN, int WM, int WN, int VECA, int VECB, bool XF = false>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, (VECA == 1 || VECB == 1) ? 1 : (XF ? 4 : ((BM == 128 && BN == 128) ? 6 : 5)))
void igemm_tn_kernel(const TnParams p) {
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    static_assert(NT >= 256 && NT <= 1024, "4 to 16 waves per workgroup");
    __shared__ __attribute__((aligned(16))) float As[BK * BM];   // [m][cout]
    __shared__ __attribute__((aligned(16))) float Bs[BK * BN];   // [m][k']

    const GatherGeom& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane((int)(t >> 6));      // (uniform: wave-level index arithmetic on the scalar ALU)
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int grp = blockIdx.z;
    // XCD-aware order: all output tiles of one m-split run consecutively on one XCD (they stream the same dy / x rows)
    const int tiles = p.tiles_m * p.tiles_n;
    const int jx = blockIdx.x >> 3;
    const int tile = jx % tiles;
    const int split = (jx / tiles) * 8 + (blockIdx.x & 7);
    const int tile_m = tile / p.tiles_n, tile_n = tile % p.tiles_n;
    const int i0 = tile_m * BM;          // cout offset inside the group
    const int k0 = tile_n * BN;          // k' offset
    const int64_t mbeg = (int64_t)split * p.m_per_split;
    int64_t mend = mbeg + p.m_per_split;
    if (mend > g.M) mend = g.M;
    const BufRsrc dyr = make_rsrc(p.dy + (int64_t)grp * p.Ng, p.dy_bytes - (int64_t)grp * p.Ng * 4);
    const BufRsrc xr = make_rsrc(p.x + (int64_t)grp * g.Cg, p.x_bytes - (int64_t)grp * g.Cg * 4);

    // A loader: dy rows.  VECA=4: col group = t % (BM/4), row = t / (BM/4) + RA*i ; VECA=1: col = t % BM
    constexpr int A_CG = BM / VECA;                 // column groups per row
    constexpr int RA = NT / A_CG;                   // rows covered per pass
    constexpr int A_IT = (BK + RA - 1) / RA;        // the last pass may run past BK rows: predicated
    static_assert(NT % A_CG == 0 && RA >= 1, "thread count must be a multiple of the A tile's column groups");
    const int a_col = (t % A_CG) * VECA, a_row = t / A_CG;
    constexpr int B_CG = BN / VECB;
    constexpr int RB = NT / B_CG;
    constexpr int B_IT = (BK + RB - 1) / RB;
    static_assert(NT % B_CG == 0 && RB >= 1, "thread count must be a multiple of the B tile's column groups");
    const int b_col = (t % B_CG) * VECB, b_row = t / B_CG;

    // this thread's k' is fixed for the whole reduction: decode its filter tap once
    const int kq = k0 + b_col;
    const bool kq_valid = kq < g.K;
    int r = 0, s = 0, ci = kq;
    if (g.R * g.S > 1) { const int tap = kq / g.Cg; ci = kq - tap * g.Cg; r = tap / g.S; s = tap - r * g.S; }
    const bool a_valid_col = i0 + a_col < p.Ng;
    const int dr = r - g.pad, ds = s - g.pad;          // wgrad always gathers x in forward form: g = o*stride - pad + tap

    // input transform (XF, pointwise convs: k' = channel): this thread's channels are fixed, so scale / shift of both segments are
    // loaded once; the per-sample scale is fetched per staged row
    float xs0[VECB], xh0[VECB], xs1[VECB], xh1[VECB], xrs[XF ? B_IT : 1][VECB];
    bool xseg[XF ? B_IT : 1];
    BufRsrc xrr = xr;
    if (XF) {
        const bool two = p.x_seg_row > 0;
#pragma unroll
        for (int v = 0; v < VECB; ++v) {
            const bool ok = kq + v < g.K;
            xs0[v] = ok ? p.x_scale[kq + v] : 0.f;
            xh0[v] = ok ? p.x_shift[kq + v] : 0.f;
            xs1[v] = (ok && two) ? p.x_scale[g.K + kq + v] : 0.f;
            xh1[v] = (ok && two) ? p.x_shift[g.K + kq + v] : 0.f;
        }
        if (p.x_rowscale) xrr = make_rsrc(p.x_rowscale, (int64_t)((g.M + p.x_hw - 1) / p.x_hw) * g.K * 4);
    }

    float a_reg[A_IT][VECA], b_reg[B_IT][VECB];
    auto load_tiles = [&](int64_t mb) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int64_t m = mb + a_row + i * RA;
            const bool ok = a_valid_col & (m < mend) & (BK % RA == 0 || a_row + i * RA < BK);
            buf_load<VECA>(dyr, ok ? ((unsigned)m * (unsigned)p.lddy + (unsigned)(i0 + a_col)) * 4u : kOob, a_reg[i]);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int64_t m = mb + b_row + i * RB;
            bool ok = kq_valid & (m < mend) & (BK % RB == 0 || b_row + i * RB < BK);
            const unsigned mu = ok ? (unsigned)m : 0u;
            const unsigned n = fast_div(mu, g.div_hw_mul, g.div_hw_shr);
            const unsigned rem = mu - n * (unsigned)(g.OH * g.OW);
            const unsigned oh = fast_div(rem, g.div_w_mul, g.div_w_shr);
            const unsigned ow = rem - oh * (unsigned)g.OW;
            const int gh = (int)oh * g.stride + dr, gw = (int)ow * g.stride + ds;
            ok = ok & ((unsigned)gh < (unsigned)g.GH) & ((unsigned)gw < (unsigned)g.GW);
            buf_load<VECB>(xr, ok ? (((unsigned)n * (unsigned)(g.GH * g.GW) + (unsigned)(gh * g.GW + gw)) * (unsigned)g.ldg + (unsigned)ci) * 4u : kOob,
                           b_reg[i]);
            if (XF) {
                xseg[i] = p.x_seg_row > 0 && m >= p.x_seg_row;
                if (p.x_rowscale) {
                    const unsigned smp = fast_div(mu, p.x_div_mul, p.x_div_shr);
                    buf_load<VECB>(xrr, ok ? (smp * (unsigned)g.K + (unsigned)kq) * 4u : kOob, xrs[i]);
                }
            }
        }
    };
    // Pointwise form (1x1, stride 1, no padding, no input transform -- every filter gradient of an MBConv expand / project conv and every Linear):
    // both operands are plain row-major matrices, so a thread's byte offsets just advance by BK rows per slab.  fp32 MFMAs run on the vector ALU
    // (tools/micro/coissue_probe.hip): the general gather above costs ~70 vector instructions per slab, 14 of them quarter-rate multiplies, beside
    // 16 MFMAs -- a third of the slab.  Here: one v_add per load; rows beyond the split are cut off by the descriptors' extents, not by compares.
    const bool pw = !XF && g.R * g.S == 1 && g.stride == 1 && g.pad == 0 && g.GH == g.OH && g.GW == g.OW;
    unsigned a_off[A_IT], b_off[B_IT];
    const BufRsrc dyr_pw = make_rsrc(p.dy + (int64_t)grp * p.Ng, mend * (int64_t)p.lddy * 4 - (int64_t)grp * p.Ng * 4 < p.dy_bytes - (int64_t)grp * p.Ng * 4
                                                                      ? mend * (int64_t)p.lddy * 4 - (int64_t)grp * p.Ng * 4 : p.dy_bytes - (int64_t)grp * p.Ng * 4);
    const BufRsrc xr_pw = make_rsrc(p.x + (int64_t)grp * g.Cg, mend * (int64_t)g.ldg * 4 - (int64_t)grp * g.Cg * 4 < p.x_bytes - (int64_t)grp * g.Cg * 4
                                                                    ? mend * (int64_t)g.ldg * 4 - (int64_t)grp * g.Cg * 4 : p.x_bytes - (int64_t)grp * g.Cg * 4);
    if (pw) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            a_off[i] = (a_valid_col && (BK % RA == 0 || a_row + i * RA < BK)) ? ((unsigned)(mbeg + a_row + i * RA) * (unsigned)p.lddy + (unsigned)(i0 + a_col)) * 4u : kOob;
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            b_off[i] = (kq_valid && (BK % RB == 0 || b_row + i * RB < BK)) ? ((unsigned)(mbeg + b_row + i * RB) * (unsigned)g.ldg + (unsigned)kq) * 4u : kOob;
    }
    const unsigned a_step = (unsigned)BK * (unsigned)p.lddy * 4u, b_step = (unsigned)BK * (unsigned)g.ldg * 4u;
    auto load_tiles_pw = [&]() {                                // (kOob + the steps of a whole split stays out of range: a split's rows span < 2^31 bytes)
#pragma unroll
        for (int i = 0; i < A_IT; ++i) { buf_load<VECA>(dyr_pw, a_off[i], a_reg[i]); a_off[i] += a_step; }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) { buf_load<VECB>(xr_pw, b_off[i], b_reg[i]); b_off[i] += b_step; }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            if (BK % RA != 0 && a_row + i * RA >= BK) continue;
            float* dst = &As[(a_row + i * RA) * BM + a_col];
            if constexpr (VECA == 4) *reinterpret_cast<float4*>(dst) = make_float4(a_reg[i][0], a_reg[i][1], a_reg[i][2], a_reg[i][3]);
            else dst[0] = a_reg[i][0];
        }
        if (XF) {                       // rows beyond the split read x = 0 -> act(shift): harmless, their dy rows are zero
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
#pragma unroll
                for (int v = 0; v < VECB; ++v) {
                    float u = xf_act(b_reg[i][v] * (xseg[i] ? xs1[v] : xs0[v]) + (xseg[i] ? xh1[v] : xh0[v]), p.x_act);
                    if (p.x_rowscale) u *= xrs[i][v];
                    b_reg[i][v] = u;
                }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            if (BK % RB != 0 && b_row + i * RB >= BK) continue;
            float* dst = &Bs[(b_row + i * RB) * BN + b_col];
            if constexpr (VECB == 4) *reinterpret_cast<float4*>(dst) = make_float4(b_reg[i][0], b_reg[i][1], b_reg[i][2], b_reg[i][3]);
            else dst[0] = b_reg[i][0];
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int r32 = lane & 31, h = lane >> 5;
    if (mbeg < mend) {
        if (pw) load_tiles_pw(); else load_tiles(mbeg);
        store_tiles();
        __syncthreads();
        for (int64_t mb = mbeg; mb < mend; mb += BK) {
            const bool more = mb + BK < mend;
            if (more) { if (pw) load_tiles_pw(); else load_tiles(mb + BK); }
            // fragments in groups of four k-pairs, two register sets: the LDS reads of group q+1 are in flight under the
            // MFMAs of group q (with one set the compiler waits for every pair of reads right before its two MFMAs)
            constexpr int GQ = 4;
            float af[2][GQ][TM], bf[2][GQ][TN];
            auto read_group = [&](int q, int set) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < GQ; ++u) {
                    const int kk = q * GQ + u;
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[set][u][i] = As[(kk * 2 + h) * BM + wm * WM + i * 32 + r32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[set][u][j] = Bs[(kk * 2 + h) * BN + wn * WN + j * 32 + r32];
                }
            };
            read_group(0, 0);
#pragma unroll
            for (int q = 0; q < BK / 2 / GQ; ++q) {
                const int cur = q & 1;
                if (q + 1 < BK / 2 / GQ) read_group(q + 1, cur ^ 1);
#pragma unroll
                for (int u = 0; u < GQ; ++u)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][u][i], bf[cur][u][j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (more) {
                store_tiles();
                __syncthreads();
            }
        }
    }
    // slab[split][grp*Ng + row][k']
    float* __restrict__ out = p.slab + ((int64_t)split * gridDim.z * p.Ng + (int64_t)grp * p.Ng) * g.K;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = k0 + wn * WN + j * 32 + r32;
        if (col >= g.K) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = i0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < p.Ng) out[(int64_t)row * g.K + col] = acc[i][j][e];
            }
    }
}

// sum_z slab[z][i] in the fixed order z = 0, 1, 2, ...; eight loads are issued before the first add (the adds stay sequential)
__device__ __forceinline__ float slab_sum(const float* __restrict__ slab, int64_t splits, int64_t count, int64_t i) {
    float s = 0.f;
    int64_t z = 0;
    for (; z + 8 <= splits; z += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = slab[(z + u) * count + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < splits; ++z) s += slab[z * count + i];
    return s;
}

// dw[i] (+)= sum_z slab[z][i], fixed order
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int splits, int64_t count, float* __restrict__ dw,
                                                          int accumulate) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = slab_sum(slab, splits, count, i);
        dw[i] = accumulate ? dw[i] + s : s;
    }
}

// The same reduction for a whole backward pass in one launch: jobs[j] = (slab, dw, count, splits), chunk_start[j] = first
// 1024-element chunk of job j (chunk_start[njobs] = total).  A workgroup takes one chunk: binary search for its job, then
// dw[i] += sum_z slab[z][i] in the same fixed order as slab_reduce_kernel (bit-identical to reducing layer by layer).
struct SlabJob { const float* slab; float* dw; int64_t count; int64_t splits; };
constexpr int kSlabChunk = 1024;
__global__ __launch_bounds__(256) void slab_reduce_batched_kernel(const SlabJob* __restrict__ jobs, const int64_t* __restrict__ chunk_start,
                                                                  int njobs) {
    const int64_t chunk = blockIdx.x;
    int lo = 0, hi = njobs;                         // largest j with chunk_start[j] <= chunk
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chunk_start[mid] <= chunk) lo = mid; else hi = mid;
    }
    const SlabJob jb = jobs[lo];
    const int64_t base = (chunk - chunk_start[lo]) * kSlabChunk;
#pragma unroll
    for (int u = 0; u < kSlabChunk / 256; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        if (i >= jb.count) break;
        jb.dw[i] += slab_sum(jb.slab, jb.splits, jb.count, i);
    }
}

// wt[g][ci][r][s][co] = w[g][co][r][s][ci]  (per group)
__global__ __launch_bounds__(256) void transpose_filter_kernel(const float* __restrict__ w, float* __restrict__ wt, int groups, int Ng,
                                                               int Cg, int RS) {
    const int64_t total = (int64_t)groups * Ng * RS * Cg;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i enumerates the destination: (g, ci, tap, co)
        const int co = (int)(i % Ng);
        const int tap = (int)((i / Ng) % RS);
        const int ci = (int)((i / ((int64_t)Ng * RS)) % Cg);
        const int gq = (int)(i / ((int64_t)Ng * RS * Cg));
        wt[i] = w[(((int64_t)gq * Ng + co) * RS + tap) * Cg + ci];
    }
}

// All transposed filters of a network in one launch (the weights move every optimiser step, so the input-gradient GEMMs of a
// backward pass need fresh transposes: ~80 launches of a few microseconds of work each otherwise).  jobs[j] = (w, wt, groups,
// Ng, Cg, RS); chunk_start as in slab_reduce_batched_kernel.
struct TransposeJob { const float* w; float* wt; int64_t groups, Ng, Cg, RS; };
__global__ __launch_bounds__(256) void transpose_filters_batched_kernel(const TransposeJob* __restrict__ jobs, const int64_t* __restrict__ chunk_start,
                                                                        int njobs) {
    const int64_t chunk = blockIdx.x;
    int lo = 0, hi = njobs;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chunk_start[mid] <= chunk) lo = mid; else hi = mid;
    }
    const TransposeJob jb = jobs[lo];
    const int Ng = (int)jb.Ng, Cg = (int)jb.Cg, RS = (int)jb.RS;
    const int64_t total = jb.groups * Ng * RS * Cg, base = (chunk - chunk_start[lo]) * kSlabChunk;
#pragma unroll
    for (int u = 0; u < kSlabChunk / 256; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        if (i >= total) break;
        const int co = (int)(i % Ng);
        const int tap = (int)((i / Ng) % RS);
        const int ci = (int)((i / ((int64_t)Ng * RS)) % Cg);
        const int gq = (int)(i / ((int64_t)Ng * RS * Cg));
        jb.wt[i] = jb.w[(((int64_t)gq * Ng + co) * RS + tap) * Cg + ci];
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int check_desc(const suhpe_conv_t* d, const char* who) {
    if (!d) { set_error("%s: null descriptor", who); return SUHPE_EINVAL; }
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cin <= 0 || d->cout <= 0 || d->r <= 0 || d->s <= 0 || d->stride <= 0 || d->pad < 0 ||
        d->groups <= 0 || d->cin % d->groups || d->cout % d->groups) {
        set_error("%s: bad conv descriptor", who);
        return SUHPE_EINVAL;
    }
    const int ho = (d->h + 2 * d->pad - d->r) / d->stride + 1, wo = (d->w + 2 * d->pad - d->s) / d->stride + 1;
    if (ho != d->ho || wo != d->wo) { set_error("%s: ho/wo (%d,%d) do not match the geometry (%d,%d)", who, d->ho, d->wo, ho, wo); return SUHPE_EINVAL; }
    const int64_t lim = (int64_t)1 << 29;      // operands are addressed through buffer descriptors with 32-bit byte offsets: < 2 GiB each
    if ((int64_t)d->n * d->h * d->w * d->cin >= lim || (int64_t)d->n * d->ho * d->wo * d->cout >= lim ||
        (int64_t)d->cout * d->r * d->s * (d->cin / d->groups) >= lim) {
        set_error("%s: tensor of 2 GiB or more (2^29 fp32 elements)", who);
        return SUHPE_EINVAL;
    }
    return SUHPE_OK;
}

static GatherGeom make_geom(int N, int GH, int GW, int OH, int OW, int ldg, int Cg, int R, int S, int stride, int pad, int mode) {
    GatherGeom g;
    g.N = N; g.GH = GH; g.GW = GW; g.OH = OH; g.OW = OW; g.ldg = ldg; g.Cg = Cg; g.R = R; g.S = S; g.stride = stride; g.pad = pad;
    g.mode = mode; g.K = R * S * Cg; g.M = (int64_t)N * OH * OW;
    if (mode == 0) { g.gm = stride; g.sgn = 1; g.off = -pad; }
    else { g.gm = 1; g.sgn = -1; g.off = pad; }
    magic_div((unsigned)(OH * OW), &g.div_hw_mul, &g.div_hw_shr);
    magic_div((unsigned)OW, &g.div_w_mul, &g.div_w_shr);
    g.om = 1; g.ph = g.pw = 0; g.YH = OH; g.YW = OW; g.r0 = g.s0 = 0; g.Sfull = S; g.off_h = g.off_w = 0;
    return g;
}

#include "conv_halo.inl"
#include "conv_stem.inl"

// fast-path selector: 1 = pointwise, 2 = affine taps, 3 = parity class of a strided dgrad, 0 = generic
static int nt_mode(const GatherGeom& g) {
    if (g.om > 1) return 3;
    if (g.R == 1 && g.S == 1 && g.stride == 1 && g.pad == 0) return 1;
    if (g.mode == 0 || g.stride == 1) return 2;
    return 0;
}

#include "conv_resident.inl"
#include "conv_dma.inl"

static int g_stagger = 0, g_stagger_first = 0;     // experiment knob (suhpe_conv2d_set_stagger)
static thread_local bool g_nt_unsupported = false;  // set when a dispatch reached a (tile, form) that has no instantiation

template <int BM, int BN, int WM, int WN, int VEC>
static void launch_nt(const NtParams& p_in, int groups, hipStream_t st) {
    NtParams p = p_in;
    p.stagger = g_stagger; p.stagger_first = g_stagger_first;
    p.col_tiles = (int)ceil_div(p.Ng, BN);
    const int64_t row_tiles8 = ceil_div(ceil_div(p.g.M, BM), 8) * 8;
    dim3 grid((unsigned)(row_tiles8 * p.col_tiles), 1, groups);
    const dim3 block((BM / WM) * (BN / WN) * 64);
    switch (nt_mode(p.g)) {
        case 1:
            if (p.a_scale) {
                // (the wide-accumulator W tiles have no input-transform form: dispatch_nt maps them to the 8-wave tiles first)
                if constexpr (VEC == 4 && !(WM == 64 && BN >= 64 && (BM / WM) * (BN / WN) == 4)) {
                    const size_t lds = 0;
                    if (p.a_rowscale) hipLaunchKernelGGL((igemm_nt_kernel<BM, BN, WM, WN, VEC, 1, false, 2>), grid, block, lds, st, p);
                    else hipLaunchKernelGGL((igemm_nt_kernel<BM, BN, WM, WN, VEC, 1, false, 1>), grid, block, lds, st, p);
                } else {
                    g_nt_unsupported = true;                 // never a silent no-op: the caller turns this into an error code
                }
            } else if (p.y2 || p.res_mode >= 4) {           // (res_mode 5 carries scale / shift of its own: tested before the folded-affine form)
                if constexpr (VEC == 4) hipLaunchKernelGGL((igemm_nt_kernel<BM, BN, WM, WN, VEC, 1, false, 0, true>), grid, block, 0, st, p);
                else g_nt_unsupported = true;
            } else if (p.scale) hipLaunchKernelGGL((igemm_nt_kernel<BM, BN, WM, WN, VEC, 1, true>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((igemm_nt_kernel<BM, BN, WM, WN, VEC, 1>), grid, block, 0, st, p);
            break;
        case 2:
            if (p.scale) hipLaunchKernelGGL((igemm_nt_kernel<BM, BN, WM, WN, VEC, 2, true>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((igemm_nt_kernel<BM, BN, WM, WN, VEC, 2>), grid, block, 0, st, p);
            break;
        case 3: hipLaunchKernelGGL((igemm_nt_kernel<BM, BN, WM, WN, VEC, 3>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((igemm_nt_kernel<BM, BN, WM, WN, VEC, 0>), grid, block, 0, st, p); break;
    }
}

// ---- tile menus.  Which tile is fastest depends on the shape in ways a padding/efficiency model does not capture
// (tools/tune_tiles.py: the small tiles win most 1x1 layers, 128x64 the large-K 3x3 ones), so the caller may pin the tile
// of a (descriptor, pass) after measuring the menu (suhpe_conv2d_set_tile; semiuhpe_amd.kernels.AUTOTUNE does that the
// first time a shape is seen).  Unpinned shapes use the heuristics below.
// NT_P*: the persistent pointwise kernel (gemm_nt_persistent); on a shape that is not a plain GEMM (3x3, strided) a pinned
// NT_P* tile falls back to the igemm tile of the same size
// NT_H*: the 3x3 / stride 1 halo-tile kernel (conv_halo.inl); on any other geometry a pinned NT_H* tile runs the igemm tile of the same size
enum NtTile { NT_64x64_S = 0, NT_64x64, NT_128x32, NT_128x128, NT_128x64, NT_256x32, NT_P64x64, NT_P128x32, NT_P128x64, NT_P128x128,
              NT_H64x64, NT_H128x32, NT_H128x64, NT_H128x128,
              NT_H128x48,                    // halo tile of 48 columns: 32 on the 32x32x2 MFMA + 16 on the 16x16x4 MFMA (outputs with 48 channels)
              NT_W128x128, NT_W128x64,       // four wavefronts with 64x64 / 64x32 accumulators each: half the LDS fragment reads per MFMA
              NT_R128x16, NT_R64x8, NT_R64x16,   // operand-resident pointwise kernel (conv_resident.inl): 128 / 64 resident rows, 16 / 8 / 16 wavefronts
              NT_G64x64,                     // Winograd F(2x2, 3x3) kernel (conv_wino.inl): 64 tiles (256 output pixels) x 64 output channels
              NT_D128x128,                   // both operands by LDS-DMA through a four-slab ring, persistent 128 x 128 tiles (conv_dma.inl): plain pointwise GEMMs, K % 32 == 0
              NT_D256x64,                    // the same pipeline as 256 x 64 tiles for outputs of at most 64 channels, with the XF = 1 loader transform on the fragments
              NT_TILES };
static const char* kNtNames[] = {"igemm_nt_kernel<64,64,32,32,1>", "igemm_nt_kernel<64,64,32,32,4>", "igemm_nt_kernel<128,32,32,32,4>",
                                 "igemm_nt_kernel<128,128,64,32,4>", "igemm_nt_kernel<128,64,32,32,4>", "igemm_nt_kernel<256,32,64,32,4>",
                                 "gemm_nt_persistent<64,64,32,32>", "gemm_nt_persistent<128,32,32,32>", "gemm_nt_persistent<128,64,32,32>",
                                 "gemm_nt_persistent<128,128,64,32>",
                                 "igemm_nt3x3_kernel<64,64,32,32>", "igemm_nt3x3_kernel<128,32,32,32>", "igemm_nt3x3_kernel<128,64,32,32>",
                                 "igemm_nt3x3_kernel<128,128,64,32>", "igemm_nt3x3_n48_kernel<128,48>",
                                 "igemm_nt_kernel<128,128,64,64,4>", "igemm_nt_kernel<128,64,64,32,4>",
                                 "gemm_nt_resident<128,2,16>", "gemm_nt_resident<64,2,8>", "gemm_nt_resident<64,2,16>",
                                 "wino_nt_kernel<64,64>", "gemm_nt_dma<128,128>", "gemm_nt_dma<256,64>"};

struct TnTile { int bm, bn; const char* name; };
static const TnTile kTnTiles[] = {
    {128, 128, "igemm_tn_kernel<128,128,64,32,4,4>"}, {128, 64, "igemm_tn_kernel<128,64,64,32,4,4>"},
    {64, 128, "igemm_tn_kernel<64,128,32,64,4,4>"},   {64, 64, "igemm_tn_kernel<64,64,32,32,4,4>"},
    {32, 128, "igemm_tn_kernel<32,128,32,32,4,4>"},
    {128, 32, "igemm_tn_kernel<128,32,32,32,4,4>"},   // narrow k' (Cin = 160: five exact column tiles instead of 2.5 of 64)
    {0, 0, "wino_wgrad_kernel<64,64>"},               // Winograd F(3x3, 2x2) filter gradient (conv_wino.hip): 3x3 / stride 1 / pad 1 convs on even sizes; pinned only
};
constexpr int TN_TILES = (int)(sizeof(kTnTiles) / sizeof(kTnTiles[0]));
constexpr int TN_WINO = TN_TILES - 1;

struct TileKey {
    int v[11];
    bool operator==(const TileKey& o) const { return memcmp(v, o.v, sizeof(v)) == 0; }
};
struct TileKeyHash {
    size_t operator()(const TileKey& k) const {
        uint64_t h = 1469598103934665603ull;
        for (int x : k.v) { h ^= (uint64_t)(uint32_t)x; h *= 1099511628211ull; }
        return (size_t)h;
    }
};
static std::unordered_map<TileKey, int, TileKeyHash>& tile_table() { static std::unordered_map<TileKey, int, TileKeyHash> t; return t; }
static std::mutex& tile_mutex() { static std::mutex m; return m; }
static TileKey tile_key(const suhpe_conv_t* d, int pass) {
    return TileKey{{pass, d->n, d->h, d->w, d->cin, d->cout, d->r, d->s, d->stride, d->pad, d->groups}};
}
static int pinned_tile(const suhpe_conv_t* d, int pass) {
    std::lock_guard<std::mutex> lock(tile_mutex());
    auto it = tile_table().find(tile_key(d, pass));
    return it == tile_table().end() ? -1 : it->second;
}

static const bool kHaloDefault = !(getenv("SUHPE_HALO") && atoi(getenv("SUHPE_HALO")) == 0);   // SUHPE_HALO=0: heuristics never pick the halo kernel
static NtTile choose_nt(int64_t M, int ng, bool vec4, int pinned, bool halo = false, bool wino = false) {
    if (!vec4) return NT_64x64_S;
    if (pinned >= 1 && pinned < (int)NT_TILES) return (NtTile)pinned;
    if (wino && wino_preferred()) return NT_G64x64;
    if (halo && kHaloDefault && M >= 8192) return ng <= 32 ? NT_H128x32 : ((ng % 128 == 0 || ng > 256) ? NT_H128x128 : NT_H128x64);
    if (M < 2048) return ng > 32 ? NT_64x64 : NT_128x32;     // small row counts (Linear, SE): more, smaller tiles
    if (ng <= 32) return NT_256x32;                           // narrow outputs are operand-bandwidth bound anyway
    // measured on the EfficientNetV2-S / RepVGG layer shapes: 128x64 (8 waves of 32x32) wins or ties nearly everywhere
    // once the grid fills the chip; below ~2 workgroups per CU the 64x64 tile does
    return ceil_div(M, 128) * ceil_div(ng, 64) >= 512 ? NT_128x64 : NT_64x64;
}

// (row-tile height = rows per BatchNorm statistics record; the resident tiles emit one record per 64-row unit)
static int nt_tile_bm(NtTile t) { if (t == NT_D256x64) return 32; return (t == NT_G64x64 || t == NT_64x64_S || t == NT_64x64 || t == NT_P64x64 || t == NT_H64x64 || t >= NT_R128x16) ? 64 : (t == NT_256x32 ? 256 : 128); }    // (the Winograd tile's record = the 16 tiles of 2x2 pixels of one wavefront)
// the halo kernel runs 3x3 / stride 1 / pad 1 forward and input-gradient GEMMs whose channel counts chunk evenly
static bool halo_ok(const GatherGeom& g, bool vec4) { return vec4 && halo_capable(g, 128) && halo_capable(g, 64); }

template <int BM, int BN, int WM, int WN>
static void launch_persistent(const NtParams& p_in, int groups, hipStream_t st) {
    static int slots = 0;                                      // resident workgroups of this instantiation on the whole device
    if (slots == 0) {
        int occ = 0, dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gemm_nt_persistent<BM, BN, WM, WN, false>, (BM / WM) * (BN / WN) * 64, 0);
        slots = (occ > 0 ? occ : 1) * (cus > 0 ? cus : 256);
    }
    NtParams p = p_in;
    p.col_tiles = (int)ceil_div(p.Ng, BN);
    const int64_t row_tiles8 = ceil_div(ceil_div(p.g.M, BM), 8) * 8;
    const int total_vb = (int)(row_tiles8 * p.col_tiles);
    int per_group = slots / groups;
    per_group = per_group / 8 * 8;
    if (per_group < 8) per_group = 8;
    const int gx = total_vb < per_group ? total_vb : per_group;     // total_vb is a multiple of 8 too
    const dim3 grid(gx, 1, groups), block((BM / WM) * (BN / WN) * 64);
    if (p.scale) hipLaunchKernelGGL((gemm_nt_persistent<BM, BN, WM, WN, true>), grid, block, 0, st, p, total_vb);
    else hipLaunchKernelGGL((gemm_nt_persistent<BM, BN, WM, WN, false>), grid, block, 0, st, p, total_vb);
}

// a pinned persistent tile on a shape that is not a plain GEMM runs the igemm tile of the same size
static NtTile effective_nt(NtTile tile, bool plain, bool halo = false, bool resident128 = false, bool resident64 = false, bool wino = false, bool dma = false) {
    if (tile == NT_D128x128) return dma ? tile : NT_64x64;     // (same statistics-record height: 64 rows)
    if (tile == NT_D256x64) return tile;                        // (checked when it was pinned: dma_ok_desc; no other tile writes 32-row records)
    if (tile == NT_G64x64) return wino ? tile : NT_128x64;
    if (tile >= NT_R128x16) return ((tile == NT_R128x16 && resident128) || (tile != NT_R128x16 && resident64)) ? tile : NT_64x64;   // (same record height)
    if (tile >= NT_W128x128) return tile;
    if (tile >= NT_H64x64) {
        if (halo) return tile;
        return tile == NT_H64x64 ? NT_64x64 : (tile == NT_H128x32 ? NT_128x32 : ((tile == NT_H128x64 || tile == NT_H128x48) ? NT_128x64 : NT_128x128));
    }
    if (tile < NT_P64x64 || plain) return tile;
    return tile == NT_P64x64 ? NT_64x64 : (tile == NT_P128x32 ? NT_128x32 : (tile == NT_P128x64 ? NT_128x64 : NT_128x128));
}

static void dispatch_nt(const NtParams& p, int groups, bool vec4, hipStream_t st, int pinned) {
    NtTile tile = choose_nt(p.g.M, p.Ng, vec4, pinned, halo_ok(p.g, vec4), wino_geom_ok(p.g, p.Ng));
    const bool act2 = p.y2 || p.res_mode >= 4 || p.o_rowscale; // activation / keep-scale epilogues live in igemm_nt_kernel's pointwise form only
    if (tile == NT_G64x64) {
        if (launch_wino(p, groups, st)) return;
        // a form the Winograd kernel lacks (eval-mode folded affine, no Winograd-domain filter handed over): the heuristic's direct tile, or with
        // statistics the direct tile of the same record height
        tile = p.stats ? NT_64x64 : choose_nt(p.g.M, p.Ng, vec4, -1, halo_ok(p.g, vec4));
    }
    if (tile == NT_D128x128) {
        if (!act2 && dma_ok(p, groups, vec4, false)) { launch_dma(p, st); note_kernel(kNtNames[NT_D128x128]); return; }
        tile = NT_64x64;                                    // an epilogue / operand form the DMA kernel lacks: the direct tile of the same record height
    }
    if (tile == NT_D256x64) {
        if (!act2 && dma_ok(p, groups, vec4, true)) { launch_dma_narrow(p, st); note_kernel(kNtNames[NT_D256x64]); return; }
        // no other tile writes 32-row statistics records: with statistics asked for, a form this kernel lacks is an error, never a silent mismatch
        if (p.stats) { g_nt_unsupported = true; return; }
        tile = NT_64x64;
    }
    if (act2 && tile >= NT_R128x16) tile = NT_64x64;
    if (act2 && tile >= NT_W128x128) tile = tile == NT_W128x128 ? NT_128x128 : NT_128x64;
    if (tile >= NT_R128x16) {
        if (tile == NT_R128x16 && resident_ok(p, vec4, 128)) { launch_resident<128, 2, 16>(p, groups, st); note_kernel(kNtNames[tile]); return; }
        if (tile == NT_R64x8 && resident_ok(p, vec4, 64)) { launch_resident<64, 2, 8>(p, groups, st); note_kernel(kNtNames[tile]); return; }
        if (tile == NT_R64x16 && resident_ok(p, vec4, 64)) { launch_resident<64, 2, 16>(p, groups, st); note_kernel(kNtNames[tile]); return; }
        tile = NT_64x64;                                    // (same statistics-record height as the resident tiles)
    }
    if (tile >= NT_W128x128 && p.a_scale) tile = tile == NT_W128x128 ? NT_128x128 : NT_128x64;     // (no input-transform form of the wide-accumulator tiles)
    if (tile == NT_W128x128) { launch_nt<128, 128, 64, 64, 4>(p, groups, st); note_kernel(kNtNames[tile]); return; }
    if (tile == NT_W128x64) { launch_nt<128, 64, 64, 32, 4>(p, groups, st); note_kernel(kNtNames[tile]); return; }
    if (tile >= NT_H64x64) {
        bool done = false;
        const NtTile halo_tile = tile;
        switch (tile) {
            case NT_H64x64: done = launch_halo<64, 64, 32, 32>(p, groups, st); tile = NT_64x64; break;
            case NT_H128x32: done = launch_halo<128, 32, 32, 32>(p, groups, st); tile = NT_128x32; break;
            case NT_H128x64: done = launch_halo<128, 64, 32, 32>(p, groups, st); tile = NT_128x64; break;
            case NT_H128x48: done = launch_halo_n48(p, groups, st); tile = NT_128x64; break;
            default: done = launch_halo<128, 128, 64, 32>(p, groups, st); tile = NT_128x128; break;
        }
        if (done) { note_kernel(kNtNames[halo_tile]); return; }   // otherwise: the igemm tile of the same size (same row-tile height)
    }
    if (tile >= NT_P64x64) {
        const bool plain = nt_mode(p.g) == 1 && p.g.M < (1LL << 31) && !p.a_scale && !act2;   // (input transform / activation epilogues live in igemm_nt_kernel)
        if (plain) {
            note_kernel(kNtNames[tile]);
            switch (tile) {
                case NT_P64x64: launch_persistent<64, 64, 32, 32>(p, groups, st); return;
                case NT_P128x32: launch_persistent<128, 32, 32, 32>(p, groups, st); return;
                case NT_P128x64: launch_persistent<128, 64, 32, 32>(p, groups, st); return;
                default: launch_persistent<128, 128, 64, 32>(p, groups, st); return;
            }
        }
        tile = effective_nt(tile, false);
    }
    note_kernel(kNtNames[tile]);
    switch (tile) {
        case NT_64x64_S: launch_nt<64, 64, 32, 32, 1>(p, groups, st); break;
        case NT_64x64: launch_nt<64, 64, 32, 32, 4>(p, groups, st); break;
        case NT_128x32: launch_nt<128, 32, 32, 32, 4>(p, groups, st); break;
        case NT_128x128: launch_nt<128, 128, 64, 32, 4>(p, groups, st); break;      // 8 wavefronts of 64x32
        case NT_128x64: launch_nt<128, 64, 32, 32, 4>(p, groups, st); break;        // 8 wavefronts of 32x32
        case NT_256x32: launch_nt<256, 32, 64, 32, 4>(p, groups, st); break;
        default: break;
    }
}

// Every entry point goes through this: a (tile, operand / epilogue form) without an instantiation is SUHPE_EINVAL, never a launch that silently
// did not happen (suhpe_conv2d_fwd_wino reached the 32-row-record rule of gemm_nt_dma<256,64> without looking at the flag).
static int dispatch_nt_checked(const NtParams& p, int groups, bool vec4, hipStream_t st, int pinned, const char* who) {
    g_nt_unsupported = false;
    dispatch_nt(p, groups, vec4, st, pinned);
    if (g_nt_unsupported) {
        g_nt_unsupported = false;
        set_error("%s: the selected tile has no instantiation for this operand / epilogue form (bias, alignment, size): unpin it (suhpe_conv2d_set_tile)", who);
        return SUHPE_EINVAL;
    }
    return SUHPE_OK;
}

// halo-kernel capability of a (descriptor, pass): pass 0 = forward, 1 = input gradient
static bool halo_ok_desc(const suhpe_conv_t* d, int pass) {
    if (!d || d->groups <= 0 || pass > 1 || d->stride != 1) return false;
    const int cg = d->cin / d->groups, ng = d->cout / d->groups;
    if (pass == 0) return halo_ok(make_geom(d->n, d->h, d->w, d->ho, d->wo, d->cin, cg, d->r, d->s, d->stride, d->pad, 0), cg % 4 == 0 && d->cin % 4 == 0);
    return halo_ok(make_geom(d->n, d->ho, d->wo, d->h, d->w, d->cout, ng, d->r, d->s, d->stride, d->pad, 1), ng % 4 == 0 && d->cout % 4 == 0);
}

// Winograd-kernel capability of a (descriptor, pass): pass 0 = forward, 1 = input gradient
static bool wino_ok_desc(const suhpe_conv_t* d, int pass) {
    if (!d || d->groups <= 0 || pass > 1 || d->stride != 1) return false;
    const int cg = d->cin / d->groups, ng = d->cout / d->groups;
    if (d->cin % 4 || d->cout % 4) return false;
    if (pass == 0) return wino_geom_ok(make_geom(d->n, d->h, d->w, d->ho, d->wo, d->cin, cg, d->r, d->s, d->stride, d->pad, 0), ng);
    return wino_geom_ok(make_geom(d->n, d->ho, d->wo, d->h, d->w, d->cout, ng, d->r, d->s, d->stride, d->pad, 1), cg);
}

// resident-kernel capability of a (descriptor, pass): pass 0 = forward, 1 = input gradient (plain pointwise GEMMs only)
static bool resident_ok_desc(const suhpe_conv_t* d, int pass, int BM) {
    if (!d || d->groups <= 0 || pass > 1 || d->r != 1 || d->s != 1 || d->stride != 1 || d->pad != 0) return false;
    const int cg = d->cin / d->groups, ng = d->cout / d->groups;
    if (cg % 4 || ng % 4 || d->cin % 4 || d->cout % 4) return false;
    const int K = pass == 0 ? cg : ng;
    return resident_lds_bytes(K, BM) <= kResidentLdsMax && (int64_t)d->n * d->h * d->w < (1LL << 31);
}

// wgrad (rows = cout, columns = k'): both extents pad; cost = padded area / measured relative efficiency
static int choose_tn(int ng, int K, int pinned) {
    if (pinned >= 0 && pinned < TN_TILES) return pinned;
    static const double eff[TN_WINO] = {0.85, 0.95, 0.95, 1.0, 0.9, 0.9};
    int best = 3;
    double best_cost = 1e30;
    for (int i = 0; i < TN_WINO; ++i) {
        const TnTile& c = kTnTiles[i];
        const double cost = (double)(ceil_div(ng, c.bm) * c.bm) * (double)(ceil_div(K, c.bn) * c.bn) / eff[i];
        if (cost < best_cost) { best_cost = cost; best = i; }
    }
    return best;
}

}  // namespace suhpe

using namespace suhpe;

extern "C" {

int suhpe_conv2d_fwd(const suhpe_conv_t* d, const float* x, const float* w, const float* bias, float* y, suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_fwd");
    if (rc) return rc;
    SUHPE_REQUIRE(x && w && y, "suhpe_conv2d_fwd: null pointer");
    if (stem_ok(d) && pinned_tile(d, 0) < 0) {               // RGB stem: direct kernel (conv_stem.inl); a pinned tile keeps the GEMM (tile menu tests)
        launch_stem(d, x, w, bias, y, nullptr, nullptr, 0, as_stream(stream));
        note_kernel(suhpe_conv2d_kernel_name(d, 0));
        return check_launch("suhpe_conv2d_fwd");
    }
    NtParams p;
    const int cg = d->cin / d->groups;
    p.g = make_geom(d->n, d->h, d->w, d->ho, d->wo, d->cin, cg, d->r, d->s, d->stride, d->pad, 0);
    p.x = x; p.f = w; p.bias = bias; p.y = y; p.Ng = d->cout / d->groups; p.ldy = d->cout; p.ldf = p.g.K;
    p.x_bytes = (int64_t)d->n * d->h * d->w * d->cin * 4; p.f_bytes = (int64_t)d->cout * p.g.K * 4;
    p.stats = nullptr; p.stats_extra = nullptr; p.seg_row = 0; p.stat_gx = 0; p.stat_cout = 0; p.scale = p.shift = p.residual = nullptr; p.act = 0; p.res_mode = 0;
    const bool vec4 = (cg % 4 == 0) && (d->cin % 4 == 0) && aligned16(x) && aligned16(w);
    if ((rc = dispatch_nt_checked(p, d->groups, vec4, as_stream(stream), pinned_tile(d, 0), "suhpe_conv2d_fwd"))) return rc;
    return check_launch("suhpe_conv2d_fwd");
}

static bool pointwise_vec4(const suhpe_conv_t* d);
int suhpe_conv2d_fwd_skip(const suhpe_conv_t* d, const float* x, const float* w, const float* bias, const float* skip, const float* rowscale,
                          int rows_per_sample, float* y, suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_fwd_skip");
    if (rc) return rc;
    SUHPE_REQUIRE(pointwise_vec4(d), "suhpe_conv2d_fwd_skip: pointwise convs / Linear layers only (1x1, stride 1, groups 1, channels %% 4 == 0)");
    SUHPE_REQUIRE(x && w && skip && y && aligned16(x) && aligned16(w), "suhpe_conv2d_fwd_skip: null or unaligned pointer");
    SUHPE_REQUIRE(!rowscale || (rows_per_sample > 0 && ((int64_t)d->n * d->h * d->w) % rows_per_sample == 0 && (int64_t)d->n * d->h * d->w < (1LL << 31)),
                  "suhpe_conv2d_fwd_skip: rows_per_sample must divide the row count");
    NtParams p;
    p.g = make_geom(d->n, d->h, d->w, d->ho, d->wo, d->cin, d->cin, 1, 1, 1, 0, 0);
    p.x = x; p.f = w; p.bias = bias; p.y = y; p.Ng = d->cout; p.ldy = d->cout; p.ldf = p.g.K;
    p.x_bytes = (int64_t)d->n * d->h * d->w * d->cin * 4; p.f_bytes = (int64_t)d->cout * p.g.K * 4;
    p.stats = nullptr; p.stats_extra = nullptr; p.seg_row = 0; p.stat_gx = 0; p.stat_cout = 0; p.scale = p.shift = nullptr; p.residual = skip; p.act = 0; p.res_mode = 3;
    if (rowscale) { p.o_rowscale = rowscale; magic_div((unsigned)rows_per_sample, &p.o_div_mul, &p.o_div_shr); }
    g_nt_unsupported = false;
    dispatch_nt(p, 1, true, as_stream(stream), pinned_tile(d, 0));
    if (g_nt_unsupported) { set_error("suhpe_conv2d_fwd_skip: the selected tile has no instantiation for this epilogue"); return SUHPE_EINVAL; }
    return check_launch("suhpe_conv2d_fwd_skip");
}

int suhpe_conv2d_fwd_affine(const suhpe_conv_t* d, const float* x, const float* w, const float* scale, const float* shift, int act,
                            const float* residual, int residual_after_act, float* y, suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_fwd_affine");
    if (rc) return rc;
    SUHPE_REQUIRE(x && w && y && scale && shift, "suhpe_conv2d_fwd_affine: null pointer");
    SUHPE_REQUIRE(act >= SUHPE_ACT_NONE && act <= SUHPE_ACT_SIGMOID, "suhpe_conv2d_fwd_affine: unknown activation %d", act);
    NtParams p;
    const int cg = d->cin / d->groups;
    p.g = make_geom(d->n, d->h, d->w, d->ho, d->wo, d->cin, cg, d->r, d->s, d->stride, d->pad, 0);
    p.x = x; p.f = w; p.bias = nullptr; p.y = y; p.Ng = d->cout / d->groups; p.ldy = d->cout; p.ldf = p.g.K;
    p.x_bytes = (int64_t)d->n * d->h * d->w * d->cin * 4; p.f_bytes = (int64_t)d->cout * p.g.K * 4;
    p.stats = nullptr; p.stats_extra = nullptr; p.seg_row = 0; p.stat_gx = 0; p.stat_cout = 0;
    p.scale = scale; p.shift = shift; p.residual = residual; p.act = act; p.res_mode = residual ? (residual_after_act ? 2 : 1) : 0;
    const bool vec4 = (cg % 4 == 0) && (d->cin % 4 == 0) && aligned16(x) && aligned16(w);
    if ((rc = dispatch_nt_checked(p, d->groups, vec4, as_stream(stream), pinned_tile(d, 0), "suhpe_conv2d_fwd_affine"))) return rc;
    return check_launch("suhpe_conv2d_fwd_affine");
}

static bool pointwise_vec4(const suhpe_conv_t* d) {
    return d && d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && d->groups == 1 && d->cin % 4 == 0 && d->cout % 4 == 0;
}

int suhpe_conv2d_fwd_act2(const suhpe_conv_t* d, const float* x, const float* w, const float* bias, int act, float* y_pre, float* y_act,
                          suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_fwd_act2");
    if (rc) return rc;
    SUHPE_REQUIRE(pointwise_vec4(d), "suhpe_conv2d_fwd_act2: pointwise convs / Linear layers only (1x1, stride 1, groups 1, channels %% 4 == 0)");
    SUHPE_REQUIRE(x && w && y_pre && y_act && aligned16(x) && aligned16(w), "suhpe_conv2d_fwd_act2: null or unaligned pointer");
    SUHPE_REQUIRE(act >= SUHPE_ACT_NONE && act <= SUHPE_ACT_SIGMOID, "suhpe_conv2d_fwd_act2: unknown activation %d", act);
    NtParams p;
    p.g = make_geom(d->n, d->h, d->w, d->ho, d->wo, d->cin, d->cin, 1, 1, 1, 0, 0);
    p.x = x; p.f = w; p.bias = bias; p.y = y_pre; p.y2 = y_act; p.Ng = d->cout; p.ldy = d->cout; p.ldf = p.g.K;
    p.x_bytes = (int64_t)d->n * d->h * d->w * d->cin * 4; p.f_bytes = (int64_t)d->cout * p.g.K * 4;
    p.stats = nullptr; p.stats_extra = nullptr; p.seg_row = 0; p.stat_gx = 0; p.stat_cout = 0; p.scale = p.shift = p.residual = nullptr; p.act = act; p.res_mode = 0;
    g_nt_unsupported = false;
    dispatch_nt(p, 1, true, as_stream(stream), pinned_tile(d, 0));
    if (g_nt_unsupported) { set_error("suhpe_conv2d_fwd_act2: the selected tile has no activation-epilogue instantiation"); return SUHPE_EINVAL; }
    return check_launch("suhpe_conv2d_fwd_act2");
}

int suhpe_conv2d_dgrad_actgrad(const suhpe_conv_t* d, const float* dy, const float* wt, const float* pre, int act, float* dx, suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_dgrad_actgrad");
    if (rc) return rc;
    SUHPE_REQUIRE(pointwise_vec4(d), "suhpe_conv2d_dgrad_actgrad: pointwise convs / Linear layers only (1x1, stride 1, groups 1, channels %% 4 == 0)");
    SUHPE_REQUIRE(dy && wt && pre && dx && aligned16(dy) && aligned16(wt), "suhpe_conv2d_dgrad_actgrad: null or unaligned pointer");
    SUHPE_REQUIRE(act >= SUHPE_ACT_NONE && act <= SUHPE_ACT_SIGMOID, "suhpe_conv2d_dgrad_actgrad: unknown activation %d", act);
    NtParams p;
    p.g = make_geom(d->n, d->ho, d->wo, d->h, d->w, d->cout, d->cout, 1, 1, 1, 0, 1);
    p.x = dy; p.f = wt; p.bias = nullptr; p.y = dx; p.Ng = d->cin; p.ldy = d->cin; p.ldf = p.g.K;
    p.x_bytes = (int64_t)d->n * d->ho * d->wo * d->cout * 4; p.f_bytes = (int64_t)d->cout * d->cin * 4;
    p.stats = nullptr; p.stats_extra = nullptr; p.seg_row = 0; p.stat_gx = 0; p.stat_cout = 0; p.scale = p.shift = nullptr; p.residual = pre; p.act = act; p.res_mode = 4;
    g_nt_unsupported = false;
    dispatch_nt(p, 1, true, as_stream(stream), pinned_tile(d, 1));
    if (g_nt_unsupported) { set_error("suhpe_conv2d_dgrad_actgrad: the selected tile has no activation-epilogue instantiation"); return SUHPE_EINVAL; }
    return check_launch("suhpe_conv2d_dgrad_actgrad");
}

// rows per statistics record of suhpe_conv2d_dgrad_bnbwd (its tile is fixed: the GEMM has K = cout of a project conv, it is HBM-bound)
constexpr int kBnbwdRows = 128;
int64_t suhpe_conv2d_dgrad_bnbwd_workspace_bytes(const suhpe_conv_t* d) {
    if (!d) return 0;
    return (int64_t)2 * ceil_div((int64_t)d->n * d->h * d->w, kBnbwdRows) * d->cin * 2 * sizeof(float);
}

int suhpe_conv2d_dgrad_bnbwd(const suhpe_conv_t* d, const float* dy, const float* wt, const float* bn_x, const float* mean, const float* rstd,
                             const float* scale, const float* shift, int act, int seg_n, float* g, float* sums, float* dgamma, float* dbeta,
                             int accumulate, void* workspace, int64_t workspace_bytes, suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_dgrad_bnbwd");
    if (rc) return rc;
    SUHPE_REQUIRE(pointwise_vec4(d), "suhpe_conv2d_dgrad_bnbwd: pointwise convs only (1x1, stride 1, groups 1, channels %% 4 == 0)");
    SUHPE_REQUIRE(dy && wt && bn_x && mean && rstd && scale && shift && g && sums && workspace && aligned16(dy) && aligned16(wt),
                  "suhpe_conv2d_dgrad_bnbwd: null or unaligned pointer");
    SUHPE_REQUIRE(act >= SUHPE_ACT_NONE && act <= SUHPE_ACT_SIGMOID && seg_n >= 0 && seg_n < d->n, "suhpe_conv2d_dgrad_bnbwd: bad activation / segment");
    if (workspace_bytes < suhpe_conv2d_dgrad_bnbwd_workspace_bytes(d)) {
        set_error("suhpe_conv2d_dgrad_bnbwd: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)suhpe_conv2d_dgrad_bnbwd_workspace_bytes(d));
        return SUHPE_EWORKSPACE;
    }
    NtParams p;
    p.g = make_geom(d->n, d->ho, d->wo, d->h, d->w, d->cout, d->cout, 1, 1, 1, 0, 1);
    p.x = dy; p.f = wt; p.bias = nullptr; p.y = g; p.Ng = d->cin; p.ldy = d->cin; p.ldf = p.g.K;
    p.x_bytes = (int64_t)d->n * d->ho * d->wo * d->cout * 4; p.f_bytes = (int64_t)d->cout * d->cin * 4;
    p.stats = static_cast<float*>(workspace); p.stats_extra = nullptr; p.seg_row = (int64_t)seg_n * d->h * d->w; p.stat_gx = 0; p.stat_cout = d->cin;
    p.scale = scale; p.shift = shift; p.residual = bn_x; p.act = act; p.res_mode = 5; p.bn_mean = mean; p.bn_rstd = rstd;
    g_nt_unsupported = false;
    dispatch_nt(p, 1, true, as_stream(stream), NT_128x64);
    if (g_nt_unsupported) { set_error("suhpe_conv2d_dgrad_bnbwd: no activation-epilogue instantiation of the tile"); return SUHPE_EINVAL; }
    const int row_tiles = (int)ceil_div(p.g.M, kBnbwdRows);
    launch_bn_bwd_finalize(p.stats, seg_n > 0 ? row_tiles : 0, seg_n > 0 ? 2 * row_tiles : row_tiles, d->cin, sums, dgamma, dbeta, accumulate, as_stream(stream));
    return check_launch("suhpe_conv2d_dgrad_bnbwd");
}

int suhpe_conv2d_fwd_stats_tiles(const suhpe_conv_t* d) {
    if (!d || d->groups <= 0) return 0;
    if (stem_ok(d) && pinned_tile(d, 0) < 0) return (int)ceil_div((int64_t)d->n * d->ho * d->wo, kStemRows);
    const int cg = d->cin / d->groups, ng = d->cout / d->groups;
    const int64_t M = (int64_t)d->n * d->ho * d->wo;
    return (int)ceil_div(M, nt_tile_bm(choose_nt(M, ng, cg % 4 == 0 && d->cin % 4 == 0, pinned_tile(d, 0), halo_ok_desc(d, 0), wino_ok_desc(d, 0))));
}

int suhpe_conv2d_fwd_stats_tile_rows(const suhpe_conv_t* d) {
    if (!d || d->groups <= 0) return 0;
    if (stem_ok(d) && pinned_tile(d, 0) < 0) return kStemRows;
    const int cg = d->cin / d->groups, ng = d->cout / d->groups;
    return nt_tile_bm(choose_nt((int64_t)d->n * d->ho * d->wo, ng, cg % 4 == 0 && d->cin % 4 == 0, pinned_tile(d, 0), halo_ok_desc(d, 0), wino_ok_desc(d, 0)));
}

static int conv2d_fwd_stats_impl(const suhpe_conv_t* d, const float* x, const float* w, float* y, float* stats, float* stats_extra,
                                 int64_t seg_row, suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_fwd_stats");
    if (rc) return rc;
    SUHPE_REQUIRE(x && w && y && stats, "suhpe_conv2d_fwd_stats: null pointer");
    if (stem_ok(d) && pinned_tile(d, 0) < 0) {
        launch_stem(d, x, w, nullptr, y, stats, stats_extra, seg_row, as_stream(stream));
        note_kernel(suhpe_conv2d_kernel_name(d, 0));
        return check_launch("suhpe_conv2d_fwd_stats");
    }
    NtParams p;
    const int cg = d->cin / d->groups;
    p.g = make_geom(d->n, d->h, d->w, d->ho, d->wo, d->cin, cg, d->r, d->s, d->stride, d->pad, 0);
    p.x = x; p.f = w; p.bias = nullptr; p.y = y; p.Ng = d->cout / d->groups; p.ldy = d->cout; p.ldf = p.g.K;
    p.x_bytes = (int64_t)d->n * d->h * d->w * d->cin * 4; p.f_bytes = (int64_t)d->cout * p.g.K * 4;
    p.stats = stats; p.stats_extra = stats_extra; p.seg_row = seg_row; p.stat_gx = suhpe_conv2d_fwd_stats_tiles(d); p.stat_cout = d->cout; p.scale = p.shift = p.residual = nullptr; p.act = 0; p.res_mode = 0;
    const bool vec4 = (cg % 4 == 0) && (d->cin % 4 == 0);
    SUHPE_REQUIRE(!vec4 || (aligned16(x) && aligned16(w)), "suhpe_conv2d_fwd_stats: x / w must be 16-byte aligned");
    g_nt_unsupported = false;
    dispatch_nt(p, d->groups, vec4, as_stream(stream), pinned_tile(d, 0));
    if (g_nt_unsupported) { set_error("suhpe_conv2d_fwd_stats: the pinned tile cannot run this call (operand alignment / size): unpin it"); return SUHPE_EINVAL; }
    return check_launch("suhpe_conv2d_fwd_stats");
}

int suhpe_conv2d_fwd_stats(const suhpe_conv_t* d, const float* x, const float* w, float* y, float* stats, suhpe_stream_t stream) {
    return conv2d_fwd_stats_impl(d, x, w, y, stats, nullptr, 0, stream);
}

int suhpe_conv2d_fwd_stats_seg(const suhpe_conv_t* d, const float* x, const float* w, float* y, float* stats, float* stats_extra,
                               int64_t seg_row, suhpe_stream_t stream) {
    SUHPE_REQUIRE(stats_extra && seg_row > 0, "suhpe_conv2d_fwd_stats_seg: stats_extra and a positive seg_row are required");
    return conv2d_fwd_stats_impl(d, x, w, y, stats, stats_extra, seg_row, stream);
}

// ---- Winograd tile with a pre-transformed filter (conv_wino.hip)
int suhpe_conv2d_wino_pinned(const suhpe_conv_t* d, int pass) {
    if (!d || d->groups <= 0 || pass < 0 || pass > 1 || !wino_ok_desc(d, pass)) return 0;
    const int cg = d->cin / d->groups, ng = d->cout / d->groups;
    const int64_t M = (int64_t)d->n * d->ho * d->wo;
    const NtTile tile = pass == 0 ? choose_nt(M, ng, true, pinned_tile(d, 0), halo_ok_desc(d, 0), true) : choose_nt(M, cg, true, pinned_tile(d, 1), halo_ok_desc(d, 1), true);
    return tile == NT_G64x64 ? 1 : 0;
}

int64_t suhpe_conv2d_wino_filter_bytes(const suhpe_conv_t* d, int pass) {
    if (!d || d->groups <= 0 || pass < 0 || pass > 1) return 0;
    const int cg = d->cin / d->groups, ng = d->cout / d->groups;
    return (pass == 0 ? wino_filter_floats(d->groups, ng, cg) : wino_filter_floats(d->groups, cg, ng)) * 4;
}

int suhpe_conv2d_wino_filter(const suhpe_conv_t* d, int pass, const float* w, float* fw, suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_wino_filter");
    if (rc) return rc;
    SUHPE_REQUIRE((pass == 0 || pass == 1) && d->r == 3 && d->s == 3 && w && fw && aligned16(fw), "suhpe_conv2d_wino_filter: 3x3 filters, pass 0 / 1, non-null 16-byte aligned output");
    const int cg = d->cin / d->groups, ng = d->cout / d->groups;
    if (pass == 0) launch_wino_filter(w, fw, d->groups, ng, cg, ng, cg, 0, as_stream(stream));
    else launch_wino_filter(w, fw, d->groups, cg, ng, ng, cg, 1, as_stream(stream));
    return check_launch("suhpe_conv2d_wino_filter");
}

int suhpe_conv2d_fwd_wino(const suhpe_conv_t* d, const float* x, const float* w, const float* fw, const float* bias, float* y, float* stats,
                          float* stats_extra, int64_t seg_row, suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_fwd_wino");
    if (rc) return rc;
    SUHPE_REQUIRE(x && w && y, "suhpe_conv2d_fwd_wino: null pointer");
    SUHPE_REQUIRE(!stats_extra || (stats && seg_row > 0), "suhpe_conv2d_fwd_wino: stats_extra needs stats and a positive seg_row");
    NtParams p;
    const int cg = d->cin / d->groups;
    p.g = make_geom(d->n, d->h, d->w, d->ho, d->wo, d->cin, cg, d->r, d->s, d->stride, d->pad, 0);
    p.x = x; p.f = w; p.bias = bias; p.y = y; p.Ng = d->cout / d->groups; p.ldy = d->cout; p.ldf = p.g.K;
    p.x_bytes = (int64_t)d->n * d->h * d->w * d->cin * 4; p.f_bytes = (int64_t)d->cout * p.g.K * 4;
    p.stats = stats; p.stats_extra = stats_extra; p.seg_row = stats ? seg_row : 0; p.stat_gx = stats ? suhpe_conv2d_fwd_stats_tiles(d) : 0; p.stat_cout = d->cout;
    p.scale = p.shift = p.residual = nullptr; p.act = 0; p.res_mode = 0;
    p.fw = fw;
    const bool vec4 = (cg % 4 == 0) && (d->cin % 4 == 0) && aligned16(x) && aligned16(w);
    if ((rc = dispatch_nt_checked(p, d->groups, vec4, as_stream(stream), pinned_tile(d, 0), "suhpe_conv2d_fwd_wino"))) return rc;
    return check_launch("suhpe_conv2d_fwd_wino");
}

int suhpe_conv2d_fwd_affine_wino(const suhpe_conv_t* d, const float* x, const float* w, const float* fw, const float* scale, const float* shift, int act,
                                 const float* residual, int residual_after_act, float* y, suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_fwd_affine_wino");
    if (rc) return rc;
    SUHPE_REQUIRE(x && w && y && scale && shift, "suhpe_conv2d_fwd_affine_wino: null pointer");
    SUHPE_REQUIRE(act >= SUHPE_ACT_NONE && act <= SUHPE_ACT_SIGMOID, "suhpe_conv2d_fwd_affine_wino: unknown activation %d", act);
    NtParams p;
    const int cg = d->cin / d->groups;
    p.g = make_geom(d->n, d->h, d->w, d->ho, d->wo, d->cin, cg, d->r, d->s, d->stride, d->pad, 0);
    p.x = x; p.f = w; p.bias = nullptr; p.y = y; p.Ng = d->cout / d->groups; p.ldy = d->cout; p.ldf = p.g.K;
    p.x_bytes = (int64_t)d->n * d->h * d->w * d->cin * 4; p.f_bytes = (int64_t)d->cout * p.g.K * 4;
    p.stats = nullptr; p.stats_extra = nullptr; p.seg_row = 0; p.stat_gx = 0; p.stat_cout = 0;
    p.scale = scale; p.shift = shift; p.residual = residual; p.act = act; p.res_mode = residual ? (residual_after_act ? 2 : 1) : 0;
    p.fw = fw;
    const bool vec4 = (cg % 4 == 0) && (d->cin % 4 == 0) && aligned16(x) && aligned16(w);
    if ((rc = dispatch_nt_checked(p, d->groups, vec4, as_stream(stream), pinned_tile(d, 0), "suhpe_conv2d_fwd_affine_wino"))) return rc;
    return check_launch("suhpe_conv2d_fwd_affine_wino");
}

int suhpe_conv2d_dgrad_wino(const suhpe_conv_t* d, const float* dy, const float* wt, const float* fw, const float* skip, float* dx,
                            suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_dgrad_wino");
    if (rc) return rc;
    SUHPE_REQUIRE(dy && dx && (wt || (fw && suhpe_conv2d_wino_pinned(d, 1))), "suhpe_conv2d_dgrad_wino: null pointer (wt may be NULL only with fw on a Winograd-pinned conv)");
    SUHPE_REQUIRE(d->stride == 1, "suhpe_conv2d_dgrad_wino: stride 1 only");
    NtParams p;
    const int ng = d->cout / d->groups, cg = d->cin / d->groups;
    const bool vec4 = (ng % 4 == 0) && (d->cout % 4 == 0) && aligned16(dy) && aligned16(wt);
    p.g = make_geom(d->n, d->ho, d->wo, d->h, d->w, d->cout, ng, d->r, d->s, d->stride, d->pad, 1);
    p.x = dy; p.f = wt; p.bias = nullptr; p.y = dx; p.Ng = cg; p.ldy = d->cin; p.ldf = p.g.K;
    p.x_bytes = (int64_t)d->n * d->ho * d->wo * d->cout * 4; p.f_bytes = (int64_t)d->cout * cg * d->r * d->s * 4;
    p.stats = nullptr; p.stats_extra = nullptr; p.seg_row = 0; p.stat_gx = 0; p.stat_cout = 0; p.scale = p.shift = nullptr; p.residual = skip; p.act = 0; p.res_mode = skip ? 3 : 0;
    p.fw = fw;
    g_nt_unsupported = false;
    if (!wt) {                                                // no direct-tile operand: the Winograd kernel must take this launch
        if (!launch_wino(p, d->groups, as_stream(stream))) { set_error("suhpe_conv2d_dgrad_wino: the Winograd kernel cannot run these operands and wt is NULL"); return SUHPE_EINVAL; }
    } else {
        if ((rc = dispatch_nt_checked(p, d->groups, vec4, as_stream(stream), pinned_tile(d, 1), "suhpe_conv2d_dgrad_wino"))) return rc;
    }
    return check_launch("suhpe_conv2d_dgrad_wino");
}

static bool xf_desc_ok(const suhpe_conv_t* d) {
    return d && d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && d->groups == 1 && d->cin % 4 == 0 && d->cout % 4 == 0;
}

int suhpe_conv2d_fwd_stats_xf(const suhpe_conv_t* d, const float* x, const float* w, float* y, float* stats, float* stats_extra, int64_t seg_row,
                              const float* a_scale, const float* a_shift, const float* a_rowscale, int a_act, int a_hw, suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_fwd_stats_xf");
    if (rc) return rc;
    SUHPE_REQUIRE(xf_desc_ok(d), "suhpe_conv2d_fwd_stats_xf: pointwise convs only (1x1, stride 1, pad 0, groups 1, channels %% 4 == 0)");
    SUHPE_REQUIRE(x && w && y && a_scale && a_shift && aligned16(x) && aligned16(w) && aligned16(a_scale) && aligned16(a_shift) && aligned16(a_rowscale),
                  "suhpe_conv2d_fwd_stats_xf: null or unaligned pointer");
    SUHPE_REQUIRE((a_act == SUHPE_ACT_NONE || a_act == SUHPE_ACT_RELU || a_act == SUHPE_ACT_SILU) && a_hw > 0 && seg_row >= 0 && (!stats_extra || seg_row > 0) &&
                      (!seg_row || stats_extra || !stats) && d->cin <= 4096,
                  "suhpe_conv2d_fwd_stats_xf: activation must be none / relu / silu; bad hw / segment arguments; cin <= 4096");
    NtParams p;
    p.g = make_geom(d->n, d->h, d->w, d->ho, d->wo, d->cin, d->cin, 1, 1, 1, 0, 0);
    p.x = x; p.f = w; p.bias = nullptr; p.y = y; p.Ng = d->cout; p.ldy = d->cout; p.ldf = p.g.K;
    p.x_bytes = (int64_t)d->n * d->h * d->w * d->cin * 4; p.f_bytes = (int64_t)d->cout * p.g.K * 4;
    p.stats = stats; p.stats_extra = stats_extra; p.seg_row = seg_row; p.stat_gx = stats ? suhpe_conv2d_fwd_stats_tiles(d) : 0; p.stat_cout = d->cout;
    p.scale = p.shift = p.residual = nullptr; p.act = 0; p.res_mode = 0;
    p.a_scale = a_scale; p.a_shift = a_shift; p.a_rowscale = a_rowscale; p.a_act = a_act; p.a_hw = a_hw; p.a_seg_row = seg_row;
    magic_div((unsigned)a_hw, &p.a_div_mul, &p.a_div_shr);
    g_nt_unsupported = false;
    dispatch_nt(p, 1, true, as_stream(stream), pinned_tile(d, 0));
    if (g_nt_unsupported) { set_error("suhpe_conv2d_fwd_stats_xf: the selected tile has no input-transform instantiation"); return SUHPE_EINVAL; }
    return check_launch("suhpe_conv2d_fwd_stats_xf");
}

int suhpe_conv2d_transpose_filter(const suhpe_conv_t* d, const float* w, float* wt, suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_transpose_filter");
    if (rc) return rc;
    SUHPE_REQUIRE(w && wt && w != wt, "suhpe_conv2d_transpose_filter: null or aliased pointers");
    const int64_t total = (int64_t)d->cout * d->r * d->s * (d->cin / d->groups);
    hipLaunchKernelGGL(transpose_filter_kernel, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), w, wt, d->groups,
                       d->cout / d->groups, d->cin / d->groups, d->r * d->s);
    return check_launch("suhpe_conv2d_transpose_filter");
}

int suhpe_transpose_filters_batched(const void* jobs_dev, const int64_t* chunk_start_dev, int njobs, int64_t total_chunks,
                                    suhpe_stream_t stream) {
    SUHPE_REQUIRE(jobs_dev && chunk_start_dev && njobs > 0 && total_chunks > 0 && total_chunks < (1LL << 31), "suhpe_transpose_filters_batched: bad arguments");
    hipLaunchKernelGGL(transpose_filters_batched_kernel, dim3((unsigned)total_chunks), dim3(256), 0, as_stream(stream),
                       static_cast<const TransposeJob*>(jobs_dev), chunk_start_dev, njobs);
    return check_launch("suhpe_transpose_filters_batched");
}

int suhpe_conv2d_dgrad(const suhpe_conv_t* d, const float* dy, const float* wt, float* dx, suhpe_stream_t stream) {
    return suhpe_conv2d_dgrad_add(d, dy, wt, nullptr, dx, stream);
}

int suhpe_conv2d_dgrad_add(const suhpe_conv_t* d, const float* dy, const float* wt, const float* skip, float* dx, suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_dgrad");
    if (rc) return rc;
    SUHPE_REQUIRE(dy && wt && dx, "suhpe_conv2d_dgrad: null pointer");
    NtParams p;
    const int ng = d->cout / d->groups, cg = d->cin / d->groups;
    // rows = input pixels (n,h,w); gathered tensor = dy (n,ho,wo,cout); filter rows = wt [cin][r][s][cout/groups]
    const bool vec4 = (ng % 4 == 0) && (d->cout % 4 == 0) && aligned16(dy) && aligned16(wt);
    if (d->stride == 1) {
        p.g = make_geom(d->n, d->ho, d->wo, d->h, d->w, d->cout, ng, d->r, d->s, d->stride, d->pad, 1);
        p.x = dy; p.f = wt; p.bias = nullptr; p.y = dx; p.Ng = cg; p.ldy = d->cin; p.ldf = p.g.K;
        p.x_bytes = (int64_t)d->n * d->ho * d->wo * d->cout * 4; p.f_bytes = (int64_t)d->cout * cg * d->r * d->s * 4;
        p.stats = nullptr; p.stats_extra = nullptr; p.seg_row = 0; p.stat_gx = 0; p.stat_cout = 0; p.scale = p.shift = nullptr; p.residual = skip; p.act = 0; p.res_mode = skip ? 3 : 0;
        if ((rc = dispatch_nt_checked(p, d->groups, vec4, as_stream(stream), pinned_tile(d, 1), "suhpe_conv2d_dgrad"))) return rc;
    } else {
        // one launch per parity class of the input pixel: only the taps r = (ph+pad) mod stride (+ k*stride) reach it,
        // so no MFMA work is spent on taps that are structurally zero (a masked gather would waste 1 - 1/stride^2).
        const int st = d->stride;
        for (int ph = 0; ph < st; ++ph)
            for (int pw = 0; pw < st; ++pw) {
                const int rh = (d->h - ph + st - 1) / st, rw = (d->w - pw + st - 1) / st;      // sub-grid extents
                if (rh <= 0 || rw <= 0) continue;
                const int r0 = (ph + d->pad) % st, s0 = (pw + d->pad) % st;
                const int nth = r0 < d->r ? (d->r - r0 + st - 1) / st : 0, ntw = s0 < d->s ? (d->s - s0 + st - 1) / st : 0;
                GatherGeom g = make_geom(d->n, d->ho, d->wo, rh, rw, d->cout, ng, nth, ntw, 1, 0, 1);
                g.K = nth * ntw * ng;
                g.om = st; g.ph = ph; g.pw = pw; g.YH = d->h; g.YW = d->w; g.r0 = r0; g.s0 = s0; g.Sfull = d->s;
                g.off_h = (ph + d->pad - r0) / st; g.off_w = (pw + d->pad - s0) / st;
                p.g = g;
                p.x = dy; p.f = wt; p.bias = nullptr; p.y = dx; p.Ng = cg; p.ldy = d->cin; p.ldf = d->r * d->s * ng;
                p.x_bytes = (int64_t)d->n * d->ho * d->wo * d->cout * 4; p.f_bytes = (int64_t)d->cout * cg * d->r * d->s * 4;
                p.stats = nullptr; p.stats_extra = nullptr; p.seg_row = 0; p.stat_gx = 0; p.stat_cout = 0; p.scale = p.shift = nullptr; p.residual = skip; p.act = 0; p.res_mode = skip ? 3 : 0;
                if ((rc = dispatch_nt_checked(p, d->groups, vec4, as_stream(stream), pinned_tile(d, 1), "suhpe_conv2d_dgrad"))) return rc;
            }
    }
    return check_launch("suhpe_conv2d_dgrad");
}

static int wgrad_splits(const suhpe_conv_t* d, int bm, int bn) {
    const int ng = d->cout / d->groups, K = d->r * d->s * (d->cin / d->groups);
    const int64_t M = (int64_t)d->n * d->ho * d->wo;
    const int64_t tiles = ceil_div(ng, bm) * ceil_div(K, bn) * d->groups;
    static const int target_blocks = getenv("SUHPE_WGRAD_BLOCKS") ? atoi(getenv("SUHPE_WGRAD_BLOCKS")) : 768;   // experiment knob
    int64_t splits = ceil_div(target_blocks, tiles);
    const int64_t max_by_m = ceil_div(M, 16 * BK);          // at least 512 rows per split
    if (splits > max_by_m) splits = max_by_m;
    const int64_t slab = (int64_t)d->cout * K * 4;
    const int64_t cap = (int64_t)384 << 20;                  // keep slabs under 384 MiB
    if (splits * slab > cap) splits = cap / slab;
    if (splits < 1) splits = 1;
    return (int)(ceil_div(splits, 8) * 8);                   // multiple of 8: one XCD per split (see igemm_tn_kernel)
}

// is the filter gradient of this conv on the Winograd kernel? (pinned by suhpe_conv2d_set_tile; never the heuristic's choice)
static bool wgrad_wino(const suhpe_conv_t* d) {
    const int pin = pinned_tile(d, 2);
    return (pin == TN_WINO || (pin < 0 && wino_preferred())) && wino_wgrad_geom_ok(d->n, d->h, d->w, d->ho, d->wo, d->cin, d->cout, d->r, d->s, d->stride, d->pad, d->groups);
}

static void wgrad_tile(const suhpe_conv_t* d, int* bm, int* bn) {
    const int ng = d->cout / d->groups, cg = d->cin / d->groups, K = d->r * d->s * cg;
    const bool va = ng % 4 == 0 && d->cout % 4 == 0, vb = cg % 4 == 0 && d->cin % 4 == 0;
    if (!(va && vb)) { *bm = 64; *bn = 64; return; }       // scalar loaders: generic 64x64 tile
    const int pin = pinned_tile(d, 2);
    const TnTile& t = kTnTiles[choose_tn(ng, K, pin == TN_WINO ? -1 : pin)];
    *bm = t.bm;
    *bn = t.bn;
}

int suhpe_conv2d_set_stagger(int quanta, int first_gen) { g_stagger = quanta; g_stagger_first = first_gen; return SUHPE_OK; }

int suhpe_conv2d_tile_count(int pass) { return pass == 2 ? TN_TILES : (int)NT_TILES; }

const char* suhpe_conv2d_tile_name(int pass, int tile) {
    if (tile < 0 || tile >= suhpe_conv2d_tile_count(pass)) return "";
    return pass == 2 ? kTnTiles[tile].name : kNtNames[tile];
}

int suhpe_conv2d_set_tile(const suhpe_conv_t* d, int pass, int tile) {
    int rc = check_desc(d, "suhpe_conv2d_set_tile");
    if (rc) return rc;
    SUHPE_REQUIRE(pass >= 0 && pass <= 2, "suhpe_conv2d_set_tile: pass must be 0 (fwd), 1 (dgrad) or 2 (wgrad)");
    SUHPE_REQUIRE(tile >= -1 && tile < suhpe_conv2d_tile_count(pass), "suhpe_conv2d_set_tile: tile %d outside the menu", tile);
    const int ng = d->cout / d->groups, cg = d->cin / d->groups;
    const bool v4 = ng % 4 == 0 && d->cout % 4 == 0 && cg % 4 == 0 && d->cin % 4 == 0;
    SUHPE_REQUIRE(tile < 0 || (v4 && !(pass != 2 && tile == (int)NT_64x64_S)), "suhpe_conv2d_set_tile: the vector tiles need channel counts that are multiples of 4");
    SUHPE_REQUIRE(pass == 2 || tile < (int)NT_H64x64 || tile >= (int)NT_W128x128 || halo_ok_desc(d, pass), "suhpe_conv2d_set_tile: the halo tiles need a 3x3 / stride 1 / pad 1 conv with evenly chunked channels");
    SUHPE_REQUIRE(pass != 2 || tile != TN_WINO || wino_wgrad_geom_ok(d->n, d->h, d->w, d->ho, d->wo, d->cin, d->cout, d->r, d->s, d->stride, d->pad, d->groups),
                  "suhpe_conv2d_set_tile: the Winograd filter-gradient tile needs a 3x3 / stride 1 / pad 1 conv on even output sizes");
    SUHPE_REQUIRE(pass == 2 || tile != (int)NT_G64x64 || wino_ok_desc(d, pass), "suhpe_conv2d_set_tile: the Winograd tile needs a 3x3 / stride 1 / pad 1 conv on even output sizes");
    SUHPE_REQUIRE(pass == 2 || tile < (int)NT_R128x16 || tile == (int)NT_G64x64 || tile == (int)NT_D128x128 || tile == (int)NT_D256x64 || resident_ok_desc(d, pass, tile == (int)NT_R128x16 ? 128 : 64),
                  "suhpe_conv2d_set_tile: the resident tiles need a 1x1 / stride 1 conv whose BM x K block fits the LDS");
    SUHPE_REQUIRE(pass == 2 || tile != (int)NT_D128x128 || dma_ok_desc(d, pass), "suhpe_conv2d_set_tile: the LDS-DMA tile needs a 1x1 / stride 1 conv, one group, reduction a multiple of 32");
    SUHPE_REQUIRE(pass == 2 || tile != (int)NT_D256x64 || dma_ok_desc(d, pass, true), "suhpe_conv2d_set_tile: the narrow LDS-DMA tile needs a 1x1 / stride 1 conv, one group, reduction a multiple of 32, at most 64 outputs");
    std::lock_guard<std::mutex> lock(tile_mutex());
    if (tile < 0) tile_table().erase(tile_key(d, pass));
    else tile_table()[tile_key(d, pass)] = tile;
    return SUHPE_OK;
}

const char* suhpe_last_kernel_name(void) { return last_kernel(); }

const char* suhpe_conv2d_kernel_name(const suhpe_conv_t* d, int pass) {
    if (!d || d->groups <= 0) return "";
    const int ng = d->cout / d->groups, cg = d->cin / d->groups;
    const bool plain = d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0;
    if (pass == 0 && stem_ok(d) && pinned_tile(d, 0) < 0) return d->cout == 24 ? "rgb_conv3x3_kernel<24>" : (d->cout == 48 ? "rgb_conv3x3_kernel<48>" : "rgb_conv3x3_kernel<64>");
    // (the Winograd tile id covers two kernels: few channels on both sides run the register-resident form of conv_wino_small.hip -- its launches
    //  are reported under their own name so that per-kernel figures, bench.py's roofline among them, do not mix the two)
    if (pass == 0) {
        const NtTile tl = effective_nt(choose_nt((int64_t)d->n * d->ho * d->wo, ng, cg % 4 == 0 && d->cin % 4 == 0, pinned_tile(d, 0), halo_ok_desc(d, 0), wino_ok_desc(d, 0)), plain, halo_ok_desc(d, 0), resident_ok_desc(d, 0, 128), resident_ok_desc(d, 0, 64), wino_ok_desc(d, 0), dma_ok_desc(d, 0));
        if (tl == NT_G64x64 && wino_small_desc_ok(cg, ng, (int64_t)d->n * d->ho * d->wo, d->ho, d->wo)) return "wino_small_kernel";
        return kNtNames[tl];
    }
    if (pass == 1) {
        const int st = d->stride;
        const NtTile tl = effective_nt(choose_nt((int64_t)d->n * ((d->h + st - 1) / st) * ((d->w + st - 1) / st), cg, ng % 4 == 0 && d->cout % 4 == 0, pinned_tile(d, 1), halo_ok_desc(d, 1), wino_ok_desc(d, 1)), plain, halo_ok_desc(d, 1), resident_ok_desc(d, 1, 128), resident_ok_desc(d, 1, 64), wino_ok_desc(d, 1), dma_ok_desc(d, 1));
        if (tl == NT_G64x64 && wino_small_desc_ok(ng, cg, (int64_t)d->n * d->h * d->w, d->h, d->w)) return "wino_small_kernel";
        return kNtNames[tl];
    }
    const bool va = ng % 4 == 0 && d->cout % 4 == 0, vb = cg % 4 == 0 && d->cin % 4 == 0;
    if (!(va && vb)) return va ? "igemm_tn_kernel<64,64,32,32,4,1>" : (vb ? "igemm_tn_kernel<64,64,32,32,1,4>" : "igemm_tn_kernel<64,64,32,32,1,1>");
    if (wgrad_wino(d)) return wino_small_wgrad_splits(d->n, d->h, d->w, cg, ng, d->groups) ? "wino_small_wgrad_kernel" : kTnTiles[TN_WINO].name;
    { const int pin = pinned_tile(d, 2); return kTnTiles[choose_tn(ng, d->r * d->s * cg, pin == TN_WINO ? -1 : pin)].name; }
}

int64_t suhpe_conv2d_wgrad_workspace_bytes(const suhpe_conv_t* d) {
    if (!d || d->groups <= 0) return 0;
    if (wgrad_wino(d)) return wino_wgrad_workspace_floats(d->n, d->h, d->w, d->cin, d->cout, d->groups) * 4;
    int bm, bn;
    wgrad_tile(d, &bm, &bn);
    const int K = d->r * d->s * (d->cin / d->groups);
    return (int64_t)wgrad_splits(d, bm, bn) * d->cout * K * 4;
}

struct XInput { const float* scale; const float* shift; const float* rowscale; int act, hw; int64_t seg_row; };

static int wgrad_partial_impl(const suhpe_conv_t* d, const float* x, const float* dy, void* workspace, int64_t workspace_bytes, const XInput* xf,
                              suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_wgrad");
    if (rc) return rc;
    SUHPE_REQUIRE(x && dy && workspace, "suhpe_conv2d_wgrad: null pointer");
    if (workspace_bytes < suhpe_conv2d_wgrad_workspace_bytes(d)) {
        set_error("suhpe_conv2d_wgrad: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)suhpe_conv2d_wgrad_workspace_bytes(d));
        return SUHPE_EWORKSPACE;
    }
    if (wgrad_wino(d) && !xf && aligned16(x) && aligned16(dy) && aligned16(workspace)) {
        // Winograd F(3x3, 2x2): the launch leaves ONE complete filter-shaped slab at the start of the workspace (suhpe_conv2d_wgrad_splits = 1)
        launch_wino_wgrad(x, dy, static_cast<float*>(workspace), d->n, d->h, d->w, d->cin, d->cout, d->groups, as_stream(stream));
        note_kernel(suhpe_conv2d_kernel_name(d, 2));
        return check_launch("suhpe_conv2d_wgrad_partial");
    }
    SUHPE_REQUIRE(!wgrad_wino(d), "suhpe_conv2d_wgrad: the Winograd filter-gradient tile needs 16-byte aligned operands and no input transform");
    int bm, bn;
    wgrad_tile(d, &bm, &bn);
    const int splits = wgrad_splits(d, bm, bn);
    const int ng = d->cout / d->groups, cg = d->cin / d->groups;
    TnParams p;
    p.g = make_geom(d->n, d->h, d->w, d->ho, d->wo, d->cin, cg, d->r, d->s, d->stride, d->pad, 0);
    p.dy = dy; p.x = x; p.slab = static_cast<float*>(workspace); p.Ng = ng; p.lddy = d->cout;
    p.dy_bytes = (int64_t)d->n * d->ho * d->wo * d->cout * 4; p.x_bytes = (int64_t)d->n * d->h * d->w * d->cin * 4;
    p.m_per_split = ceil_div(ceil_div(p.g.M, splits), BK) * BK;
    p.tiles_n = (int)ceil_div(p.g.K, bn);
    p.tiles_m = (int)ceil_div(ng, bm);
    const bool va = (ng % 4 == 0) && (d->cout % 4 == 0) && aligned16(dy);
    const bool vb = (cg % 4 == 0) && (d->cin % 4 == 0) && aligned16(x);
    if (xf) {
        SUHPE_REQUIRE(va && vb, "suhpe_conv2d_wgrad_partial_xf: 16-byte aligned operands required");
        p.x_scale = xf->scale; p.x_shift = xf->shift; p.x_rowscale = xf->rowscale; p.x_act = xf->act; p.x_hw = xf->hw; p.x_seg_row = xf->seg_row;
        magic_div((unsigned)xf->hw, &p.x_div_mul, &p.x_div_shr);
    }
    dim3 grid(p.tiles_m * p.tiles_n * splits, 1, d->groups);
    hipStream_t st = as_stream(stream);
#define TN_LAUNCH(BM_, BN_, WM_, WN_, VA_, VB_) \
    hipLaunchKernelGGL((igemm_tn_kernel<BM_, BN_, WM_, WN_, VA_, VB_>), grid, dim3(((BM_) / (WM_)) * ((BN_) / (WN_)) * 64), 0, st, p)
#define TN_LAUNCH_XF(BM_, BN_, WM_, WN_) \
    hipLaunchKernelGGL((igemm_tn_kernel<BM_, BN_, WM_, WN_, 4, 4, true>), grid, dim3(((BM_) / (WM_)) * ((BN_) / (WN_)) * 64), 0, st, p)
    if (xf) {
        if (bm == 128 && bn == 128) TN_LAUNCH_XF(128, 128, 64, 32);
        else if (bm == 128 && bn == 64) TN_LAUNCH_XF(128, 64, 64, 32);
        else if (bm == 64 && bn == 128) TN_LAUNCH_XF(64, 128, 32, 64);
        else if (bm == 32 && bn == 128) TN_LAUNCH_XF(32, 128, 32, 32);
        else if (bm == 128 && bn == 32) TN_LAUNCH_XF(128, 32, 32, 32);
        else TN_LAUNCH_XF(64, 64, 32, 32);
    } else if (va && vb) {
        if (bm == 128 && bn == 128) TN_LAUNCH(128, 128, 64, 32, 4, 4);       // 8 wavefronts of 64x32
        else if (bm == 128 && bn == 64) TN_LAUNCH(128, 64, 64, 32, 4, 4);
        else if (bm == 64 && bn == 128) TN_LAUNCH(64, 128, 32, 64, 4, 4);
        else if (bm == 32 && bn == 128) TN_LAUNCH(32, 128, 32, 32, 4, 4);
        else if (bm == 128 && bn == 32) TN_LAUNCH(128, 32, 32, 32, 4, 4);
        else TN_LAUNCH(64, 64, 32, 32, 4, 4);
    } else {
        // odd channel counts (RGB stem, 9-D head): scalar loaders, generic 64x64 tile (wgrad_tile returned 64x64)
        if (va) TN_LAUNCH(64, 64, 32, 32, 4, 1);
        else if (vb) TN_LAUNCH(64, 64, 32, 32, 1, 4);
        else TN_LAUNCH(64, 64, 32, 32, 1, 1);
    }
#undef TN_LAUNCH
#undef TN_LAUNCH_XF
    note_kernel(suhpe_conv2d_kernel_name(d, 2));
    return check_launch("suhpe_conv2d_wgrad_partial");
}

int suhpe_conv2d_wgrad_partial(const suhpe_conv_t* d, const float* x, const float* dy, void* workspace, int64_t workspace_bytes,
                               suhpe_stream_t stream) {
    return wgrad_partial_impl(d, x, dy, workspace, workspace_bytes, nullptr, stream);
}

int suhpe_conv2d_wgrad_partial_xf(const suhpe_conv_t* d, const float* x, const float* dy, void* workspace, int64_t workspace_bytes,
                                  const float* a_scale, const float* a_shift, const float* a_rowscale, int a_act, int a_hw, int64_t seg_row,
                                  suhpe_stream_t stream) {
    SUHPE_REQUIRE(xf_desc_ok(d), "suhpe_conv2d_wgrad_partial_xf: pointwise convs only (1x1, stride 1, pad 0, groups 1, channels %% 4 == 0)");
    SUHPE_REQUIRE(a_scale && a_shift && (a_act == SUHPE_ACT_NONE || a_act == SUHPE_ACT_RELU || a_act == SUHPE_ACT_SILU) && a_hw > 0 && seg_row >= 0 &&
                      aligned16(a_rowscale),
                  "suhpe_conv2d_wgrad_partial_xf: bad transform arguments (activation: none / relu / silu)");
    const XInput xf{a_scale, a_shift, a_rowscale, a_act, a_hw, seg_row};
    return wgrad_partial_impl(d, x, dy, workspace, workspace_bytes, &xf, stream);
}

int suhpe_conv2d_wgrad_reduce(const suhpe_conv_t* d, const void* workspace, float* dw, int accumulate, suhpe_stream_t stream) {
    int rc = check_desc(d, "suhpe_conv2d_wgrad");
    if (rc) return rc;
    SUHPE_REQUIRE(workspace && dw, "suhpe_conv2d_wgrad: null pointer");
    int bm, bn;
    wgrad_tile(d, &bm, &bn);
    const int splits = wgrad_wino(d) ? 1 : wgrad_splits(d, bm, bn);
    const int64_t count = (int64_t)d->cout * d->r * d->s * (d->cin / d->groups);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(grid_for(count, 256)), dim3(256), 0, as_stream(stream), static_cast<const float*>(workspace),
                       splits, count, dw, accumulate);
    return check_launch("suhpe_conv2d_wgrad_reduce");
}

int suhpe_conv2d_wgrad_splits(const suhpe_conv_t* d) {
    if (!d || d->groups <= 0) return 0;
    if (wgrad_wino(d)) return 1;
    int bm, bn;
    wgrad_tile(d, &bm, &bn);
    return wgrad_splits(d, bm, bn);
}

int suhpe_wgrad_reduce_batched(const void* jobs_dev, const int64_t* chunk_start_dev, int njobs, int64_t total_chunks, suhpe_stream_t stream) {
    SUHPE_REQUIRE(jobs_dev && chunk_start_dev && njobs > 0 && total_chunks > 0 && total_chunks < (1LL << 31), "suhpe_wgrad_reduce_batched: bad arguments");
    hipLaunchKernelGGL(slab_reduce_batched_kernel, dim3((unsigned)total_chunks), dim3(256), 0, as_stream(stream), static_cast<const SlabJob*>(jobs_dev),
                       chunk_start_dev, njobs);
    return check_launch("suhpe_wgrad_reduce_batched");
}

int suhpe_conv2d_wgrad(const suhpe_conv_t* d, const float* x, const float* dy, float* dw, int accumulate, void* workspace,
                       int64_t workspace_bytes, suhpe_stream_t stream) {
    SUHPE_REQUIRE(dw, "suhpe_conv2d_wgrad: null pointer");
    int rc = suhpe_conv2d_wgrad_partial(d, x, dy, workspace, workspace_bytes, stream);
    if (rc) return rc;
    return suhpe_conv2d_wgrad_reduce(d, workspace, dw, accumulate, stream);
}

}  // extern "C"
