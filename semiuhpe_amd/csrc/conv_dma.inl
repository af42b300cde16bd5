// Pointwise GEMMs with BOTH operands on the LDS-DMA path (included by conv_igemm.hip; round 5).
//
//   out[m][j] = sum_k G[m][k] * F[j][k]      1x1 / stride 1 convolutions, one group, K % 32 == 0
//
// What VERDICT r4 item 1 asked for: conv_wino.hip's pipeline for the plain GEMMs.  A persistent workgroup (8 wavefronts, one per CU) walks output
// tiles in the XCD-aware order of the other tiles; the (tile, k-slab) pairs of a workgroup form ONE stream whose slabs (32 k of both operands)
// travel global -> LDS by global_load_lds_dwordx4 into a ring of STAGES slabs, STAGES - 1 slabs ahead of the MFMAs and across tile boundaries.
// LDS rows are 128 bytes, unpadded: the DMA lane that fills chunk position c' of row r fetches chunk c' ^ ((r >> 1) & 7), which makes every
// ds_read_b128 of the fragments conflict-free (lane groups of 16 cover the 64 banks once).  ONE barrier per slab, in the MIDDLE of the slab's MFMAs:
// it publishes slab g + 1 (every wavefront has waited for its own DMA pieces: counted vmcnt) and frees slab g - 1's stage for the DMA of slab
// g + STAGES - 1; fragment reads run one k-group ahead, also across the slab boundary.  No vector-memory operation of the loop is visible to hipcc
// (DMA, output stores, statistics records and waits are inline asm): see conv_wino.hip on what a compiler-placed s_waitcnt vmcnt does to such a loop.
// Every wavefront issues EXACTLY NST output stores per tile (a lane that must not store gets an out-of-range offset, the hardware drops it): the
// hand-counted waits rely on it.
//
// Two shapes of the same kernel:
//   WIDE   128 x 128 tiles, wavefronts 2 (rows) x 4 (columns) of 64 x 32, ring of four 32 KB slabs: the MBConv expand convs of EfficientNetV2-S
//          (128 -> 512, 256 -> 1536 ...), the input gradients of their project convs, TinyViT's token GEMMs.  Menu tile "gemm_nt_dma<128,128>".
//   NARROW 256 x 64 tiles, wavefronts 8 (rows) x 1 of 32 x 64, ring of three 40 KB slabs: outputs of at most 64 channels -- the project convs of
//          torchvision's FusedMBConv (192 -> 48 @ 56^2, 256 -> 64 @ 28^2), which are HBM-bound streams (19 flop per byte).  Every element of the
//          gathered operand is read from LDS by exactly ONE wavefront here, so the producer's train-mode BatchNorm + activation (XF: a <-
//          act(a * scale[s][k] + shift[s][k]), what igemm_nt_kernel<..., XF = 1> does in its register stage) is applied to the FRAGMENT, once per
//          element, on the way from LDS to the matrix pipe, under the previous k-group's MFMAs; the scale / shift vectors sit in LDS behind the ring.
//          Menu tile "gemm_nt_dma<256,64>".
//
// Same fragment order as igemm_nt_kernel (lane half h takes k = 8 g + 4 h + j of a slab): plain results are BIT-IDENTICAL to the igemm / persistent /
// resident tiles, and the XF form to igemm_nt_kernel's XF form (same fmaf + xf_act per element) -- tests/test_gpu_kernels.py.  Epilogues: plain,
// BatchNorm statistics records per (32 * TM)-row unit of a wavefront ((S, Q, k, n), segment rule of gemm_nt_resident), from registers, no LDS, no
// barrier.  Everything else (bias, skip add, folded eval BatchNorm, activation epilogues, per-sample loader scale, groups) keeps the older tiles.
//
// Measured (tools/proto/pw_lab.hip, profiles/r05_pw_lab.txt, profiles/r05_retune_dma.txt; cold caches): the WIDE form is ~5 % ahead of the best
// older tile on the short reductions (7 840 x 1 536 x 256: 59 against 62 us) and 10-30 % on TinyViT's 56^2 token GEMMs.  The ablations say why it is
// not more: with stores, DMA and barrier all compiled out the loop reaches 118-121 TF/s (0.76 of the nominal fp32-MFMA peak) -- launch ramp, tile
// quantisation, the clock the chip holds under load and LDS fragment latency are what is left.

typedef unsigned dma_u32x4 __attribute__((ext_vector_type(4)));

template <int TM, int TN, int WAVES_M, int WAVES_N, int STAGES>
struct DmaCfg {
    static constexpr int BM = 32 * TM * WAVES_M, BN = 32 * TN * WAVES_N;
    static constexpr int NW = WAVES_M * WAVES_N;
    static constexpr int PIECES = (BM + BN) / 8, PPW = PIECES / NW;       // 1 KiB DMA pieces of a slab, per wavefront
    static constexpr int NST = 16 * TM * TN;                                // output stores per wavefront and tile
    static constexpr int STAGE_FLOATS = (BM + BN) * 32;
    static constexpr int RING_BYTES = STAGES * STAGE_FLOATS * 4;
    static_assert(NW == 8 && PIECES % NW == 0 && STAGES >= 3 && STAGES <= 4, "eight wavefronts, whole pieces per wavefront");
};

// XF: 0 = none; 1 = the gathered operand is act(x * a_scale[s][k] + a_shift[s][k]) of the stored tensor, act = p.a_act; 2 = the same with act = SiLU
// at compile time (WAVES_N == 1 only)
template <int TM, int TN, int WAVES_M, int WAVES_N, int STAGES, bool STATS, int XF>
__global__ __launch_bounds__(512) void gemm_nt_dma(const NtParams p, int total_vb, int nk) {
    using C = DmaCfg<TM, TN, WAVES_M, WAVES_N, STAGES>;
    constexpr int BM = C::BM, BN = C::BN, PPW = C::PPW, NST = C::NST, STAGE_FLOATS = C::STAGE_FLOATS, APIECES = BM / 8;
    static_assert(XF == 0 || WAVES_N == 1, "the fragment transform needs every gathered element read by one wavefront");
    // EARLY (the narrow form): the slab's barrier sits at the slab's START instead of its middle.  It publishes slab g AND frees slab g - 1's stage,
    // and slab g + 2's DMA is issued right behind it: every slab has TWO slab times to land and two slabs are in flight at any moment inside a ring
    // of three.  (With the mid-slab barrier a ring of three keeps ONE slab in flight for one slab time: on these HBM-bound shapes the first narrow
    // version ran in exactly MFMA time + HBM time -- 185 us = 89 + 96 on 192 -> 48 @ 56^2 -- i.e. without any overlap.)  The price: no fragment
    // read across the slab boundary, so the matrix pipe waits out one LDS round trip per slab -- nothing on a kernel that waits for memory.
    constexpr bool EARLY = STAGES == 3;
    extern __shared__ __attribute__((aligned(16))) float dma_smem[];
    const GatherGeom& g = p.g;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int gstride = gridDim.x;
    const int M = (int)g.M;
    auto tile_of = [&](int vb, int* m0, int* n0, int* rt) __attribute__((always_inline)) {
        const int jx = vb >> 3;
        *n0 = (jx % p.col_tiles) * BN;
        *rt = (jx / p.col_tiles) * 8 + (vb & 7);
        *m0 = *rt * BM;
    };
    // Row tiles are padded to a multiple of 8 for the XCD-aware order; a workgroup (fixed XCD index blockIdx.x & 7) meets its padding tiles only at
    // the END of its sequence, so it simply stops before them: its own tile limit replaces the launch's
    {
        const int row_tiles = (M + BM - 1) / BM, xcd = (int)blockIdx.x & 7;
        const int groups8 = (row_tiles >> 3) + (xcd < (row_tiles & 7) ? 1 : 0);
        total_vb = groups8 * p.col_tiles * 8;
    }
    if ((int)blockIdx.x >= total_vb) return;

    // ---- input transform: scale / shift of both batch segments behind the ring: [segment][scale K | shift K]
    float* xf_tab = dma_smem + STAGES * STAGE_FLOATS;
    if constexpr (XF != 0) {
        const int nseg = p.a_seg_row > 0 ? 2 : 1;
        for (int i = threadIdx.x; i < nseg * g.K; i += 512) {
            const int s = i / g.K, k = i - s * g.K;
            xf_tab[s * 2 * g.K + k] = p.a_scale[i];
            xf_tab[s * 2 * g.K + g.K + k] = p.a_shift[i];
        }
        __syncthreads();
    }

    // ---- loader cursor over the stream of (tile, slab) pairs; piece q of a slab: q < BM / 8 -> rows 8 q .. 8 q + 7 of the gathered operand, else of the filter
    int l_vb = blockIdx.x, l_kt = 0, l_stage = 0;
    unsigned voff[PPW];
    auto setup_loader = [&]() __attribute__((always_inline)) {
        int m0, n0, rt;
        tile_of(l_vb, &m0, &n0, &rt);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave * PPW + i;                        // (uniform)
            const int rl = 8 * (q < APIECES ? q : q - APIECES) + (lane >> 3);
            const int chunk = (lane & 7) ^ ((rl >> 1) & 7);      // the chunk that belongs at this lane's LDS position
            if (q < APIECES) {
                int r = m0 + rl; r = r < M ? r : M - 1;          // (rows beyond M repeat the last row: never stored)
                voff[i] = ((unsigned)r * (unsigned)g.ldg + 4u * chunk) * 4u;
            } else {
                int r = n0 + rl; r = r < p.Ng ? r : p.Ng - 1;
                voff[i] = ((unsigned)r * (unsigned)p.ldf + 4u * chunk) * 4u;
            }
        }
    };
    auto issue_slab = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave * PPW + i;
            const float* base = (q < APIECES ? p.x : p.f) + l_kt * 32;
            const unsigned lds_dst = __builtin_amdgcn_readfirstlane(
                (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(dma_smem + l_stage * STAGE_FLOATS + q * 256));
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %2\n\t"
                "s_nop 2\n\t"
                "global_load_lds_dwordx4 %1, %3\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep) : "v"(voff[i]), "s"(lds_dst), "s"(base) : "memory");
        }
        l_stage = l_stage == STAGES - 1 ? 0 : l_stage + 1;
        if (++l_kt == nk) {
            l_kt = 0;
            l_vb += gstride;
            if (l_vb < total_vb) setup_loader();
        }
    };

    // ---- fragments: lane (r32, h) reads row r32 of its 32-row block, k-group kg -> logical chunk 2 kg + h, i.e. position (2 kg) ^ (h ^ swizzle)
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
        b_rd[j] = BM * 32 + row * 32 + ((h ^ ((row >> 1) & 7)) << 2);
    }
    float4 af[2][TM], bf[2][TN];
    auto read_frags = [&](int stage, int kg, int set) __attribute__((always_inline)) {
        const float* S = dma_smem + stage * STAGE_FLOATS;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const float4*>(&S[a_rd[i] ^ (kg << 3)]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[set][j] = *reinterpret_cast<const float4*>(&S[b_rd[j] ^ (kg << 3)]);
    };
    // the producer's BatchNorm + activation on fragment set `set` = k-group kg of slab kt of the tile under the MFMAs.  Branch-free: a lane reads the
    // scale / shift vectors of ITS row's batch segment (xf_row[i] = float offset of that segment's block in the table), so the tile that straddles the
    // boundary needs no other code, and the whole k-group -- eight MFMAs of the previous fragments, this transform, the next fragment reads -- is ONE
    // scheduling region: interleave() below pins an MFMA / vector-ALU pattern on it.  (Left to itself hipcc keeps the MFMAs of a k-group together
    // and the transform behind them; the two wavefronts of a SIMD run in phase, so for that long neither fed the matrix pipe: the transform cost
    // 192 -> 48 @ 56^2 40 us on top of 146.)  XF == 2: the activation is SiLU at compile time (no selects).
    int xf_row[XF ? TM : 1];
    auto transform = [&](int kt, int kg, int set) __attribute__((always_inline)) {
        if constexpr (XF != 0) {
            const int k0 = kt * 32 + kg * 8 + h * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float* t0 = xf_tab + xf_row[i] + k0;
                const float4 sc = *reinterpret_cast<const float4*>(t0), sh = *reinterpret_cast<const float4*>(t0 + g.K);
                auto act1 = [&](float a, float c, float d) __attribute__((always_inline)) {
                    const float u = a * c + d;
                    if constexpr (XF == 2) return u * __builtin_amdgcn_rcpf(1.f + __expf(-u));
                    else return xf_act(u, p.a_act);
                };
                af[set][i].x = act1(af[set][i].x, sc.x, sh.x);
                af[set][i].y = act1(af[set][i].y, sc.y, sh.y);
                af[set][i].z = act1(af[set][i].z, sc.z, sh.z);
                af[set][i].w = act1(af[set][i].w, sc.w, sh.w);
            }
        }
    };
    // one MFMA, then a handful of vector-ALU instructions, eight times: the transform of the next fragments rides in the issue slots between this
    // k-group's MFMAs (a 64-cycle fp32 MFMA leaves its wavefront ~48 cycles of vector issue)
    auto interleave = [&]() __attribute__((always_inline)) {
        if constexpr (XF != 0) {
#pragma unroll
            for (int r = 0; r < 4 * TM * TN; ++r) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, XF == 2 ? 4 : 6, 0);      // vector ALU (the transcendentals count among them)
            }
        }
    };
    auto setup_xf = [&](int vb) __attribute__((always_inline)) {
        if constexpr (XF != 0) {
            int m0, n0, rt;
            tile_of(vb, &m0, &n0, &rt);
#pragma unroll
            for (int i = 0; i < TM; ++i) xf_row[i] = (p.a_seg_row > 0 && m0 + (wm * TM + i) * 32 + r32 >= p.a_seg_row) ? 2 * (int)g.K : 0;
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
    auto mma = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i].x, bf[set][j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i].y, bf[set][j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i].z, bf[set][j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i].w, bf[set][j].w, acc[i][j], 0, 0, 0);
            }
    };
    // output / statistics buffers as descriptors of four scalars (out-of-range offset = dropped store)
    auto make4 = [&](const void* ptr, int64_t nb) __attribute__((always_inline)) {
        dma_u32x4 r;
        const uint64_t addr = reinterpret_cast<uint64_t>(ptr);
        r.x = __builtin_amdgcn_readfirstlane((uint32_t)addr);
        r.y = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32) & 0xffffu);
        r.z = __builtin_amdgcn_readfirstlane((uint32_t)(nb > 0x7fffffff ? 0x7fffffff : nb));
        r.w = 0x00020000u;
        return r;
    };
    const dma_u32x4 yrs = make4(p.y, (int64_t)M * p.ldy * 4);
    const dma_u32x4 srs = make4(STATS ? (const void*)p.stats : (const void*)p.y, 0x7fffffff);
    const dma_u32x4 ers = make4(STATS && p.stats_extra ? (const void*)p.stats_extra : (const void*)p.y, 0x7fffffff);

    auto epilogue = [&](int vb) __attribute__((always_inline)) {
        int m0, n0, rt;
        tile_of(vb, &m0, &n0, &rt);
        // (the stores are asm statements: hipcc's hazard recognizer does not pad an MFMA result -> vector-memory read for them; 20 wait states cover
        //  the 16-pass MFMA that wrote the last accumulator)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
        const int rowb = m0 + wm * TM * 32;                     // first row of this wavefront's unit
        const unsigned ldy4 = (unsigned)p.ldy * 4u;
        // a row beyond M lands beyond the descriptor's extent by itself (offset >= M * ldy * 4)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + r32;
            const unsigned off0 = col < p.Ng ? ((unsigned)(rowb + 4 * h) * (unsigned)p.ldy + (unsigned)col) * 4u : kOob;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned o = off0 + (unsigned)(i * 32 + (e & 3) + 8 * (e >> 2)) * ldy4;
                    const float v = acc[i][j][e];
                    // (s_nop 4: the descriptor may reach its scalar registers through a v_readlane / v_readfirstlane the compiler places right in front of
                    //  this statement -- it parks scalars in vector registers under pressure -- and a vector-memory instruction must not read a scalar
                    //  register within five wait states of a vector instruction writing it; hipcc pads that hazard only for instructions it can see)
                    asm volatile("s_nop 4\n\tbuffer_store_dword %1, %0, %2, 0 offen" : : "v"(o), "v"(v), "s"(yrs) : "memory");
                }
        }
        if constexpr (STATS) {
            // BatchNorm statistics record(s) of the unit: (S, Q, k, n) per column over its n valid rows, shifted by k = the unit's first-row value;
            // a unit that contains the segment boundary emits its first-segment rows to `stats` and the rest to `stats_extra` (gemm_nt_resident's rule)
            if (rowb < M) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                constexpr int UR = 32 * TM;
                const int nrows = M - rowb < UR ? M - rowb : UR;
                const int64_t cut64 = p.seg_row - rowb;
                const bool two = p.stats_extra && cut64 > 0 && cut64 < nrows;
                const int cut = two ? (int)cut64 : nrows;
                const int unit = rt * WAVES_M + wm;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + (wn * TN + j) * 32 + r32;
                    const bool okc = col < p.Ng;
                    const float k = __shfl(acc[0][j][0], r32, 64);
                    auto emit = [&](int lo, int hi, const dma_u32x4& rs, unsigned off) __attribute__((always_inline)) {
                        float a = 0.f, q = 0.f;
                        if (lo <= 0 && hi >= UR) {               // every row of the unit counts: no per-element range tests (the vector ALU is the matrix pipe)
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
                                    const float v = (lr >= lo && lr < hi) ? acc[i][j][e] - k : 0.f;
                                    a += v;
                                    q += v * v;
                                }
                        }
                        a += __shfl_xor(a, 32, 64);
                        q += __shfl_xor(q, 32, 64);
                        const f4 rec = {a, q, k, (float)(hi - lo)};
                        const unsigned o = (h == 0 && okc) ? off : kOob;
                        asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %1, %0, %2, 0 offen\n\ts_nop 1" : : "v"(o), "v"(rec), "s"(rs) : "memory");
                    };
                    emit(0, cut, srs, (unsigned)(((int64_t)unit * p.stat_cout + col) * 16));
                    if (two) emit(cut, nrows, ers, (unsigned)col * 16u);
                }
            }
        }
    };

    const int my_tiles = (total_vb - (int)blockIdx.x + gstride - 1) / gstride;
    const int n_slabs = my_tiles * nk;
    setup_loader();
    int issued = 0;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (issued < n_slabs) { issue_slab(); ++issued; }
    if constexpr (!EARLY) {
        if (n_slabs >= STAGES - 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"((STAGES - 2) * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    zero_acc();
    int c_vb = blockIdx.x, c_kt = 0, c_stage = 0;
    setup_xf(c_vb);
    // bit b: a tile ended (its NST stores were issued) at the end of slab g - 1 - b.
    //   mid-slab barrier: slab g + 1's DMA was issued at mid-slab g + 2 - STAGES, so the stores of tiles that ended at slabs g + 2 - STAGES .. g - 1
    //   (bits 0 .. STAGES - 3) are younger than it;   EARLY: slab g's DMA was issued at the top of slab g - 2: bits 0 and 1.
    // At least NST stores each: the statistics records come on top and are not counted (a smaller count only waits for more).
    unsigned ended = 0;
    constexpr unsigned ENDED_MASK = EARLY ? 3u : (1u << (STAGES - 2)) - 1u;
    constexpr int IN_FLIGHT = EARLY ? PPW : (STAGES - 3) * PPW;
    constexpr int WAIT_PLAIN = IN_FLIGHT, WAIT_ENDED = IN_FLIGHT + NST > 63 ? 63 : IN_FLIGHT + NST;
    if constexpr (!EARLY) {
        read_frags(0, 0, 0);
        transform(0, 0, 0);
    }
    for (int gs = 0; gs < n_slabs; ++gs) {
        const int n_stage = c_stage == STAGES - 1 ? 0 : c_stage + 1;
        const bool last = c_kt == nk - 1;
        if constexpr (EARLY) {
            if (gs + 1 < issued) {                               // slab gs + 1 is in flight behind slab gs
                if (ended & ENDED_MASK) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WAIT_ENDED) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WAIT_PLAIN) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (issued < n_slabs) { issue_slab(); ++issued; }
            read_frags(c_stage, 0, 0);
            transform(c_kt, 0, 0);
        }
        read_frags(c_stage, 1, 1);
        mma(0);
        transform(c_kt, 1, 1);
        interleave();
        read_frags(c_stage, 2, 0);
        mma(1);
        transform(c_kt, 2, 0);
        interleave();
        if constexpr (!EARLY) {
            if (gs + 1 < n_slabs) {
                if (issued - (gs + 2) == STAGES - 3) {
                    if (ended & ENDED_MASK) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WAIT_ENDED) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(WAIT_PLAIN) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __builtin_amdgcn_s_barrier();
            if (issued < n_slabs) { issue_slab(); ++issued; }
        }
        read_frags(c_stage, 3, 1);
        mma(0);
        transform(c_kt, 3, 1);
        interleave();
        if constexpr (!EARLY) { if (gs + 1 < n_slabs) read_frags(n_stage, 0, 0); }
        mma(1);
        ended <<= 1;
        if (last) {
            epilogue(c_vb);
            zero_acc();
            c_kt = 0;
            c_vb += gstride;
            ended |= 1u;
            if (gs + 1 < n_slabs) setup_xf(c_vb);
        } else {
            ++c_kt;
        }
        if constexpr (!EARLY) { if (gs + 1 < n_slabs) transform(c_kt, 0, 0); }      // (k-group 0 of the next slab: after the tile switch, its segment may differ)
        c_stage = n_stage;
    }
}

// ---- host side
using DmaWide = DmaCfg<2, 1, 2, 4, 4>;        // 128 x 128, ring of four
using DmaNarrow = DmaCfg<1, 2, 8, 1, 3>;      // 256 x 64, ring of three

// plain pointwise GEMM, one group, reduction a multiple of 32, 16-byte rows, every operand below 2 GiB (32-bit byte offsets), no epilogue operand;
// the narrow form also takes the XF = 1 loader transform (no per-sample scale) for outputs of at most 64 channels
static bool dma_ok(const NtParams& p, int groups, bool vec4, bool narrow) {
    const bool xf_ok = !p.a_scale || (narrow && p.a_shift && !p.a_rowscale && p.g.K <= 1024);
    const int rec_rows = narrow ? 32 : 64;
    return vec4 && groups == 1 && nt_mode(p.g) == 1 && xf_ok && !p.bias && !p.scale && !p.residual && p.res_mode == 0 && !p.y2 && !p.o_rowscale &&
           (!narrow || p.Ng <= 64) && p.g.K % 32 == 0 && p.g.K >= 32 && p.g.ldg % 4 == 0 && p.ldf % 4 == 0 && p.g.M >= 1 && p.Ng >= 1 &&
           p.g.M * (int64_t)p.g.ldg * 4 < (1LL << 31) && p.g.M * (int64_t)p.ldy * 4 < (1LL << 31) && (int64_t)p.Ng * p.ldf * 4 < (1LL << 31) &&
           aligned16(p.x) && aligned16(p.f) && (!p.stats || aligned16(p.stats)) && (!p.stats_extra || aligned16(p.stats_extra)) &&
           (!p.stats || (int64_t)ceil_div(p.g.M, rec_rows) * p.stat_cout * 16 < (1LL << 31));
}

template <typename Cfg, int TM, int TN, int WAVES_M, int WAVES_N, int STAGES>
static void launch_dma_cfg(const NtParams& p_in, hipStream_t st) {
    NtParams p = p_in;
    p.col_tiles = (int)ceil_div(p.Ng, Cfg::BN);
    const int64_t row_tiles8 = ceil_div(ceil_div(p.g.M, Cfg::BM), 8) * 8;
    const int total_vb = (int)(row_tiles8 * p.col_tiles);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (cus <= 0) cus = 256;
    }
    int gx = cus / 8 * 8;
    if (gx < 8) gx = 8;
    if (gx > total_vb) gx = total_vb;                           // (total_vb is a multiple of 8)
    const int nk = (int)(p.g.K / 32);
    const int lds = Cfg::RING_BYTES + (p.a_scale ? 4 * (int)p.g.K * 4 : 0);
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3(gx), dim3(512), lds, st, p, total_vb, nk);
    };
    if constexpr (WAVES_N == 1) {
        if (p.a_scale) {
            if (p.a_act == SUHPE_ACT_SILU) {
                if (p.stats) go(gemm_nt_dma<TM, TN, WAVES_M, WAVES_N, STAGES, true, 2>); else go(gemm_nt_dma<TM, TN, WAVES_M, WAVES_N, STAGES, false, 2>);
            } else {
                if (p.stats) go(gemm_nt_dma<TM, TN, WAVES_M, WAVES_N, STAGES, true, 1>); else go(gemm_nt_dma<TM, TN, WAVES_M, WAVES_N, STAGES, false, 1>);
            }
            return;
        }
    }
    if (p.stats) go(gemm_nt_dma<TM, TN, WAVES_M, WAVES_N, STAGES, true, 0>); else go(gemm_nt_dma<TM, TN, WAVES_M, WAVES_N, STAGES, false, 0>);
}
static void launch_dma(const NtParams& p, hipStream_t st) { launch_dma_cfg<DmaWide, 2, 1, 2, 4, 4>(p, st); }
static void launch_dma_narrow(const NtParams& p, hipStream_t st) { launch_dma_cfg<DmaNarrow, 1, 2, 8, 1, 3>(p, st); }

// capability of a (descriptor, pass): pass 0 = forward, 1 = input gradient
static bool dma_ok_desc(const suhpe_conv_t* d, int pass, bool narrow = false) {
    if (!d || d->groups != 1 || pass > 1 || d->r != 1 || d->s != 1 || d->stride != 1 || d->pad != 0) return false;
    if (d->cin % 4 || d->cout % 4) return false;
    const int K = pass == 0 ? d->cin : d->cout, N = pass == 0 ? d->cout : d->cin;
    if (narrow && N > 64) return false;
    return K % 32 == 0 && (int64_t)d->n * d->h * d->w * (d->cin > d->cout ? d->cin : d->cout) * 4 < (1LL << 31);
}
