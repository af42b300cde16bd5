// 3x3 / stride 1 / pad 1 implicit GEMM with the gathered operand's HALO TILE staged once in LDS (included by conv_igemm.hip).
//
// igemm_nt_kernel gathers the A operand tap by tap: every input pixel of a row tile travels global -> LDS nine times (once per
// filter tap), and a tile of consecutive pixels needs the image rows above and below it as well -- rocprof showed the 3x3
// launches pulling 2.7-4.3x their operand through L2 (profiles/README.md, round 1).  Here a tile is a set of output pixels whose
// input footprint (the tile + a one-pixel ring) is copied into LDS ONCE per channel chunk; the nine taps of that chunk then read
// their A fragments from the same LDS image at a constant pixel shift: fragment(row i, tap (r,s)) = halo[slot(i) + (r-1)*P + (s-1)].
// Padding needs no predicate in the main loop: pixels outside the image are zeros in the LDS image (out-of-range descriptor loads).
//
//   PATCH tiles (OH, OW multiples of 8): 8x8 output patches, one (BM = 64) or two consecutive ones (BM = 128), halo 10x10 per patch
//     (1.56x the outputs instead of 2.75x for 64 consecutive pixels of a 56-wide image), LDS pitch P = 12 pixels;
//   RUN tiles (any size): BM consecutive pixels in (n, oh, ow) order; the LDS image holds the covered image rows + one above / below,
//     each with a zero column left and right (P = OW + 2) and a zero row between consecutive images.
//
// Reduction order: channel chunk (CC = 32 / 48 / 24 channels) outermost, then tap, then channel -- one slab = one tap of one chunk.
// Lane -> pixel permutation of the PATCH tiles: ds_read_b128 serves a wave in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}
// (+32); with the LDS row pitch of CC+4 floats a group is conflict-free iff its 16 pixel slots are distinct mod 16.  With P = 12 the
// four rows of a 4x8 pixel block sit at slots {0..7}, {12..19}, {24..31}, {36..43} = {0..7}, {12..15,0..3}, {8..15}, {4..11} mod 16,
// so rows 0+2 form one conflict-free group and rows 1+3 the other: lane rho of a 32-row MFMA block maps to
//   g = bit (rho>>2) of 0x96, idx = 4*(rho>>3) + (rho&3), pixel (iy, ix) = (2*(idx>>3) + g, idx & 7),
// a constant tap shift moves all slots alike, and the epilogue inverts the same map.
//
// Replaces (with igemm_nt_kernel as the fallback for other geometries) the cuDNN forward / input-gradient kernels behind the 3x3
// convolutions of torchvision's FusedMBConv (src/networks.py:303) and of RepVGGBlock.rbr_dense (src/repvgg.py:28-32,67-76).

struct HaloGeom {
    int patch;            // 1: PATCH tiles, 0: RUN tiles
    int PX, PXY;          // PATCH: patches per image row, patches per image
    unsigned div_pxy_mul, div_pxy_shr, div_px_mul, div_px_shr;
    int P;                // LDS pixels per halo row (12 or OW + 2)
    int rows_cov;         // RUN: LDS halo rows per tile (upper bound)
    unsigned div_p_mul, div_p_shr;         // / (OW + 2)
    unsigned div_h1_mul, div_h1_shr;       // / (OH + 1)
    int nslots;           // LDS pixel slots of the A image
    int nchunks;          // ceil(Cg / CC)
};

__device__ __forceinline__ void halo_pi(int rho, int* iy, int* ix) {
    const int g = (0x96 >> (rho >> 2)) & 1, idx = ((rho >> 3) << 2) + (rho & 3);
    *iy = 2 * (idx >> 3) + g;
    *ix = idx & 7;
}

// tile row (0..BM) -> output pixel index in (n, oh, ow) order (PATCH tiles; RUN tiles are the identity m0 + row)
__device__ __forceinline__ int64_t halo_patch_orow(const GatherGeom& g, const HaloGeom& hg, int64_t m0, int row) {
    const unsigned pidx = (unsigned)(m0 >> 6) + (unsigned)(row >> 6);
    const unsigned n = fast_div(pidx, hg.div_pxy_mul, hg.div_pxy_shr);
    const unsigned pr = pidx - n * (unsigned)hg.PXY;
    const unsigned py = fast_div(pr, hg.div_px_mul, hg.div_px_shr), px = pr - py * (unsigned)hg.PX;
    int iy, ix;
    halo_pi(row & 31, &iy, &ix);
    iy += ((row >> 5) & 1) * 4;
    return ((int64_t)n * g.OH + py * 8 + iy) * g.OW + px * 8 + ix;
}

template <int BM, int BN, int WM, int WN, int CC, bool PATCH, bool AFF>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, ((BM / WM) * (BN / WN) > 4) ? 4 : 2)
void igemm_nt3x3_kernel(const NtParams p, const HaloGeom hg) {
    constexpr int WAVES_N = BN / WN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    constexpr int LDP = CC + 4;                      // LDS row pitch (floats): (CC+4)/4 is odd for CC = 24, 32, 48
    constexpr int VC = CC / 4;                       // 16-byte vectors per row
    constexpr int KG = CC / 8;                       // k-groups of 8 per slab
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                // [nslots][LDP]
    float* Bs = smem + (size_t)hg.nslots * LDP;      // [BN][LDP]

    const GatherGeom& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int grp = blockIdx.z;
    // XCD-aware order as igemm_nt_kernel (spatial footprint: each XCD takes one contiguous run of row tiles)
    const int b = blockIdx.x, jx = b >> 3, col_tile = jx % p.col_tiles;
    int row_tile;
    {
        const int row_tiles = (int)((g.M + BM - 1) / BM), q = row_tiles >> 3, rem = row_tiles & 7, xcd = b & 7, jl = jx / p.col_tiles;
        if (jl >= q + (xcd < rem ? 1 : 0)) return;
        row_tile = xcd * q + (xcd < rem ? xcd : rem) + jl;
    }
    const int64_t m0 = (int64_t)row_tile * BM;
    const int n0 = col_tile * BN;
    const BufRsrc xr = make_rsrc(p.x + (int64_t)grp * g.Cg, p.x_bytes - (int64_t)grp * g.Cg * 4);
    const BufRsrc fr = make_rsrc(p.f + (int64_t)grp * p.Ng * p.ldf, p.f_bytes - (int64_t)grp * p.Ng * p.ldf * 4);
    const int sgn = g.mode == 0 ? 1 : -1;            // forward: input pixel = o + (tap - 1); dgrad: dy pixel = o - (tap - 1)
    const int H1 = g.OH + 1;

    // ---- the tile's first "virtual row" (RUN tiles): v = n*(OH+1) + oh + 1, v = 0 mod (OH+1) is the zero row between images
    int v_base = 0;
    if (!PATCH) {
        const unsigned n = fast_div((unsigned)m0, g.div_hw_mul, g.div_hw_shr);
        const unsigned rem = (unsigned)m0 - n * (unsigned)(g.OH * g.OW);
        const unsigned oh = fast_div(rem, g.div_w_mul, g.div_w_shr);
        v_base = (int)n * H1 + (int)oh + 1;
    }

    // ---- A image loader: pair index = t + i*NT over (slot, vector column); global pixel offsets are fixed for the whole kernel
    const int nload = PATCH ? (BM / 64) * 100 : hg.rows_cov * hg.P;
    constexpr int LPT_MAX = PATCH ? ((BM / 64) * 100 * VC + NT - 1) / NT : (288 * VC + NT - 1) / NT;
    int a_goff[LPT_MAX];          // pixel index * ldg (elements) of the pair's pixel, -1: zero (outside the image / beyond the list)
    int a_loff[LPT_MAX];          // LDS float offset of the pair
#pragma unroll
    for (int i = 0; i < LPT_MAX; ++i) {
        const int pair = t + i * NT;
        const int idx = pair / VC, vcol = pair - idx * VC;
        a_goff[i] = -1;
        a_loff[i] = 0;
        if (idx < nload) {
            int n, oh, ow, slot;
            if (PATCH) {
                const int bq = idx / 100, rem = idx - bq * 100, py = rem / 10, px = rem - py * 10;
                const unsigned pidx = (unsigned)(m0 >> 6) + (unsigned)bq;
                const unsigned nn = fast_div(pidx, hg.div_pxy_mul, hg.div_pxy_shr);
                const unsigned pr = pidx - nn * (unsigned)hg.PXY;
                const unsigned pyy = fast_div(pr, hg.div_px_mul, hg.div_px_shr), pxx = pr - pyy * (unsigned)hg.PX;
                n = (int)nn; oh = (int)pyy * 8 + py - 1; ow = (int)pxx * 8 + px - 1;
                slot = bq * 120 + py * 12 + px;
            } else {
                const unsigned vr = fast_div((unsigned)idx, hg.div_p_mul, hg.div_p_shr);
                const int c = idx - (int)vr * hg.P;
                const int v = v_base - 1 + (int)vr;
                const unsigned nn = fast_div((unsigned)(v < 0 ? 0 : v), hg.div_h1_mul, hg.div_h1_shr);
                n = (int)nn; oh = v - (int)nn * H1 - 1; ow = c - 1;
                if (v < 0) oh = -1;
                slot = idx;
            }
            const bool ok = (unsigned)oh < (unsigned)g.OH && (unsigned)ow < (unsigned)g.OW && n < g.N;
            a_goff[i] = ok ? ((n * g.OH + oh) * g.OW + ow) * g.ldg + vcol * 4 : -1;
            a_loff[i] = slot * LDP + vcol * 4;
        } else {
            a_loff[i] = -1;
        }
    }
    // ---- B loader: pair index over (filter row, vector column)
    constexpr int BPT = (BN * VC + NT - 1) / NT;
    int b_goff[BPT], b_loff[BPT];
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
        const int pair = t + i * NT;
        const int brow = pair / VC, vcol = pair - brow * VC;
        const int col = n0 + brow;
        b_goff[i] = (brow < BN && col < p.Ng) ? col * p.ldf + vcol * 4 : -1;
        b_loff[i] = brow < BN ? brow * LDP + vcol * 4 : -1;
    }

    // ---- A fragment bases: LDS float offset of this lane's rows (tap (1,1) position)
    const int r32 = lane & 31, h = lane >> 5;
    int arow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * WM + i * 32 + r32;
        int slot;
        if (PATCH) {
            int iy, ix;
            halo_pi(r32, &iy, &ix);
            iy += ((row >> 5) & 1) * 4;
            slot = (row >> 6) * 120 + (iy + 1) * 12 + ix + 1;
        } else {
            int64_t m = m0 + row;
            if (m >= g.M) m = g.M - 1;
            const unsigned n = fast_div((unsigned)m, g.div_hw_mul, g.div_hw_shr);
            const unsigned rem = (unsigned)m - n * (unsigned)(g.OH * g.OW);
            const unsigned oh = fast_div(rem, g.div_w_mul, g.div_w_shr), ow = rem - oh * (unsigned)g.OW;
            const int v = (int)n * H1 + (int)oh + 1;
            slot = (v - v_base + 1) * hg.P + (int)ow + 1;
        }
        arow[i] = slot * LDP + h * 4;
    }
    int brow_off[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) brow_off[j] = (wn * WN + j * 32 + r32) * LDP + h * 4;

    float a_stage[LPT_MAX][4];
    float b_stage[2][BPT][4];
    auto load_a = [&](int chunk) __attribute__((always_inline)) {
        const int c0 = chunk * CC;
#pragma unroll
        for (int i = 0; i < LPT_MAX; ++i) {
            const int vcol4 = ((t + i * NT) % VC) * 4;
            const bool ok = (a_goff[i] >= 0) & (c0 + vcol4 < g.Cg);
            buf_load<4>(xr, ok ? ((unsigned)a_goff[i] + (unsigned)c0) * 4u : kOob, a_stage[i]);
        }
    };
    auto store_a = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LPT_MAX; ++i) {
            if (a_loff[i] < 0) continue;
            *reinterpret_cast<float4*>(&As[a_loff[i]]) = make_float4(a_stage[i][0], a_stage[i][1], a_stage[i][2], a_stage[i][3]);
        }
    };
    auto load_b = [&](auto stage, int chunk, int tap) __attribute__((always_inline)) {
        auto& b_reg = b_stage[decltype(stage)::value];
        const int c0 = chunk * CC, k0 = tap * g.Cg + c0;
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int vcol4 = ((t + i * NT) % VC) * 4;
            const bool ok = (b_goff[i] >= 0) & (c0 + vcol4 < g.Cg);
            buf_load<4>(fr, ok ? ((unsigned)b_goff[i] + (unsigned)k0) * 4u : kOob, b_reg[i]);
        }
    };
    auto store_b = [&](auto stage) __attribute__((always_inline)) {
        auto& b_reg = b_stage[decltype(stage)::value];
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            if (b_loff[i] < 0) continue;
            *reinterpret_cast<float4*>(&Bs[b_loff[i]]) = make_float4(b_reg[i][0], b_reg[i][1], b_reg[i][2], b_reg[i][3]);
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto mma_slab = [&](int shift) __attribute__((always_inline)) {          // shift: LDS float offset of the tap
        float4 af[2][TM], bf[2][TN];
        auto read_frags = [&](int kg, int set) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[set][i] = *reinterpret_cast<const float4*>(&As[arow[i] + shift + kg * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[set][j] = *reinterpret_cast<const float4*>(&Bs[brow_off[j] + kg * 8]);
        };
        read_frags(0, 0);
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const int cur = kg & 1;
            if (kg + 1 < KG) read_frags(kg + 1, cur ^ 1);
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

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int nslabs = hg.nchunks * 9;
    // cursors: (chunk, tap) of the slab being multiplied, and of the slab two ahead (B loads)
    int c_chunk = 0, c_tap = 0, l_chunk = 0, l_tap = 0;
    auto advance = [](int& chunk, int& tap) __attribute__((always_inline)) { if (++tap == 9) { tap = 0; ++chunk; } };
    auto step = [&](int s, auto cur, auto nxt) __attribute__((always_inline)) {
        if (s + 2 < nslabs) { load_b(cur, l_chunk, l_tap); advance(l_chunk, l_tap); }     // slab s+2 into the stage slab s came from
        if (c_tap == 5 && c_chunk + 1 < hg.nchunks) load_a(c_chunk + 1);                   // next chunk's halo image, three taps ahead
        const int tr = c_tap / 3, ts = c_tap - tr * 3;
        mma_slab(sgn * ((tr - 1) * hg.P + (ts - 1)) * LDP);
        __syncthreads();
        if (s + 1 < nslabs) {
            store_b(nxt);
            if (c_tap == 8) store_a();
            __syncthreads();
        }
        advance(c_chunk, c_tap);
    };
    load_a(0);
    load_b(S0{}, 0, 0);
    advance(l_chunk, l_tap);
    store_a();
    store_b(S0{});
    __syncthreads();
    if (nslabs > 1) { load_b(S1{}, l_chunk, l_tap); advance(l_chunk, l_tap); }
    for (int s = 0; s < nslabs; s += 2) {
        step(s, S0{}, S1{});
        if (s + 1 < nslabs) step(s + 1, S1{}, S0{});
    }

    // ---- epilogue (igemm_nt_kernel's, with the tile-row -> pixel map of the PATCH tiles)
    float* __restrict__ y = p.y + (int64_t)grp * p.Ng;
    auto store_out = [&](auto aff) __attribute__((always_inline)) {
        constexpr bool AFF_ = decltype(aff)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int64_t orow[16];
            bool rok[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                rok[e] = m0 + row < g.M;
                orow[e] = PATCH ? halo_patch_orow(g, hg, m0, row) : m0 + row;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * WN + j * 32 + r32;
                if (col >= p.Ng) continue;
                // residual loads (skip gradient of the block) ahead of the stores: they may alias y as far as the compiler knows
                float rv[16];
                if (p.res_mode) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) rv[e] = rok[e] ? p.residual[orow[e] * p.ldy + grp * p.Ng + col] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (!rok[e]) continue;
                    if constexpr (!AFF_) {
                        y[orow[e] * p.ldy + col] = acc[i][j][e] + (p.bias ? p.bias[grp * p.Ng + col] : 0.f) + (p.res_mode == 3 ? rv[e] : 0.f);
                    } else {
                        float v = acc[i][j][e] * p.scale[grp * p.Ng + col] + p.shift[grp * p.Ng + col];
                        if (p.res_mode == 1) v += rv[e];
                        v = act_apply(v, p.act);
                        if (p.res_mode == 2) v += rv[e];
                        y[orow[e] * p.ldy + col] = v;
                    }
                }
            }
        }
    };
    store_out(std::integral_constant<bool, AFF>{});

    if (p.stats) {                                          // BatchNorm statistics records, as in igemm_nt_kernel
        __syncthreads();
        float* red = As;                                    // [BM/WM][BN][2]
        float* kk = As + (BM / WM) * BN * 2;                // [BN]
        if (wm == 0 && h == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) kk[wn * WN + j * 32 + r32] = acc[0][j][0];
        }
        __syncthreads();
        auto emit = [&](int64_t lo, int64_t hi, float4* dst0) __attribute__((always_inline)) {
            const int64_t wlo = lo - wm * WM, whi = hi - wm * WM;
            const bool full = wlo <= 0 && whi >= WM;            // (as in igemm_nt_kernel: no range tests where every row counts)
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
        const int64_t cut = p.seg_row - m0;                 // tile rows are image-major in both tile kinds: rows < cut = first segment
        if (p.stats_extra && cut > 0 && cut < nrows) {
            emit(0, cut, rec);
            emit(cut, nrows, reinterpret_cast<float4*>(p.stats_extra));
        } else {
            emit(0, nrows, rec);
        }
    }
}

// ---- 48-column PATCH tile: 128 rows x 48 columns, four wavefronts of 32 rows each: columns 0-31 on the 32x32x2 MFMA as above, columns
// 32-47 on two v_mfma_f32_16x16x4_f32 blocks (rows 0-15 / 16-31 of the wavefront's 32) -- an output with 48 channels (the input gradient
// of EfficientNetV2-S's 48 -> 192 FusedMBConv expand convs: 1.7 TFLOP per step) wastes a quarter of a 64-column tile's matrix work.
// 16x16x4 operands: lane l supplies A[row l & 15][k] and B[k][column l & 15] for the k of its 16-lane group q = l >> 4; a ds_read_b128 hands
// it four consecutive k, used by four MFMAs: group q of read g takes k = 4 (q * CC/16 + g) .. +3 of the slab (any assignment works as long as
// A and B agree).  Two-way LDS conflicts on these reads (no lane -> row permutation serves both fragment shapes), which the matrix pipe hides.
// Forward or input gradient without statistics / folded affine; bias and the skip-connection add (res_mode 3) in the epilogue.
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int CC>
__global__ __launch_bounds__(256, 2) void igemm_nt3x3_n48_kernel(const NtParams p, const HaloGeom hg) {
    constexpr int BM = 128, BN = 48, NT = 256;
    constexpr int LDP = CC + 4, VC = CC / 4, KG = CC / 8, NG16 = CC / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                // [240][LDP]
    float* Bs = smem + (size_t)hg.nslots * LDP;      // [48][LDP]
    const GatherGeom& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wm = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const int grp = blockIdx.z;
    const int b = blockIdx.x, jx = b >> 3, col_tile = jx % p.col_tiles;
    int row_tile;
    {
        const int row_tiles = (int)((g.M + BM - 1) / BM), q = row_tiles >> 3, rem = row_tiles & 7, xcd = b & 7, jl = jx / p.col_tiles;
        if (jl >= q + (xcd < rem ? 1 : 0)) return;
        row_tile = xcd * q + (xcd < rem ? xcd : rem) + jl;
    }
    const int64_t m0 = (int64_t)row_tile * BM;
    const int n0 = col_tile * BN;
    const BufRsrc xr = make_rsrc(p.x + (int64_t)grp * g.Cg, p.x_bytes - (int64_t)grp * g.Cg * 4);
    const BufRsrc fr = make_rsrc(p.f + (int64_t)grp * p.Ng * p.ldf, p.f_bytes - (int64_t)grp * p.Ng * p.ldf * 4);
    const int sgn = g.mode == 0 ? 1 : -1;
    // ---- A image loader (PATCH tiles: two 10 x 10 halo patches)
    constexpr int LPT_MAX = (2 * 100 * VC + NT - 1) / NT;
    int a_goff[LPT_MAX], a_loff[LPT_MAX];
#pragma unroll
    for (int i = 0; i < LPT_MAX; ++i) {
        const int pair = t + i * NT;
        const int idx = pair / VC, vcol = pair - idx * VC;
        a_goff[i] = -1;
        a_loff[i] = -1;
        if (idx < 200) {
            const int bq = idx / 100, rem = idx - bq * 100, py = rem / 10, px = rem - py * 10;
            const unsigned pidx = (unsigned)(m0 >> 6) + (unsigned)bq;
            const unsigned nn = fast_div(pidx, hg.div_pxy_mul, hg.div_pxy_shr);
            const unsigned pr = pidx - nn * (unsigned)hg.PXY;
            const unsigned pyy = fast_div(pr, hg.div_px_mul, hg.div_px_shr), pxx = pr - pyy * (unsigned)hg.PX;
            const int n = (int)nn, oh = (int)pyy * 8 + py - 1, ow = (int)pxx * 8 + px - 1;
            const bool ok = (unsigned)oh < (unsigned)g.OH && (unsigned)ow < (unsigned)g.OW && n < g.N;
            a_goff[i] = ok ? ((n * g.OH + oh) * g.OW + ow) * g.ldg + vcol * 4 : -1;
            a_loff[i] = (bq * 120 + py * 12 + px) * LDP + vcol * 4;
        }
    }
    constexpr int BPT = (BN * VC + NT - 1) / NT;
    int b_goff[BPT], b_loff[BPT];
#pragma unroll
    for (int i = 0; i < BPT; ++i) {
        const int pair = t + i * NT;
        const int brow = pair / VC, vcol = pair - brow * VC;
        const int col = n0 + brow;
        b_goff[i] = (brow < BN && col < p.Ng) ? col * p.ldf + vcol * 4 : -1;
        b_loff[i] = brow < BN ? brow * LDP + vcol * 4 : -1;
    }
    // ---- fragment bases (tap (1,1) position): the 32-row block of the wavefront, and its two 16-row blocks in the 16x16x4 lane order
    const int r32 = lane & 31, h = lane >> 5, i16 = lane & 15, q16 = lane >> 4;
    auto slot_of = [&](int row) __attribute__((always_inline)) {
        int iy, ix;
        halo_pi(row & 31, &iy, &ix);
        iy += ((row >> 5) & 1) * 4;
        return (row >> 6) * 120 + (iy + 1) * 12 + ix + 1;
    };
    const int arow32 = slot_of(wm * 32 + r32) * LDP + h * 4;
    int arow16[2];
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) arow16[bb] = slot_of(wm * 32 + bb * 16 + i16) * LDP + q16 * NG16 * 4;
    const int brow32 = r32 * LDP + h * 4, brow16 = (32 + i16) * LDP + q16 * NG16 * 4;

    float a_stage[LPT_MAX][4];
    float b_stage[2][BPT][4];
    auto load_a = [&](int chunk) __attribute__((always_inline)) {
        const int c0 = chunk * CC;
#pragma unroll
        for (int i = 0; i < LPT_MAX; ++i) {
            const int vcol4 = ((t + i * NT) % VC) * 4;
            const bool ok = (a_goff[i] >= 0) & (c0 + vcol4 < g.Cg);
            buf_load<4>(xr, ok ? ((unsigned)a_goff[i] + (unsigned)c0) * 4u : kOob, a_stage[i]);
        }
    };
    auto store_a = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LPT_MAX; ++i) {
            if (a_loff[i] < 0) continue;
            *reinterpret_cast<float4*>(&As[a_loff[i]]) = make_float4(a_stage[i][0], a_stage[i][1], a_stage[i][2], a_stage[i][3]);
        }
    };
    auto load_b = [&](auto stage, int chunk, int tap) __attribute__((always_inline)) {
        auto& b_reg = b_stage[decltype(stage)::value];
        const int c0 = chunk * CC, k0 = tap * g.Cg + c0;
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int vcol4 = ((t + i * NT) % VC) * 4;
            const bool ok = (b_goff[i] >= 0) & (c0 + vcol4 < g.Cg);
            buf_load<4>(fr, ok ? ((unsigned)b_goff[i] + (unsigned)k0) * 4u : kOob, b_reg[i]);
        }
    };
    auto store_b = [&](auto stage) __attribute__((always_inline)) {
        auto& b_reg = b_stage[decltype(stage)::value];
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            if (b_loff[i] < 0) continue;
            *reinterpret_cast<float4*>(&Bs[b_loff[i]]) = make_float4(b_reg[i][0], b_reg[i][1], b_reg[i][2], b_reg[i][3]);
        }
    };
    floatx16 acc32;
    floatx4 acc16[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc32[e] = 0.f;
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc16[bb][e] = 0.f;
    auto mma_slab = [&](int shift) __attribute__((always_inline)) {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const float4 af = *reinterpret_cast<const float4*>(&As[arow32 + shift + kg * 8]);
            const float4 bf = *reinterpret_cast<const float4*>(&Bs[brow32 + kg * 8]);
            acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc32, 0, 0, 0);
            acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc32, 0, 0, 0);
            acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc32, 0, 0, 0);
            acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc32, 0, 0, 0);
        }
#pragma unroll
        for (int gg = 0; gg < NG16; ++gg) {
            const float4 bf = *reinterpret_cast<const float4*>(&Bs[brow16 + gg * 4]);
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const float4 af = *reinterpret_cast<const float4*>(&As[arow16[bb] + shift + gg * 4]);
                acc16[bb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bf.x, acc16[bb], 0, 0, 0);
                acc16[bb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bf.y, acc16[bb], 0, 0, 0);
                acc16[bb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bf.z, acc16[bb], 0, 0, 0);
                acc16[bb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bf.w, acc16[bb], 0, 0, 0);
            }
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int nslabs = hg.nchunks * 9;
    int c_chunk = 0, c_tap = 0, l_chunk = 0, l_tap = 0;
    auto advance = [](int& chunk, int& tap) __attribute__((always_inline)) { if (++tap == 9) { tap = 0; ++chunk; } };
    auto step = [&](int s_, auto cur, auto nxt) __attribute__((always_inline)) {
        if (s_ + 2 < nslabs) { load_b(cur, l_chunk, l_tap); advance(l_chunk, l_tap); }
        if (c_tap == 5 && c_chunk + 1 < hg.nchunks) load_a(c_chunk + 1);
        const int tr = c_tap / 3, ts = c_tap - tr * 3;
        mma_slab(sgn * ((tr - 1) * hg.P + (ts - 1)) * LDP);
        __syncthreads();
        if (s_ + 1 < nslabs) {
            store_b(nxt);
            if (c_tap == 8) store_a();
            __syncthreads();
        }
        advance(c_chunk, c_tap);
    };
    load_a(0);
    load_b(S0{}, 0, 0);
    advance(l_chunk, l_tap);
    store_a();
    store_b(S0{});
    __syncthreads();
    if (nslabs > 1) { load_b(S1{}, l_chunk, l_tap); advance(l_chunk, l_tap); }
    for (int s_ = 0; s_ < nslabs; s_ += 2) {
        step(s_, S0{}, S1{});
        if (s_ + 1 < nslabs) step(s_ + 1, S1{}, S0{});
    }
    // ---- epilogue: the 32-column block in the 32x32 C layout, the 16-column blocks in the 16x16 one (row = 4 (lane >> 4) + e, column = lane & 15)
    float* __restrict__ y = p.y + (int64_t)grp * p.Ng;
    {
        const int col = n0 + r32;
        int64_t orow[16];
        bool rok[16];
        float rv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            rok[e] = m0 + row < g.M && col < p.Ng;
            orow[e] = halo_patch_orow(g, hg, m0, row);
            rv[e] = (p.res_mode == 3 && rok[e]) ? p.residual[orow[e] * p.ldy + grp * p.Ng + col] : 0.f;
        }
        const float bv = (p.bias && col < p.Ng) ? p.bias[grp * p.Ng + col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (rok[e]) y[orow[e] * p.ldy + col] = acc32[e] + bv + rv[e];
    }
    {
        const int col = n0 + 32 + i16;
        int64_t orow[8];
        bool rok[8];
        float rv[8];
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = wm * 32 + bb * 16 + 4 * q16 + e;
                rok[bb * 4 + e] = m0 + row < g.M && col < p.Ng;
                orow[bb * 4 + e] = halo_patch_orow(g, hg, m0, row);
                rv[bb * 4 + e] = (p.res_mode == 3 && rok[bb * 4 + e]) ? p.residual[orow[bb * 4 + e] * p.ldy + grp * p.Ng + col] : 0.f;
            }
        const float bv = (p.bias && col < p.Ng) ? p.bias[grp * p.Ng + col] : 0.f;
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (rok[bb * 4 + e]) y[orow[bb * 4 + e] * p.ldy + col] = acc16[bb][e] + bv + rv[bb * 4 + e];
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
// eligible: 3x3, stride 1, pad 1 forward (mode 0) or input gradient (mode 1), 16-byte vector channels
static bool halo_eligible(const GatherGeom& g) {
    return g.om == 1 && g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g.GH == g.OH && g.GW == g.OW && g.M < (1LL << 31);
}
static int halo_chunk(int Cg) { return Cg % 32 == 0 ? 32 : (Cg % 48 == 0 ? 48 : (Cg % 24 == 0 ? 24 : 0)); }

static bool halo_geom(const GatherGeom& g, int BM, int CC, HaloGeom* hg) {
    HaloGeom h;
    memset(&h, 0, sizeof(h));
    h.patch = (g.OH % 8 == 0 && g.OW % 8 == 0) ? 1 : 0;
    h.nchunks = (g.Cg + CC - 1) / CC;
    if (h.patch) {
        h.PX = g.OW / 8; h.PXY = (g.OH / 8) * h.PX;
        magic_div((unsigned)h.PXY, &h.div_pxy_mul, &h.div_pxy_shr);
        magic_div((unsigned)h.PX, &h.div_px_mul, &h.div_px_shr);
        h.P = 12;
        h.nslots = (BM / 64) * 120;
    } else {
        if (CC != 32) return false;
        h.P = g.OW + 2;
        const int pix_rows = (BM - 1) / g.OW + 2;                            // image rows a run of BM pixels can touch
        const int zero_rows = (BM - 1) / (g.OW * g.OH) + 1;                  // zero rows between the images it spans
        h.rows_cov = pix_rows + zero_rows + 2;
        h.nslots = h.rows_cov * h.P;
        if (h.nslots > 288) return false;
        magic_div((unsigned)h.P, &h.div_p_mul, &h.div_p_shr);
        magic_div((unsigned)(g.OH + 1), &h.div_h1_mul, &h.div_h1_shr);
    }
    *hg = h;
    return true;
}

template <int BM, int BN, int WM, int WN, int CC>
static bool launch_halo_cc(const NtParams& p_in, int groups, hipStream_t st) {
    HaloGeom hg;
    if (!halo_geom(p_in.g, BM, CC, &hg)) return false;
    NtParams p = p_in;
    p.col_tiles = (int)ceil_div(p.Ng, BN);
    const int64_t row_tiles8 = ceil_div(ceil_div(p.g.M, BM), 8) * 8;
    const dim3 grid((unsigned)(row_tiles8 * p.col_tiles), 1, groups), block((BM / WM) * (BN / WN) * 64);
    const size_t lds = ((size_t)hg.nslots + BN) * (CC + 4) * sizeof(float);
#define HALO_LAUNCH(PATCH_, AFF_)                                                                                               \
    do {                                                                                                                        \
        auto kern = igemm_nt3x3_kernel<BM, BN, WM, WN, CC, PATCH_, AFF_>;                                                       \
        /* the attribute is per device: set before every launch (no process-wide flag to go stale on a second GPU) */            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
        hipLaunchKernelGGL(kern, grid, block, lds, st, p, hg);                                                                  \
    } while (0)
    if (hg.patch) { if (p.scale) HALO_LAUNCH(true, true); else HALO_LAUNCH(true, false); }
    else if constexpr (CC == 32) { if (p.scale) HALO_LAUNCH(false, true); else HALO_LAUNCH(false, false); }
    else return false;
#undef HALO_LAUNCH
    return true;
}

// the 48-column tile: PATCH geometries (output sizes that are multiples of 8), 32- or 48-channel chunks, no statistics / folded affine
static bool launch_halo_n48(const NtParams& p_in, int groups, hipStream_t st) {
    if (!halo_eligible(p_in.g) || p_in.stats || p_in.scale || p_in.y2 || p_in.a_scale || (p_in.res_mode != 0 && p_in.res_mode != 3)) return false;
    const int cc = halo_chunk(p_in.g.Cg);
    HaloGeom hg;
    if (!(cc == 32 || cc == 48) || !halo_geom(p_in.g, 128, cc, &hg) || !hg.patch) return false;
    NtParams p = p_in;
    p.col_tiles = (int)ceil_div(p.Ng, 48);
    const int64_t row_tiles8 = ceil_div(ceil_div(p.g.M, 128), 8) * 8;
    const dim3 grid((unsigned)(row_tiles8 * p.col_tiles), 1, groups), block(256);
    const size_t lds = ((size_t)hg.nslots + 48) * (cc + 4) * sizeof(float);
    if (cc == 32) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_nt3x3_n48_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(igemm_nt3x3_n48_kernel<32>, grid, block, lds, st, p, hg);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_nt3x3_n48_kernel<48>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(igemm_nt3x3_n48_kernel<48>, grid, block, lds, st, p, hg);
    }
    return true;
}

// can this (geometry, tile height) run on the halo kernel at all?  (used by the tile chooser and by suhpe_conv2d_set_tile)
static bool halo_capable(const GatherGeom& g, int BM) {
    const int cc = halo_chunk(g.Cg);
    HaloGeom hg;
    return halo_eligible(g) && cc != 0 && halo_geom(g, BM, cc, &hg);
}

template <int BM, int BN, int WM, int WN>
static bool launch_halo(const NtParams& p, int groups, hipStream_t st) {
    if (!halo_eligible(p.g)) return false;
    switch (halo_chunk(p.g.Cg)) {
        case 32: return launch_halo_cc<BM, BN, WM, WN, 32>(p, groups, st);
        case 48: return launch_halo_cc<BM, BN, WM, WN, 48>(p, groups, st);
        case 24: return launch_halo_cc<BM, BN, WM, WN, 24>(p, groups, st);
        default: return false;
    }
}
