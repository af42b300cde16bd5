// Operand-resident pointwise GEMM (included by conv_igemm.hip; round 3).
//
//   out[m][j] = sum_k G[m][k] * F[j][k]      1x1 / stride 1 convolutions and nn.Linear with a short reduction (K <= 608)
//
// igemm_nt_kernel gives every 64x64 output tile its own workgroup: with 5-8 slabs of 32 per tile most of a workgroup's life is
// prologue (first loads), epilogue (stores) and the two barriers per slab, and the matrix pipes measured 62 % busy over the launch
// (gemm_lab / lab_pmc.sh: 31360 x 960 x 160).  Here a workgroup stages its BM x K block of the gathered operand in LDS ONCE
// (rows padded by 4 floats: conflict-free ds_read_b128), and from then on every wavefront works alone, without barriers:
//   unit of work = (32-column stripe of the output, TMU*32 rows of the resident block), dealt round-robin to the NW waves
//   (waves w and w+4 share a SIMD, so the SIMDs stay balanced whenever the unit count is a multiple of 4);
//   the stripe's filter rows come straight from global memory in MFMA-operand shape -- lane (j, h) loads F[n0+j][8g+4h .. +3],
//   one 16-byte load per four v_mfma_f32_32x32x2_f32 -- through a two-slab register ring (32 VGPRs), so the filter never touches
//   LDS, the loads of slab s+1 are in flight under the MFMAs of slab s, and nothing synchronises the waves: they drift apart and
//   one wave's stores and waits sit under its SIMD partners' MFMAs (4 waves per SIMD at NW = 16).
// Traffic: the gathered operand is read once per column range instead of once per 64-column tile; the filter is read once per
// (row tile, unit) from L2 (fragment-shaped loads touch 32 lines of 32 bytes each: fine at the fp32 MFMA's 16 flop per operand byte).
// Epilogue per unit, from registers: bias / skip add (res_mode 3) / folded BatchNorm + activation (AFF), buffer stores with one lane
// offset and a scalar row offset per store, and the BatchNorm statistics records (S, Q, k, n) of the unit's rows -- a unit is a
// statistics tile of TMU*32 rows, reduced with one shuffle, no LDS, no barrier.
//
// Replaces cuBLAS / cuDNN behind the 1x1 expand convolutions of torchvision's MBConv (src/networks.py:303), the 256 -> 1280 head
// conv, and TinyViT's qkv / fc1 Linear layers (src/tiny_vit.py:205-213,262-286), forward and input gradient.

template <int BM, int TMU, int NW, bool AFF>
__global__ __launch_bounds__(NW * 64, NW / 4) void gemm_nt_resident(const NtParams p, int col_ranges, int stripes_per_range) {
    constexpr int NT = NW * 64, RU = BM / (32 * TMU), UR = TMU * 32;
    extern __shared__ __attribute__((aligned(16))) float res_As[];        // [BM][Kp + 4]
    const GatherGeom& g = p.g;
    const int Kp = (g.K + 31) & ~31, pitch = Kp + 4;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const int grp = blockIdx.z;
    // workgroups that share a row tile (its column ranges) are multiples of 8 apart: one XCD, one L2 copy of the row tile
    const int b = blockIdx.x, jx = b >> 3;
    const int range = jx % col_ranges;
    const int row_tile = (jx / col_ranges) * 8 + (b & 7);
    const int M = (int)g.M;
    const int m0 = row_tile * BM;
    if (m0 >= M) return;
    const BufRsrc xr = make_rsrc(p.x + (int64_t)grp * g.Cg, p.x_bytes - (int64_t)grp * g.Cg * 4);
    const BufRsrc fr = make_rsrc(p.f + (int64_t)grp * p.Ng * p.ldf, p.f_bytes - (int64_t)grp * p.Ng * p.ldf * 4);
    const int64_t y_bytes = (int64_t)M * p.ldy * 4;
    const BufRsrc yr = make_rsrc(p.y, y_bytes);
    const BufRsrc rr = make_rsrc(p.residual ? p.residual : p.y, y_bytes);

    // ---- stage the resident block: 8 threads cover one 128-byte row chunk; rows beyond M and columns beyond K read zero
    {
        const int lcol = (t & 7) * 4, lrow = t >> 3;
        constexpr int RSTEP = NT / 8;
        for (int kc = 0; kc < Kp; kc += 32) {
#pragma unroll
            for (int i = 0; i < (BM + RSTEP - 1) / RSTEP; ++i) {
                const int row = lrow + i * RSTEP;
                if (BM % RSTEP != 0 && row >= BM) continue;
                const bool ok = (m0 + row < M) & (kc + lcol < g.K);
                float v[4];
                buf_load<4>(xr, ok ? ((unsigned)(m0 + row) * (unsigned)g.ldg + (unsigned)(kc + lcol)) * 4u : kOob, v);
                *reinterpret_cast<float4*>(&res_As[row * pitch + kc + lcol]) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    __syncthreads();

    const int nslab = Kp >> 5;
    const int st0 = range * stripes_per_range;
    int nst = (p.Ng + 31) / 32 - st0;
    if (nst > stripes_per_range) nst = stripes_per_range;
    const int nunits = nst * RU;
    const int mine = nunits > wave ? (nunits - wave + NW - 1) / NW : 0;
    if (mine == 0) return;
    const int total = mine * nslab;

    float bq0[4][4], bq1[4][4];
    int l_u = wave, l_slab = 0;
    auto load_filter = [&](float (&bq)[4][4]) __attribute__((always_inline)) {
        const int col = (st0 + l_u / RU) * 32 + r32;
        const unsigned base = (unsigned)col * (unsigned)p.ldf + (unsigned)(l_slab * 32 + h * 4);
        const bool okc = col < p.Ng;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok = okc & (l_slab * 32 + q * 8 + h * 4 < g.K);
            buf_load<4>(fr, ok ? (base + (unsigned)(q * 8)) * 4u : kOob, bq[q]);
        }
        if (++l_slab == nslab) { l_slab = 0; l_u += NW; }
    };

    floatx16 acc[TMU];
#pragma unroll
    for (int i = 0; i < TMU; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    int c_u = wave, c_slab = 0;
    const float* arow = res_As + r32 * pitch + h * 4;

    auto epilogue = [&]() __attribute__((always_inline)) {
        // C layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 h
        const int col = (st0 + c_u / RU) * 32 + r32;
        const int part = c_u % RU;
        const int rowb = m0 + part * UR;                       // first row of the unit
        const int rowl = rowb + 4 * h;                         // this lane's first row
        const bool okc = col < p.Ng;
        const unsigned voff = okc ? ((unsigned)rowl * (unsigned)p.ldy + (unsigned)(grp * p.Ng + col)) * 4u : kOob;
        const int lim = M - rowl;                              // row offsets ro < lim are inside the tensor
        const unsigned ldy4 = (unsigned)p.ldy * 4u;
        if constexpr (!AFF) {
            const float bv = (p.bias && okc) ? p.bias[grp * p.Ng + col] : 0.f;
            if (p.res_mode == 3) {                               // skip-connection gradient added to the input gradient
                // eight loads in flight, then their eight stores (a full set of 32 skip values would cost the kernel a wave per SIMD)
#pragma unroll
                for (int i = 0; i < TMU; ++i)
#pragma unroll
                    for (int e0 = 0; e0 < 16; e0 += 8) {
                        float rv[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int ro = i * 32 + ((e0 + e) & 3) + 8 * ((e0 + e) >> 2);
                            rv[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, ro < lim ? voff : kOob, (unsigned)ro * ldy4, 0));
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int ro = i * 32 + ((e0 + e) & 3) + 8 * ((e0 + e) >> 2);
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][e0 + e] + bv + rv[e]), yr, ro < lim ? voff : kOob, (unsigned)ro * ldy4, 0);
                        }
                    }
            } else {
#pragma unroll
                for (int i = 0; i < TMU; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int ro = i * 32 + (e & 3) + 8 * (e >> 2);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][e] + bv), yr, ro < lim ? voff : kOob, (unsigned)ro * ldy4, 0);
                    }
            }
        } else {
            const float sc = okc ? p.scale[grp * p.Ng + col] : 0.f, sh = okc ? p.shift[grp * p.Ng + col] : 0.f;
#pragma unroll
            for (int i = 0; i < TMU; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ro = i * 32 + (e & 3) + 8 * (e >> 2);
                    float v = acc[i][e] * sc + sh;
                    const float rv = p.res_mode ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, ro < lim ? voff : kOob, (unsigned)ro * ldy4, 0)) : 0.f;
                    if (p.res_mode == 1) v += rv;
                    v = act_apply(v, p.act);
                    if (p.res_mode == 2) v += rv;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, ro < lim ? voff : kOob, (unsigned)ro * ldy4, 0);
                }
        }
        // ---- BatchNorm statistics record(s) of the unit: (S, Q, k, n) per column over its n valid rows, shifted by k = the
        // unit's first-row value (the finaliser re-centres in fp64); a unit that contains the segment boundary emits its
        // first-segment rows to `stats` and the rest to `stats_extra`
        if (!AFF && p.stats && rowb < M) {
            const float k = __shfl(acc[0][0], r32, 64);
            const int nrows = M - rowb < UR ? M - rowb : UR;
            const int64_t cut64 = p.seg_row - rowb;
            const bool two = p.stats_extra && cut64 > 0 && cut64 < nrows;
            const int cut = two ? (int)cut64 : nrows;           // rows [0, cut) -> record 1, [cut, nrows) -> record 2
            float4* rec = reinterpret_cast<float4*>(p.stats) + (int64_t)(row_tile * RU + part) * p.stat_cout + grp * p.Ng + col;
            // sums over the unit's rows [lo, hi)
            auto emit = [&](int lo, int hi, float4* dst) __attribute__((always_inline)) {
                float a = 0.f, q = 0.f;
                if (lo <= 0 && hi >= UR) {                     // every row of the unit counts: no per-element range tests (the vector ALU is the matrix pipe)
#pragma unroll
                    for (int i = 0; i < TMU; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) { const float v = acc[i][e] - k; a += v; q += v * v; }
                } else {
#pragma unroll
                    for (int i = 0; i < TMU; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int lr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                            const float v = (lr >= lo && lr < hi) ? acc[i][e] - k : 0.f;
                            a += v;
                            q += v * v;
                        }
                }
                a += __shfl_xor(a, 32, 64);
                q += __shfl_xor(q, 32, 64);
                if (h == 0 && okc) *dst = make_float4(a, q, k, (float)(hi - lo));
            };
            emit(0, cut, rec);
            if (two) emit(cut, nrows, reinterpret_cast<float4*>(p.stats_extra) + grp * p.Ng + col);
        }
    };

    auto compute = [&](const float (&bq)[4][4]) __attribute__((always_inline)) {
        const float* ap = arow + (c_u % RU) * UR * pitch + c_slab * 32;
        float4 af[2][TMU];
#pragma unroll
        for (int i = 0; i < TMU; ++i) af[0][i] = *reinterpret_cast<const float4*>(ap + i * 32 * pitch);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cur = q & 1;
            if (q + 1 < 4) {
#pragma unroll
                for (int i = 0; i < TMU; ++i) af[cur ^ 1][i] = *reinterpret_cast<const float4*>(ap + i * 32 * pitch + (q + 1) * 8);
            }
#pragma unroll
            for (int i = 0; i < TMU; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].x, bq[q][0], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TMU; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].y, bq[q][1], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TMU; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].z, bq[q][2], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TMU; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i].w, bq[q][3], acc[i], 0, 0, 0);
        }
        if (++c_slab == nslab) {
            epilogue();
#pragma unroll
            for (int i = 0; i < TMU; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
            c_slab = 0;
            c_u += NW;
        }
    };
    load_filter(bq0);
    for (int it = 0; it < total; it += 2) {
        if (it + 1 < total) load_filter(bq1);
        compute(bq0);
        if (it + 1 >= total) break;
        if (it + 2 < total) load_filter(bq0);
        compute(bq1);
    }
}

// ---- host side
constexpr int kResidentLdsMax = 160 * 1024;
static int resident_lds_bytes(int K, int BM) { return BM * (((K + 31) & ~31) + 4) * 4; }

// plain pointwise GEMM (rows < 2^31, 16-byte rows) whose BM x K block fits the LDS, no loader transform
static bool resident_ok(const NtParams& p, bool vec4, int BM) {
    return vec4 && nt_mode(p.g) == 1 && !p.a_scale && p.g.M < (1LL << 31) && p.g.K % 4 == 0 && p.ldf % 4 == 0 &&
           resident_lds_bytes(p.g.K, BM) <= kResidentLdsMax;
}

// column ranges per row tile: enough workgroups to fill the chip once (BM x K of LDS and NW waves each), no more -- every extra
// range stages the row tile again
static int resident_ranges(int64_t M, int Ng, int K, int BM, int NW) {
    const int lds = resident_lds_bytes(K, BM);
    int per_cu = kResidentLdsMax / lds;
    if (per_cu > 16 / NW) per_cu = 16 / NW;
    if (per_cu < 1) per_cu = 1;
    const int64_t row_tiles = ceil_div(M, BM), target = 256 * per_cu;
    const int stripes = (int)ceil_div(Ng, 32);
    int64_t r = (target + row_tiles / 2) / row_tiles;
    if (r < 1) r = 1;
    if (r > stripes) r = stripes;
    return (int)r;
}

template <int BM, int TMU, int NW>
static void launch_resident(const NtParams& p, int groups, hipStream_t st) {
    const int ranges = resident_ranges(p.g.M, p.Ng, p.g.K, BM, NW);
    const int stripes = (int)ceil_div(p.Ng, 32), spr = (int)ceil_div(stripes, ranges);
    const int64_t row_tiles8 = ceil_div(ceil_div(p.g.M, BM), 8) * 8;
    const dim3 grid((unsigned)(row_tiles8 * ranges), 1, groups), block(NW * 64);
    const size_t lds = (size_t)resident_lds_bytes(p.g.K, BM);
    // the attribute is per device and per kernel: set it before every launch (cheap; no static flag to go stale on a second GPU)
    if (p.scale) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_resident<BM, TMU, NW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kResidentLdsMax);
        hipLaunchKernelGGL((gemm_nt_resident<BM, TMU, NW, true>), grid, block, lds, st, p, ranges, spr);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_resident<BM, TMU, NW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kResidentLdsMax);
        hipLaunchKernelGGL((gemm_nt_resident<BM, TMU, NW, false>), grid, block, lds, st, p, ranges, spr);
    }
}
