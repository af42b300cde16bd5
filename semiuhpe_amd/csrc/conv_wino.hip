// 3x3 / stride 1 / pad 1 convolution (forward and input gradient) by Winograd's minimal filtering F(2x2, 3x3) on the fp32 matrix cores
// (its own translation unit; conv_igemm.hip's dispatch calls launch_wino).
//
// Every other dense kernel of this library spends 9 multiply-adds per (output pixel, input channel, output channel) of a 3x3 conv.
// F(2x2, 3x3) computes a 2x2 output patch from a 4x4 input patch with 16 multiplies per channel pair instead of 36: 2.25x less matrix
// work, which is the one resource these layers are bound by (profiles/README.md: the 3x3 GEMMs run at 105-118 TF/s of fp32 MFMA and are
// 44 % of EfficientNetV2-S's FLOPs, nearly all of RepVGG's).  cuDNN takes the same route for the reference's 3x3 convolutions
// (src/networks.py:303 FusedMBConv via torchvision, src/repvgg.py:28-32 rbr_dense); the arithmetic differs from the direct sum by
// rounding only (the transforms add / subtract / halve; measured error against float64 is BELOW the direct kernels': fewer
// accumulation steps).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A          d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs, (.) summed over input channels
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// so per transform position xi = (i, jx) of the 4x4 grid there is one GEMM  M_xi[cout][tile] = sum_c U_xi[cout][c] V_xi[tile][c]
// over the input channels, 16 of them sharing their rows and columns.  (Row i = 3 of both U and V is stored NEGATED -- the product
// does not notice and V's vertical pass becomes one multiply-add per value.)  One launch does everything: a PERSISTENT workgroup
// (8 wavefronts, one per CU: 2 x (32 + 32) KB of LDS) walks work items = 64 tiles (2x2 output patches in (n, ty, tx) order = 256 output
// pixels) x 64 output channels, and inside an item the input channels in chunks of 8:
//
//   stage   thread (patch row i = t&3, channel vector v = (t>>2)&1, tile = t>>3) holds one row of a 4x4 input patch (four 16-byte
//           descriptor loads, zero outside the image): the horizontal half of B^T d B is in-thread, the vertical half runs across
//           the four lanes of a quad with a DPP quad_perm operand -- no LDS pass for the raw pixels, every lane the same work -- and
//           the 16 values go to the LDS image V[xi][tile][8 ch] (16-byte stores).  The filter side comes pre-transformed
//           (suhpe_conv2d_wino_filter: one small launch per layer and step, the weights move every step) in exactly the LDS image
//           order, so a chunk's U image is one contiguous 32 KB block copied global -> LDS by LDS-DMA (global_load_lds_dwordx4:
//           no registers, no VALU, no ds_write).  Without a pre-transformed filter (plain suhpe_conv2d_fwd / _dgrad with the tile
//           pinned) the same threads load a filter row each and run G g G^T the same way (three quad broadcasts per value);
//   MFMA    wavefront (tile block tb = w&3, cout half ch = w>>2) owns 16 tiles x 32 couts for ALL 16 positions: 32 accumulators of
//           v_mfma_f32_16x16x4_f32 (A = U rows, B = V rows; one ds_read_b64 hands a lane two k-steps), 64 MFMAs per chunk, so the
//           output transform A^T M A happens in registers: a lane ends up with the 2x2 pixels of one tile x 4 consecutive couts
//           and stores them as 16-byte vectors;
//   the two LDS buffers alternate per chunk with ONE barrier per chunk; the next chunk's loads (DMA + patch rows) are issued right
//   after the barrier and land under the MFMAs, its transform sits in the second half of the MFMA block, and the stream of chunks runs
//   ACROSS item boundaries: the first chunk of the next item is staged while the current item's last chunk is multiplied, so an
//   item's prologue costs nothing and only its epilogue (output transform + stores) interrupts the matrix pipe.
//
// LDS images are unpadded; bank conflicts are avoided by two XORs of the address: row r of position xi sits at row r ^ (xi >> 2), and
// the two channel vectors of rows 8-15 (mod 16) are swapped.  Reads (ds_read_b64, 32-lane groups: 16 rows x 2 channel pairs) then
// cover all 64 banks once, writes (ds_write_b128, 8-lane groups: one row index, 2 vectors, 4 positions) all 32.
//
// Epilogues: bias, skip-connection add (res_mode 3), BatchNorm statistics records per 64-tile block (256 rows per record, the same
// (S, Q, k, n) format and segment rule as igemm_nt_kernel's).  No folded-affine (eval) form: eval passes keep the direct kernels, so the
// eval-mode parity figures are untouched.
#include "conv_params.h"
#include <cstdlib>
#include <type_traits>

namespace suhpe {

typedef float floatx4w __attribute__((ext_vector_type(4)));

struct WinoGeom {
    int TX, TPI;                 // tiles per image row, tiles per image
    unsigned div_tpi_mul, div_tpi_shr, div_tx_mul, div_tx_shr;
    int MT;                      // tiles of the batch = N * TPI
    int nchunks;                 // ceil(Cg / 8)
    int flip;                    // input gradient: the filter's taps are mirrored (dx = dy correlated with the 180-degree rotated filter)
    int seg_tile;                // tiles of the first batch segment (0: a single segment)
    int row_tiles;               // ceil(MT / 64)
    int fw_group_stride;         // floats of one group's pre-transformed filter = col_tiles * nchunks * WN_IMG
};

constexpr int WN_XI = 512;               // floats per transform position of one LDS image: 64 rows x 8 channels
constexpr int WN_IMG = 16 * WN_XI;       // floats per image (V or U) = 32 KB
constexpr int WN_SCRATCH = 64 + 4 * 64 * 4;   // statistics epilogue: k[64] + partial sums [4 tile blocks][64 couts][4]

template <int CTRL> __device__ __forceinline__ float quad_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, false));
}

// float offset of (row, channel vector v, position xi) inside an LDS image (see the header comment)
__host__ __device__ __forceinline__ int wino_img_off(int xi, int row, int v) {
    return xi * WN_XI + ((row ^ (xi >> 2)) << 3) + ((v ^ ((row >> 3) & 1)) << 2);
}

template <bool PRE>
__global__ __launch_bounds__(512, 2) void wino_nt_kernel(const NtParams p, const WinoGeom wg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];       // [buffer][V image | U image], then the statistics scratch
    const GatherGeom& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int grp = blockIdx.z;
    const BufRsrc xr = make_rsrc(p.x + (int64_t)grp * g.Cg, p.x_bytes - (int64_t)grp * g.Cg * 4);
    const BufRsrc fr = make_rsrc(p.f + (int64_t)grp * p.Ng * p.ldf, p.f_bytes - (int64_t)grp * p.Ng * p.ldf * 4);

    // ---- work items of this workgroup: item = blockIdx.x + k * gridDim.x (gridDim.x is a multiple of 8, so xcd = item & 7 is fixed);
    // an XCD takes one contiguous run of tile blocks, the cout blocks of a tile block consecutively (they re-read the same pixels from L2)
    const int xcd = blockIdx.x & 7;
    const int rt_q = wg.row_tiles >> 3, rt_rem = wg.row_tiles & 7;
    const int rt_first = xcd * rt_q + (xcd < rt_rem ? xcd : rt_rem);
    const int xcd_items = (rt_q + (xcd < rt_rem ? 1 : 0)) * p.col_tiles;
    auto item_valid = [&](int item) __attribute__((always_inline)) { return (item >> 3) < xcd_items; };

    // ---- staging role of this thread
    const int si = t & 3, sv = (t >> 2) & 1, stl = t >> 3;
    const float sxv = si == 1 ? 1.f : -1.f;                    // V's vertical pass: out_i = own + sxv * quad_perm[2,2,1,1](own)
    //   i = 0: H0 - H2, 1: H1 + H2, 2: H2 - H1, 3: H3 - H1 (= -(H1 - H3): row 3 negated)
    const float ua = si == 0 ? 1.f : (si == 3 ? 0.f : 0.5f), ub = si == 1 ? 0.5f : (si == 2 ? -0.5f : 0.f), uc = si == 3 ? -1.f : (si == 0 ? 0.f : 0.5f);
    const int wbase = wino_img_off(si * 4, stl, sv);            // + jx * WN_XI

    // loader cursor (one chunk ahead of the MFMAs) and the item under the MFMAs
    int l_item = blockIdx.x, l_chunk = 0;
    unsigned voff[4], uoff[3];
    const float* ufw = nullptr;                                 // pre-transformed filter image of (l_item's cout block, chunk 0) + this thread's slice
    int l_tile0 = 0, l_n0 = 0, l_row_tile = 0;
    auto setup_item = [&]() __attribute__((always_inline)) {
        const int jx = l_item >> 3;
        const int col_tile = jx % p.col_tiles;
        l_row_tile = rt_first + jx / p.col_tiles;
        l_tile0 = l_row_tile * 64; l_n0 = col_tile * 64;
        const int T = l_tile0 + stl;
        const bool tv = T < wg.MT;
        const unsigned Tn = (unsigned)(tv ? T : 0);
        const unsigned n = fast_div(Tn, wg.div_tpi_mul, wg.div_tpi_shr);
        const unsigned rem = Tn - n * (unsigned)wg.TPI;
        const unsigned ty = fast_div(rem, wg.div_tx_mul, wg.div_tx_shr), tx = rem - ty * (unsigned)wg.TX;
        const int iy = 2 * (int)ty - 1 + si, ix0 = 2 * (int)tx - 1;
        const bool rowok = tv && (unsigned)iy < (unsigned)g.GH;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool ok = rowok && (unsigned)(ix0 + c) < (unsigned)g.GW;
            voff[c] = ok ? ((unsigned)(((int)n * g.GH + iy) * g.GW + ix0 + c) * (unsigned)g.ldg + 4u * sv) * 4u : kOob;
        }
        if constexpr (PRE) {
            ufw = p.fw + (int64_t)grp * wg.fw_group_stride + (int64_t)col_tile * wg.nchunks * WN_IMG + wave * 1024 + lane * 4;
        } else {
            const int co = l_n0 + stl;
            const bool uok = co < p.Ng && si < 3;
            const int r = wg.flip ? 2 - si : si;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int tap = r * 3 + (wg.flip ? 2 - s : s);
                uoff[s] = uok ? ((unsigned)co * (unsigned)p.ldf + (unsigned)(tap * g.Cg) + 4u * sv) * 4u : kOob;
            }
        }
    };

    float vreg[4][4], ureg[PRE ? 1 : 3][4];
    auto issue_loads = [&](int buf) __attribute__((always_inline)) {
        if constexpr (PRE) {
            // this wavefront's four 1 KiB pieces of the chunk's U image, global -> LDS (LDS address = M0 + instruction offset + 16 * lane; the
            // instruction offset also advances the global address).  Inline asm on purpose: behind the builtin the compiler cannot tell the DMA's
            // LDS destination (the OTHER buffer) from the fragment reads that follow and parks an s_waitcnt vmcnt(0) in front of the MFMAs; this way
            // the DMA is invisible to its wait counting, and the wait that retires it is the one for the patch rows issued right behind it (loads
            // return in order), backed by the explicit vmcnt(0) in front of the chunk's barrier.
            const unsigned lds_dst = __builtin_amdgcn_readfirstlane(
                (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(smem + buf * 2 * WN_IMG + WN_IMG + wave * 1024));
            const float* src = ufw + (int64_t)l_chunk * WN_IMG;
            unsigned keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b32 m0, %2\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %1, off\n\t"
                "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                "global_load_lds_dwordx4 %1, off offset:3072\n\t"
                "s_mov_b32 m0, %0"
                : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
        }
        const bool cok = l_chunk * 8 + 4 * sv < g.Cg;
        const unsigned cb = (unsigned)l_chunk * 32u;
#pragma unroll
        for (int c = 0; c < 4; ++c) buf_load<4>(xr, cok ? voff[c] + cb : kOob, vreg[c]);
        if constexpr (!PRE) {
#pragma unroll
            for (int s = 0; s < 3; ++s) buf_load<4>(fr, cok ? uoff[s] + cb : kOob, ureg[s]);
        }
    };
    float ov[4][4], ou[PRE ? 1 : 4][4];                          // transformed values of the staged chunk: [jx][channel]
    auto transform = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h0 = vreg[0][e] - vreg[2][e], h1 = vreg[1][e] + vreg[2][e], h2 = vreg[2][e] - vreg[1][e], h3 = vreg[1][e] - vreg[3][e];
            ov[0][e] = fmaf(sxv, quad_dpp<0x5A>(h0), h0);
            ov[1][e] = fmaf(sxv, quad_dpp<0x5A>(h1), h1);
            ov[2][e] = fmaf(sxv, quad_dpp<0x5A>(h2), h2);
            ov[3][e] = fmaf(sxv, quad_dpp<0x5A>(h3), h3);
        }
        if constexpr (!PRE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s02 = ureg[0][e] + ureg[2][e];
                const float hu[4] = {ureg[0][e], 0.5f * (s02 + ureg[1][e]), 0.5f * (s02 - ureg[1][e]), ureg[2][e]};
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    const float a0 = quad_dpp<0x00>(hu[jx]), a1 = quad_dpp<0x55>(hu[jx]), a2 = quad_dpp<0xAA>(hu[jx]);
                    ou[jx][e] = fmaf(uc, a2, fmaf(ub, a1, ua * a0));
                }
            }
        }
    };
    auto write_staged = [&](int buf) __attribute__((always_inline)) {
        float* Vs = smem + buf * 2 * WN_IMG;
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            *reinterpret_cast<float4*>(&Vs[wbase + jx * WN_XI]) = make_float4(ov[jx][0], ov[jx][1], ov[jx][2], ov[jx][3]);
            if constexpr (!PRE) *reinterpret_cast<float4*>(&Vs[WN_IMG + wbase + jx * WN_XI]) = make_float4(ou[jx][0], ou[jx][1], ou[jx][2], ou[jx][3]);
        }
    };
    auto advance_loader = [&]() __attribute__((always_inline)) {
        if (++l_chunk == wg.nchunks) {
            l_chunk = 0;
            l_item += gridDim.x;
            if (item_valid(l_item)) setup_item();
        }
    };

    // ---- MFMA role: tile block tb (16 tiles), cout half ch (2 blocks of 16); lane = (row j of the block, k-group kg)
    const int tb = wave & 3, ch = wave >> 2, j16 = lane & 15, kg = lane >> 4;
    int vrd[4], urd[2][4];
    {
        const int vpos = (((kg >> 1) ^ ((j16 >> 3) & 1)) << 2) + ((kg & 1) << 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vrd[i] = (((16 * tb + j16) ^ i) << 3) + vpos;
            urd[0][i] = (((32 * ch + j16) ^ i) << 3) + vpos;
            urd[1][i] = (((32 * ch + 16 + j16) ^ i) << 3) + vpos;
        }
    }
    floatx4w acc[16][2];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[xi][bb][e] = 0.f;
    };
    // the chunk's 64 MFMAs; the staged chunk's transform (registers only) sits behind position 7, under the second half
    auto mma_chunk = [&](int buf, auto stage_t) __attribute__((always_inline)) {
        constexpr bool stage = decltype(stage_t)::value;
        const float* Vs = smem + buf * 2 * WN_IMG;
        const float* Us = Vs + WN_IMG;
        float2 vf[2], uf[2][2];
        auto read_frags = [&](int xi, int set) __attribute__((always_inline)) {
            vf[set] = *reinterpret_cast<const float2*>(&Vs[xi * WN_XI + vrd[xi >> 2]]);
            uf[set][0] = *reinterpret_cast<const float2*>(&Us[xi * WN_XI + urd[0][xi >> 2]]);
            uf[set][1] = *reinterpret_cast<const float2*>(&Us[xi * WN_XI + urd[1][xi >> 2]]);
        };
        read_frags(0, 0);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const int cur = xi & 1;
            if (xi + 1 < 16) read_frags(xi + 1, cur ^ 1);
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][0].x, vf[cur].x, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][1].x, vf[cur].x, acc[xi][1], 0, 0, 0);
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][0].y, vf[cur].y, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][1].y, vf[cur].y, acc[xi][1], 0, 0, 0);
            if constexpr (stage) {
                if (xi == 7) {
                    __builtin_amdgcn_sched_barrier(0);           // the transform (and the wait for its loads) stays behind the first 32 MFMAs
                    transform();
                }
            }
        }
    };

    // ---- epilogue of the item under the MFMAs: output transform in registers, stores, statistics
    auto epilogue = [&](int tile0, int n0, int row_tile) __attribute__((always_inline)) {
        float out[4][2][4];                                      // [pixel (py, px) = (q >> 1, q & 1)][cout block][e]
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t0[4], t1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float m0 = acc[4 * i][bb][e], m1 = acc[4 * i + 1][bb][e], m2 = acc[4 * i + 2][bb][e], m3 = acc[4 * i + 3][bb][e];
                    t0[i] = m0 + m1 + m2;
                    t1[i] = m1 - m2 - m3;
                }
                out[0][bb][e] = t0[0] + t0[1] + t0[2];
                out[1][bb][e] = t1[0] + t1[1] + t1[2];
                out[2][bb][e] = t0[1] - t0[2] - t0[3];
                out[3][bb][e] = t1[1] - t1[2] - t1[3];
            }
        const int Tj = tile0 + 16 * tb + j16;
        const bool tvalid = Tj < wg.MT;
        int64_t pix0;
        {
            const unsigned Tn = (unsigned)(tvalid ? Tj : 0);
            const unsigned n = fast_div(Tn, wg.div_tpi_mul, wg.div_tpi_shr);
            const unsigned rem = Tn - n * (unsigned)wg.TPI;
            const unsigned ty = fast_div(rem, wg.div_tx_mul, wg.div_tx_shr), tx = rem - ty * (unsigned)wg.TX;
            pix0 = ((int64_t)n * g.OH + 2 * ty) * g.OW + 2 * tx;
        }
        const int colb[2] = {n0 + 32 * ch + 4 * kg, n0 + 32 * ch + 16 + 4 * kg};
        float* __restrict__ y = p.y + (int64_t)grp * p.Ng;
        {
            float rv[4][2][4];
            if (p.res_mode == 3) {                               // the skip operand, every load ahead of the stores (it may alias y for the compiler)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const int64_t pix = pix0 + (q >> 1) * g.OW + (q & 1);
                        const bool ok = tvalid && colb[bb] < p.Ng;
                        const float4 r4 = ok ? *reinterpret_cast<const float4*>(&p.residual[pix * p.ldy + grp * p.Ng + colb[bb]]) : make_float4(0.f, 0.f, 0.f, 0.f);
                        rv[q][bb][0] = r4.x; rv[q][bb][1] = r4.y; rv[q][bb][2] = r4.z; rv[q][bb][3] = r4.w;
                    }
            }
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                if (!(tvalid && colb[bb] < p.Ng)) continue;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias) bv = *reinterpret_cast<const float4*>(&p.bias[grp * p.Ng + colb[bb]]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t pix = pix0 + (q >> 1) * g.OW + (q & 1);
                    float4 o = make_float4(out[q][bb][0] + bv.x, out[q][bb][1] + bv.y, out[q][bb][2] + bv.z, out[q][bb][3] + bv.w);
                    if (p.res_mode == 3) { o.x += rv[q][bb][0]; o.y += rv[q][bb][1]; o.z += rv[q][bb][2]; o.w += rv[q][bb][3]; }
                    *reinterpret_cast<float4*>(&y[pix * p.ldy + colb[bb]]) = o;
                }
            }
        }
        // BatchNorm statistics: one (S, Q, k, n) record per (tile block, cout), sums over the block's valid pixels shifted by k = the block's
        // first output; the block that contains the segment boundary emits its first-segment tiles to `stats`, the rest to `stats_extra`
        if (p.stats) {
            float* kk = smem + 4 * WN_IMG;                       // [64]    (its own region: the image buffers hold the next item's first chunk)
            float* red = kk + 64;                                // [4 tile blocks][64 couts][4]
            if (tb == 0 && j16 == 0) {
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) kk[32 * ch + 16 * bb + 4 * kg + e] = out[0][bb][e];
            }
            __syncthreads();
            const int cut = wg.seg_tile > tile0 && wg.seg_tile < tile0 + 64 && p.stats_extra ? wg.seg_tile : 0x7fffffff;    // first tile of the second part
            const bool first = tvalid && Tj < cut, second = tvalid && Tj >= cut;
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float k = kk[32 * ch + 16 * bb + 4 * kg + e];
                    float a = 0.f, qq = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float v = out[q][bb][e] - k; a += v; qq += v * v; }
                    float a1 = first ? a : 0.f, q1 = first ? qq : 0.f, a2 = second ? a : 0.f, q2 = second ? qq : 0.f;
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        a1 += __shfl_xor(a1, o, 64); q1 += __shfl_xor(q1, o, 64);
                        if (cut != 0x7fffffff) { a2 += __shfl_xor(a2, o, 64); q2 += __shfl_xor(q2, o, 64); }
                    }
                    if (j16 == 0) *reinterpret_cast<float4*>(&red[(tb * 64 + 32 * ch + 16 * bb + 4 * kg + e) * 4]) = make_float4(a1, q1, a2, q2);
                }
            __syncthreads();
            if (t < 64 && n0 + t < p.Ng) {
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float4 r4 = *reinterpret_cast<const float4*>(&red[(w * 64 + t) * 4]);
                    s.x += r4.x; s.y += r4.y; s.z += r4.z; s.w += r4.w;
                }
                const int nvalid = wg.MT - tile0 < 64 ? wg.MT - tile0 : 64;
                const int n1 = cut == 0x7fffffff ? nvalid : cut - tile0;
                float4* rec = reinterpret_cast<float4*>(p.stats) + (int64_t)row_tile * p.stat_cout;
                rec[grp * p.Ng + n0 + t] = make_float4(s.x, s.y, kk[t], 4.f * (float)n1);
                if (cut != 0x7fffffff) reinterpret_cast<float4*>(p.stats_extra)[grp * p.Ng + n0 + t] = make_float4(s.z, s.w, kk[t], 4.f * (float)(nvalid - n1));
            }
            // (the next item's statistics write kk / red only after a later __syncthreads of the chunk loop: no hazard with these reads)
        }
    };

    if (!item_valid(l_item)) return;
    setup_item();
    int c_tile0 = l_tile0, c_n0 = l_n0, c_row_tile = l_row_tile, c_chunk = 0;
    issue_loads(0);
    transform();
    write_staged(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    advance_loader();
    zero_acc();
    int buf = 0;
    auto finish_chunk = [&]() __attribute__((always_inline)) {
        if (++c_chunk == wg.nchunks) {
            epilogue(c_tile0, c_n0, c_row_tile);
            zero_acc();
            c_chunk = 0;
            c_tile0 = l_tile0; c_n0 = l_n0; c_row_tile = l_row_tile;      // (the loader is inside the next item by now)
        }
    };
    while (item_valid(l_item)) {                                 // the loader has a chunk to stage beside this one's MFMAs
        issue_loads(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);                       // (left alone the scheduler sinks the patch-row loads down to their first use)
        mma_chunk(buf, std::integral_constant<bool, PRE>{});     // (the self-transforming form keeps its longer transform out of the MFMA block: registers)
        if constexpr (!PRE) transform();
        write_staged(buf ^ 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        finish_chunk();
        advance_loader();
        buf ^= 1;
    }
    mma_chunk(buf, std::false_type{});                           // the last chunk of the last item
    finish_chunk();
}

// ---- filter transform: U = G g G^T (row 3 negated) of every (output row R, reduction channel k) of a group, written in the LDS image order
// of the main kernel: fw[group][R / 64][k / 8][image of 16 x 512 floats].  pass 0: g[r][s] = w[(group, R)][r][s][k] (the filter as stored,
// [cout][r][s][cin / groups]); pass 1 (input gradient): R = input channel, k = output channel of the group, g[r][s] = w[(group, k)][2 - r][2 - s][R].
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ fw, int groups, int NR, int NK, int ng_w, int cg_w,
                                                          int pass, int rblocks, int kchunks) {
    const int64_t total = (int64_t)groups * rblocks * 64 * kchunks * 8;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % (kchunks * 8));
        const int R = (int)((idx / (kchunks * 8)) % (rblocks * 64));
        const int grp = (int)(idx / ((int64_t)kchunks * 8 * rblocks * 64));
        float gq[3][3];
        const bool ok = R < NR && k < NK;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                float v = 0.f;
                if (ok) {
                    if (pass == 0) v = w[(((int64_t)grp * ng_w + R) * 9 + r * 3 + s) * cg_w + k];
                    else v = w[(((int64_t)grp * ng_w + k) * 9 + (2 - r) * 3 + (2 - s)) * cg_w + R];
                }
                gq[r][s] = v;
            }
        float hg[4][3];                                          // G g
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const float s02 = gq[0][s] + gq[2][s];
            hg[0][s] = gq[0][s]; hg[1][s] = 0.5f * (s02 + gq[1][s]); hg[2][s] = 0.5f * (s02 - gq[1][s]); hg[3][s] = -gq[2][s];
        }
        float* img = fw + (((int64_t)grp * rblocks + R / 64) * kchunks + k / 8) * WN_IMG;
        const int row = R & 63, v4 = (k & 7) >> 2, e = k & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s02 = hg[i][0] + hg[i][2];
            const float u[4] = {hg[i][0], 0.5f * (s02 + hg[i][1]), 0.5f * (s02 - hg[i][1]), hg[i][2]};
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) img[wino_img_off(4 * i + jx, row, v4) + e] = u[jx];
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
static const bool kWinoEnabled = !(getenv("SUHPE_WINOGRAD") && atoi(getenv("SUHPE_WINOGRAD")) == 0);   // SUHPE_WINOGRAD=0: the tile is never eligible

// geometry: 3x3 / stride 1 / pad 1 forward (mode 0) or input gradient (mode 1) on even output sizes, 16-byte channel vectors on both sides
bool wino_geom_ok(const GatherGeom& g, int Ng) {
    return kWinoEnabled && g.om == 1 && g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g.GH == g.OH && g.GW == g.OW && g.OH % 2 == 0 && g.OW % 2 == 0 &&
           g.M < (1LL << 31) && g.Cg % 4 == 0 && Ng % 4 == 0 && g.ldg % 4 == 0;
}
// epilogue forms the kernel has: bias, skip add, statistics records
static bool wino_params_ok(const NtParams& p) {
    return wino_geom_ok(p.g, p.Ng) && !p.scale && !p.y2 && !p.a_scale && !p.o_rowscale && (p.res_mode == 0 || p.res_mode == 3) && p.ldy % 4 == 0 &&
           aligned16(p.y) && aligned16(p.x) && (p.fw ? aligned16(p.fw) : aligned16(p.f)) && (!p.residual || aligned16(p.residual)) && (!p.bias || aligned16(p.bias));
}

int64_t wino_filter_floats(int groups, int NR, int NK) { return (int64_t)groups * ceil_div(NR, 64) * ceil_div(NK, 8) * WN_IMG; }

void launch_wino_filter(const float* w, float* fw, int groups, int NR, int NK, int ng_w, int cg_w, int pass, hipStream_t st) {
    const int rblocks = (int)ceil_div(NR, 64), kchunks = (int)ceil_div(NK, 8);
    const int64_t total = (int64_t)groups * rblocks * 64 * kchunks * 8;
    hipLaunchKernelGGL(wino_filter_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, w, fw, groups, NR, NK, ng_w, cg_w, pass, rblocks, kchunks);
}

bool launch_wino(const NtParams& p_in, int groups, hipStream_t st) {
    if (!wino_params_ok(p_in)) return false;
    NtParams p = p_in;
    const GatherGeom& g = p.g;
    WinoGeom wg;
    memset(&wg, 0, sizeof(wg));
    wg.TX = g.OW / 2; wg.TPI = (g.OH / 2) * wg.TX;
    magic_div((unsigned)wg.TPI, &wg.div_tpi_mul, &wg.div_tpi_shr);
    magic_div((unsigned)wg.TX, &wg.div_tx_mul, &wg.div_tx_shr);
    wg.MT = g.N * wg.TPI;
    wg.nchunks = (g.Cg + 7) / 8;
    wg.flip = g.mode == 0 ? 0 : 1;
    wg.seg_tile = (p.stats && p.seg_row > 0) ? (int)(p.seg_row / 4) : 0;
    wg.row_tiles = (int)ceil_div(wg.MT, 64);
    p.col_tiles = (int)ceil_div(p.Ng, 64);
    wg.fw_group_stride = p.col_tiles * wg.nchunks * WN_IMG;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (cus <= 0) cus = 256;
    }
    const int64_t items8 = ceil_div((int64_t)wg.row_tiles, 8) * 8 * p.col_tiles;       // item space (every XCD's list padded to the longest)
    int per_group = cus / groups / 8 * 8;
    if (per_group < 8) per_group = 8;
    const int gx = (int)(items8 < per_group ? items8 : per_group);
    const dim3 grid((unsigned)gx, 1, groups), block(512);
    const size_t lds = ((size_t)4 * WN_IMG + WN_SCRATCH) * sizeof(float);
    if (p.fw) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_nt_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(wino_nt_kernel<true>, grid, block, lds, st, p, wg);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_nt_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(wino_nt_kernel<false>, grid, block, lds, st, p, wg);
    }
    return true;
}

}  // namespace suhpe
