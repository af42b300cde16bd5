// Winograd F(2x2, 3x3) / F(3x3, 2x2) for 3x3 / stride 1 / pad 1 convolutions with FEW channels on both sides (forward / input gradient: 16 or 24 in, at
// most 32 out; filter gradient: at most 32 on both sides): the 24 -> 24 stage of EfficientNetV2-S at 112 x 112 (torchvision FusedMBConv with expand
// ratio 1 via src/networks.py:303).  The same arithmetic and the same Winograd-domain filter image as wino_nt_kernel / wino_wgrad_kernel
// (conv_wino.hip), another data path.
//
// wino_nt_kernel streams U and V images of 8-channel chunks through LDS; with 24 channels an item has three or four chunks and then an epilogue
// in which all eight wavefronts stop, the matrix work is padded 24 -> 32 on BOTH sides (56 % useful), and the workgroup owns a compute unit
// (160 KB of LDS).  Here the whole transformed filter lives in REGISTERS and nothing but a 32 KB exchange buffer in LDS.  fp32 MFMAs execute on the
// vector ALU (tools/micro/coissue_probe.hip: a vector instruction behind an MFMA costs its full issue time), so this file counts vector instructions:
//
// wino_small_kernel<KS, STATS, RES> (forward / input gradient):
//   wavefront w of a workgroup (4 wavefronts) owns patch row i = w: the four transform positions (i, jx), jx = 0..3, of every tile.  Its A operands
//   U_(i,jx)[cout][c] -- 4 positions x 2 blocks of 16 output channels x KS k-steps = 8 KS registers per lane -- are read ONCE from the filter image.
//   A lane is (tile = lane & 15, k-group kg = lane >> 4) and carries channels 4 kg .. 4 kg + 3 and (24 channels) 16 + 2 kg, 17 + 2 kg: the four lanes of
//   a tile read one aligned 64-byte and one aligned 32-byte piece of a pixel.  The B operand V_(i,jx)[tile][c] is made in the lane that feeds it to the
//   MFMA: row i of B^T d B takes two patch rows (0: r0 - r2, 1: r1 + r2, 2: r2 - r1, 3: r3 - r1; row 3 negated on both operands as in conv_wino.hip);
//   a lane loads only its tile's OWN two pixel columns, the outer two are the neighbour tiles' own columns and arrive by DPP row shifts (lanes 0 / 15 of
//   a 16-tile row load theirs; at the image border the column is zero, which is also where consecutive tiles stop being neighbours); vertical pass =
//   one multiply-add, horizontal pass in-lane.  No LDS image, no barrier in front of the MFMAs; 8 KS MFMAs (v_mfma_f32_16x16x4_f32, 8 independent
//   accumulators) per 16-tile block.  Padding: output channels 24 -> 32 only (75 % useful), the reduction is exact.
//   Output transform: the horizontal half (over jx) in registers; the vertical half sums over the patch rows = over the wavefronts: every wavefront
//   leaves its two half-transformed columns in LDS (16 floats per lane as four 16-byte stores, double-buffered: ONE barrier per block), then a lane of
//   wavefront w finishes output row py = kg & 1 of tile j16 for the four channels 16 (w >> 1) + 4 (2 (w & 1) + (kg >> 1)) + e: two 16-byte stores, skip
//   add, BatchNorm statistics records (the (S, Q, k, n) format and segment rule of wino_nt_kernel: one record per 16-tile block and channel).
//   The block loop is ONE basic block: absent lanes / blocks / records store to an out-of-range descriptor offset instead of branching (only the one
//   block that holds the segment boundary branches, to its second record).  Loads and stores are inline asm behind hand-counted waits, and the loop is
//   skewed -- outputs stored one block later, skip values requested one block earlier -- because the vector-memory queue completes in order.
//   Blocks of 16 tiles are dealt to persistent workgroups in XCD-contiguous runs (vertically adjacent tiles share two of four patch rows in L2).
//
// wino_small_wgrad_kernel (filter gradient): at the end of the file.
#include "conv_params.h"
#include <cstdlib>
#include <type_traits>
// timing experiments (any bit set = wrong results): 1 no patch-row loads, 2 no MFMAs, 4 no stores / statistics / skip loads, 8 no barrier
#ifndef WS_ABLATE
#define WS_ABLATE 0
#endif

namespace suhpe {

typedef float floatx4s __attribute__((ext_vector_type(4)));
typedef unsigned u32x2s __attribute__((ext_vector_type(2)));

struct WinoSmallGeom {
    int TX, TPI;
    unsigned div_tpi_mul, div_tpi_shr, div_tx_mul, div_tx_shr;
    int MT;                      // tiles of the batch
    int nblocks;                 // ceil(MT / 16)
    int per_xcd;                 // blocks of one XCD's run
    int seg_tile;                // first tile of the second batch segment (0: one segment)
    int fw_group_stride;
};

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
// buffer descriptor as four scalars for the asm stores below (num_records 0 when the operand is absent: every access is out of range and dropped)
__device__ __forceinline__ u32x4s rsrc_scalars(const void* base, int64_t bytes) {
    const uint64_t addr = reinterpret_cast<uint64_t>(base);
    u32x4s r;
    r.x = __builtin_amdgcn_readfirstlane((uint32_t)addr);
    r.y = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane((uint32_t)(bytes < 0 ? 0 : (bytes > 0x7fffffff ? 0x7fffffff : bytes)));
    r.w = 0x00020000u;
    return r;
}
// Stores hipcc does not see (conv_params.h, store4_unseen), through a descriptor: a lane that has nothing to store passes kOob and the hardware drops
// it -- no branch around the store, so the loop body stays ONE basic block.  s_nop 4: the descriptor's scalar registers may come straight from a
// v_readlane (conv_wino.hip, issue_v); s_nop 1: the data registers are read right after issue.
__device__ __forceinline__ void store2_buf(u32x4s rs, unsigned off, float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    asm volatile("s_nop 4\n\tbuffer_store_dwordx2 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(v), "v"(off), "s"(rs) : "memory");
}
__device__ __forceinline__ void store4_buf(u32x4s rs, unsigned off, float a, float b, float c, float d) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {a, b, c, d};
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(v), "v"(off), "s"(rs) : "memory");
}
__device__ __forceinline__ unsigned div_nz(unsigned n, unsigned mul, unsigned shr) { return __umulhi(n, mul) >> shr; }      // (divisor > 1: the launcher checks)

// k-step s of k-group kg <-> input channel: 16-byte vectors first (channel 4 kg + s), then (24 channels) 8-byte pairs from channels 16 .. 23, so that the
// four lanes of a tile read one aligned 64-byte piece and one aligned 32-byte piece of a pixel
template <int KS> __host__ __device__ constexpr int ws_channel(int kg, int s) { return s < 4 ? 4 * kg + s : 16 + 2 * kg + (s - 4); }

// STATS: BatchNorm statistics records; RES: skip-connection add (res_mode 3)
template <int KS, bool STATS, bool RES>
__global__ __launch_bounds__(256, 2) void wino_small_kernel(const NtParams p, const WinoSmallGeom wg) {
    static_assert(KS == 4 || KS == 6, "16 or 24 input channels");
    __shared__ __attribute__((aligned(16))) float xch[2 * 16 * 64 * 4];       // [buffer][slot (i, px, blk)][lane] float4
    const GatherGeom& g = p.g;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int grp = blockIdx.z;
    const int j16 = lane & 15, kg = lane >> 4;

    // ---- this workgroup's blocks: XCD x takes blocks [x per_xcd, (x + 1) per_xcd), its workgroups interleaved inside the run
    const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3, nj = gridDim.x >> 3;
    int b = xcd * wg.per_xcd + wj;
    int b_end = (xcd + 1) * wg.per_xcd;
    if (b_end > wg.nblocks) b_end = wg.nblocks;
    if (b >= b_end) return;

    // ---- A operands: U_(i, jx)[16 blk + j16][channel of (kg, s)] from the filter image of conv_wino.hip's wino_filter_kernel
    float ureg[4][2][KS];
    {
        const float* fwg = p.fw + (int64_t)grp * wg.fw_group_stride;
#pragma unroll
        for (int jx = 0; jx < 4; ++jx)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int c = ws_channel<KS>(kg, s), R = 16 * blk + j16, xi = 4 * wave + jx;
                    const int chunk = 2 * (c >> 4) + ((c >> 2) & 1), kc = 4 * ((c >> 3) & 1) + (c & 3);
                    ureg[jx][blk][s] = fwg[(int64_t)chunk * WN_IMG + wino_slot_off(xi >> 1, R, kc >> 1) + 2 * (xi & 1) + (kc & 1)];
                }
    }
    // patch rows of this wavefront's row of B^T d B: ra + sgn * rb
    const int ra = wave, rb = wave < 2 ? 2 : 1;
    const float sgn = wave == 1 ? 1.f : -1.f;
    const unsigned ldg4 = (unsigned)g.ldg * 4u, ldy4 = (unsigned)p.ldy * 4u;
    const unsigned row_a = (unsigned)(((ra - 1) * g.GW - 1) * g.ldg * 4), row_b = (unsigned)(((rb - 1) * g.GW - 1) * g.ldg * 4);      // (wrap-around offsets)

    // epilogue role of a lane: cout block oblk = wave >> 1, channel quad 2 (wave & 1) + (kg >> 1) of it, output row py = kg & 1 of tile j16: both pixels
    // of that row x 4 consecutive channels = two 16-byte stores
    const int oblk = wave >> 1, oquad = 2 * (wave & 1) + (kg >> 1), opy = kg & 1;
    const int co0 = 16 * oblk + 4 * oquad;                      // first of this lane's four output channels (of the group)
    const bool col_ok = co0 < p.Ng;
    const float spy = opy ? -1.f : 1.f;
    const u32x4s yrs = rsrc_scalars(p.y + (int64_t)grp * p.Ng, (int64_t)g.M * p.ldy * 4 - (int64_t)grp * p.Ng * 4);
    const u32x4s srs = rsrc_scalars(STATS ? p.stats : nullptr, STATS ? (int64_t)wg.nblocks * p.stat_cout * 16 : 0);
    const u32x4s srs2 = rsrc_scalars(STATS && p.stats_extra ? p.stats_extra : nullptr, STATS && p.stats_extra ? (int64_t)p.stat_cout * 16 : 0);
    // statistics: the lane with tile 0 writes the records of channels co0 + 2 py + {0, 1}
    const int rco = co0 + 2 * opy;
    const unsigned rec_lane = (j16 == 0) & (rco < p.Ng) ? (unsigned)(grp * p.Ng + rco) * 16u : kOob;

    // [own pixel 1, own pixel 2, halo pixel]: channels 4 kg .. 4 kg + 3 and (24 channels) 16 + 2 kg, 17 + 2 kg.  Written by inline-asm loads hipcc does not
    // see and read only behind the hand-written waits below (conv_wino.hip, issue_v, has the reasons; tests/test_codeobj_audit.py checks the code object)
    floatx4s ra4[3], rb4[3];
    typedef float floatx2s __attribute__((ext_vector_type(2)));
    floatx2s ra2[3], rb2[3];
    const u32x4s xrs = rsrc_scalars(p.x + (int64_t)grp * g.Cg, p.x_bytes - (int64_t)grp * g.Cg * 4);
    const u32x4s rrs = rsrc_scalars(RES ? p.residual + (int64_t)grp * p.Ng : nullptr, RES ? (int64_t)g.M * p.ldy * 4 - (int64_t)grp * p.Ng * 4 : 0);
    constexpr int NL = KS == 6 ? 12 : 6;                         // loads of a block's patch rows
    constexpr int NS = 2 + (STATS ? 2 : 0);                      // stores of a block that ALWAYS issue (absent lanes / records at kOob); the block with the segment
                                                                 // boundary issues two more: the counted waits then leave two more operations in flight than they
                                                                 // assume -- they wait LONGER, never shorter (a count is an upper bound on what may be pending)
    constexpr int NR = RES ? 2 : 0;                              // skip-value loads of a block
    // Requests the patch rows of block blk_id (live = false: a block that does not exist -- every lane reads zeros from beyond the descriptor, so the
    // loop body has no conditional load).  A tile's patch spans pixel columns 2 tx - 1 .. 2 tx + 2; columns 2 tx and 2 tx + 1 are the tile's OWN, the
    // outer two are the neighbour tiles' own columns -- the neighbour lanes' registers (DPP row shifts in the transform below).  Only lane 0 of a
    // 16-tile row loads its left column and lane 15 its right one (`halo`: one more load instruction in which 2 of 16 lanes are in range), and at the
    // image border the outer column is zero, which is also where consecutive tiles stop being neighbours.  Half the cache-line requests of loading
    // four columns per lane.  Returns the byte offset of the tile's first output of this lane's epilogue role (kOob: none) and the border flags.
    unsigned flags_n = 0;                                        // bit 0: a left column exists (tx != 0), bit 1: a right column exists
    auto issue_loads = [&](int blk_id, bool live) __attribute__((always_inline)) {
        const int T = blk_id * 16 + j16;
        const bool tv = live & (T < wg.MT);
        const unsigned Tn = (unsigned)(tv ? T : 0);
        const unsigned n = div_nz(Tn, wg.div_tpi_mul, wg.div_tpi_shr);
        const unsigned rem = Tn - __umul24(n, (unsigned)wg.TPI);
        const unsigned ty = div_nz(rem, wg.div_tx_mul, wg.div_tx_shr), tx = rem - __umul24(ty, (unsigned)wg.TX);
        const unsigned pix0 = __umul24(__umul24(n, (unsigned)g.OH) + 2u * ty, (unsigned)g.OW) + 2u * tx;
        const unsigned base = __umul24(pix0, ldg4) + (unsigned)(kg * 16);
        const bool oka = tv & ((ra != 0) | (ty != 0)) & ((ra != 3) | ((int)ty != (g.GH >> 1) - 1));
        const bool okb = tv;                                     // (patch rows 1 and 2 lie inside the image)
        const bool x0 = tx != 0, x3 = (int)tx != wg.TX - 1;
        flags_n = (x0 ? 1u : 0u) | (x3 ? 2u : 0u);
        const bool hl = ((j16 == 0) & x0) | ((j16 == 15) & x3);
        const unsigned hq = j16 == 0 ? 0u : 3u * ldg4;
        unsigned oa[3], ob[3];
        oa[0] = oka ? base + row_a + ldg4 : kOob; oa[1] = oka ? base + row_a + 2u * ldg4 : kOob; oa[2] = oka & hl ? base + row_a + hq : kOob;
        ob[0] = okb ? base + row_b + ldg4 : kOob; ob[1] = okb ? base + row_b + 2u * ldg4 : kOob; ob[2] = okb & hl ? base + row_b + hq : kOob;
#if WS_ABLATE & 1
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { ra4[q][e] = __uint_as_float(oa[q] + e); rb4[q][e] = __uint_as_float(ob[q] + e); }
            ra2[q][0] = ra2[q][1] = 1.f; rb2[q][0] = rb2[q][1] = 2.f;
        }
#else
        // (an offset at or beyond kOob stays out of range with the small constants added to it: the descriptors hold < 2^31 bytes)
        if constexpr (KS == 6) {
            const unsigned d2 = 64u - 8u * (unsigned)kg;         // channel 16 + 2 kg sits at byte 64 + 8 kg of the pixel; oa / ob hold 16 kg
            const unsigned pa0 = oa[0] + d2, pa1 = oa[1] + d2, pa2 = oa[2] + d2, pb0 = ob[0] + d2, pb1 = ob[1] + d2, pb2 = ob[2] + d2;
            asm volatile(
                "s_nop 4\n\t"
                "buffer_load_dwordx4 %0, %12, %24, 0 offen\n\t"
                "buffer_load_dwordx4 %3, %15, %24, 0 offen\n\t"
                "buffer_load_dwordx4 %1, %13, %24, 0 offen\n\t"
                "buffer_load_dwordx4 %4, %16, %24, 0 offen\n\t"
                "buffer_load_dwordx2 %6, %18, %24, 0 offen\n\t"
                "buffer_load_dwordx2 %9, %21, %24, 0 offen\n\t"
                "buffer_load_dwordx2 %7, %19, %24, 0 offen\n\t"
                "buffer_load_dwordx2 %10, %22, %24, 0 offen\n\t"
                "buffer_load_dwordx4 %2, %14, %24, 0 offen\n\t"
                "buffer_load_dwordx4 %5, %17, %24, 0 offen\n\t"
                "buffer_load_dwordx2 %8, %20, %24, 0 offen\n\t"
                "buffer_load_dwordx2 %11, %23, %24, 0 offen"
                : "=&v"(ra4[0]), "=&v"(ra4[1]), "=&v"(ra4[2]), "=&v"(rb4[0]), "=&v"(rb4[1]), "=&v"(rb4[2]),
                  "=&v"(ra2[0]), "=&v"(ra2[1]), "=&v"(ra2[2]), "=&v"(rb2[0]), "=&v"(rb2[1]), "=&v"(rb2[2])
                : "v"(oa[0]), "v"(oa[1]), "v"(oa[2]), "v"(ob[0]), "v"(ob[1]), "v"(ob[2]),
                  "v"(pa0), "v"(pa1), "v"(pa2), "v"(pb0), "v"(pb1), "v"(pb2), "s"(xrs)
                : "memory");
        } else {
            asm volatile(
                "s_nop 4\n\t"
                "buffer_load_dwordx4 %0, %6, %12, 0 offen\n\t"
                "buffer_load_dwordx4 %3, %9, %12, 0 offen\n\t"
                "buffer_load_dwordx4 %1, %7, %12, 0 offen\n\t"
                "buffer_load_dwordx4 %4, %10, %12, 0 offen\n\t"
                "buffer_load_dwordx4 %2, %8, %12, 0 offen\n\t"
                "buffer_load_dwordx4 %5, %11, %12, 0 offen"
                : "=&v"(ra4[0]), "=&v"(ra4[1]), "=&v"(ra4[2]), "=&v"(rb4[0]), "=&v"(rb4[1]), "=&v"(rb4[2])
                : "v"(oa[0]), "v"(oa[1]), "v"(oa[2]), "v"(ob[0]), "v"(ob[1]), "v"(ob[2]), "s"(xrs)
                : "memory");
        }
#endif
        return tv & col_ok ? __umul24(pix0, ldy4) + (unsigned)opy * (unsigned)g.OW * ldy4 + (unsigned)(co0 * 4) : kOob;
    };

    // The vector-memory queue completes in order, and the stores below are invisible to hipcc (a store hipcc issues itself makes it drain the whole
    // queue before the data registers are reused), so every wait for a load also waits for the stores issued before it.  The loop is therefore
    // skewed: a block's outputs are stored one block LATER, right after the loads of the block after next were issued -- by the time anything waits
    // behind those stores a whole block of MFMAs has passed -- and the skip-connection values of a block are requested one block EARLIER than they
    // are added.  Nothing in the body branches (absent lanes / blocks / records go to kOob), so the address arithmetic, the stores and the
    // statistics of the neighbouring blocks can be scheduled into the shadow of this block's MFMAs.
    float outp[2][4];                                            // finished outputs of the previous block: [pixel px of row py][cout]
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) outp[q][e] = 0.f;
    floatx4s resp[2];                                            // its skip values (asm loads, see above)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) resp[q][e] = 0.f;
    auto issue_res = [&](unsigned yoff) __attribute__((always_inline)) {
        const unsigned y1 = yoff + ldy4;                         // (kOob + a row pitch is still out of range)
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %4, 0 offen\n\tbuffer_load_dwordx4 %1, %3, %4, 0 offen"
                     : "=&v"(resp[0]), "=&v"(resp[1]) : "v"(yoff), "v"(y1), "s"(rrs) : "memory");
    };
    // stores + statistics of block blk_id (blk_id < 0: nothing -- every offset is kOob)
    auto finish_block = [&](int blk_id, unsigned yoff) __attribute__((always_inline)) {
        const bool lane_ok = yoff != kOob;                       // a tile of the batch and channels of the group
        {
            float o[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) o[q][e] = outp[q][e];
            if constexpr (RES) {
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[q][e] += resp[q][e];
            }
            store4_buf(yrs, yoff, o[0][0], o[0][1], o[0][2], o[0][3]);
            store4_buf(yrs, lane_ok ? yoff + ldy4 : kOob, o[1][0], o[1][1], o[1][2], o[1][3]);
        }
        if constexpr (STATS) {
            // one (S, Q, k, n) record per (block, cout): sums over the block's valid pixels shifted by k = the block's first output; the block that
            // contains the segment boundary emits its first-segment tiles to `stats`, the rest to `stats_extra` (both computed always: no branch).
            // A lane holds one output row of its tile: the other row's sums come from lane ^ 16 (ds_swizzle), then the row py = 0 lanes carry on with
            // channels 0, 1 of the quad and the py = 1 lanes with channels 2, 3; the sums over the 16 tiles by DPP row rotations.
            const int T = blk_id * 16 + j16;
            const int blk0 = blk_id * 16;
            const bool straddle = (wg.seg_tile > blk0) & (wg.seg_tile < blk0 + 16) & (p.stats_extra != nullptr);
            const int cut = straddle ? wg.seg_tile : 0x7fffffff;
            const bool tile_ok = T < wg.MT;
            const bool first = tile_ok & (T < cut), second = tile_ok & (T >= cut);
            int nvalid = wg.MT - blk0;
            nvalid = nvalid < 0 ? 0 : (nvalid > 16 ? 16 : nvalid);
            const int n1 = straddle ? cut - blk0 : nvalid;
            const unsigned rec = (blk_id >= 0) & (nvalid > 0) & (rec_lane != kOob) ? (unsigned)blk_id * (unsigned)p.stat_cout * 16u + rec_lane : kOob;
            const unsigned rec2 = straddle ? rec_lane : kOob;
            float ks[4], as[4], qs[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ks[e] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(outp[0][e]), 0x0000));      // lane 0 of the 32-lane half: tile 0, row 0, this quad
                const float v0 = outp[0][e] - ks[e], v1 = outp[1][e] - ks[e];
                const float a = v0 + v1, qq = v0 * v0 + v1 * v1;
                as[e] = a + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a), 0x401F));           // + lane ^ 16: the tile's other output row
                qs[e] = qq + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(qq), 0x401F));
            }
            float sa[2], sq[2], sk[2];
#pragma unroll
            for (int ee = 0; ee < 2; ++ee) { sa[ee] = opy ? as[2 + ee] : as[ee]; sq[ee] = opy ? qs[2 + ee] : qs[ee]; sk[ee] = opy ? ks[2 + ee] : ks[ee]; }
#pragma unroll
            for (int ee = 0; ee < 2; ++ee) {
                float a1 = first ? sa[ee] : 0.f, q1 = first ? sq[ee] : 0.f;
                a1 += quad_dpp<0x128>(a1); q1 += quad_dpp<0x128>(q1);       // row_ror 8, 4, 2, 1: every lane of the row ends with the row's sum
                a1 += quad_dpp<0x124>(a1); q1 += quad_dpp<0x124>(q1);
                a1 += quad_dpp<0x122>(a1); q1 += quad_dpp<0x122>(q1);
                a1 += quad_dpp<0x121>(a1); q1 += quad_dpp<0x121>(q1);
                store4_buf(srs, (rco + ee < p.Ng) & (rec != kOob) ? rec + 16u * ee : kOob, a1, q1, sk[ee], 4.f * (float)n1);
            }
            if (straddle) {                                      // (one block of the launch at most: the only branch of the loop body, behind everything else)
#pragma unroll
                for (int ee = 0; ee < 2; ++ee) {
                    float a2 = second ? sa[ee] : 0.f, q2 = second ? sq[ee] : 0.f;
                    a2 += quad_dpp<0x128>(a2); q2 += quad_dpp<0x128>(q2);
                    a2 += quad_dpp<0x124>(a2); q2 += quad_dpp<0x124>(q2);
                    a2 += quad_dpp<0x122>(a2); q2 += quad_dpp<0x122>(q2);
                    a2 += quad_dpp<0x121>(a2); q2 += quad_dpp<0x121>(q2);
                    store4_buf(srs2, (rco + ee < p.Ng) & (rec2 != kOob) ? rec2 + 16u * ee : kOob, a2, q2, sk[ee], 4.f * (float)(nvalid - n1));
                }
            }
        }
    };

    unsigned yoff_n = issue_loads(b, true), yoff_c = kOob, yoff_p = kOob;
    // (the first pass' counted wait assumes a full queue behind the patch rows: drain once here.  The builtin is for hipcc: it has the filter loads of
    // the prologue pending and would otherwise wait for them at their first use -- inside the loop, on every pass, with vmcnt(0))
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if constexpr (KS == 6)
        asm volatile("s_waitcnt vmcnt(0) ; covers-loads" : "+v"(ra4[0]), "+v"(ra4[1]), "+v"(ra4[2]), "+v"(rb4[0]), "+v"(rb4[1]), "+v"(rb4[2]),
                     "+v"(ra2[0]), "+v"(ra2[1]), "+v"(ra2[2]), "+v"(rb2[0]), "+v"(rb2[1]), "+v"(rb2[2]) :: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0) ; covers-loads" : "+v"(ra4[0]), "+v"(ra4[1]), "+v"(ra4[2]), "+v"(rb4[0]), "+v"(rb4[1]), "+v"(rb4[2]) :: "memory");
    int bp = -1;                                                 // the block whose outputs wait in outp (-1: none yet)
    int buf = 0;
    for (;;) {
        // ---- B operands of the block: vertical pass on the three loaded columns, outer columns from the neighbour lanes, horizontal pass
        float V[4][KS];
        {
            // the patch rows requested one pass ago; behind them in the queue: the stores of the block before last and the skip loads of the last one
            if constexpr (KS == 6)
                asm volatile("s_waitcnt vmcnt(%12) ; covers-loads" : "+v"(ra4[0]), "+v"(ra4[1]), "+v"(ra4[2]), "+v"(rb4[0]), "+v"(rb4[1]), "+v"(rb4[2]),
                             "+v"(ra2[0]), "+v"(ra2[1]), "+v"(ra2[2]), "+v"(rb2[0]), "+v"(rb2[1]), "+v"(rb2[2]) : "n"(NS + NR) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%6) ; covers-loads" : "+v"(ra4[0]), "+v"(ra4[1]), "+v"(ra4[2]), "+v"(rb4[0]), "+v"(rb4[1]), "+v"(rb4[2]) : "n"(NS + NR) : "memory");
            const bool x0 = flags_n & 1u, x3 = (flags_n & 2u) != 0;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float t1 = s < 4 ? fmaf(sgn, rb4[0][s & 3], ra4[0][s & 3]) : fmaf(sgn, rb2[0][s & 1], ra2[0][s & 1]);
                const float t2 = s < 4 ? fmaf(sgn, rb4[1][s & 3], ra4[1][s & 3]) : fmaf(sgn, rb2[1][s & 1], ra2[1][s & 1]);
                const float th = s < 4 ? fmaf(sgn, rb4[2][s & 3], ra4[2][s & 3]) : fmaf(sgn, rb2[2][s & 1], ra2[2][s & 1]);
                // row_shr:1 -- a lane takes lane - 1's value, lane 0 of the row keeps `old` (its own halo load); row_shl:1 -- lane + 1's, lane 15 keeps its halo
                float t0 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(th), __float_as_int(t2), 0x111, 0xf, 0xf, false));
                float t3 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(th), __float_as_int(t1), 0x101, 0xf, 0xf, false));
                t0 = x0 ? t0 : 0.f;
                t3 = x3 ? t3 : 0.f;
                V[0][s] = t0 - t2; V[1][s] = t1 + t2; V[2][s] = t2 - t1; V[3][s] = t1 - t3;
            }
        }
        yoff_c = yoff_n;
        const int bn = b + nj;
        const bool has_next = bn < b_end;
        yoff_n = issue_loads(bn, has_next);                      // (the patch-row registers are free: a whole block of MFMAs until they are read)
        __builtin_amdgcn_sched_barrier(0);                       // (left alone the scheduler sinks the loads behind the MFMAs: a block later they are waited for at once)
#if !(WS_ABLATE & 4)
        if constexpr (RES) asm volatile("s_waitcnt vmcnt(%2) ; covers-loads" : "+v"(resp[0]), "+v"(resp[1]) : "n"(NL) : "memory");      // behind the skip values: this pass' patch rows
        finish_block(bp, yoff_p);
        if constexpr (RES) issue_res(yoff_c);
#endif
        floatx4s acc[4][2];
#pragma unroll
        for (int jx = 0; jx < 4; ++jx)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[jx][blk][e] = (WS_ABLATE & 2) ? V[jx][e] : 0.f;
#if !(WS_ABLATE & 2)
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                acc[jx][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ureg[jx][0][s], V[jx][s], acc[jx][0], 0, 0, 0);
                acc[jx][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ureg[jx][1][s], V[jx][s], acc[jx][1], 0, 0, 0);
            }
#endif
        // ---- horizontal half of A^T M A, to LDS: slot (i 2 + px) 2 + blk holds the lane's four couts
        float* xw = xch + buf * (16 * 64 * 4);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            float r0[4], r1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r0[e] = acc[0][blk][e] + acc[1][blk][e] + acc[2][blk][e];
                r1[e] = acc[1][blk][e] - acc[2][blk][e] - acc[3][blk][e];
            }
            *reinterpret_cast<float4*>(&xw[(((wave * 2 + 0) * 2 + blk) * 64 + lane) * 4]) = make_float4(r0[0], r0[1], r0[2], r0[3]);
            *reinterpret_cast<float4*>(&xw[(((wave * 2 + 1) * 2 + blk) * 64 + lane) * 4]) = make_float4(r1[0], r1[1], r1[2], r1[3]);
        }
#if !(WS_ABLATE & 8)
        __syncthreads();
#endif
        // ---- vertical half for this lane's (row py, quad) of tile j16: rows py, py + 1, py + 2 of the exchange (py = 0: T0 + T1 + T2, py = 1: T1 - T2 - T3);
        // kept in registers until the next pass stores it
        {
            const int src = oquad * 16 + j16;
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const float4 ta = *reinterpret_cast<const float4*>(&xw[((((opy + 0) * 2 + px) * 2 + oblk) * 64 + src) * 4]);
                const float4 tb = *reinterpret_cast<const float4*>(&xw[((((opy + 1) * 2 + px) * 2 + oblk) * 64 + src) * 4]);
                const float4 tc = *reinterpret_cast<const float4*>(&xw[((((opy + 2) * 2 + px) * 2 + oblk) * 64 + src) * 4]);
                outp[px][0] = fmaf(spy, tb.x + tc.x, ta.x); outp[px][1] = fmaf(spy, tb.y + tc.y, ta.y);
                outp[px][2] = fmaf(spy, tb.z + tc.z, ta.z); outp[px][3] = fmaf(spy, tb.w + tc.w, ta.w);
            }
        }
        bp = b;
        yoff_p = yoff_c;
        if (!has_next) break;
        b = bn;
        buf ^= 1;
    }
    // the last pass requested the patch rows of a block that does not exist (zeros from beyond the descriptor): they must have landed before their
    // registers are used for anything else
    if constexpr (KS == 6)
        asm volatile("s_waitcnt vmcnt(0) ; covers-loads" : "+v"(ra4[0]), "+v"(ra4[1]), "+v"(ra4[2]), "+v"(rb4[0]), "+v"(rb4[1]), "+v"(rb4[2]),
                     "+v"(ra2[0]), "+v"(ra2[1]), "+v"(ra2[2]), "+v"(rb2[0]), "+v"(rb2[1]), "+v"(rb2[2]), "+v"(resp[0]), "+v"(resp[1]) :: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0) ; covers-loads" : "+v"(ra4[0]), "+v"(ra4[1]), "+v"(ra4[2]), "+v"(rb4[0]), "+v"(rb4[1]), "+v"(rb4[2]), "+v"(resp[0]), "+v"(resp[1]) :: "memory");
#if !(WS_ABLATE & 4)
    finish_block(bp, yoff_p);
#endif
}

static const bool kWinoSmall = !(getenv("SUHPE_WINO_SMALL") && atoi(getenv("SUHPE_WINO_SMALL")) == 0);
static const int kWinoSmallWgs = getenv("SUHPE_WINO_SMALL_WGS") ? atoi(getenv("SUHPE_WINO_SMALL_WGS")) : 2;      // workgroups per compute unit

// descriptor-level eligibility (channels of the gathered tensor / of the output, per group; output rows and image size): what suhpe_conv2d_kernel_name
// reports and launch_wino_small requires (the launch adds: no bias, no eval-mode epilogue, an 8-byte aligned skip tensor)
bool wino_small_desc_ok(int cg, int ng, int64_t M, int oh, int ow) {
    return kWinoSmall && (cg == 16 || cg == 24) && ng <= 32 && ng % 2 == 0 && M < (1 << 24) && ow >= 4 && oh >= 4;
}

// the caller (launch_wino) has checked wino_params_ok: geometry, alignment, fw present, res_mode 0 / 3 or the eval-mode form
bool launch_wino_small(const NtParams& p_in, int groups, hipStream_t st) {
    const GatherGeom& g = p_in.g;
    if (!wino_small_desc_ok(g.Cg, p_in.Ng, g.M, g.OH, g.OW) || p_in.scale || p_in.bias) return false;
    if (p_in.residual && (reinterpret_cast<uintptr_t>(p_in.residual) & 7)) return false;
    NtParams p = p_in;
    WinoSmallGeom wg;
    memset(&wg, 0, sizeof(wg));
    wg.TX = g.OW / 2; wg.TPI = (g.OH / 2) * wg.TX;
    magic_div((unsigned)wg.TPI, &wg.div_tpi_mul, &wg.div_tpi_shr);
    magic_div((unsigned)wg.TX, &wg.div_tx_mul, &wg.div_tx_shr);
    wg.MT = g.N * wg.TPI;
    wg.nblocks = (int)ceil_div(wg.MT, 16);
    wg.per_xcd = (int)ceil_div(wg.nblocks, 8);
    wg.seg_tile = (p.stats && p.seg_row > 0) ? (int)(p.seg_row / 4) : 0;
    if (p.stats && (int64_t)wg.nblocks * p.stat_cout * 16 >= (1LL << 31)) return false;      // (32-bit record offsets)
    wg.fw_group_stride = (int)ceil_div(p.Ng, 64) * (2 * ((g.Cg + 15) / 16)) * WN_IMG;      // (wino_filter_kernel's image: row blocks of 64 x chunks of 8 channels)
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (cus <= 0) cus = 256;
    }
    int per_group = cus * kWinoSmallWgs / groups / 8 * 8;
    if (per_group < 8) per_group = 8;
    const int need = (int)ceil_div(wg.per_xcd, 1) * 8;           // one workgroup per block of the longest run at most
    const int gx = need < per_group ? need : per_group;
    const dim3 grid((unsigned)gx, 1, groups), block(256);
    auto go = [&](auto ks) {
        constexpr int KS = decltype(ks)::value;
        const bool st_ = p.stats != nullptr, rs_ = p.res_mode == 3;
        if (st_ && rs_) hipLaunchKernelGGL((wino_small_kernel<KS, true, true>), grid, block, 0, st, p, wg);
        else if (st_) hipLaunchKernelGGL((wino_small_kernel<KS, true, false>), grid, block, 0, st, p, wg);
        else if (rs_) hipLaunchKernelGGL((wino_small_kernel<KS, false, true>), grid, block, 0, st, p, wg);
        else hipLaunchKernelGGL((wino_small_kernel<KS, false, false>), grid, block, 0, st, p, wg);
    };
    if (g.Cg == 16) go(std::integral_constant<int, 4>()); else go(std::integral_constant<int, 6>());
    return true;
}

// ================================================================================================================================
// Filter gradient of the same few-channel convolutions, F(3x3, 2x2) (conv_wino.hip, wino_wgrad_kernel, has the algebra and the sign conventions:
// dU_xi[cout][cin] = sum over tiles of dM_xi[tile][cout] V_xi[tile][cin], rows 3 / columns 3 negated on both operands), with both operands made in
// registers: no LDS, no barrier.  Wavefront w owns patch row i = w (positions (i, jx)) for ALL channel pairs: 4 positions x 2 x 2 blocks of 16 x 16
// = 16 accumulators; the reduction index of v_mfma_f32_16x16x4_f32 is the tile (k-group kg = lane >> 4 = tile 4 step + kg), lane & 15 the channel:
// a lane reads the 2x2 block of dy of its tile at output channel 16 cb + (lane & 15) and the two patch rows of x at input channel 16 nb + (lane & 15)
// (dword loads, 64 contiguous bytes per tile; zero beyond the channels, the image and the slab through the descriptor), forms row i of A dY A^T and of
// B^T d B in-lane and issues 16 MFMAs per step; the loads of the next step are requested before them.  Matrix work is padded 24 -> 32 on both sides
// (56 % useful) and still a third of the direct kernel's.  One slab of partial sums per workgroup (split-K over the tiles, two workgroups per compute
// unit), added in a fixed order by wino_wgrad_finish_kernel.
__global__ __launch_bounds__(256, 2) void wino_small_wgrad_kernel(const WinoWgradParams p) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int grp = blockIdx.z, split = blockIdx.x;
    const int m = lane & 15, kg = lane >> 4;
    const BufRsrc xr = make_rsrc(p.x + (int64_t)grp * p.Cg, p.x_bytes - (int64_t)grp * p.Cg * 4);
    const BufRsrc yr = make_rsrc(p.dy + (int64_t)grp * p.Ng, p.dy_bytes - (int64_t)grp * p.Ng * 4);
    const int tile_lo = split * p.tiles_per_split;
    int tile_hi = tile_lo + p.tiles_per_split;
    if (tile_hi > p.MT) tile_hi = p.MT;
    const int nsteps = tile_lo < tile_hi ? (tile_hi - tile_lo + 3) >> 2 : 0;
    const int ra = wave, rb = wave < 2 ? 2 : 1;
    const float sgn = wave == 1 ? 1.f : -1.f;                   // V's row i: patch row ra + sgn * patch row rb (row 3 negated)
    const float c0 = wave == 3 ? 0.f : 1.f, c1 = wave == 0 ? 0.f : (wave == 2 ? -1.f : 1.f);      // (A dY)'s row i: c0 dy row 0 + c1 dy row 1 (row 3 negated)
    const unsigned ldx4 = (unsigned)p.ldx * 4u, ldy4 = (unsigned)p.lddy * 4u;
    // (patch rows start one pixel left of the tile: that pixel gets its own offset register -- a load's scalar offset is not part of the bounds check,
    //  so the register offset itself must never be 'negative')
    const unsigned row_a = (unsigned)((ra - 1) * p.W * p.ldx * 4), row_b = (unsigned)((rb - 1) * p.W * p.ldx * 4);

    struct Raw { float dyv[2][4], xa[2][4], xb[2][4]; };        // [channel block][pixel]
    // Loader state of this lane's tile (tile_lo + 4 step + kg), advanced by four tiles per step WITHOUT divisions: the tile's column, row and the
    // byte offsets of its first pixel in dy and x.  fp32 MFMAs run on the vector ALU itself (tools/micro/coissue_probe.hip: a v_fma behind a
    // v_mfma_f32_16x16x4_f32 costs its full four cycles), so every vector instruction of the loop is paid for in matrix throughput: the pixel /
    // channel-block offsets ride in the loads' scalar offset and immediate fields, and channels beyond Cg / Ng are NOT masked -- a padding lane of
    // A (B) only feeds rows (columns) of the accumulators that are never stored, and what it reads is mapped memory or the descriptor's zero.
    int lT = tile_lo + kg;
    unsigned ltx, lty, lxb, lyb;
    {
        const unsigned Tn = (unsigned)(lT < p.MT ? lT : 0);
        const unsigned n = div_nz(Tn, p.div_tpi_mul, p.div_tpi_shr);
        const unsigned rem = Tn - __umul24(n, (unsigned)p.TPI);
        lty = div_nz(rem, p.div_tx_mul, p.div_tx_shr);
        ltx = rem - __umul24(lty, (unsigned)p.TX);
        const unsigned pix0 = __umul24(__umul24(n, (unsigned)p.H) + 2u * lty, (unsigned)p.W) + 2u * ltx;
        lyb = __umul24(pix0, ldy4) + (unsigned)m * 4u;
        lxb = __umul24(pix0, ldx4) + (unsigned)m * 4u;
    }
    const unsigned TYm1 = (unsigned)(p.H >> 1) - 1u;
    const int sy1 = (int)ldy4, sy2 = (int)((unsigned)p.W * ldy4), sy3 = (int)((unsigned)(p.W + 1) * ldy4);
    const int sx1 = (int)ldx4, sx2 = (int)(2u * ldx4);
    auto load_step = [&](Raw& r) __attribute__((always_inline)) {
        const bool tv = lT < tile_hi;
        const bool oka = tv & ((ra != 0) | (lty != 0)) & ((ra != 3) | (lty != TYm1));
        const bool x0 = ltx != 0, x3 = ltx != (unsigned)p.TX - 1u;
        const unsigned vy = tv ? lyb : kOob;
        const unsigned va = oka ? lxb + row_a : kOob, vb = tv ? lxb + row_b : kOob;
        const unsigned va0 = x0 ? va - ldx4 : kOob, va3 = x3 ? va : kOob, vb0 = x0 ? vb - ldx4 : kOob, vb3 = x3 ? vb : kOob;      // (kOob - a pixel: x0 is false there)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            r.dyv[cb][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, vy + 64u * cb, 0, 0));
            r.dyv[cb][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, vy + 64u * cb, sy1, 0));
            r.dyv[cb][2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, vy + 64u * cb, sy2, 0));
            r.dyv[cb][3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, vy + 64u * cb, sy3, 0));
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            r.xa[nb][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, va0 + 64u * nb, 0, 0));
            r.xa[nb][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, va + 64u * nb, 0, 0));
            r.xa[nb][2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, va + 64u * nb, sx1, 0));
            r.xa[nb][3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, va3 + 64u * nb, sx2, 0));
            r.xb[nb][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, vb0 + 64u * nb, 0, 0));
            r.xb[nb][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, vb + 64u * nb, 0, 0));
            r.xb[nb][2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, vb + 64u * nb, sx1, 0));
            r.xb[nb][3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, vb3 + 64u * nb, sx2, 0));
        }
        // four tiles on (TX >= 4: at most one row of tiles is crossed): the next row of tiles starts one image row further than 8 pixels on
        // (pixel indices run on linearly from the last rows of an image into the next image)
        lT += 4;
        ltx += 4u;
        const bool wrap = ltx >= (unsigned)p.TX;
        ltx -= wrap ? (unsigned)p.TX : 0u;
        lty = wrap ? (lty == TYm1 ? 0u : lty + 1u) : lty;
        lyb += 8u * ldy4 + (wrap ? (unsigned)p.W * ldy4 : 0u);
        lxb += 8u * ldx4 + (wrap ? (unsigned)p.W * ldx4 : 0u);
    };
    floatx4s acc[4][2][2];
#pragma unroll
    for (int jx = 0; jx < 4; ++jx)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[jx][cb][nb][e] = 0.f;
    // one step: operands from the registers requested two steps ago, the request for the step after next into the same registers, 16 MFMAs
    auto do_step = [&](Raw& r) __attribute__((always_inline)) {
        float A[4][2], B[4][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const float r0 = fmaf(c1, r.dyv[cb][2], c0 * r.dyv[cb][0]), r1 = fmaf(c1, r.dyv[cb][3], c0 * r.dyv[cb][1]);
            A[0][cb] = r0; A[1][cb] = r0 + r1; A[2][cb] = r0 - r1; A[3][cb] = r1;                       // (column 3 negated)
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float t0 = fmaf(sgn, r.xb[nb][0], r.xa[nb][0]), t1 = fmaf(sgn, r.xb[nb][1], r.xa[nb][1]);
            const float t2 = fmaf(sgn, r.xb[nb][2], r.xa[nb][2]), t3 = fmaf(sgn, r.xb[nb][3], r.xa[nb][3]);
            B[0][nb] = t0 - t2; B[1][nb] = t1 + t2; B[2][nb] = t2 - t1; B[3][nb] = t3 - t1;             // (column 3 negated)
        }
        load_step(r);                                            // (the step after next; past the slab: every lane out of range, zeros)
#pragma unroll
        for (int jx = 0; jx < 4; ++jx)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[jx][cb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[jx][cb], B[jx][nb], acc[jx][cb][nb], 0, 0, 0);
    };
    Raw r0, r1;
    load_step(r0);
    load_step(r1);
    for (int step = 0; step < nsteps; step += 2) {               // (an odd count runs one step of zeros: its loads are out of range)
        do_step(r0);
        do_step(r1);
    }
    // ---- partial sums: part[split][position][cout][cin]; lane holds couts 16 cb + 4 kg + e, cin 16 nb + m
    float* part = p.part + (int64_t)split * 16 * p.cout_all * p.Cg;
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) {
        const int xi = 4 * wave + jx;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = 16 * cb + 4 * kg + e;
                if (co >= p.Ng) continue;
                float* dst = &part[((int64_t)xi * p.cout_all + grp * p.Ng + co) * p.Cg];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    if (16 * nb + m < p.Cg) dst[16 * nb + m] = acc[jx][cb][nb][e];
            }
    }
}

static const bool kWinoSmallWgrad = !(getenv("SUHPE_WINO_SMALL_WGRAD") && atoi(getenv("SUHPE_WINO_SMALL_WGRAD")) == 0);

// slabs of the small filter-gradient kernel (0: the conv is not its shape); the caller has checked wino_wgrad_geom_ok
int wino_small_wgrad_splits(int n, int h, int w, int cg, int ng, int groups) {
    if (!kWinoSmall || !kWinoSmallWgrad || cg > 32 || ng > 32 || h < 4 || w < 8 || (int64_t)n * h * w >= (1 << 24)) return 0;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (cus <= 0) cus = 256;
    }
    const int64_t MT = (int64_t)n * (h / 2) * (w / 2);
    static const int wgs = getenv("SUHPE_WINO_SMALL_WGRAD_WGS") ? atoi(getenv("SUHPE_WINO_SMALL_WGRAD_WGS")) : 3;
    int64_t splits = (int64_t)cus * wgs / groups;               // workgroups per compute unit
    const int64_t max_by_k = ceil_div(MT, 64);                   // at least 16 steps per slab
    if (splits > max_by_k) splits = max_by_k;
    if (splits < 1) splits = 1;
    return (int)splits;
}

void launch_wino_small_wgrad(const WinoWgradParams& p_in, int groups, hipStream_t st) {
    WinoWgradParams p = p_in;
    p.tiles_per_split = (int)(ceil_div(ceil_div(p.MT, p.splits), 4) * 4);
    hipLaunchKernelGGL(wino_small_wgrad_kernel, dim3((unsigned)p.splits, 1, groups), dim3(256), 0, st, p);
}

}  // namespace suhpe
