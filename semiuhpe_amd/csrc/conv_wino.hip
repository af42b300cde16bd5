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
// (8 wavefronts, one per CU: all 160 KB of LDS = 2 V images + 3 U images of 32 KB) walks work items = 64 tiles (2x2 output patches in
// (n, ty, tx) order = 256 output pixels) x 64 output channels, and inside an item the input channels in chunks of 8:
//
//   stage   thread (patch row i = t&3, channel vector v = (t>>2)&1, tile = t>>3) holds one row of a 4x4 input patch -- per SUPER-chunk of
//           16 channels two 16-byte descriptor loads per pixel (32 contiguous bytes; zero outside the image), one half staged per chunk --:
//           the horizontal half of B^T d B is in-thread, the vertical half runs across the four lanes of a quad with a DPP quad_perm
//           operand -- no LDS pass for the raw pixels, every lane the same work -- and the 16 values go to the LDS image of the chunk
//           (16-byte stores).  The filter side comes pre-transformed (suhpe_conv2d_wino_filter: one small launch per layer and weight
//           update) in exactly the LDS image order, so a chunk's U image is one contiguous 32 KB block copied global -> LDS by LDS-DMA
//           (global_load_lds_dwordx4: no registers, no VALU, no ds_write), two chunks ahead;
//   MFMA    wavefront (tile block tb = w&3, cout half ch = w>>2) owns 16 tiles x 32 couts for ALL 16 positions: 32 accumulators of
//           v_mfma_f32_16x16x4_f32 (A = U rows, B = V rows; one ds_read_b128 hands a lane two k-steps of two positions), 64 MFMAs per
//           chunk, so the output transform A^T M A happens in registers: a lane ends up with the 2x2 pixels of one tile x 4 consecutive
//           couts and stores them as 16-byte vectors;
//   the chunks of ALL items of a workgroup form one stream: V images alternate, U images rotate through three, ONE barrier per chunk;
//   every memory operation of the loop is inline asm with hand-counted waits (see issue_v / issue_u); the first chunks of the next item
//   are staged while the current item's last chunks are multiplied, so an item's prologue costs nothing and only its epilogue (output
//   transform + stores + statistics, all from registers) interrupts the matrix pipe.
//
// LDS images are unpadded; bank conflicts are avoided by XOR swizzles of the slot address (wino_slot_off): reads (ds_read_b128) cover all
// 64 banks once per lane group, writes (ds_write_b128, 8-lane groups: one row, 2 channel vectors, 4 patch rows) all 32.
//
// Epilogues: bias, skip-connection add (res_mode 3), BatchNorm statistics records per wavefront block of 16 tiles (64 rows per record, the
// same (S, Q, k, n) format and segment rule as igemm_nt_kernel's).  MODE bit 2 (instantiations 4-6, `AFFINE`) is the eval-mode form: folded
// BatchNorm scale / shift + activation + skip add in the output-transform epilogue, no statistics (suhpe_conv2d_fwd_affine_wino), so eval passes
// on database-pinned shapes run on Winograd rounding too (parity re-measured: tests/test_gpu_winograd.py, bench.py mae_parity).  There is no
// self-transforming form: without a Winograd-domain filter (fw NULL, or plain suhpe_conv2d_fwd with the tile pinned) the dispatch runs a
// direct tile.
#include "conv_params.h"
#include <cstdlib>
#include <type_traits>
// timing experiments (tools/wino_ablate.sh builds variant libraries; any bit set = wrong results):
//   1 no barrier in the chunk loop, 2 no loads (DMA + patch rows), 4 no transform / LDS stores, 8 no MFMAs, 16 no epilogue, 32 no output stores,
//   64 no statistics
#ifndef WINO_ABLATE
#define WINO_ABLATE 0
#endif
#ifndef WINO_STAGGER
#define WINO_STAGGER 0     // 1: wavefronts 4-7 stage the next chunk half a chunk later than 0-3 (measured: +1.5 % on 6-chunk items, -0.8 % on 64-chunk items: off)
#endif

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

typedef unsigned u32x4w __attribute__((ext_vector_type(4)));

// LDS map (floats): V images [2] at 0, U images [3] at 2 * WN_IMG: 160 KB, all of a CU.
constexpr int WN_U0 = 2 * WN_IMG;
constexpr int WN_LDS_FLOATS = 5 * WN_IMG;

// HALF: at most 48 output channels in all (one cout block per item whose upper half ch = 1 holds a single 16-channel block)
//   NARROW: at most 32 output channels in all (EfficientNetV2-S's 24 -> 24 stage): each wavefront holds ONE 16-channel block (cout 16 ch + j), half the
//   MFMAs and accumulators, so a chunk is as short as its staging allows
template <int MODE>
__global__ __launch_bounds__(512, 2) void wino_nt_kernel(const NtParams p, const WinoGeom wg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GatherGeom& g = p.g;
    // The wavefront index is uniform (a scalar register); the lane index is RE-MADE where the rarely run parts need it (setup_v, the epilogue) by two
    // v_mbcnt instructions instead of being kept in a register across the chunk loop: every register kept alive there is one the 48-output form spills,
    // and a spill reload anywhere in the loop makes the compiler drain the vector-memory queue (prefetches included) at the head of every pass.
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    auto lane_now = [&]() __attribute__((always_inline)) {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    const int lane = lane_now(), t = wave * 64 + lane;
    const int grp = blockIdx.z;
    // buffer descriptor of the gathered tensor as four scalars (the patch-row loads are inline asm: see issue_loads)
    u32x4w xrs;
    {
        const uint64_t addr = reinterpret_cast<uint64_t>(p.x + (int64_t)grp * g.Cg);
        const int64_t nb = p.x_bytes - (int64_t)grp * g.Cg * 4;
        xrs.x = __builtin_amdgcn_readfirstlane((uint32_t)addr);
        xrs.y = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32) & 0xffffu);
        xrs.z = __builtin_amdgcn_readfirstlane((uint32_t)(nb > 0x7fffffff ? 0x7fffffff : nb));
        xrs.w = 0x00020000u;
    }

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
    int woff[2][2];                                             // [pair p of this patch row: m = 2 i + p][half h: k-group 2 v + h]
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) woff[pp][hh] = wino_slot_off(2 * si + pp, stl, 2 * sv + hh);

    // ---- loader cursors over the flat stream of (item, chunk) pairs.  Channels come in SUPER-CHUNKS of 16: a thread loads 32 contiguous bytes of each of
    // its four pixels (channels 16 s + 8 v .. + 7) once and stages one half per chunk, so chunk c = 2 s + h holds channels 16 s + 8 v + 4 h + e at its
    // in-chunk position 4 v + e (the filter transform writes U in the same order).  With 8-channel loads every chunk pulled a whole 128-byte line per
    // pixel through L2 -> L1 for 32 bytes of it, and the chunk loop was bound by that traffic (1.75 us of staging per chunk beside 1.86 us of MFMAs,
    // tools/wino_ablate.sh); items have an even number of chunks, so a super-chunk never straddles two items.
    int lu_item = blockIdx.x, lu_chunk = 0, lu_ub = 0;          // U cursor: next chunk to copy, into U image lu_ub (0..2)
    int lv_item = blockIdx.x, lv_sc = 0;                        // V cursor: next super-chunk to load
    unsigned voff[4];
    // channels 16 s + 8 sv + 4 hh .. + 3 of the last super-chunk s = nchunks / 2 - 1 do not exist (launch constants of the lane)
    const bool tail_off[2] = {((wg.nchunks >> 1) - 1) * 16 + 8 * sv >= g.Cg, ((wg.nchunks >> 1) - 1) * 16 + 8 * sv + 4 >= g.Cg};
    const float* ufw = nullptr;                                 // Winograd-domain filter image of (lu_item's cout block, chunk 0)
    auto setup_v = [&]() __attribute__((always_inline)) {
        const int t = wave * 64 + lane_now(), si = t & 3, sv = (t >> 2) & 1, stl = t >> 3;      // (re-made, see above)
        const int jx = lv_item >> 3;
        const int tile0 = (rt_first + jx / p.col_tiles) * 64;
        const int T = tile0 + stl;
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
            voff[c] = ok ? ((unsigned)(((int)n * g.GH + iy) * g.GW + ix0 + c) * (unsigned)g.ldg + 8u * sv) * 4u : kOob;
        }
    };
    auto setup_u = [&]() __attribute__((always_inline)) {
        const int col_tile = (lu_item >> 3) % p.col_tiles;
        ufw = p.fw + (int64_t)grp * wg.fw_group_stride + (int64_t)col_tile * wg.nchunks * WN_IMG;      // (workgroup-uniform: lives in scalar registers)
    };
    const unsigned u_lane_off = (unsigned)(wave * 1024 + lane * 4) * 4u;      // this thread's 16 bytes of each 1 KiB piece of a U image
    // Every vector-memory operation of the chunk loop is inline asm and every wait is written by hand.  Behind the builtins the compiler (1) cannot tell
    // an LDS-DMA's destination image from the fragment reads that follow and parks an s_waitcnt vmcnt(0) in front of the MFMAs, and (2) counts only the
    // loads it knows when it waits for a register load, so beside DMAs it drains the whole queue.  (Loads return in issue order; the waits below count
    // what may still be in flight behind the operation they need.)
    floatx4w raw[4][2];                                          // [pixel of the patch row][half of the 8 channels]
    auto issue_v = [&]() __attribute__((always_inline)) {        // the V cursor's super-chunk: 8 loads
        // The super-chunk's byte offset rides in the load's SCALAR offset and the half's 16 bytes in its immediate field; a lane whose channels do not
        // exist -- only in the LAST super-chunk of a layer with Cg % 16 != 0 (24 / 48 channels) -- is sent out of range by ONE select on a launch-constant
        // lane mask: one vector instruction per load instead of an add, a compare and a select (fp32 MFMAs run on the vector ALU: tools/micro/
        // coissue_probe.hip).  The scalar offset is outside the descriptor's bounds check, but voff is in range or kOob by itself.
        const unsigned cb = (unsigned)lv_sc * 64u;
        const bool last_sc = lv_sc + 1 == (wg.nchunks >> 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned o0 = last_sc & tail_off[0] ? kOob : voff[c], o1 = last_sc & tail_off[1] ? kOob : voff[c];
            // (s_nop 4: five wait states between a vector instruction that may have just written the descriptor's scalar registers -- the compiler
            //  parks scalars in vector registers under pressure and brings them back with v_readlane -- and the load that reads them; hipcc pads
            //  that hazard only for instructions it can see)
            asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(raw[c][0]) : "v"(o0), "s"(xrs), "s"(cb) : "memory");
            asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(raw[c][1]) : "v"(o1), "s"(xrs), "s"(cb) : "memory");
        }
        if (++lv_sc == (wg.nchunks >> 1)) {
            lv_sc = 0;
            lv_item += gridDim.x;
            if (item_valid(lv_item)) setup_v();
        }
    };
    auto issue_u = [&]() __attribute__((always_inline)) {        // the U cursor's chunk: this wavefront's four 1 KiB pieces of the image, global -> LDS
        // (LDS address = M0 + instruction offset + 16 * lane; the instruction offset also advances the global address)
        const unsigned lds_dst = __builtin_amdgcn_readfirstlane(
            (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(smem + WN_U0 + lu_ub * WN_IMG + wave * 1024));
        // scalar base + one 32-bit lane offset (a 64-bit per-lane address costs two more registers, and the 48-output form spilled exactly such pairs)
        const float* src = ufw + (int64_t)lu_chunk * WN_IMG;
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "s_nop 2\n\t"                                       // 5 wait states in all: the scalar base may come straight from a v_readlane (an SGPR the compiler parked in a VGPR)
            "global_load_lds_dwordx4 %1, %3\n\t"
            "global_load_lds_dwordx4 %1, %3 offset:1024\n\t"
            "global_load_lds_dwordx4 %1, %3 offset:2048\n\t"
            "global_load_lds_dwordx4 %1, %3 offset:3072\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "v"(u_lane_off), "s"(lds_dst), "s"(src) : "memory");
        lu_ub = lu_ub == 2 ? 0 : lu_ub + 1;
        if (++lu_chunk == wg.nchunks) {
            lu_chunk = 0;
            lu_item += gridDim.x;
            if (item_valid(lu_item)) setup_u();
        }
    };
    float ov[4][4];                                              // transformed values of the staged chunk: [jx][channel]
    auto transform = [&](auto half) __attribute__((always_inline)) {
        constexpr int H = decltype(half)::value;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h0 = raw[0][H][e] - raw[2][H][e], h1 = raw[1][H][e] + raw[2][H][e], h2 = raw[2][H][e] - raw[1][H][e], h3 = raw[1][H][e] - raw[3][H][e];
            ov[0][e] = fmaf(sxv, quad_dpp<0x5A>(h0), h0);
            ov[1][e] = fmaf(sxv, quad_dpp<0x5A>(h1), h1);
            ov[2][e] = fmaf(sxv, quad_dpp<0x5A>(h2), h2);
            ov[3][e] = fmaf(sxv, quad_dpp<0x5A>(h3), h3);
        }
    };
    auto write_staged = [&](int vb) __attribute__((always_inline)) {
        float* Vs = smem + vb * WN_IMG;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                *reinterpret_cast<float4*>(&Vs[woff[pp][hh]]) = make_float4(ov[2 * pp][2 * hh], ov[2 * pp][2 * hh + 1], ov[2 * pp + 1][2 * hh], ov[2 * pp + 1][2 * hh + 1]);
    };

    // ---- MFMA role: tile block tb (16 tiles), cout half ch (2 blocks of 16); lane = (row j of the block, k-group kg)
    constexpr bool HALF = (MODE & 3) == 1, NARROW = (MODE & 3) == 2;
    constexpr bool AFFINE = (MODE & 4) != 0;                    // eval-mode epilogue: y = act(acc * scale + shift [+ res]) [+ res] (no bias, no statistics)
    constexpr int NBLK = NARROW ? 1 : 2;                        // cout blocks of a wavefront
    const int tb = wave & 3, ch = wave >> 2, j16 = lane & 15, kg = lane >> 4;
    // slot offsets of this lane's fragment rows for patch row i = 0 (pair m adds m * 1024).  The swizzle of wino_slot_off only XORs bit 4 with i & 1 and
    // bits 2-3 with i >> 1, so the other three patch rows are one v_xor with a constant away (vrd(i)) and the second cout block (row + 16: same swizzle)
    // sits 256 floats behind the first: 2 registers instead of 12 -- the 48-output form spilled with the table, and its reloads put s_waitcnt vmcnt(0)
    // between the prefetches of the chunk loop.
    const int vrd0 = wino_slot_off(0, 16 * tb + j16, kg), urd0 = wino_slot_off(0, (NARROW ? 16 : 32) * ch + j16, kg);
    auto swz = [](int base, int i) __attribute__((always_inline)) { return base ^ (((i & 1) << 4) | ((i >> 1) << 2)); };
    floatx4w acc[16][NBLK];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
#pragma unroll
            for (int bb = 0; bb < NBLK; ++bb)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[xi][bb][e] = 0.f;
    };
    // MFMAs of position pairs [m0, m1) of the chunk in V image vb / U image ub (8 pairs = 64 MFMAs per chunk); the fragments of pair m + 1 are read
    // from LDS before the MFMAs of pair m issue (two register sets, carried across the calls of one chunk)
    float4 vf[2], uf[2][2];
    // second_block: this wavefront's second cout block exists (48 output channels -- the input gradient of EfficientNetV2-S's 48 -> 192 convs -- leave
    // the cout half ch = 1 with one block: its wavefronts skip the padding block's MFMAs, and since wavefronts w and w + 4 share a SIMD every SIMD
    // then carries 96 MFMAs per chunk instead of 128)
    const bool second_block = !NARROW && (!HALF || ch == 0);
    auto mma_pairs = [&](int vb, int ub, auto m0_t, auto m1_t) __attribute__((always_inline)) {
        constexpr int M0 = decltype(m0_t)::value, M1 = decltype(m1_t)::value;
        const float* Vs = smem + vb * WN_IMG;
        const float* Us = smem + WN_U0 + ub * WN_IMG;
        auto read_frags = [&](int m, int set) __attribute__((always_inline)) {
            vf[set] = *reinterpret_cast<const float4*>(&Vs[m * 1024 + swz(vrd0, m >> 1)]);
            uf[set][0] = *reinterpret_cast<const float4*>(&Us[m * 1024 + swz(urd0, m >> 1)]);
            if (!NARROW) uf[set][1] = *reinterpret_cast<const float4*>(&Us[m * 1024 + 256 + swz(urd0, m >> 1)]);
        };
        if (M0 == 0) read_frags(0, 0);
#pragma unroll
        for (int m = M0; m < M1; ++m) {
            const int cur = m & 1;
            if (m + 1 < 8) read_frags(m + 1, cur ^ 1);
            if constexpr (!NARROW) if (second_block) {
                acc[2 * m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][0].x, vf[cur].x, acc[2 * m][0], 0, 0, 0);
                acc[2 * m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][1].x, vf[cur].x, acc[2 * m][1], 0, 0, 0);
                acc[2 * m + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][0].z, vf[cur].z, acc[2 * m + 1][0], 0, 0, 0);
                acc[2 * m + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][1].z, vf[cur].z, acc[2 * m + 1][1], 0, 0, 0);
                acc[2 * m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][0].y, vf[cur].y, acc[2 * m][0], 0, 0, 0);
                acc[2 * m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][1].y, vf[cur].y, acc[2 * m][1], 0, 0, 0);
                acc[2 * m + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][0].w, vf[cur].w, acc[2 * m + 1][0], 0, 0, 0);
                acc[2 * m + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][1].w, vf[cur].w, acc[2 * m + 1][1], 0, 0, 0);
                continue;
            }
            acc[2 * m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][0].x, vf[cur].x, acc[2 * m][0], 0, 0, 0);
            acc[2 * m + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][0].z, vf[cur].z, acc[2 * m + 1][0], 0, 0, 0);
            acc[2 * m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][0].y, vf[cur].y, acc[2 * m][0], 0, 0, 0);
            acc[2 * m + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[cur][0].w, vf[cur].w, acc[2 * m + 1][0], 0, 0, 0);
        }
    };

    // ---- epilogue of the item under the MFMAs: output transform in registers, stores, statistics
    // LOADS: the launch has epilogue operands (skip tensor, bias, or the eval-mode scale / shift).  Two compile-time copies of the epilogue behind one
    // uniform branch, and in the copy WITH operands every load is an unseen asm load from an always-valid address (a dummy when the operand is absent or
    // the lane out of range), waited for ONCE by hand and selected afterwards.  Why: any load or store hipcc itself issues inside the persistent loop puts
    // its s_waitcnt vmcnt(0) there -- at the use, at the join of a divergent branch, or (a use under another run-time test than its load) carried around
    // the back edge to the head of EVERY pass, right behind the U image's DMA issue, where the round-4 build waited out the whole DMA latency before the
    // first MFMA of every second chunk -- and each of them drains the hand-managed prefetch queue (tools/debug/wino_waits.py lists them: none now).
    auto epilogue = [&](int tile0, int n0, int row_tile, auto loads_t) __attribute__((always_inline)) {
        constexpr bool LOADS = decltype(loads_t)::value;
        const int lane = lane_now(), j16 = lane & 15, kg = lane >> 4;      // (re-made, see the top of the kernel)
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
        const int colb[2] = {n0 + (NARROW ? 16 : 32) * ch + 4 * kg, n0 + 32 * ch + 16 + 4 * kg};
        float* __restrict__ y = p.y + (int64_t)grp * p.Ng;
        const bool has_res = AFFINE ? p.res_mode != 0 : p.res_mode == 3;
        floatx4w rvq[LOADS ? 4 : 1][2], opa[2], opb[2];          // skip values [pixel][block]; bias or scale; shift
        if constexpr (LOADS) {
            const float* dummy = p.y;                            // (16-byte aligned, always mapped; what it holds is never used)
            auto ld4 = [&](const float* ptr, floatx4w& dst) __attribute__((always_inline)) {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
            };
#pragma unroll
            for (int bb = 0; bb < NBLK; ++bb) {
                const bool col_ok = colb[bb] < p.Ng;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t pix = pix0 + (q >> 1) * g.OW + (q & 1);
                    ld4(has_res && tvalid && col_ok ? &p.residual[pix * p.ldy + grp * p.Ng + colb[bb]] : dummy, rvq[q][bb]);
                }
                if constexpr (AFFINE) {
                    ld4(col_ok ? &p.scale[grp * p.Ng + colb[bb]] : dummy, opa[bb]);
                    ld4(col_ok ? &p.shift[grp * p.Ng + colb[bb]] : dummy, opb[bb]);
                } else {
                    ld4(p.bias && col_ok ? &p.bias[grp * p.Ng + colb[bb]] : dummy, opa[bb]);
                }
            }
        }
        // statistics bookkeeping of this wavefront's 16-tile block (see below)
        const int blk0 = tile0 + 16 * tb;                        // first tile of the block
        const int cut = wg.seg_tile > blk0 && wg.seg_tile < blk0 + 16 && p.stats_extra ? wg.seg_tile : 0x7fffffff;    // first tile of the second part
        const bool first = tvalid && Tj < cut, second = tvalid && Tj >= cut;
        int nvalid = wg.MT - blk0;
        nvalid = nvalid < 0 ? 0 : (nvalid > 16 ? 16 : nvalid);
        const int n1 = cut == 0x7fffffff ? nvalid : cut - blk0;
        // One cout block at a time -- output transform A^T M A in registers, stores, statistics -- so that only 16 transformed values are live beside the
        // accumulators and the next item's prefetched patch rows (both blocks at once spilled in the 48-output form, and a spill reload inside this loop
        // makes the compiler drain the whole vector-memory queue, prefetches included, at the head of every pass).
#pragma unroll
        for (int bb = 0; bb < NBLK; ++bb) {
            float out[4][4];                                     // [pixel (py, px) = (q >> 1, q & 1)][e]
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t0[4], t1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float m0 = acc[4 * i][bb][e], m1 = acc[4 * i + 1][bb][e], m2 = acc[4 * i + 2][bb][e], m3 = acc[4 * i + 3][bb][e];
                    t0[i] = m0 + m1 + m2;
                    t1[i] = m1 - m2 - m3;
                }
                out[0][e] = t0[0] + t0[1] + t0[2];
                out[1][e] = t1[0] + t1[1] + t1[2];
                out[2][e] = t0[1] - t0[2] - t0[3];
                out[3][e] = t1[1] - t1[2] - t1[3];
            }
            const bool col_ok = colb[bb] < p.Ng;
            const bool store_ok = tvalid && col_ok && !((WINO_ABLATE & 32) && out[0][0] != 12345.678f);      // (timing experiment: the transform stays, the stores go)
            if constexpr (LOADS) {
                if (bb == 0) {                                   // the one wait of the epilogue: every operand of both blocks (issued before block 0's transform)
                    if constexpr (NBLK == 2) {
                        if constexpr (AFFINE)
                            asm volatile("s_waitcnt vmcnt(0) ; covers-loads" : "+v"(rvq[0][0]), "+v"(rvq[1][0]), "+v"(rvq[2][0]), "+v"(rvq[3][0]), "+v"(rvq[0][1]), "+v"(rvq[1][1]), "+v"(rvq[2][1]),
                                         "+v"(rvq[3][1]), "+v"(opa[0]), "+v"(opa[1]), "+v"(opb[0]), "+v"(opb[1]) :: "memory");
                        else
                            asm volatile("s_waitcnt vmcnt(0) ; covers-loads" : "+v"(rvq[0][0]), "+v"(rvq[1][0]), "+v"(rvq[2][0]), "+v"(rvq[3][0]), "+v"(rvq[0][1]), "+v"(rvq[1][1]), "+v"(rvq[2][1]),
                                         "+v"(rvq[3][1]), "+v"(opa[0]), "+v"(opa[1]) :: "memory");
                    } else {
                        if constexpr (AFFINE)
                            asm volatile("s_waitcnt vmcnt(0) ; covers-loads" : "+v"(rvq[0][0]), "+v"(rvq[1][0]), "+v"(rvq[2][0]), "+v"(rvq[3][0]), "+v"(opa[0]), "+v"(opb[0]) :: "memory");
                        else
                            asm volatile("s_waitcnt vmcnt(0) ; covers-loads" : "+v"(rvq[0][0]), "+v"(rvq[1][0]), "+v"(rvq[2][0]), "+v"(rvq[3][0]), "+v"(opa[0]) :: "memory");
                    }
                }
            }
            float o[4][4];
            if constexpr (AFFINE) {
                const bool r1 = p.res_mode == 1 && tvalid && col_ok, r2 = p.res_mode == 2 && tvalid && col_ok;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = fmaf(out[q][e], col_ok ? opa[bb][e] : 0.f, col_ok ? opb[bb][e] : 0.f);
                        v += r1 ? rvq[q][bb][e] : 0.f;
                        v = act_apply(v, p.act);
                        v += r2 ? rvq[q][bb][e] : 0.f;
                        o[q][e] = v;
                    }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[q][e] = out[q][e];
                        if constexpr (LOADS) {
                            o[q][e] += p.bias && col_ok ? opa[bb][e] : 0.f;
                            o[q][e] += has_res && tvalid && col_ok ? rvq[q][bb][e] : 0.f;
                        }
                    }
            }
            if (store_ok) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t pix = pix0 + (q >> 1) * g.OW + (q & 1);
                    store4_unseen(&y[pix * p.ldy + colb[bb]], o[q][0], o[q][1], o[q][2], o[q][3]);
                }
            }
            // BatchNorm statistics: one (S, Q, k, n) record per (16-tile block of a wavefront = 64 output pixels, cout): sums over the block's valid pixels
            // shifted by k = the block's first output.  Everything stays inside the wavefront -- k by a 16-lane broadcast, the sums over the 16 tiles by
            // DPP row rotations, the record written by the row's first lane -- no LDS, no barrier (the first version went through LDS with three barriers
            // and 128 ds_bpermute shuffles per item: 60-85 us of a 480 us launch, tools/wino_ablate.sh).  The block that contains the segment boundary
            // emits its first-segment tiles to `stats`, the rest to `stats_extra`.
            if (!AFFINE && p.stats && !(WINO_ABLATE & 64)) {
                float4* rec = reinterpret_cast<float4*>(p.stats) + ((int64_t)row_tile * 4 + tb) * p.stat_cout + grp * p.Ng;
                float4* rec2 = reinterpret_cast<float4*>(p.stats_extra) + grp * p.Ng;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // lane 0 of each 16-lane row -> the row (ds_swizzle, bit mode: lane' = lane & 0x10 inside each half of the wavefront)
                    const float k = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(out[0][e]), 0x0010));
                    float a = 0.f, qq = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float v = out[q][e] - k; a += v; qq += v * v; }
                    float a1 = first ? a : 0.f, q1 = first ? qq : 0.f, a2 = second ? a : 0.f, q2 = second ? qq : 0.f;
                    a1 += quad_dpp<0x128>(a1); q1 += quad_dpp<0x128>(q1);       // row_ror:8, 4, 2, 1: every lane of the row ends with the row's sum
                    a1 += quad_dpp<0x124>(a1); q1 += quad_dpp<0x124>(q1);
                    a1 += quad_dpp<0x122>(a1); q1 += quad_dpp<0x122>(q1);
                    a1 += quad_dpp<0x121>(a1); q1 += quad_dpp<0x121>(q1);
                    if (cut != 0x7fffffff) {
                        a2 += quad_dpp<0x128>(a2); q2 += quad_dpp<0x128>(q2);
                        a2 += quad_dpp<0x124>(a2); q2 += quad_dpp<0x124>(q2);
                        a2 += quad_dpp<0x122>(a2); q2 += quad_dpp<0x122>(q2);
                        a2 += quad_dpp<0x121>(a2); q2 += quad_dpp<0x121>(q2);
                    }
                    const int co = colb[bb] + e;
                    if (j16 == 0 && co < p.Ng && nvalid > 0) {            // (a block beyond the last tile has no record)
                        store4_unseen(reinterpret_cast<float*>(rec + co), a1, q1, k, 4.f * (float)n1);
                        if (cut != 0x7fffffff) store4_unseen(reinterpret_cast<float*>(rec2 + co), a2, q2, k, 4.f * (float)(nvalid - n1));
                    }
                }
            }
        }
    };

    if (!item_valid(lu_item)) return;
    setup_v();
    setup_u();
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I6 = std::integral_constant<int, 6>;
    using I8 = std::integral_constant<int, 8>;
    // Prologue: super-chunk 0, chunks 0 and 1 of the stream.  Queue: V x 8, U(0) x 4, U(1) x 4 -> vmcnt(4) leaves U(1) in flight.
    issue_v();
    issue_u();
    issue_u();
    asm volatile("s_waitcnt vmcnt(4) ; covers-loads" : "+v"(raw[0][0]), "+v"(raw[1][0]), "+v"(raw[2][0]), "+v"(raw[3][0]), "+v"(raw[0][1]), "+v"(raw[1][1]), "+v"(raw[2][1]), "+v"(raw[3][1]) :: "memory");
    transform(I0{});
    write_staged(0);
    __syncthreads();
    zero_acc();
    int c_item = blockIdx.x, c_chunk = 0;                       // item / chunk under the MFMAs

    int ub = 0;                                                  // U image of chunk k
    const bool late = WINO_STAGGER && wave >= 4;                 // (see the stagger note in the loop)
    // One pass = chunks k (even) and k + 1 of the stream; S = k / 2 is the super-chunk in `raw` (half 0 already staged as chunk k's V image 0).
    //   even: DMA U(k+2) | MFMAs of chunk k, transform of raw half 1 under them -> V image 1 | loads of super-chunk S+1 | wait: U(k+1) landed | barrier
    //   odd : DMA U(k+3) | MFMAs of chunk k+1; behind three quarters of them wait for super-chunk S+1, transform its half 0 -> V image 0 | barrier | epilogue?
    for (;;) {
        const bool e2 = item_valid(lu_item);                     // chunk k + 2 exists (the U cursor stands on it), and with it chunk k + 3 and super-chunk S + 1
        // (WINO_STAGGER: the two wavefronts of a SIMD, w and w + 4, run the same program between the same barriers; with the switch on wavefronts 4-7 stage
        // the next chunk behind the third quarter of their MFMAs instead of the first, so that one computes while the other stages.  Measured neutral.)
        // ---- even chunk
        if (e2 && !(WINO_ABLATE & 2)) issue_u();
        auto stage_even = [&]() __attribute__((always_inline)) {
            if (!(WINO_ABLATE & 4)) { transform(I1{}); write_staged(1); }
            if (e2 && !(WINO_ABLATE & 2)) issue_v();             // (the registers are free from here: at least a chunk until the data is needed)
        };
        if (!(WINO_ABLATE & 8)) mma_pairs(0, ub, I0{}, I2{});
        if (!late) stage_even();
        if (!(WINO_ABLATE & 8)) mma_pairs(0, ub, I2{}, I6{});
        if (late) stage_even();
        if (!(WINO_ABLATE & 8)) mma_pairs(0, ub, I6{}, I8{});
        if (e2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");    // behind U(k+1): U(k+2) x 4, super-chunk S+1 x 8
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(WINO_ABLATE & 1)) __syncthreads();
        // ---- odd chunk
        const int ub1 = ub == 2 ? 0 : ub + 1;
        if (e2 && !(WINO_ABLATE & 2)) issue_u();
        auto stage_odd = [&]() __attribute__((always_inline)) {
            if (e2) {
                // behind super-chunk S+1: U(k+3) x 4 (U(k+2), issued before it, has landed with it)
                asm volatile("s_waitcnt vmcnt(4) ; covers-loads" : "+v"(raw[0][0]), "+v"(raw[1][0]), "+v"(raw[2][0]), "+v"(raw[3][0]), "+v"(raw[0][1]), "+v"(raw[1][1]), "+v"(raw[2][1]), "+v"(raw[3][1]) :: "memory");
                if (!(WINO_ABLATE & 4)) { transform(I0{}); write_staged(0); }
            }
        };
        if (!(WINO_ABLATE & 8)) mma_pairs(1, ub1, I0{}, I2{});
        if (!late) stage_odd();
        if (!(WINO_ABLATE & 8)) mma_pairs(1, ub1, I2{}, I6{});
        if (late) stage_odd();
        if (!(WINO_ABLATE & 8)) mma_pairs(1, ub1, I6{}, I8{});
        if (!(WINO_ABLATE & 1)) __syncthreads();
        c_chunk += 2;
        if (c_chunk == wg.nchunks) {
            const int jx = c_item >> 3;
            const int row_tile = rt_first + jx / p.col_tiles;
            if (!(WINO_ABLATE & 16)) {
                if (AFFINE || p.res_mode == 3 || p.bias) epilogue(row_tile * 64, (jx % p.col_tiles) * 64, row_tile, std::true_type{});
                else epilogue(row_tile * 64, (jx % p.col_tiles) * 64, row_tile, std::false_type{});
            }
            zero_acc();
            c_chunk = 0;
            c_item += gridDim.x;
        }
        if (!e2) break;
        ub = ub1 == 2 ? 0 : ub1 + 1;
    }
}

// ---- filter transform: U = G g G^T (row 3 negated) of every (output row R, reduction channel k) of a group, written in the LDS image order
// of the main kernel: fw[group][R / 64][chunk][image of 16 x 512 floats] (chunks of 8 channels in super-chunk order, see below).  pass 0: g[r][s] = w[(group, R)][r][s][k] (the filter as stored,
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
        // channel k -> chunk 2 (k / 16) + ((k / 4) & 1), in-chunk position 4 ((k / 8) & 1) + (k & 3): the order the main kernel's super-chunk loads produce
        const int chunk = 2 * (k >> 4) + ((k >> 2) & 1), kc = 4 * ((k >> 3) & 1) + (k & 3);
        float* img = fw + (((int64_t)grp * rblocks + R / 64) * kchunks + chunk) * WN_IMG;
        const int row = R & 63, kgq = kc >> 1, e = kc & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s02 = hg[i][0] + hg[i][2];
            const float u[4] = {hg[i][0], 0.5f * (s02 + hg[i][1]), 0.5f * (s02 - hg[i][1]), hg[i][2]};
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) img[wino_slot_off((4 * i + jx) >> 1, row, kgq) + 2 * (jx & 1) + e] = u[jx];
        }
    }
}

// ================================================================================================================================
// Filter gradient of the same convolutions by the transposed algorithm, F(3x3, 2x2):
//   dU_xi[cout][cin] = sum over tiles of dM_xi[tile][cout] * V_xi[tile][cin],   dM = A dY A^T (4x4 from the 2x2 tile of dy), V = B^T d B as above,
//   dW = G^T dU G   (per (cout, cin): 4x4 -> 3x3, linear, so it is applied once after the sum over tiles)
// again 16 multiplies per tile and channel pair where the direct sum has 36.  A workgroup (8 wavefronts) owns a 64 cout x 64 cin block of all 16
// positions over a slab of tiles (split-K: the partial sums go to a workspace, a second small kernel adds the slabs in a fixed order and applies
// G^T . G, writing one filter-shaped slab that the library's deferred reduction adds to the gradient like any other filter gradient).
//   stage   chunk = 8 tiles; thread (patch row i = t&3, channel vector v = (t>>2)&15, tile = t>>6 = its wavefront): one row of the tile's 4x4 input
//           patch (four 16-byte loads: the 16 vectors of a pixel are 256 contiguous bytes) and the tile's 2x2 block of dy (four loads, the same for
//           the four lanes of a quad); horizontal halves in-thread, vertical halves: V across the quad by DPP, dM per lane from its coefficients;
//           both images [position][tile][64 channels] (rows 3 / columns 3 negated on both sides: the products do not notice);
//   MFMA    wavefront w owns positions 2w, 2w+1 for the whole 64 x 64 block: reduction index = tile (4 per v_mfma_f32_16x16x4_f32), a lane's
//           ds_read_b128 of a tile row hands it four channel sub-blocks at once (channel 4 * (lane & 15) + a <-> sub-block a), 16 MFMAs per pair
//           of reads, 64 MFMAs per chunk; accumulators 2 x 16 sub-blocks x 4 registers.

constexpr int WW_IMG = 16 * 8 * 64;      // floats per image (dM or V): 16 positions x 8 tiles x 64 channels = 32 KB

// NB = input-channel sub-blocks of a workgroup's block: 4 (64 input channels: channel 4 (lane & 15) + b <-> sub-block b, one ds_read_b128) or 3 (48 input
// channels, EfficientNetV2-S's 48 -> 192 expand convs: channel 16 b + (lane & 15), three ds_read_b32 -- no matrix work on 16 padding channels)
#ifndef WW_STAGE_AT
#define WW_STAGE_AT 3          // MFMA quarters of a chunk in front of the next chunk's staging (4 = staging behind all of them)
#endif
template <int NB>
__global__ __launch_bounds__(512, 2) void wino_wgrad_kernel(const WinoWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];       // [buffer][dM image | V image]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int grp = blockIdx.z;
    const int b = blockIdx.x;
    const int split = b % p.splits, cb = b / p.splits;
    const int ci_blk = cb % p.ci_blocks, co_blk = cb / p.ci_blocks;
    const int cin0 = ci_blk * 64, cout0 = co_blk * 64;
    const BufRsrc xr = make_rsrc(p.x + (int64_t)grp * p.Cg, p.x_bytes - (int64_t)grp * p.Cg * 4);
    const BufRsrc yr = make_rsrc(p.dy + (int64_t)grp * p.Ng, p.dy_bytes - (int64_t)grp * p.Ng * 4);
    const int si = t & 3, sv = (t >> 2) & 15;
    const bool civ = cin0 + 4 * sv < p.Cg, cov = cout0 + 4 * sv < p.Ng;
    const float sxv = si == 1 ? 1.f : -1.f;                    // V's vertical pass (row 3 negated), as in wino_nt_kernel
    const float c0 = si == 3 ? 0.f : 1.f, c1 = si == 0 ? 0.f : (si == 2 ? -1.f : 1.f);      // dM rows: dy0, dy0 + dy1, dy0 - dy1, +dy1 (row 3 negated)
    // LDS float offset of this thread's vector inside an image: [position][tile][vector ^ 2 i] (the XOR spreads the four patch rows of a quad, which write
    // four different position planes at the same offset, over the banks; a read of one position sees a fixed permutation of its 16 vectors)
    const int wvec = ((sv ^ (2 * si)) << 2) + wave * 64 + (4 * si) * 512;       // + jx * 512
    const int tile_lo = split * p.tiles_per_split;
    int tile_hi = tile_lo + p.tiles_per_split;
    if (tile_hi > p.MT) tile_hi = p.MT;
    const int nchunks = tile_lo < tile_hi ? (tile_hi - tile_lo + 7) >> 3 : 0;

    float xr_[4][4], dr_[4][4];
    // A chunk's tile is its WAVEFRONT's (tile_lo + 8 c + wave): decomposing it is scalar work, and the lane-dependent part of every offset -- patch row,
    // channel vector -- is a constant of the launch.  fp32 MFMAs run on the vector ALU (tools/micro/coissue_probe.hip), and the first version of this
    // loader paid 28 quarter-rate vector multiplies + ~50 vector instructions per chunk beside its 64 MFMAs (a third of the chunk): now one vector
    // add per load, the validity of rows / columns / the slab's end as scalar conditions on precomputed lane constants.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned ldx4 = (unsigned)p.ldx * 4u, lddy4 = (unsigned)p.lddy * 4u;
    const unsigned lane_x = (unsigned)(((si - 1) * p.W - 1) * p.ldx * 4 + (cin0 + 4 * sv) * 4);      // (may wrap: the sum with the tile's pixel offset does not, where the load is valid)
    const unsigned lane_y = (unsigned)((cout0 + 4 * sv) * 4);
    const bool l_top = si == 0, l_bot = si == 3;
    auto load_chunk = [&](int c) __attribute__((always_inline)) {
        const int T = tile_lo + c * 8 + wave_u;                    // (scalar)
        const bool tv = T < tile_hi;
        const unsigned Tn = (unsigned)(tv ? T : 0);
        const unsigned n = fast_div(Tn, p.div_tpi_mul, p.div_tpi_shr);
        const unsigned rem = Tn - n * (unsigned)p.TPI;
        const unsigned ty = fast_div(rem, p.div_tx_mul, p.div_tx_shr), tx = rem - ty * (unsigned)p.TX;
        const unsigned pix = (n * (unsigned)p.H + 2u * ty) * (unsigned)p.W + 2u * tx;      // the tile's first own pixel (scalar)
        const bool top = ty == 0, bot = (int)ty == (p.H >> 1) - 1, left = tx == 0, right = (int)tx == p.TX - 1;
        const bool rowok = tv & civ & !(l_top & top) & !(l_bot & bot);
        const unsigned xb = pix * ldx4 + lane_x;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const bool ok = rowok & !(c4 == 0 && left) & !(c4 == 3 && right);
            buf_load<4>(xr, ok ? xb + (unsigned)c4 * ldx4 : kOob, xr_[c4]);
        }
        const unsigned yb = pix * lddy4 + lane_y;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            buf_load<4>(yr, tv & cov ? yb + (unsigned)((q >> 1) * p.W + (q & 1)) * lddy4 : kOob, dr_[q]);
    };
    auto stage_chunk = [&](int buf) __attribute__((always_inline)) {
        float* Ds = smem + buf * 2 * WW_IMG;
        float* Vs = Ds + WW_IMG;
        float ov[4][4], od[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h0 = xr_[0][e] - xr_[2][e], h1 = xr_[1][e] + xr_[2][e], h2 = xr_[2][e] - xr_[1][e], h3 = xr_[3][e] - xr_[1][e];      // (column 3 negated)
            ov[0][e] = fmaf(sxv, quad_dpp<0x5A>(h0), h0);
            ov[1][e] = fmaf(sxv, quad_dpp<0x5A>(h1), h1);
            ov[2][e] = fmaf(sxv, quad_dpp<0x5A>(h2), h2);
            ov[3][e] = fmaf(sxv, quad_dpp<0x5A>(h3), h3);
            const float r0 = fmaf(c1, dr_[2][e], c0 * dr_[0][e]), r1 = fmaf(c1, dr_[3][e], c0 * dr_[1][e]);          // this lane's row of A dY: columns 0, 1
            od[0][e] = r0; od[1][e] = r0 + r1; od[2][e] = r0 - r1; od[3][e] = r1;                                      // (column 3 negated)
        }
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            *reinterpret_cast<float4*>(&Ds[wvec + jx * 512]) = make_float4(od[jx][0], od[jx][1], od[jx][2], od[jx][3]);
            *reinterpret_cast<float4*>(&Vs[wvec + jx * 512]) = make_float4(ov[jx][0], ov[jx][1], ov[jx][2], ov[jx][3]);
        }
    };
    // ---- MFMA role: positions 2 wave, 2 wave + 1; lane = (vector idx = lane & 15, tile-in-group kg = lane >> 4)
    const int idx = lane & 15, kg = lane >> 4;
    floatx4w acc[2][4][NB];
#pragma unroll
    for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bb = 0; bb < NB; ++bb)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[x2][a][bb][e] = 0.f;
    // quarters [q0, q1) of a chunk's MFMAs: quarter = (position x2, k-step pair s4)
    auto mma_chunk = [&](int buf, auto q0_t, auto q1_t) __attribute__((always_inline)) {
        constexpr int Q0 = decltype(q0_t)::value, Q1 = decltype(q1_t)::value;
        const float* Ds = smem + buf * 2 * WW_IMG;
        const float* Vs = Ds + WW_IMG;
#pragma unroll
        for (int qq = Q0; qq < Q1; ++qq) {
            const int x2 = qq >> 1;
            const int xi = 2 * wave + x2;
            const int rd = xi * 512 + ((idx ^ (2 * (xi >> 2))) << 2) + kg * 64;
            {
                const int s4 = qq & 1;
                const float4 af = *reinterpret_cast<const float4*>(&Ds[rd + s4 * 256]);
                const float av[4] = {af.x, af.y, af.z, af.w};
                float bv[NB];
                if constexpr (NB == 4) {
                    const float4 bf = *reinterpret_cast<const float4*>(&Vs[rd + s4 * 256]);
                    bv[0] = bf.x; bv[1] = bf.y; bv[2] = bf.z; bv[3] = bf.w;
                } else {
#pragma unroll
                    for (int bb = 0; bb < NB; ++bb) {              // channel 16 bb + idx: vector 4 bb + (idx >> 2) (at its swizzled place), element idx & 3
                        const int vecp = (4 * bb + (idx >> 2)) ^ (2 * (xi >> 2));
                        bv[bb] = Vs[xi * 512 + (vecp << 2) + (idx & 3) + kg * 64 + s4 * 256];
                    }
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int bb = 0; bb < NB; ++bb) acc[x2][a][bb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[bb], acc[x2][a][bb], 0, 0, 0);
            }
        }
    };
    if (nchunks > 0) {
        using Q0 = std::integral_constant<int, 0>;
        using Q3 = std::integral_constant<int, WW_STAGE_AT>;
        using Q4 = std::integral_constant<int, 4>;
        load_chunk(0);
        stage_chunk(0);
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            if (c + 1 < nchunks) load_chunk(c + 1);
            __builtin_amdgcn_sched_barrier(0);                   // (left alone the scheduler sinks the loads down to their first use)
            // The next chunk is staged (transforms + LDS stores: VALU) between this chunk's MFMA quarters, not behind them: the two wavefronts of a SIMD
            // run the same program between two barriers, so a VALU block at the end is a stretch in which neither of them feeds the matrix pipe
            mma_chunk(c & 1, Q0(), Q3());
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < nchunks) stage_chunk((c + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk(c & 1, Q3(), Q4());
            __syncthreads();
        }
    }
    // ---- partial sums: part[split][position][cout][cin]; lane holds cout = cout0 + 16 (lane >> 4) + 4 e + a, cins = cin0 + 4 (lane & 15) + {0..3} (sub-block bb)
    float* part = p.part + (int64_t)split * 16 * p.cout_all * p.Cg;
#pragma unroll
    for (int x2 = 0; x2 < 2; ++x2) {
        const int xi = 2 * wave + x2;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = cout0 + 16 * kg + 4 * e + a;
                if (co >= p.Ng) continue;
                float* dst = &part[((int64_t)xi * p.cout_all + grp * p.Ng + co) * p.Cg];
                if constexpr (NB == 4) {
                    const int ci = cin0 + 4 * idx;
                    if (ci < p.Cg) *reinterpret_cast<float4*>(&dst[ci]) = make_float4(acc[x2][a][0][e], acc[x2][a][1][e], acc[x2][a][2][e], acc[x2][a][3][e]);
                } else {
#pragma unroll
                    for (int bb = 0; bb < NB; ++bb) {
                        const int ci = cin0 + 16 * bb + idx;
                        if (ci < p.Cg) dst[ci] = acc[x2][a][bb][e];
                    }
                }
            }
    }
}

// sum of the slabs (fixed order) and dW = G^T dU G per (cout, cin), written as ONE filter-shaped slab [cout][3][3][Cg]
// G^T U G of one (cout, cin) pair's sixteen sums -> the nine filter taps of `slab`
__device__ __forceinline__ void wino_wgrad_store_taps(const float (&u)[16], float* __restrict__ slab, int64_t i, int Cg) {
    float hcol[3][4];                                            // G^T U: rows (1, .5, .5, 0), (0, .5, -.5, 0), (0, .5, .5, 1)
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) {
        const float u0 = u[jx], u1 = u[4 + jx], u2 = u[8 + jx], u3 = u[12 + jx];
        hcol[0][jx] = u0 + 0.5f * (u1 + u2);
        hcol[1][jx] = 0.5f * (u1 - u2);
        hcol[2][jx] = 0.5f * (u1 + u2) + u3;
    }
    const int co = (int)(i / Cg), ci = (int)(i - (int64_t)co * Cg);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float g0 = hcol[r][0] + 0.5f * (hcol[r][1] + hcol[r][2]);
        const float g1 = 0.5f * (hcol[r][1] - hcol[r][2]);
        const float g2 = 0.5f * (hcol[r][1] + hcol[r][2]) + hcol[r][3];
        float* dst = slab + ((int64_t)co * 9 + r * 3) * Cg + ci;
        dst[0] = g0; dst[Cg] = g1; dst[2 * Cg] = g2;
    }
}
// Sum of the split-K slabs + output transform.  SL = 1: a thread adds all the slabs of its (cout, cin) pair -- fine when cout x cin fills the chip and
// the splits are few (RepVGG's wide layers: 4 splits).  EfficientNetV2-S's 48 -> 192 and 64 -> 256 layers are the opposite: 12 k - 16 k pairs and 64 - 85
// splits, so that form ran 36 - 64 workgroups each walking a chain of 64 - 85 dependent rounds of 16 loads (225 us for 67 MB, kernel trace of the C2
// step).  SL = 16: sixteen wavefronts per 64 pairs take every sixteenth slab each and meet in LDS (fixed order: the result does not depend on timing).
template <int SL>
__global__ __launch_bounds__(SL == 1 ? 256 : 64 * SL) void wino_wgrad_finish_kernel(const float* __restrict__ part, float* __restrict__ slab, int splits, int cout_all, int Cg) {
    const int64_t total = (int64_t)cout_all * Cg;
    if (SL == 1) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
            float u[16];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) u[xi] = 0.f;
            for (int s = 0; s < splits; ++s)
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) u[xi] += part[((int64_t)s * 16 + xi) * total + i];
            wino_wgrad_store_taps(u, slab, i, Cg);
        }
        return;
    }
    __shared__ float red[SL > 1 ? SL * 16 * 64 : 1];             // [split lane][position][pair]
    const int il = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + il;
    const bool ok = i < total;
    float u[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) u[xi] = 0.f;
    if (ok)
        for (int s = sl; s < splits; s += SL)
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) u[xi] += part[((int64_t)s * 16 + xi) * total + i];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) red[(sl * 16 + xi) * 64 + il] = u[xi];
    __syncthreads();
    // wavefront sl adds position sl over the split lanes and leaves it in lane 0's plane; the first wavefront then holds all sixteen
    static_assert(SL == 1 || SL == 16, "one wavefront per Winograd position in the second round");
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < SL; ++k) t += red[(k * 16 + sl) * 64 + il];
    __syncthreads();
    red[sl * 64 + il] = t;
    __syncthreads();
    if (sl != 0 || !ok) return;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) u[xi] = red[xi * 64 + il];
    wino_wgrad_store_taps(u, slab, i, Cg);
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
// SUHPE_WINOGRAD=0: the tiles are never eligible; 2: the heuristics pick them wherever the geometry allows (the parity suite's second pass:
// tests/test_gpu_winograd.py runs the network-level tests that way); default 1: only where a caller / the tile database pinned them
static const int kWinoMode = getenv("SUHPE_WINOGRAD") ? atoi(getenv("SUHPE_WINOGRAD")) : 1;
static const bool kWinoEnabled = kWinoMode != 0;
bool wino_preferred() { return kWinoMode == 2; }

// geometry: 3x3 / stride 1 / pad 1 forward (mode 0) or input gradient (mode 1) on even output sizes, 16-byte channel vectors on both sides
bool wino_geom_ok(const GatherGeom& g, int Ng) {
    return kWinoEnabled && g.om == 1 && g.R == 3 && g.S == 3 && g.stride == 1 && g.pad == 1 && g.GH == g.OH && g.GW == g.OW && g.OH % 2 == 0 && g.OW % 2 == 0 &&
           g.M < (1LL << 31) && g.Cg % 4 == 0 && Ng % 4 == 0 && g.ldg % 4 == 0 &&
           g.M * (int64_t)g.ldg * 4 < (1LL << 31);      // 32-bit byte offsets against a descriptor of < 2 GiB: larger inputs take the direct tiles
}
// epilogue forms the kernel has: bias, skip add, statistics records; or (AFFINE instantiations) the folded eval-mode BatchNorm + activation + skip
static bool wino_params_ok(const NtParams& p) {
    const bool affine = p.scale != nullptr;                      // eval-mode form: scale AND shift, activation, skip before / after it; no bias, no statistics
    const bool epi_ok = affine ? (p.shift && !p.bias && !p.stats && p.res_mode >= 0 && p.res_mode <= 2 && aligned16(p.scale) && aligned16(p.shift))
                               : (p.res_mode == 0 || p.res_mode == 3);
    return wino_geom_ok(p.g, p.Ng) && epi_ok && !p.y2 && !p.a_scale && !p.o_rowscale && p.ldy % 4 == 0 && p.g.M * (int64_t)p.ldy * 4 < (1LL << 31) &&
           p.fw && aligned16(p.y) && aligned16(p.x) && aligned16(p.fw) && (!p.residual || aligned16(p.residual)) && (!p.bias || aligned16(p.bias));
}

int64_t wino_filter_floats(int groups, int NR, int NK) { return (int64_t)groups * ceil_div(NR, 64) * 2 * ceil_div(NK, 16) * WN_IMG; }

void launch_wino_filter(const float* w, float* fw, int groups, int NR, int NK, int ng_w, int cg_w, int pass, hipStream_t st) {
    const int rblocks = (int)ceil_div(NR, 64), kchunks = 2 * (int)ceil_div(NK, 16);
    const int64_t total = (int64_t)groups * rblocks * 64 * kchunks * 8;
    hipLaunchKernelGGL(wino_filter_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, w, fw, groups, NR, NK, ng_w, cg_w, pass, rblocks, kchunks);
}

bool launch_wino(const NtParams& p_in, int groups, hipStream_t st) {
    if (!wino_params_ok(p_in)) return false;
    if (launch_wino_small(p_in, groups, st)) { note_kernel("wino_small_kernel"); return true; }        // few channels on both sides: the register-resident form (conv_wino_small.hip)
    NtParams p = p_in;
    const GatherGeom& g = p.g;
    WinoGeom wg;
    memset(&wg, 0, sizeof(wg));
    wg.TX = g.OW / 2; wg.TPI = (g.OH / 2) * wg.TX;
    magic_div((unsigned)wg.TPI, &wg.div_tpi_mul, &wg.div_tpi_shr);
    magic_div((unsigned)wg.TX, &wg.div_tx_mul, &wg.div_tx_shr);
    wg.MT = g.N * wg.TPI;
    wg.nchunks = 2 * ((g.Cg + 15) / 16);               // super-chunks of 16 channels, two chunks each
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
    const size_t lds = (size_t)WN_LDS_FLOATS * sizeof(float);
    auto go = [&](auto mode) {
        constexpr int M = decltype(mode)::value;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_nt_kernel<M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(wino_nt_kernel<M>, grid, block, lds, st, p, wg);
    };
    const int form = p.Ng <= 32 ? 2 : (p.Ng <= 48 ? 1 : 0);
    if (p.scale) {
        if (form == 2) go(std::integral_constant<int, 6>()); else if (form == 1) go(std::integral_constant<int, 5>()); else go(std::integral_constant<int, 4>());
    } else {
        if (form == 2) go(std::integral_constant<int, 2>()); else if (form == 1) go(std::integral_constant<int, 1>()); else go(std::integral_constant<int, 0>());
    }
    note_kernel("wino_nt_kernel<64,64>");
    return true;
}

// ---- filter gradient
bool wino_wgrad_geom_ok(int n, int h, int w, int ho, int wo, int cin, int cout, int r, int sfilt, int stride, int pad, int groups) {
    return kWinoEnabled && r == 3 && sfilt == 3 && stride == 1 && pad == 1 && ho == h && wo == w && h % 2 == 0 && w % 2 == 0 && groups > 0 && cin % groups == 0 &&
           cout % groups == 0 && (cin / groups) % 4 == 0 && (cout / groups) % 4 == 0 && cin % 4 == 0 && cout % 4 == 0 && (int64_t)n * h * w < (1LL << 31) &&
           (int64_t)n * h * w * cin * 4 < (1LL << 31) && (int64_t)n * h * w * cout * 4 < (1LL << 31);      // x and dy extents: 32-bit byte offsets
}
static int wino_wgrad_splits(int n, int h, int w, int cin, int cout, int groups) {
    const int cg = cin / groups, ng = cout / groups;
    if (const int ss = wino_small_wgrad_splits(n, h, w, cg, ng, groups)) return ss;      // few channels on both sides: conv_wino_small.hip
    const int64_t blocks = ceil_div(cg, 64) * ceil_div(ng, 64) * groups;
    const int64_t MT = (int64_t)n * (h / 2) * (w / 2);
    int64_t splits = 256 / blocks;                            // ONE generation of workgroups (a workgroup per CU: 128 KB of LDS); never a 257th
    const int64_t max_by_k = ceil_div(MT, 64);                // at least 8 chunks per slab
    if (splits > max_by_k) splits = max_by_k;
    if (splits < 1) splits = 1;
    return (int)splits;
}
int64_t wino_wgrad_workspace_floats(int n, int h, int w, int cin, int cout, int groups) {
    const int cg = cin / groups;
    return (int64_t)cout * 9 * cg + (int64_t)wino_wgrad_splits(n, h, w, cin, cout, groups) * 16 * cout * cg;
}
// x (n,h,w,cin), dy (n,h,w,cout) -> workspace: [filter-shaped slab | partial sums]; the slab is complete when the second kernel has run
void launch_wino_wgrad(const float* x, const float* dy, float* workspace, int n, int h, int w, int cin, int cout, int groups, hipStream_t st) {
    WinoWgradParams p;
    memset(&p, 0, sizeof(p));
    const int cg = cin / groups, ng = cout / groups;
    p.x = x; p.dy = dy; p.part = workspace + (int64_t)cout * 9 * cg;
    p.x_bytes = (int64_t)n * h * w * cin * 4; p.dy_bytes = (int64_t)n * h * w * cout * 4;
    p.N = n; p.H = h; p.W = w; p.ldx = cin; p.lddy = cout; p.Cg = cg; p.Ng = ng;
    p.TX = w / 2; p.TPI = (h / 2) * p.TX; p.MT = n * p.TPI;
    magic_div((unsigned)p.TPI, &p.div_tpi_mul, &p.div_tpi_shr);
    magic_div((unsigned)p.TX, &p.div_tx_mul, &p.div_tx_shr);
    p.ci_blocks = (int)ceil_div(cg, 64); p.co_blocks = (int)ceil_div(ng, 64);
    p.splits = wino_wgrad_splits(n, h, w, cin, cout, groups);
    p.tiles_per_split = (int)(ceil_div(ceil_div(p.MT, p.splits), 8) * 8);
    p.cout_all = cout;
    const dim3 grid((unsigned)(p.ci_blocks * p.co_blocks * p.splits), 1, groups), block(512);
    const size_t lds = (size_t)4 * WW_IMG * sizeof(float);
    if (wino_small_wgrad_splits(n, h, w, cg, ng, groups)) {
        launch_wino_small_wgrad(p, groups, st);
    } else if (cg % 64 == 48 && p.ci_blocks == 1) {                     // (one block of 48 input channels)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_wgrad_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(wino_wgrad_kernel<3>, grid, block, lds, st, p);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wino_wgrad_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(wino_wgrad_kernel<4>, grid, block, lds, st, p);
    }
    const int64_t total = (int64_t)cout * cg;
    if (p.splits >= 16 && total <= 64 * 1024)                    // many slabs over few pairs: sixteen wavefronts share a pair's chain
        hipLaunchKernelGGL(wino_wgrad_finish_kernel<16>, dim3((unsigned)ceil_div(total, 64)), dim3(1024), 0, st, p.part, workspace, p.splits, cout, cg);
    else
        hipLaunchKernelGGL(wino_wgrad_finish_kernel<1>, dim3(grid_for(total, 256)), dim3(256), 0, st, p.part, workspace, p.splits, cout, cg);
}

}  // namespace suhpe
