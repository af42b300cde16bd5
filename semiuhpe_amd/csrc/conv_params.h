// Parameter blocks shared by the dense-convolution translation units (conv_igemm.hip, conv_wino.hip).
#pragma once
#include "common.h"
#include <cstring>

namespace suhpe {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;          // reduction chunk staged per LDS tile
constexpr int LDK = BK + 4;     // padded LDS row (floats): 144 B keeps 16-B alignment and spreads banks

// activation of the input transforms (XF loaders): SiLU on the hardware exponential (v_exp_f32, ~1 ulp of exp2) -- the accurate
// expf / erff paths of act_apply cost the GEMM loaders dozens of VGPRs and scratch; none / ReLU / SiLU are what the producers of
// a pointwise conv's input use (torchvision MBConv / FusedMBConv)
__device__ __forceinline__ float xf_act(float u, int act) {
    const float s = u * __builtin_amdgcn_rcpf(1.f + __expf(-u));       // v_exp_f32 + v_rcp_f32: ~1 ulp each
    return act == SUHPE_ACT_SILU ? s : (act == SUHPE_ACT_RELU ? fmaxf(u, 0.f) : u);
}

struct GatherGeom {
    int N, GH, GW;       // gathered tensor spatial dims
    int OH, OW;          // GEMM-row enumeration: m = (n, oh, ow)
    int ldg;             // gathered tensor pixel stride (elements)
    int Cg;              // channels per group in the gathered tensor (k = tap*Cg + c)
    int R, S;
    int stride, pad;
    int mode;            // 0: gh = oh*stride - pad + r      (forward / wgrad gather of x)
                         // 1: gh = (oh + pad - r)/stride     (dgrad gather of dy; only exact multiples)
    int K;               // R*S*Cg
    int64_t M;           // N*OH*OW
    // affine form used by the fast paths: coordinate = o*gm + sgn*tap + off
    //   forward (any stride): gm = stride, sgn = +1, off = -pad ;  dgrad stride 1: gm = 1, sgn = -1, off = +pad
    int gm, sgn, off;
    // magic numbers for n / (OH*OW) and rem / OW on dividends < 2^31 (q = umulhi(n, mul) >> shr)
    unsigned div_hw_mul, div_hw_shr, div_w_mul, div_w_shr;
    // MODE 3 (dgrad of a strided conv, one launch per output parity class (ph,pw)): rows enumerate the sub-grid
    // (n, a, b) -> output pixel (a*om+ph, b*om+pw) of a (YH,YW) tensor; taps enumerate only the filter rows/cols
    // r = r0 + om*th, s = s0 + om*tw that hit this parity (R,S hold the enumerated counts, Sfull the filter width);
    // gathered coordinate = a - th + off_h, b - tw + off_w.
    int om, ph, pw, YH, YW, r0, s0, Sfull, off_h, off_w;
};

__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned mul, unsigned shr) {
    return mul ? (__umulhi(n, mul) >> shr) : n;       // mul == 0 encodes divisor 1
}

struct NtParams {
    GatherGeom g;
    const float* x;      // gathered tensor
    const float* f;      // filter rows [Ng_total][K]
    const float* bias;   // per output column or null
    float* y;            // [M][ldy]
    int Ng;              // output columns per group
    int ldy;
    int col_tiles;       // ceil(Ng / BN), set by the launcher
    int ldf;             // filter row length in elements (== g.K except in MODE 3)
    int64_t x_bytes, f_bytes;   // extents of the gathered tensor and of the filter matrix (buffer descriptors)
    const float* scale;  // optional per-channel affine + activation + residual in the epilogue (eval-mode BatchNorm folded into
    const float* shift;  // the conv): y = act(acc*scale + shift [+ res]) [+ res], res_mode 1 = before / 2 = after the activation
    const float* residual;
    int act, res_mode;
    float* stats;        // optional BatchNorm statistics from the epilogue: records [stat_gx][stat_cout][4] = (S, Q, k, n)
    float* stats_extra;  // [stat_cout][4]: second record of the one row tile that straddles seg_row (see below), or NULL
    int64_t seg_row;     // batch segments (two sub-batches normalised separately): rows < seg_row belong to the first; a tile
                         // that contains the boundary emits its first-segment rows to `stats` and the rest to `stats_extra`
    int stat_cout;       // channels of the whole output (record pitch)
    int stat_gx;         // row tiles = ceil(M / BM)
    // optional transform of the GATHERED operand while it is staged (pointwise GEMMs only, XF instantiations):
    //   a[m][k] <- act(a[m][k] * a_scale[s][k] + a_shift[s][k]) * a_rowscale[m / a_hw][k]      (s = segment of row m, a_rowscale may be NULL)
    // i.e. the producer's BatchNorm + activation (+ SqueezeExcitation scale) applied by the consumer's loader: the activated
    // tensor is never written to or read from HBM (a_scale / a_shift are [segments][K] as suhpe_bn_affine_from_stats writes them)
    const float* a_scale = nullptr;
    const float* a_shift = nullptr;
    const float* a_rowscale = nullptr;
    int a_act = 0, a_hw = 1;
    int64_t a_seg_row = 0;
    unsigned a_div_mul = 0, a_div_shr = 0;
    // activation epilogues of the pointwise ACT2 instantiations (Linear -> activation chains, TinyViT's Mlp):
    //   y2 != NULL : y = acc + bias (the pre-activation, kept for the backward) AND y2 = act(y), from one epilogue;
    //   res_mode 4 : y = acc * act'(residual)  -- an input gradient multiplied by the activation's derivative at the saved pre-activation
    float* y2 = nullptr;
    //   res_mode 5 : the conv's INPUT was act(BatchNorm(residual)) in train mode: y = g = acc * act'(residual * scale[s][c] + shift[s][c]), the
    //                gradient of that BatchNorm's output, AND its backward sums (sum g, sum g xhat), xhat = (residual - bn_mean[s][c]) bn_rstd[s][c],
    //                per (row tile, channel) as float2 records in `stats`: [plane s][row tile][channel] (s = batch segment of the row, seg_row)
    const float* bn_mean = nullptr;
    const float* bn_rstd = nullptr;
    // residual connection with a per-sample keep scale (timm DropPath / torchvision StochasticDepth 'row') in the epilogue of the plain igemm
    // tiles: y = (acc + bias) * o_rowscale[row / rows per sample] + residual   (res_mode 3; o_rowscale == NULL: scale 1)
    const float* o_rowscale = nullptr;
    unsigned o_div_mul = 0, o_div_shr = 0;
    // Winograd-domain filter of the conv (suhpe_conv2d_wino_filter), or NULL: the Winograd tile then transforms the filter rows itself
    const float* fw = nullptr;
    // start-up stagger (see stagger_wait): workgroups of the first resident generation delay their first barrier by
    // (hardware wave slot) * stagger quanta so that co-resident workgroups run out of phase
    int stagger = 0, stagger_first = 0;
};


inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline void magic_div(unsigned d, unsigned* mul, unsigned* shr) {
    if (d <= 1) { *mul = 0; *shr = 0; return; }
    unsigned lg = 0;
    while ((1ull << lg) < d) ++lg;
    const unsigned pw = 31 + lg;
    *mul = (unsigned)(((1ull << pw) + d - 1) / d);
    *shr = pw - 32;
}


// ---- shared by the Winograd translation units (conv_wino.hip, conv_wino_small.hip)
constexpr int WN_XI = 512;               // floats per transform position of one LDS image: 64 rows x 8 channels
constexpr int WN_IMG = 16 * WN_XI;       // floats per image (V or U) = 32 KB

// A 16-byte store hipcc does not see.  The compiler waits (s_waitcnt vmcnt(0)) before it reuses the data registers of a store IT issued, and inside the
// persistent chunk loop every such wait also drains the hand-managed prefetch queue: the round-4 epilogue sat through eight of them per item (one per
// statistics record round).  The hardware reads a store's data registers right after issue: two wait states (s_nop 1) cover it.
__device__ __forceinline__ void store4_unseen(float* ptr, float a, float b, float c, float d) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(ptr), "v"(v) : "memory");
}

template <int CTRL> __device__ __forceinline__ float quad_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, false));
}

// LDS image of one operand chunk: 8 position PAIRS m = xi >> 1, each [64 rows][4 slots of 16 bytes]; slot kg of a row holds channels (2 kg, 2 kg + 1)
// of position 2 m followed by the same channels of position 2 m + 1, so that ONE ds_read_b128 hands a lane (row = lane & 15, k-group kg = lane >> 4)
// the operands of four MFMAs (two k-steps of two positions).  Swizzle (i = m >> 1 = the patch row that writes the pair):
//   slot (row, kg) of pair m sits at row ^ (i & 1), slot kg ^ g(row) ^ (i >> 1), g(row) = (-(row >> 2)) & 3.
// Reads (ds_read_b128: 16-lane groups {rows 0-3, 12-15 of k-group a, rows 4-11 of k-group a + 1}) then cover the 64 banks exactly once, writes
// (ds_write_b128: 8-lane groups = one row, 2 channel vectors, 4 patch rows) the 32 banks of their window.  (A first version used ds_read_b64 per
// position: the compiler fused pairs of them into ds_read2st64_b64, which runs at half the LDS rate and broke the swizzle: 45 % of the LDS cycles
// were bank conflicts.)
__host__ __device__ __forceinline__ int wino_slot_off(int m, int row, int kg) {
    const int i = m >> 1;
    return m * 1024 + ((row ^ (i & 1)) << 4) + (((kg ^ ((-(row >> 2)) & 3) ^ (i >> 1)) & 3) << 2);
}


// conv_wino.hip: Winograd F(2x2, 3x3) kernel for 3x3 / stride 1 / pad 1 forward and input-gradient GEMMs
bool wino_geom_ok(const GatherGeom& g, int Ng);
bool wino_preferred();          // SUHPE_WINOGRAD=2: unpinned eligible shapes run the Winograd tiles
bool launch_wino(const NtParams& p, int groups, hipStream_t st);
// conv_wino_small.hip: the same arithmetic with the transformed filter in registers (16 / 24 / 32 input channels, at most 32 output channels; training forms);
// false = not this form's shape / epilogue (launch_wino, which has checked wino_params_ok, goes on to wino_nt_kernel)
bool launch_wino_small(const NtParams& p, int groups, hipStream_t st);
bool wino_small_desc_ok(int cg, int ng, int64_t M, int oh, int ow);      // the shapes of that form (per-group channels in / out, output rows, output image)
int64_t wino_filter_floats(int groups, int NR, int NK);
// w: the filter as stored, [groups * ng_w][3][3][cg_w]; pass 0: NR = ng_w rows, NK = cg_w channels; pass 1 (input gradient): NR = cg_w, NK = ng_w
void launch_wino_filter(const float* w, float* fw, int groups, int NR, int NK, int ng_w, int cg_w, int pass, hipStream_t st);

struct WinoWgradParams {
    const float* x; const float* dy; float* part;      // part: [splits][16][cout_all][Cg]
    int64_t x_bytes, dy_bytes;
    int N, H, W;                 // image size (input = output size)
    int ldx, lddy;               // pixel strides (all channels of all groups)
    int Cg, Ng;                  // channels per group: input, output
    int TX, TPI, MT;
    unsigned div_tpi_mul, div_tpi_shr, div_tx_mul, div_tx_shr;
    int ci_blocks, co_blocks, splits, tiles_per_split;      // tiles_per_split: multiple of 8
    int cout_all;
};
// conv_wino_small.hip: the filter gradient of convs with at most 32 channels on both sides, operands made in registers; splits = 0: not that form
int wino_small_wgrad_splits(int n, int h, int w, int cg, int ng, int groups);
void launch_wino_small_wgrad(const WinoWgradParams& p, int groups, hipStream_t st);      // p.splits / p.tiles_per_split as wino_small_wgrad_splits / launch_wino_wgrad set them

// Winograd filter gradient F(3x3, 2x2): workspace = one filter-shaped slab [cout][3][3][cin/groups] (complete after the launch) + the split-K partial sums
bool wino_wgrad_geom_ok(int n, int h, int w, int ho, int wo, int cin, int cout, int r, int s, int stride, int pad, int groups);
int64_t wino_wgrad_workspace_floats(int n, int h, int w, int cin, int cout, int groups);
void launch_wino_wgrad(const float* x, const float* dy, float* workspace, int n, int h, int w, int cin, int cout, int groups, hipStream_t st);

}  // namespace suhpe
