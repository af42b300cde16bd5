// Depthwise 3x3 convolution (groups == channels) on gfx950: forward, input gradient, filter gradient.
// HBM-bound VALU work (0.03 of EfficientNetV2-S's 2.85 GMAC but 28% of its conv activation bytes), so the
// kernels are organised around 16-byte channel vectors of pixel-major NHWC tensors: a workgroup owns a chunk of
// channels (filter taps staged once in LDS as [tap][channel]) and a slab of output pixels; a lane handles 4
// channels of one pixel per step.  Replaces nn.Conv2d(groups=C) of torchvision MBConv (src/networks.py:303) and
// src/tiny_vit.py:91-93,130-131,323-324.
#include "common.h"

namespace suhpe {

constexpr int kMaxTaps = 9;
constexpr int kDwMaxRowBlocks = 128;

struct DwGeom {
    int N, GH, GW;       // gathered tensor dims (x for fwd/wgrad, dy for dgrad)
    int OH, OW;          // enumerated output pixels
    int C;
    int R, S, stride, pad;
    int mode;            // 0: g = o*stride - pad + r ; 1: g = (o + pad - r)/stride (exact multiples only)
    int cw, rp;          // columns (float4) per block, pixel rows per pass
    int64_t P;           // N*OH*OW
    int64_t pix_per_block;
};

// Optional input transform of the forward / filter-gradient kernels: the depthwise conv of an MBConv reads the expand
// conv's raw output x and applies that conv's BatchNorm + activation while loading, v -> act(v*sc[c] + sh[c]) with
// sc = gamma*rstd, sh = beta - mean*sc, so the normalised tensor is never written to or read from HBM.  Padding taps stay 0
// (the padding is applied to the activated tensor).  mean == NULL: no transform.  Batch segments (seg_n > 0): the samples
// [0, seg_n) were normalised with mean[0..C) / rstd[0..C), the others with mean[C..2C) / rstd[C..2C) (gamma / beta are the
// layer's); every thread walks its samples in increasing order, so it reloads its scale / shift once, when it crosses seg_n.
struct DwXform { const float* mean; const float* rstd; const float* gamma; const float* beta; int act; int seg_n; int C; };
struct DwXf4 { float sc[4], sh[4]; int act; bool on; };
__device__ __forceinline__ DwXf4 dw_xf_load(const DwXform& xf, int ch, int seg = 0) {
    DwXf4 r;
    r.on = xf.mean != nullptr;
    r.act = xf.act;
    const int sch = seg * xf.C + ch;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float g = (r.on && xf.gamma) ? xf.gamma[ch + v] : 1.f, b = (r.on && xf.beta) ? xf.beta[ch + v] : 0.f;
        r.sc[v] = r.on ? xf.rstd[sch + v] * g : 1.f;
        r.sh[v] = r.on ? b - xf.mean[sch + v] * r.sc[v] : 0.f;
    }
    return r;
}
template <bool XF>
__device__ __forceinline__ void dw_xf_cross(DwXf4& xf, bool& second, const DwXform& xform, int ch, int n) {
    if (XF && xform.seg_n > 0 && !second && n >= xform.seg_n) { xf = dw_xf_load(xform, ch, 1); second = true; }
}
// activation of the input transform: SiLU on the hardware exponential / reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each) -- the
// depthwise kernels evaluate it 1.5x (forward) to 3x (filter gradient) per element, which the accurate expf path (dozens of VALU
// instructions) turned into the kernels' bound; other activations keep act_apply
__device__ __forceinline__ float dw_act(float u, int act) {
    if (act == SUHPE_ACT_SILU) return u * __builtin_amdgcn_rcpf(1.f + __expf(-u));
    return act_apply(u, act);
}
template <bool XF>
__device__ __forceinline__ float4 dw_xf(const DwXf4& t, float4 v) {
    if (!XF) return v;
    return make_float4(dw_act(v.x * t.sc[0] + t.sh[0], t.act), dw_act(v.y * t.sc[1] + t.sh[1], t.act),
                       dw_act(v.z * t.sc[2] + t.sh[2], t.act), dw_act(v.w * t.sc[3] + t.sh[3], t.act));
}

__device__ __forceinline__ bool dw_coord(const DwGeom& g, int o, int tap, int extent, int* out) {
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
    return q * g.stride == t && q < extent;
}

// y[p][c] = sum_taps G[gather(p,tap)][c] * w[c][tap]
template <bool XF>
__global__ __launch_bounds__(256) void dw_conv_kernel(const DwGeom g, const float* __restrict__ src, const float* __restrict__ w,
                                                      float* __restrict__ dst, const DwXform xform) {
    __shared__ __attribute__((aligned(16))) float wl[kMaxTaps * 256 * 4];
    const int t = threadIdx.x, col = t % g.cw, rsub = t / g.cw;
    const int cv = g.C / 4, col0 = blockIdx.y * g.cw;
    const int taps = g.R * g.S;
    // stage the filter chunk: wl[tap][cw*4]
    for (int i = t; i < taps * g.cw * 4; i += 256) {
        const int tap = i / (g.cw * 4), cc = i - tap * g.cw * 4;
        const int ch = col0 * 4 + cc;
        wl[i] = ch < g.C ? w[(int64_t)ch * taps + tap] : 0.f;
    }
    __syncthreads();
    if (rsub >= g.rp || col0 + col >= cv) return;
    DwXf4 xf = dw_xf_load(xform, (col0 + col) * 4);
    bool xsecond = false;
    const int64_t p0 = (int64_t)blockIdx.x * g.pix_per_block;
    int64_t p1 = p0 + g.pix_per_block;
    if (p1 > g.P) p1 = g.P;
    for (int64_t p = p0 + rsub; p < p1; p += g.rp) {
        const int n = (int)(p / ((int64_t)g.OH * g.OW));
        dw_xf_cross<XF>(xf, xsecond, xform, (col0 + col) * 4, n);
        const int rem = (int)(p - (int64_t)n * g.OH * g.OW);
        const int oh = rem / g.OW, ow = rem - oh * g.OW;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < g.R; ++r) {
            int gh;
            if (!dw_coord(g, oh, r, g.GH, &gh)) continue;
            for (int s = 0; s < g.S; ++s) {
                int gw;
                if (!dw_coord(g, ow, s, g.GW, &gw)) continue;
                const float4 xv = dw_xf<XF>(xf, *reinterpret_cast<const float4*>(src + (((int64_t)n * g.GH + gh) * g.GW + gw) * g.C + (col0 + col) * 4));
                const float4 wv = *reinterpret_cast<const float4*>(&wl[((r * g.S + s) * g.cw + col) * 4]);
                acc.x += xv.x * wv.x; acc.y += xv.y * wv.y; acc.z += xv.z * wv.z; acc.w += xv.w * wv.w;
            }
        }
        *reinterpret_cast<float4*>(dst + p * g.C + (col0 + col) * 4) = acc;
    }
}

// ---- 3x3, pad 1 fast path: a thread walks one output row with a sliding 3x3 window of float4 registers, so an
// output pixel costs 3 (stride 1) or 6 (stride 2) new loads instead of 9.  `flip` reverses the taps: the input
// gradient of a stride-1 depthwise conv is the same convolution with the filter rotated by 180 degrees.
__device__ __forceinline__ float4 ld4_or_zero(const float* __restrict__ base, bool ok, int64_t off) {
    float4 v = *reinterpret_cast<const float4*>(base + (ok ? off : 0));
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
}
__device__ __forceinline__ void fma4(float4& a, const float4& x, const float4& w) {
    a.x += x.x * w.x; a.y += x.y * w.y; a.z += x.z * w.z; a.w += x.w * w.w;
}

template <bool XF>
__device__ __forceinline__ float4 ld4_xf(const float* __restrict__ base, bool ok, int64_t off, const DwXf4& xf) {
    float4 v = *reinterpret_cast<const float4*>(base + (ok ? off : 0));
    v = dw_xf<XF>(xf, v);
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
}


// ---- BatchNorm statistics of the OUTPUT from the same pass (the depthwise conv of an MBConv block is followed by a train-mode
// BatchNorm: a separate statistics pass would read the tensor once more).  A thread keeps shifted sums (S, Q, k, n) of the
// output values it produces -- k = its first value, per channel -- for the current batch segment; when its rows cross into
// the second segment (sample index >= seg_n) the sums move to the first segment's registers and restart.  At the end the
// threads of a workgroup that share a channel vector are combined in LDS in a fixed order, re-centred on the first
// contributing thread's shift, and ONE record per (workgroup, channel) and segment is written: stats[seg][blockIdx.x][C] as
// float4 (S, Q, k, n), the format of the GEMM epilogue's records (conv_igemm.hip) and of bn_stats_partial (nn_elem.hip).
struct DwStat {
    float S[4], Q[4], k[4];
    float n;
};
__device__ __forceinline__ void dw_stat_zero(DwStat& a) {
#pragma unroll
    for (int v = 0; v < 4; ++v) { a.S[v] = 0.f; a.Q[v] = 0.f; a.k[v] = 0.f; }
    a.n = 0.f;
}
__device__ __forceinline__ void dw_stat_add(DwStat& a, const float* o) {
    if (a.n == 0.f) {
#pragma unroll
        for (int v = 0; v < 4; ++v) a.k[v] = o[v];
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) { const float d = o[v] - a.k[v]; a.S[v] += d; a.Q[v] += d * d; }
    a.n += 1.f;
}
// all 256 threads call this (no early exits before it); scratch: 256 * 13 floats of LDS
__device__ __forceinline__ void dw_stat_emit(const DwStat& a, bool active, int cw, int rp, int col, int col0, int C, float* __restrict__ rec,
                                             float* scratch) {
    const int t = threadIdx.x;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 4; ++v) { scratch[t * 13 + v] = a.S[v]; scratch[t * 13 + 4 + v] = a.Q[v]; scratch[t * 13 + 8 + v] = a.k[v]; }
    scratch[t * 13 + 12] = active ? a.n : 0.f;
    __syncthreads();
    if (t < cw && (col0 + t) * 4 < C) {
        float S[4] = {0.f, 0.f, 0.f, 0.f}, Q[4] = {0.f, 0.f, 0.f, 0.f}, kb[4] = {0.f, 0.f, 0.f, 0.f}, n = 0.f;
        for (int r = 0; r < rp; ++r) {
            const float* e = &scratch[(r * cw + t) * 13];
            const float nt = e[12];
            if (nt == 0.f) continue;
            if (n == 0.f) {
#pragma unroll
                for (int v = 0; v < 4; ++v) kb[v] = e[8 + v];
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float d = e[8 + v] - kb[v];
                S[v] += e[v] + nt * d;
                Q[v] += e[4 + v] + 2.f * d * e[v] + nt * d * d;
            }
            n += nt;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) reinterpret_cast<float4*>(rec)[(col0 + t) * 4 + v] = make_float4(S[v], Q[v], kb[v], n);
    }
}

template <bool XF, bool STATS = false>
__global__ __launch_bounds__(256) void dw3x3_row_kernel(const float* __restrict__ src, const float* __restrict__ w, float* __restrict__ dst,
                                                        int N, int H, int W, int OH, int OW, int C, int stride, int flip, int cw, int rp,
                                                        const DwXform xform, float* __restrict__ stats = nullptr, int seg_n = 0) {
    __shared__ __attribute__((aligned(16))) float wl[9 * 256 * 4];
    const int t = threadIdx.x, col = t % cw, rsub = t / cw;
    const int cv = C / 4, col0 = blockIdx.y * cw;
    for (int i = t; i < 9 * cw * 4; i += 256) {
        const int tap = i / (cw * 4), cc = i - tap * cw * 4;
        const int ch = col0 * 4 + cc;
        wl[i] = ch < C ? w[(int64_t)ch * 9 + (flip ? 8 - tap : tap)] : 0.f;
    }
    __syncthreads();
    const bool active = rsub < rp && col0 + col < cv;
    if (!STATS && !active) return;
    float4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const float4*>(&wl[(k * cw + (active ? col : 0)) * 4]);
    const int coff = (col0 + col) * 4;
    DwXf4 xf = dw_xf_load(xform, active ? coff : 0);
    DwStat st_a, st;
    bool crossed = false, xsecond = false;
    if (STATS) { dw_stat_zero(st_a); dw_stat_zero(st); }
    for (int row = blockIdx.x * rp + rsub; active && row < N * OH; row += gridDim.x * rp) {
        const int n = row / OH, oh = row - n * OH;
        dw_xf_cross<XF>(xf, xsecond, xform, coff, n);
        if (STATS && seg_n > 0 && !crossed && n >= seg_n) { st_a = st; dw_stat_zero(st); crossed = true; }
        const int ih0 = oh * stride - 1;
        bool rok[3];
        int64_t rbase[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { const int ih = ih0 + r; rok[r] = ih >= 0 && ih < H; rbase[r] = ((int64_t)n * H + ih) * W * C + coff; }
        float4 x[3][3];
        // prime the window: columns -1, 0 of the first output (column -1 is padding)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            x[r][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            x[r][2] = ld4_xf<XF>(src, rok[r], rbase[r], xf);                      // iw = 0
        }
        for (int ow = 0; ow < OW; ++ow) {
            const int iw0 = ow * stride - 1;                                    // window columns iw0, iw0+1, iw0+2
            if (stride == 1) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    x[r][0] = x[r][1]; x[r][1] = x[r][2];
                    x[r][2] = ld4_xf<XF>(src, rok[r] && iw0 + 2 < W, rbase[r] + (int64_t)(iw0 + 2) * C, xf);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    x[r][0] = ow == 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : x[r][2];
                    if (ow > 0) x[r][1] = ld4_xf<XF>(src, rok[r] && iw0 + 1 < W, rbase[r] + (int64_t)(iw0 + 1) * C, xf);
                    else x[r][1] = x[r][2];
                    x[r][2] = ld4_xf<XF>(src, rok[r] && iw0 + 2 < W, rbase[r] + (int64_t)(iw0 + 2) * C, xf);
                }
            }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) fma4(acc, x[r][c], wv[r * 3 + c]);
            *reinterpret_cast<float4*>(dst + ((int64_t)row * OW + ow) * C + coff) = acc;
            if (STATS) { const float o[4] = {acc.x, acc.y, acc.z, acc.w}; dw_stat_add(st, o); }
        }
    }
    if (STATS) {
        // rows of the first segment: st_a when the thread crossed the boundary, else st if it never left the first segment
        const bool two = seg_n > 0;
        if (two && !crossed) { st_a = st; dw_stat_zero(st); }          // everything this thread produced lies in the first segment
        float* rec = stats + (int64_t)blockIdx.x * C * 4;
        if (two) {
            dw_stat_emit(st_a, active, cw, rp, col, col0, C, rec, wl);
            dw_stat_emit(st, active, cw, rp, col, col0, C, rec + (int64_t)gridDim.x * C * 4, wl);
        } else {
            dw_stat_emit(st, active, cw, rp, col, col0, C, rec, wl);
        }
    }
}

// Stride-1 variant: a thread produces FOUR vertically adjacent outputs per column step from a 6 x 3 register window, so an
// input row is fetched 1.5 times instead of 3 (with one output row per thread the halo rows came back from L2 / MALL:
// FETCH_SIZE measured 2.7x the input).  Padding taps read zero through the buffer descriptor's bounds check, and the
// column ahead is loaded one step early (the wait lands at its first use).
constexpr int kDwRows = 4;
// BNB (input-gradient use): the outputs are the gradient d of act(BN(bn_x)) -- the depthwise conv's input was the activated output
// of a train-mode BatchNorm over bn_x (torchvision MBConv: 1x1 expand -> BatchNorm -> SiLU -> depthwise).  The kernel then also
// accumulates that BatchNorm's backward sums over the outputs it produces, g = d * act'(gamma xhat + beta): (sum g, sum g xhat) per
// channel, in the [row block][channel] float2 records of bn_bwd_partial (nn_elem.hip), one record plane per batch segment -- so the
// BatchNorm backward needs no reduction pass of its own over (bn_x, d): one read of bn_x here instead of a read of both there.
struct DwBnb { const float* x; const float* mean; const float* rstd; const float* gamma; const float* beta; int act; int seg_n; float* part; };

template <bool XF, bool STATS = false, bool BNB = false>
__global__ __launch_bounds__(256) void dw3x3_s1_kernel(const float* __restrict__ src, int64_t src_bytes, const float* __restrict__ w,
                                                       float* __restrict__ dst, int N, int H, int W, int C, int flip, int cw, int rp, const DwXform xform,
                                                       float* __restrict__ stats = nullptr, int seg_n = 0, const DwBnb bnb = DwBnb{}) {
    __shared__ __attribute__((aligned(16))) float wl[9 * 256 * 4];
    const int t = threadIdx.x, col = t % cw, rsub = t / cw;
    const int cv = C / 4, col0 = blockIdx.y * cw;
    for (int i = t; i < 9 * cw * 4; i += 256) {
        const int tap = i / (cw * 4), cc = i - tap * cw * 4;
        const int ch = col0 * 4 + cc;
        wl[i] = ch < C ? w[(int64_t)ch * 9 + (flip ? 8 - tap : tap)] : 0.f;
    }
    __syncthreads();
    const bool active = rsub < rp && col0 + col < cv;
    if (!STATS && !BNB && !active) return;
    float4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const float4*>(&wl[(k * cw + (active ? col : 0)) * 4]);
    const BufRsrc sr = make_rsrc(src, src_bytes);
    const unsigned coff = (unsigned)(col0 + col) * 16u, pix = (unsigned)C * 4u;       // byte offsets
    DwXf4 xf = dw_xf_load(xform, active ? (col0 + col) * 4 : 0);
    const int groups_h = (H + kDwRows - 1) / kDwRows;                                  // row groups per image
    DwStat st_a, st;
    bool crossed = false, xsecond = false;
    if (STATS) { dw_stat_zero(st_a); dw_stat_zero(st); }
    // BNB: this thread's four channels of the BatchNorm (statistics of the current segment) and its running sums
    float b_mu[4], b_rs[4], b_ga[4], b_be[4], b_s0[4] = {0.f, 0.f, 0.f, 0.f}, b_s1[4] = {0.f, 0.f, 0.f, 0.f}, b_a0[4] = {0.f, 0.f, 0.f, 0.f},
          b_a1[4] = {0.f, 0.f, 0.f, 0.f};
    bool b_crossed = false;
    if (BNB) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int ch = active ? (col0 + col) * 4 + v : 0;
            b_mu[v] = bnb.mean[ch]; b_rs[v] = bnb.rstd[ch];
            b_ga[v] = bnb.gamma ? bnb.gamma[ch] : 1.f; b_be[v] = bnb.beta ? bnb.beta[ch] : 0.f;
        }
    }
    for (int rg = blockIdx.x * rp + rsub; active && rg < N * groups_h; rg += gridDim.x * rp) {
        const int n = rg / groups_h, oh0 = (rg - n * groups_h) * kDwRows;
        dw_xf_cross<XF>(xf, xsecond, xform, (col0 + col) * 4, n);
        if (STATS && seg_n > 0 && !crossed && n >= seg_n) { st_a = st; dw_stat_zero(st); crossed = true; }
        if (BNB && bnb.seg_n > 0 && !b_crossed && n >= bnb.seg_n) {           // second batch segment: its statistics, fresh sums
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                b_a0[v] = b_s0[v]; b_a1[v] = b_s1[v]; b_s0[v] = 0.f; b_s1[v] = 0.f;
                b_mu[v] = bnb.mean[C + (col0 + col) * 4 + v]; b_rs[v] = bnb.rstd[C + (col0 + col) * 4 + v];
            }
            b_crossed = true;
        }
        unsigned rbase[kDwRows + 2];
        bool rok[kDwRows + 2];
#pragma unroll
        for (int r = 0; r < kDwRows + 2; ++r) {
            const int ih = oh0 - 1 + r;
            rok[r] = (unsigned)ih < (unsigned)H;
            rbase[r] = (unsigned)((n * H + ih) * W) * pix + coff;
        }
        float x[kDwRows + 2][4][4];                          // window columns ow-1, ow, ow+1 and the prefetched ow+2
        auto ld = [&](int r, int p, int c) __attribute__((always_inline)) {
            buf_load<4>(sr, (rok[r] & ((unsigned)c < (unsigned)W)) ? rbase[r] + (unsigned)c * pix : kOob, x[r][p]);
        };
        // the input transform is applied when a loaded column enters the window (one step after its load was issued, so the
        // wait still lands late); padding taps (row or column outside) stay zero
        auto enter = [&](int r, int p, int c) __attribute__((always_inline)) {
            if (!XF) return;
            const bool ok = rok[r] & ((unsigned)c < (unsigned)W);
            const float4 v = dw_xf<XF>(xf, make_float4(x[r][p][0], x[r][p][1], x[r][p][2], x[r][p][3]));
            x[r][p][0] = ok ? v.x : 0.f; x[r][p][1] = ok ? v.y : 0.f; x[r][p][2] = ok ? v.z : 0.f; x[r][p][3] = ok ? v.w : 0.f;
        };
#pragma unroll
        for (int r = 0; r < kDwRows + 2; ++r) {
#pragma unroll
            for (int v = 0; v < 4; ++v) x[r][1][v] = 0.f;    // becomes column -1 after the first shift
            ld(r, 2, 0);
            ld(r, 3, 1);
        }
#pragma unroll
        for (int r = 0; r < kDwRows + 2; ++r) enter(r, 2, 0);
        for (int ow = 0; ow < W; ++ow) {
#pragma unroll
            for (int r = 0; r < kDwRows + 2; ++r) {
#pragma unroll
                for (int v = 0; v < 4; ++v) { x[r][0][v] = x[r][1][v]; x[r][1][v] = x[r][2][v]; x[r][2][v] = x[r][3][v]; }
                enter(r, 2, ow + 1);
                ld(r, 3, ow + 2);                            // needed at the next step
            }
            float4 bx[kDwRows];
            if (BNB) {                                       // the BatchNorm input at the output positions: loads issued ahead of the taps' arithmetic
#pragma unroll
                for (int j = 0; j < kDwRows; ++j) {
                    const int oh = oh0 + j < H ? oh0 + j : H - 1;
                    bx[j] = *reinterpret_cast<const float4*>(bnb.x + (((int64_t)n * H + oh) * W + ow) * C + (col0 + col) * 4);
                }
            }
#pragma unroll
            for (int j = 0; j < kDwRows; ++j) {
                if (oh0 + j >= H) break;
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float4 wk = wv[r * 3 + c];
                        acc[0] += x[j + r][c][0] * wk.x; acc[1] += x[j + r][c][1] * wk.y;
                        acc[2] += x[j + r][c][2] * wk.z; acc[3] += x[j + r][c][3] * wk.w;
                    }
                *reinterpret_cast<float4*>(dst + (((int64_t)n * H + oh0 + j) * W + ow) * C + (col0 + col) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                if (STATS) dw_stat_add(st, acc);
                if (BNB) {
                    const float xv[4] = {bx[j].x, bx[j].y, bx[j].z, bx[j].w};
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float xh = (xv[v] - b_mu[v]) * b_rs[v];
                        const float g = acc[v] * act_grad(xh * b_ga[v] + b_be[v], bnb.act);
                        b_s0[v] += g;
                        b_s1[v] += g * xh;
                    }
                }
            }
        }
    }
    if (BNB) {
        // threads of the workgroup that share a channel vector, added in a fixed order; records [plane][blockIdx.x][C] float2
        const bool two = bnb.seg_n > 0;
        if (two && !b_crossed) {
#pragma unroll
            for (int v = 0; v < 4; ++v) { b_a0[v] = b_s0[v]; b_a1[v] = b_s1[v]; b_s0[v] = 0.f; b_s1[v] = 0.f; }
        }
        float* scratch = wl;
        for (int plane = 0; plane < (two ? 2 : 1); ++plane) {
            const float* q0 = (two && plane == 0) ? b_a0 : b_s0;
            const float* q1 = (two && plane == 0) ? b_a1 : b_s1;
            __syncthreads();
#pragma unroll
            for (int v = 0; v < 4; ++v) { scratch[t * 8 + v] = active ? q0[v] : 0.f; scratch[t * 8 + 4 + v] = active ? q1[v] : 0.f; }
            __syncthreads();
            if (t < cw && (col0 + t) * 4 < C) {
                float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
                for (int r = 0; r < rp; ++r) {
                    const float* e = &scratch[(r * cw + t) * 8];
#pragma unroll
                    for (int v = 0; v < 4; ++v) { a0[v] += e[v]; a1[v] += e[4 + v]; }
                }
                float2* rec = reinterpret_cast<float2*>(bnb.part) + ((int64_t)plane * gridDim.x + blockIdx.x) * C + (col0 + t) * 4;
#pragma unroll
                for (int v = 0; v < 4; ++v) rec[v] = make_float2(a0[v], a1[v]);
            }
        }
    }
    if (STATS) {
        const bool two = seg_n > 0;
        if (two && !crossed) { st_a = st; dw_stat_zero(st); }          // everything this thread produced lies in the first segment
        float* rec = stats + (int64_t)blockIdx.x * C * 4;
        if (two) {
            dw_stat_emit(st_a, active, cw, rp, col, col0, C, rec, wl);
            dw_stat_emit(st, active, cw, rp, col, col0, C, rec + (int64_t)gridDim.x * C * 4, wl);
        } else {
            dw_stat_emit(st, active, cw, rp, col, col0, C, rec, wl);
        }
    }
}

// ---- small square feature maps (14 x 14, 7 x 7: every MBConv stage of EfficientNetV2-S, the last two TinyViT stages): whole images in LDS.
// The sliding-window kernel above keeps one column step of loads in flight per thread -- a dependent chain of W steps -- and
// reached 2.6-3.7 TB/s on these shapes (tools/ewbench.py).  Here a workgroup owns a 64-channel chunk and walks groups of whole images:
// the group's pixels are loaded with EVERY load issued up front (HW*HW*imgs/16 independent 16-byte loads per thread), written
// into an LDS image with a zero ring (padding needs no predicate, the input transform is applied once per element on the way in),
// and while the 3 x 3 windows of that group are computed from LDS -- thread = (channel vector, image row), the row is walked with
// the window in registers, three ds_read_b128 per output -- the NEXT group's loads are already in flight.  tools/proto/dw_lab.hip:
// 160 x 14 x 14 x 960: 63 -> 45 us.  LDS: 16 consecutive lanes read 256 contiguous bytes, rows are a multiple of 256 B apart: no conflicts.
constexpr int kTileSlots = 16;                       // image rows per workgroup pass = 256 threads / 16 channel vectors
constexpr int kTileAhead = 6;                        // column steps between a per-row operand's load (BatchNorm input, dy) and its use
template <int HW> struct DwTile {
    static constexpr int imgs = kTileSlots / HW;                                   // whole images per group (14 -> 1, 7 -> 2)
    static constexpr int npix = imgs * HW * HW;
    static constexpr int lpt = (npix + kTileSlots - 1) / kTileSlots;               // interior loads per thread and group
    static constexpr int wp = HW + 2;
    static constexpr int lds_bytes = imgs * wp * wp * 64 * 4;
};

template <int HW, bool XF, bool STATS, bool BNB, int ACTC>
__global__ __launch_bounds__(256, (HW == 14 || BNB || (XF && STATS)) ? 2 : 3) void dw3x3_tile_kernel(const float* __restrict__ src, const float* __restrict__ w, float* __restrict__ dst, int N, int C,
                                                         int flip, int ngroups, const DwXform xform, float* __restrict__ stats, int seg_n,
                                                         const DwBnb bnb) {
    using T = DwTile<HW>;
    constexpr int WP = T::wp, LPT = T::lpt;
    extern __shared__ __attribute__((aligned(16))) float tile[];          // [imgs][HW+2][HW+2][64]
    const int t = threadIdx.x, vc = t & 15, slot = t >> 4;
    const int col0 = blockIdx.y * 16, c0 = (col0 + vc) * 4;
    const bool cok = c0 < C;
    for (int i = t; i < T::imgs * WP * WP * 16; i += 256) reinterpret_cast<float4*>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int kk = flip ? 8 - k : k;
        wv[k] = cok ? make_float4(w[(int64_t)(c0 + 0) * 9 + kk], w[(int64_t)(c0 + 1) * 9 + kk], w[(int64_t)(c0 + 2) * 9 + kk], w[(int64_t)(c0 + 3) * 9 + kk])
                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // input transform of the two batch segments (XF) / BatchNorm of the backward sums (BNB): both sets stay in registers
    const int xact = ACTC >= 0 ? ACTC : xform.act;
    DwXf4 xfa = dw_xf_load(xform, cok ? c0 : 0, 0), xfb = xfa;
    if (XF && xform.seg_n > 0) xfb = dw_xf_load(xform, cok ? c0 : 0, 1);
    // (a thread's images come in increasing order: the statistics set and the running sums switch once, at the segment boundary)
    float b_mu[4], b_rs[4], b_ga[4], b_be[4], b_s0[4] = {0.f, 0.f, 0.f, 0.f}, b_s1[4] = {0.f, 0.f, 0.f, 0.f}, b_a0[4] = {0.f, 0.f, 0.f, 0.f},
          b_a1[4] = {0.f, 0.f, 0.f, 0.f};
    bool b_crossed = false;
    if (BNB) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int ch = cok ? c0 + v : 0;
            b_mu[v] = bnb.mean[ch]; b_rs[v] = bnb.rstd[ch];
            b_ga[v] = bnb.gamma ? bnb.gamma[ch] : 1.f; b_be[v] = bnb.beta ? bnb.beta[ch] : 0.f;
        }
    }
    const int bact = ACTC >= 0 ? ACTC : bnb.act;
    DwStat st_a, st;
    bool crossed = false;
    if (STATS) { dw_stat_zero(st_a); dw_stat_zero(st); }

    float4 pf[LPT];
    auto issue = [&](int g) __attribute__((always_inline)) {
        const int64_t p0 = (int64_t)g * T::npix;
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int q = slot + kTileSlots * k;
            const int64_t pix = p0 + q;
            const bool ok = cok && q < T::npix && pix < (int64_t)N * HW * HW;
            pf[k] = ok ? *reinterpret_cast<const float4*>(src + pix * C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int g = blockIdx.x;
    if (g < ngroups) issue(g);
    __syncthreads();
    const int img = slot / HW, row = slot - img * HW;
    const bool crow = slot < T::imgs * HW;
    for (; g < ngroups; g += gridDim.x) {
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int q = slot + kTileSlots * k;
            if (q >= T::npix) continue;
            const int qi = q / (HW * HW), rem = q - qi * HW * HW, h = rem / HW, x = rem - h * HW;
            float4 v = pf[k];
            if (XF) {
                const int64_t n = (int64_t)g * T::imgs + qi;
                const bool second = xform.seg_n > 0 && n >= xform.seg_n;
                float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = dw_act(o[e] * (second ? xfb.sc[e] : xfa.sc[e]) + (second ? xfb.sh[e] : xfa.sh[e]), xact);
                const bool ok = cok && n < N;                                  // images past the batch stay zero, not act(shift)
                v = ok ? make_float4(o[0], o[1], o[2], o[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            *reinterpret_cast<float4*>(&tile[((qi * WP + h + 1) * WP + x + 1) * 64 + vc * 4]) = v;
        }
        __syncthreads();
        if (g + (int)gridDim.x < ngroups) issue(g + gridDim.x);
        const int64_t n = (int64_t)g * T::imgs + img;
        if (crow && cok && n < N) {
            if (STATS && seg_n > 0 && !crossed && n >= seg_n) { st_a = st; dw_stat_zero(st); crossed = true; }
            if (BNB && bnb.seg_n > 0 && !b_crossed && n >= bnb.seg_n) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    b_a0[v] = b_s0[v]; b_a1[v] = b_s1[v]; b_s0[v] = 0.f; b_s1[v] = 0.f;
                    b_mu[v] = bnb.mean[C + c0 + v]; b_rs[v] = bnb.rstd[C + c0 + v];
                }
                b_crossed = true;
            }
            // BNB: the BatchNorm input at the output positions, kTileAhead column steps ahead of its use
            const float* bxp = BNB ? bnb.x + ((n * HW + row) * HW) * C + c0 : nullptr;
            float4 bq[kTileAhead];
            if (BNB) {
#pragma unroll
                for (int d = 0; d < kTileAhead; ++d) bq[d] = *reinterpret_cast<const float4*>(bxp + (int64_t)(d < HW ? d : HW - 1) * C);
            }
            const float* base = &tile[((img * WP + row) * WP) * 64 + vc * 4];      // padded (row, column 0): the window's top-left corner
            float4 a[3], b[3], c[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                a[r] = *reinterpret_cast<const float4*>(base + (r * WP + 0) * 64);
                b[r] = *reinterpret_cast<const float4*>(base + (r * WP + 1) * 64);
            }
            float* out = dst + ((n * HW + row) * HW) * C + c0;
#pragma unroll 1
            for (int ow = 0; ow < HW; ++ow) {
#pragma unroll
                for (int r = 0; r < 3; ++r) c[r] = *reinterpret_cast<const float4*>(base + (r * WP + ow + 2) * 64);
                const float4 bxc = BNB ? bq[0] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (BNB) {
#pragma unroll
                    for (int d = 0; d + 1 < kTileAhead; ++d) bq[d] = bq[d + 1];
                    bq[kTileAhead - 1] = *reinterpret_cast<const float4*>(bxp + (int64_t)(ow + kTileAhead < HW ? ow + kTileAhead : HW - 1) * C);
                }
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < 3; ++r) { fma4(acc, a[r], wv[r * 3]); fma4(acc, b[r], wv[r * 3 + 1]); fma4(acc, c[r], wv[r * 3 + 2]); }
                *reinterpret_cast<float4*>(out + (int64_t)ow * C) = acc;
                const float o[4] = {acc.x, acc.y, acc.z, acc.w};
                if (STATS) dw_stat_add(st, o);
                if (BNB) {
                    const float xv[4] = {bxc.x, bxc.y, bxc.z, bxc.w};
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float xh = (xv[v] - b_mu[v]) * b_rs[v];
                        const float gg = o[v] * act_grad(xh * b_ga[v] + b_be[v], bact);
                        b_s0[v] += gg;
                        b_s1[v] += gg * xh;
                    }
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) { a[r] = b[r]; b[r] = c[r]; }
            }
        }
        __syncthreads();
    }
    const bool active = crow && cok;
    if (BNB) {
        // threads that share a channel vector, added in a fixed order; records [plane][blockIdx.x][C] float2 (bn_bwd_partial's format)
        const bool two = bnb.seg_n > 0;
        if (two && !b_crossed) {
#pragma unroll
            for (int v = 0; v < 4; ++v) { b_a0[v] = b_s0[v]; b_a1[v] = b_s1[v]; b_s0[v] = 0.f; b_s1[v] = 0.f; }
        }
        float* scratch = tile;
        for (int plane = 0; plane < (two ? 2 : 1); ++plane) {
            const float* q0 = (two && plane == 0) ? b_a0 : b_s0;
            const float* q1 = (two && plane == 0) ? b_a1 : b_s1;
            __syncthreads();
#pragma unroll
            for (int v = 0; v < 4; ++v) { scratch[t * 8 + v] = active ? q0[v] : 0.f; scratch[t * 8 + 4 + v] = active ? q1[v] : 0.f; }
            __syncthreads();
            if (t < 16 && (col0 + t) * 4 < C) {
                float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
                for (int r = 0; r < kTileSlots; ++r) {
                    const float* e = &scratch[(r * 16 + t) * 8];
#pragma unroll
                    for (int v = 0; v < 4; ++v) { a0[v] += e[v]; a1[v] += e[4 + v]; }
                }
                float2* rec = reinterpret_cast<float2*>(bnb.part) + ((int64_t)plane * gridDim.x + blockIdx.x) * C + (col0 + t) * 4;
#pragma unroll
                for (int v = 0; v < 4; ++v) rec[v] = make_float2(a0[v], a1[v]);
            }
        }
    }
    if (STATS) {
        const bool two = seg_n > 0;
        if (two && !crossed) { st_a = st; dw_stat_zero(st); }          // everything this thread produced lies in the first segment
        float* rec = stats + (int64_t)blockIdx.x * C * 4;
        if (two) {
            dw_stat_emit(st_a, active, 16, kTileSlots, vc, col0, C, rec, tile);
            dw_stat_emit(st, active, 16, kTileSlots, vc, col0, C, rec + (int64_t)gridDim.x * C * 4, tile);
        } else {
            dw_stat_emit(st, active, 16, kTileSlots, vc, col0, C, rec, tile);
        }
    }
}

// Filter gradient on the same image tiles: x (through the input transform, once per element) is staged in LDS with its zero ring, a
// thread = (channel vector, image row) walks its row with the 3 x 3 window of x in registers and the row's dy values arriving kTileAhead steps
// ahead, and keeps the nine tap sums of its four channels in registers ACROSS the image groups of its persistent workgroup: one LDS
// reduction and one part[bx][tap][c] record set per workgroup at the end (dw_wgrad_finalize adds the workgroups in fp64).
// The row kernel below loads and uses x / dy in the same step (no prefetch): 2.4-3.0 TB/s on these shapes.
template <int HW, bool XF, int ACTC>
__global__ __launch_bounds__(256, 2) void dw3x3_wgrad_tile_kernel(const float* __restrict__ src, const float* __restrict__ dy, float* __restrict__ part, int N,
                                                                  int C, int ngroups, const DwXform xform) {
    using T = DwTile<HW>;
    constexpr int WP = T::wp, LPT = T::lpt;
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int t = threadIdx.x, vc = t & 15, slot = t >> 4;
    const int col0 = blockIdx.y * 16, c0 = (col0 + vc) * 4;
    const bool cok = c0 < C;
    for (int i = t; i < T::imgs * WP * WP * 16; i += 256) reinterpret_cast<float4*>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int xact = ACTC >= 0 ? ACTC : xform.act;
    DwXf4 xfa = dw_xf_load(xform, cok ? c0 : 0, 0), xfb = xfa;
    if (XF && xform.seg_n > 0) xfb = dw_xf_load(xform, cok ? c0 : 0, 1);
    float4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pf[LPT];
    auto issue = [&](int g) __attribute__((always_inline)) {
        const int64_t p0 = (int64_t)g * T::npix;
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int q = slot + kTileSlots * k;
            const int64_t pix = p0 + q;
            const bool ok = cok && q < T::npix && pix < (int64_t)N * HW * HW;
            pf[k] = ok ? *reinterpret_cast<const float4*>(src + pix * C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int g = blockIdx.x;
    if (g < ngroups) issue(g);
    __syncthreads();
    const int img = slot / HW, row = slot - img * HW;
    const bool crow = slot < T::imgs * HW;
    for (; g < ngroups; g += gridDim.x) {
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int q = slot + kTileSlots * k;
            if (q >= T::npix) continue;
            const int qi = q / (HW * HW), rem = q - qi * HW * HW, h = rem / HW, x = rem - h * HW;
            float4 v = pf[k];
            if (XF) {
                const int64_t n = (int64_t)g * T::imgs + qi;
                const bool second = xform.seg_n > 0 && n >= xform.seg_n;
                float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = dw_act(o[e] * (second ? xfb.sc[e] : xfa.sc[e]) + (second ? xfb.sh[e] : xfa.sh[e]), xact);
                const bool ok = cok && n < N;
                v = ok ? make_float4(o[0], o[1], o[2], o[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            *reinterpret_cast<float4*>(&tile[((qi * WP + h + 1) * WP + x + 1) * 64 + vc * 4]) = v;
        }
        __syncthreads();
        if (g + (int)gridDim.x < ngroups) issue(g + gridDim.x);
        const int64_t n = (int64_t)g * T::imgs + img;
        if (crow && cok && n < N) {
            const float* dp = dy + ((n * HW + row) * HW) * C + c0;
            float4 dq[kTileAhead];
#pragma unroll
            for (int d = 0; d < kTileAhead; ++d) dq[d] = *reinterpret_cast<const float4*>(dp + (int64_t)(d < HW ? d : HW - 1) * C);
            const float* base = &tile[((img * WP + row) * WP) * 64 + vc * 4];
            float4 a[3], b[3], c[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                a[r] = *reinterpret_cast<const float4*>(base + (r * WP + 0) * 64);
                b[r] = *reinterpret_cast<const float4*>(base + (r * WP + 1) * 64);
            }
#pragma unroll 1
            for (int ow = 0; ow < HW; ++ow) {
#pragma unroll
                for (int r = 0; r < 3; ++r) c[r] = *reinterpret_cast<const float4*>(base + (r * WP + ow + 2) * 64);
                const float4 dv = dq[0];
#pragma unroll
                for (int d = 0; d + 1 < kTileAhead; ++d) dq[d] = dq[d + 1];
                dq[kTileAhead - 1] = *reinterpret_cast<const float4*>(dp + (int64_t)(ow + kTileAhead < HW ? ow + kTileAhead : HW - 1) * C);
#pragma unroll
                for (int r = 0; r < 3; ++r) { fma4(acc[r * 3], a[r], dv); fma4(acc[r * 3 + 1], b[r], dv); fma4(acc[r * 3 + 2], c[r], dv); }
#pragma unroll
                for (int r = 0; r < 3; ++r) { a[r] = b[r]; b[r] = c[r]; }
            }
        }
        __syncthreads();
    }
    // the 16 row slots of a channel vector, added in a fixed order
    float* red = tile;                                   // [tap][256][4]: 36 KB of the tile
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) *reinterpret_cast<float4*>(&red[(tap * 256 + t) * 4]) = acc[tap];
    __syncthreads();
    if (t < 16 && (col0 + t) * 4 < C) {
        for (int tap = 0; tap < 9; ++tap) {
            float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int rr = 0; rr < kTileSlots; ++rr) {
                const float4 v = *reinterpret_cast<const float4*>(&red[(tap * 256 + rr * 16 + t) * 4]);
                s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
            }
            *reinterpret_cast<float4*>(&part[((int64_t)blockIdx.x * 9 + tap) * C + (col0 + t) * 4]) = s4;
        }
    }
}

// ---- the depthwise conv's WHOLE backward on the image tiles (round 6): input gradient + the producer BatchNorm's backward sums (what
// dw3x3_tile_kernel<HW, false, false, true> does) AND the filter gradient from the same staged tile, optionally with the conv's OUTPUT gradient formed
// while the tile is filled (DYF: the apply pass of the depthwise BatchNorm + SqueezeExcitation section, se_bn_bwd_apply_kernel's arithmetic, never
// written).  torchvision MBConv (expand 1x1 -> BatchNorm_e -> SiLU -> depthwise 3x3 -> BatchNorm_d -> SiLU -> SE -> project), per block and step:
//   before   se_bn_bwd_apply  reads y, dout          writes dy          (3 passes over the expanded tensor)
//            dw wgrad         reads e0, dy                              (2)
//            dw dgrad + sums  reads dy, e0           writes d           (3)        = 8 passes
//   here     reads dout, y (DYF) or dy, and e0;      writes d                      = 4 passes (3 without DYF), two launches fewer
// One LDS tile holds the output gradient dy of the group's image(s) with a zero ring, as in the input-gradient kernel.  Both gradients are sums over
// the SAME 3 x 3 window of dy around an input position q:   d(q) = sum_ij dy(q + (i-1, j-1)) w[2-i][2-j]   and, with z = act(BN_e(e0)) the conv's
// input,   dW[r][s] = sum_q z(q) dy(q - (r-1, s-1)) = sum_q z(q) win[2-r][2-s]   -- so the filter gradient needs z only at the window's CENTRE, and
// e0(q) is already streamed there for the BatchNorm_e sums (xhat and act' come from it): ONE sigmoid per element serves z, act' and the sums.
// DYF fill: dy = gamma_d rstd_d (g - m0 - xhat_d m1),  g = (dout s[n] + dpool[n] / hw) act'(xhat_d gamma_d + beta_d),  xhat_d = (y - mean_d) rstd_d,
// (m0, m1) = the section's per-channel sums over the segment / rows (se_bn_bwd_combine_kernel's `chan`); the per-channel constants of both batch
// segments sit in a small LDS table behind the tile and are read per pixel (a group of two 7 x 7 images may straddle the segment boundary).
struct DwMid {
    const float* y;          // the depthwise conv's raw output (N, HW, HW, C); NULL: `src` IS dy (no DYF)
    const float* s;          // (N, C) excitation
    const float* dpool;      // (N, C)
    const float* chan;       // [sets][2 C]: (sum g, sum g xhat) per channel
    const float* mean; const float* rstd; const float* gamma; const float* beta;      // BatchNorm_d: mean / rstd [sets][C]
    int act, seg_n, train_stats;
    float inv_rows_a, inv_rows_b, inv_hw;
};
template <int HW, bool DYF, int ACTC>
__global__ __launch_bounds__(256, 2) void dw3x3_bwd_tile_kernel(const float* __restrict__ src, const float* __restrict__ w, float* __restrict__ dst, int N, int C,
                                                                int ngroups, const DwBnb bnb, const DwMid mid, float* __restrict__ wpart) {
    using T = DwTile<HW>;
    constexpr int WP = T::wp, LPT = T::lpt;
    extern __shared__ __attribute__((aligned(16))) float tile[];          // [imgs][HW+2][HW+2][64] + DYF: [2 sets][mean | rstd | m0 | m1][64]
    float* ctab = tile + T::imgs * WP * WP * 64;
    float* wtab = ctab + 2 * 4 * 64;                       // the flipped filter [9][64 channels]: read per window step (the LDS is 3 % busy; 36 registers are not free)
    const int t = threadIdx.x, vc = t & 15, slot = t >> 4;
    const int col0 = blockIdx.y * 16, c0 = (col0 + vc) * 4;
    const bool cok = c0 < C;
    const int cc = cok ? c0 : 0;
    for (int i = t; i < T::imgs * WP * WP * 16; i += 256) reinterpret_cast<float4*>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (DYF && t < 128) {                                // thread (set, quantity, channel vector) fills one float4 of the table
        const int set = t >> 6, qty = (t >> 4) & 3, v4 = t & 15;
        const int ch0 = (col0 + v4) * 4;
        const bool live = ch0 < C && (set == 0 || mid.seg_n > 0);
        const float inv = set ? mid.inv_rows_b : mid.inv_rows_a;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (live) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ch = ch0 + v;
                o[v] = qty == 0 ? mid.mean[set * C + ch] : qty == 1 ? mid.rstd[set * C + ch]
                     : (mid.train_stats ? mid.chan[set * 2 * C + 2 * ch + (qty - 2)] * inv : 0.f);
            }
        }
        *reinterpret_cast<float4*>(&ctab[((set * 4 + qty) * 16 + v4) * 4]) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (t < 9 * 16) {                                    // flipped: the input gradient is a correlation with the rotated filter
        const int kq = t >> 4, v4 = t & 15, ch0 = (col0 + v4) * 4, kk = 8 - kq;
        const bool okw = ch0 < C;
        *reinterpret_cast<float4*>(&wtab[(kq * 16 + v4) * 4]) =
            okw ? make_float4(w[(int64_t)(ch0 + 0) * 9 + kk], w[(int64_t)(ch0 + 1) * 9 + kk], w[(int64_t)(ch0 + 2) * 9 + kk], w[(int64_t)(ch0 + 3) * 9 + kk])
                : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int bact = ACTC >= 0 ? ACTC : bnb.act;
    const int mact = ACTC >= 0 ? ACTC : mid.act;
    float b_mu[4], b_rs[4], b_ga[4], b_be[4], b_s0[4] = {0.f, 0.f, 0.f, 0.f}, b_s1[4] = {0.f, 0.f, 0.f, 0.f}, b_a0[4] = {0.f, 0.f, 0.f, 0.f},
          b_a1[4] = {0.f, 0.f, 0.f, 0.f};
    bool b_crossed = false;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        b_mu[v] = bnb.mean[cc + v]; b_rs[v] = bnb.rstd[cc + v];
        b_ga[v] = bnb.gamma ? bnb.gamma[cc + v] : 1.f; b_be[v] = bnb.beta ? bnb.beta[cc + v] : 0.f;
    }
    float m_ga[4] = {1.f, 1.f, 1.f, 1.f}, m_be[4] = {0.f, 0.f, 0.f, 0.f};
    if (DYF) {
#pragma unroll
        for (int v = 0; v < 4; ++v) { m_ga[v] = mid.gamma ? mid.gamma[cc + v] : 1.f; m_be[v] = mid.beta ? mid.beta[cc + v] : 0.f; }
    }
    float4 accw[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) accw[k] = make_float4(0.f, 0.f, 0.f, 0.f);

    // The group's pixels are loaded and written to the tile in chunks of CHK per thread, every load of a chunk issued before the first use; nothing is
    // prefetched across the window loop: with the nine filter-gradient accumulators beside the window and the filter, a group's worth of loads held
    // in registers there spilled (256 VGPRs at two workgroups per CU) -- the second resident workgroup covers the load phase instead.
    constexpr int CHK = 7;
    float4 sv[DYF ? T::imgs : 1], pv[DYF ? T::imgs : 1];
    int g = blockIdx.x;
    __syncthreads();
    const int img = slot / HW, row = slot - img * HW;
    const bool crow = slot < T::imgs * HW;
    for (; g < ngroups; g += gridDim.x) {
        const int64_t p0 = (int64_t)g * T::npix;
        if (DYF) {
#pragma unroll
            for (int qi = 0; qi < T::imgs; ++qi) {
                const int64_t n = (int64_t)g * T::imgs + qi;
                const bool ok = cok && n < N;
                sv[qi] = ok ? *reinterpret_cast<const float4*>(mid.s + n * C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
                pv[qi] = ok ? *reinterpret_cast<const float4*>(mid.dpool + n * C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int k0 = 0; k0 < LPT; k0 += CHK) {
            float4 pf[CHK], pg[DYF ? CHK : 1];
#pragma unroll
            for (int k = 0; k < CHK; ++k) {
                const int q = slot + kTileSlots * (k0 + k);
                const int64_t pix = p0 + q;
                const bool ok = cok && k0 + k < LPT && q < T::npix && pix < (int64_t)N * HW * HW;
                pf[k] = ok ? *reinterpret_cast<const float4*>(src + pix * C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (DYF) pg[k] = ok ? *reinterpret_cast<const float4*>(mid.y + pix * C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < CHK; ++k) {
                const int q = slot + kTileSlots * (k0 + k);
                if (k0 + k >= LPT || q >= T::npix) continue;
                const int qi = q / (HW * HW), rem = q - qi * HW * HW, h = rem / HW, x = rem - h * HW;
                float4 v = pf[k];
                if (DYF) {
                    const int64_t n = (int64_t)g * T::imgs + qi;
                    const int set = (mid.seg_n > 0 && n >= mid.seg_n) ? 1 : 0;
                    const float4 mu4 = *reinterpret_cast<const float4*>(&ctab[((set * 4 + 0) * 16 + vc) * 4]);
                    const float4 rs4 = *reinterpret_cast<const float4*>(&ctab[((set * 4 + 1) * 16 + vc) * 4]);
                    const float4 m04 = *reinterpret_cast<const float4*>(&ctab[((set * 4 + 2) * 16 + vc) * 4]);
                    const float4 m14 = *reinterpret_cast<const float4*>(&ctab[((set * 4 + 3) * 16 + vc) * 4]);
                    const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, rs[4] = {rs4.x, rs4.y, rs4.z, rs4.w};
                    const float m0[4] = {m04.x, m04.y, m04.z, m04.w}, m1[4] = {m14.x, m14.y, m14.z, m14.w};
                    const float dv[4] = {v.x, v.y, v.z, v.w}, yv[4] = {pg[k].x, pg[k].y, pg[k].z, pg[k].w};
                    const float4 s4 = sv[T::imgs > 1 ? qi : 0], p4 = pv[T::imgs > 1 ? qi : 0];
                    const float ss[4] = {s4.x, s4.y, s4.z, s4.w}, pp[4] = {p4.x, p4.y, p4.z, p4.w};
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (yv[e] - mu[e]) * rs[e];
                        float gq = (dv[e] * ss[e] + pp[e] * mid.inv_hw) * act_grad(xh * m_ga[e] + m_be[e], mact);
                        gq = gq - m0[e] - xh * m1[e];
                        o[e] = m_ga[e] * rs[e] * gq;
                    }
                    const bool ok = cok && n < N;
                    v = ok ? make_float4(o[0], o[1], o[2], o[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                *reinterpret_cast<float4*>(&tile[((qi * WP + h + 1) * WP + x + 1) * 64 + vc * 4]) = v;
            }
        }
        __syncthreads();
        const int64_t n = (int64_t)g * T::imgs + img;
        if (crow && cok && n < N) {
            if (bnb.seg_n > 0 && !b_crossed && n >= bnb.seg_n) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    b_a0[v] = b_s0[v]; b_a1[v] = b_s1[v]; b_s0[v] = 0.f; b_s1[v] = 0.f;
                    b_mu[v] = bnb.mean[C + c0 + v]; b_rs[v] = bnb.rstd[C + c0 + v];
                }
                b_crossed = true;
            }
            // the conv's raw input e0 at the window's centre, kTileAhead column steps ahead of its use
            const float* bxp = bnb.x + ((n * HW + row) * HW) * C + c0;
            float4 bq[kTileAhead];
#pragma unroll
            for (int d = 0; d < kTileAhead; ++d) bq[d] = *reinterpret_cast<const float4*>(bxp + (int64_t)(d < HW ? d : HW - 1) * C);
            const float* base = &tile[((img * WP + row) * WP) * 64 + vc * 4];      // padded (row, column 0): the window's top-left corner
            float4 a[3], b[3], c[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                a[r] = *reinterpret_cast<const float4*>(base + (r * WP + 0) * 64);
                b[r] = *reinterpret_cast<const float4*>(base + (r * WP + 1) * 64);
            }
            float* out = dst + ((n * HW + row) * HW) * C + c0;
#pragma unroll 1
            for (int ow = 0; ow < HW; ++ow) {
#pragma unroll
                for (int r = 0; r < 3; ++r) c[r] = *reinterpret_cast<const float4*>(base + (r * WP + ow + 2) * 64);
                const float4 bxc = bq[0];
#pragma unroll
                for (int d = 0; d + 1 < kTileAhead; ++d) bq[d] = bq[d + 1];
                bq[kTileAhead - 1] = *reinterpret_cast<const float4*>(bxp + (int64_t)(ow + kTileAhead < HW ? ow + kTileAhead : HW - 1) * C);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float4 w0 = *reinterpret_cast<const float4*>(&wtab[((r * 3 + 0) * 16 + vc) * 4]);
                    const float4 w1 = *reinterpret_cast<const float4*>(&wtab[((r * 3 + 1) * 16 + vc) * 4]);
                    const float4 w2 = *reinterpret_cast<const float4*>(&wtab[((r * 3 + 2) * 16 + vc) * 4]);
                    fma4(acc, a[r], w0); fma4(acc, b[r], w1); fma4(acc, c[r], w2);
                }
                *reinterpret_cast<float4*>(out + (int64_t)ow * C) = acc;
                const float o[4] = {acc.x, acc.y, acc.z, acc.w}, xv[4] = {bxc.x, bxc.y, bxc.z, bxc.w};
                float zc[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float xh = (xv[v] - b_mu[v]) * b_rs[v];
                    const float u = xh * b_ga[v] + b_be[v];
                    float ag;
                    if (bact == SUHPE_ACT_SILU) {           // one sigmoid serves the conv's input z = u sigma(u) and act' = sigma (1 + u (1 - sigma))
                        const float sg = fast_sigmoid(u);
                        zc[v] = u * sg;
                        ag = sg * (1.f + u * (1.f - sg));
                    } else {
                        zc[v] = act_apply(u, bact);
                        ag = act_grad(u, bact);
                    }
                    const float gg = o[v] * ag;
                    b_s0[v] += gg;
                    b_s1[v] += gg * xh;
                }
                const float4 z4 = make_float4(zc[0], zc[1], zc[2], zc[3]);
#pragma unroll
                for (int r = 0; r < 3; ++r) { fma4(accw[r * 3], a[r], z4); fma4(accw[r * 3 + 1], b[r], z4); fma4(accw[r * 3 + 2], c[r], z4); }
#pragma unroll
                for (int r = 0; r < 3; ++r) { a[r] = b[r]; b[r] = c[r]; }
            }
        }
        __syncthreads();
    }
    const bool active = crow && cok;
    {
        // BatchNorm_e sums: threads that share a channel vector, added in a fixed order; records [plane][blockIdx.x][C] float2 (bn_bwd_partial's format)
        const bool two = bnb.seg_n > 0;
        if (two && !b_crossed) {
#pragma unroll
            for (int v = 0; v < 4; ++v) { b_a0[v] = b_s0[v]; b_a1[v] = b_s1[v]; b_s0[v] = 0.f; b_s1[v] = 0.f; }
        }
        float* scratch = tile;
        for (int plane = 0; plane < (two ? 2 : 1); ++plane) {
            const float* q0 = (two && plane == 0) ? b_a0 : b_s0;
            const float* q1 = (two && plane == 0) ? b_a1 : b_s1;
            __syncthreads();
#pragma unroll
            for (int v = 0; v < 4; ++v) { scratch[t * 8 + v] = active ? q0[v] : 0.f; scratch[t * 8 + 4 + v] = active ? q1[v] : 0.f; }
            __syncthreads();
            if (t < 16 && (col0 + t) * 4 < C) {
                float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
                for (int r = 0; r < kTileSlots; ++r) {
                    const float* e = &scratch[(r * 16 + t) * 8];
#pragma unroll
                    for (int v = 0; v < 4; ++v) { a0[v] += e[v]; a1[v] += e[4 + v]; }
                }
                float2* rec = reinterpret_cast<float2*>(bnb.part) + ((int64_t)plane * gridDim.x + blockIdx.x) * C + (col0 + t) * 4;
#pragma unroll
                for (int v = 0; v < 4; ++v) rec[v] = make_float2(a0[v], a1[v]);
            }
        }
    }
    // filter gradient: the 16 row slots of a channel vector, added in a fixed order; window position k = 3 i + j holds tap 8 - k
    __syncthreads();
    float* red = tile;                                   // [k][256][4]: 36 KB of the tile
#pragma unroll
    for (int k = 0; k < 9; ++k) *reinterpret_cast<float4*>(&red[(k * 256 + t) * 4]) = active ? accw[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (t < 16 && (col0 + t) * 4 < C) {
        for (int k = 0; k < 9; ++k) {
            float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int rr = 0; rr < kTileSlots; ++rr) {
                const float4 v = *reinterpret_cast<const float4*>(&red[(k * 256 + rr * 16 + t) * 4]);
                s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
            }
            *reinterpret_cast<float4*>(&wpart[((int64_t)blockIdx.x * 9 + (8 - k)) * C + (col0 + t) * 4]) = s4;
        }
    }
}

// filter-gradient partials with the same sliding window: part[bx][tap][c]
template <bool XF>
__global__ __launch_bounds__(256) void dw3x3_wgrad_row_kernel(const float* __restrict__ x_, const float* __restrict__ dy, float* __restrict__ part,
                                                              int N, int H, int W, int OH, int OW, int C, int stride, int cw, int rp, const DwXform xform) {
    __shared__ __attribute__((aligned(16))) float red[kMaxTaps * 256 * 4];
    const int t = threadIdx.x, col = t % cw, rsub = t / cw;
    const int cv = C / 4, col0 = blockIdx.y * cw;
    float4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rsub < rp && col0 + col < cv) {
        const int coff = (col0 + col) * 4;
        DwXf4 xf = dw_xf_load(xform, coff);
        bool xsecond = false;
        for (int row = blockIdx.x * rp + rsub; row < N * OH; row += gridDim.x * rp) {
            const int n = row / OH, oh = row - n * OH;
            dw_xf_cross<XF>(xf, xsecond, xform, coff, n);
            const int ih0 = oh * stride - 1;
            bool rok[3];
            int64_t rbase[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) { const int ih = ih0 + r; rok[r] = ih >= 0 && ih < H; rbase[r] = ((int64_t)n * H + ih) * W * C + coff; }
            float4 x[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r) { x[r][1] = make_float4(0.f, 0.f, 0.f, 0.f); x[r][2] = ld4_xf<XF>(x_, rok[r], rbase[r], xf); }
            for (int ow = 0; ow < OW; ++ow) {
                const int iw0 = ow * stride - 1;
                if (stride == 1) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        x[r][0] = x[r][1]; x[r][1] = x[r][2];
                        x[r][2] = ld4_xf<XF>(x_, rok[r] && iw0 + 2 < W, rbase[r] + (int64_t)(iw0 + 2) * C, xf);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        x[r][0] = ow == 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : x[r][2];
                        if (ow > 0) x[r][1] = ld4_xf<XF>(x_, rok[r] && iw0 + 1 < W, rbase[r] + (int64_t)(iw0 + 1) * C, xf);
                        else x[r][1] = x[r][2];
                        x[r][2] = ld4_xf<XF>(x_, rok[r] && iw0 + 2 < W, rbase[r] + (int64_t)(iw0 + 2) * C, xf);
                    }
                }
                const float4 dv = *reinterpret_cast<const float4*>(dy + ((int64_t)row * OW + ow) * C + coff);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) fma4(acc[r * 3 + c], x[r][c], dv);
            }
        }
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) *reinterpret_cast<float4*>(&red[(tap * 256 + t) * 4]) = acc[tap];
    __syncthreads();
    if (t < cw && col0 + t < cv) {
        for (int tap = 0; tap < 9; ++tap) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int rr = 0; rr < rp; ++rr) {
                const float4 v = *reinterpret_cast<const float4*>(&red[(tap * 256 + rr * cw + t) * 4]);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            *reinterpret_cast<float4*>(&part[((int64_t)blockIdx.x * 9 + tap) * C + (col0 + t) * 4]) = a;
        }
    }
}

// partial filter gradient: part[bx][tap][c] = sum over the block's pixels of dy[p][c] * x[gather(p,tap)][c]
template <bool XF>
__global__ __launch_bounds__(256) void dw_wgrad_partial(const DwGeom g, const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ part, const DwXform xform) {
    __shared__ __attribute__((aligned(16))) float red[kMaxTaps * 256 * 4];
    const int t = threadIdx.x, col = t % g.cw, rsub = t / g.cw;
    const int cv = g.C / 4, col0 = blockIdx.y * g.cw;
    const int taps = g.R * g.S;
    float4 acc[kMaxTaps];
#pragma unroll
    for (int i = 0; i < kMaxTaps; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rsub < g.rp && col0 + col < cv) {
        DwXf4 xf = dw_xf_load(xform, (col0 + col) * 4);
        bool xsecond = false;
        const int64_t p0 = (int64_t)blockIdx.x * g.pix_per_block;
        int64_t p1 = p0 + g.pix_per_block;
        if (p1 > g.P) p1 = g.P;
        for (int64_t p = p0 + rsub; p < p1; p += g.rp) {
            const int n = (int)(p / ((int64_t)g.OH * g.OW));
            dw_xf_cross<XF>(xf, xsecond, xform, (col0 + col) * 4, n);
            const int rem = (int)(p - (int64_t)n * g.OH * g.OW);
            const int oh = rem / g.OW, ow = rem - oh * g.OW;
            const float4 dv = *reinterpret_cast<const float4*>(dy + p * g.C + (col0 + col) * 4);
#pragma unroll
            for (int tap = 0; tap < kMaxTaps; ++tap) {
                const int r = tap / g.S, s = tap - r * g.S;
                int gh = 0, gw = 0;
                const bool ok = tap < taps && dw_coord(g, oh, r, g.GH, &gh) && dw_coord(g, ow, s, g.GW, &gw);
                if (ok) {
                    const float4 xv = dw_xf<XF>(xf, *reinterpret_cast<const float4*>(x + (((int64_t)n * g.GH + gh) * g.GW + gw) * g.C + (col0 + col) * 4));
                    acc[tap].x += dv.x * xv.x; acc[tap].y += dv.y * xv.y; acc[tap].z += dv.z * xv.z; acc[tap].w += dv.w * xv.w;
                }
            }
        }
    }
#pragma unroll
    for (int tap = 0; tap < kMaxTaps; ++tap) *reinterpret_cast<float4*>(&red[(tap * 256 + t) * 4]) = acc[tap];
    __syncthreads();
    if (t < g.cw && col0 + t < cv) {
        for (int tap = 0; tap < taps; ++tap) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int rr = 0; rr < g.rp; ++rr) {
                const float4 v = *reinterpret_cast<const float4*>(&red[(tap * 256 + rr * g.cw + t) * 4]);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            *reinterpret_cast<float4*>(&part[((int64_t)blockIdx.x * taps + tap) * g.C + (col0 + t) * 4]) = a;
        }
    }
}

// dw[c][tap] (+)= sum_bx part[bx][tap][c]; a block owns 64 consecutive (tap,c) outputs, 4 threads share the row blocks
__global__ __launch_bounds__(256) void dw_wgrad_finalize(const float* __restrict__ part, int gx, int taps, int C, float* __restrict__ dw,
                                                         int accumulate) {
    __shared__ double red[256];
    const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;             // enumerates (tap, c) with c fastest
    double s = 0.0;
    if (i < taps * C)
        for (int b = q; b < gx; b += 4) s += (double)part[(int64_t)b * taps * C + i];
    red[threadIdx.x] = s;
    __syncthreads();
    if (q == 0 && i < taps * C) {
        const double tot = red[o] + red[64 + o] + red[128 + o] + red[192 + o];
        const int tap = i / C, c = i - tap * C;
        float* out = dw + (int64_t)c * taps + tap;
        *out = accumulate ? *out + (float)tot : (float)tot;
    }
}

static int dw_check(const suhpe_conv_t* d, const char* who) {
    if (!d || d->groups != d->cin || d->cin != d->cout || d->cin % 4 || d->r * d->s > kMaxTaps || d->r <= 0 || d->s <= 0 || d->stride <= 0) {
        set_error("%s: needs groups == cin == cout, cin %% 4 == 0 and r*s <= %d", who, kMaxTaps);
        return SUHPE_EINVAL;
    }
    const int ho = (d->h + 2 * d->pad - d->r) / d->stride + 1, wo = (d->w + 2 * d->pad - d->s) / d->stride + 1;
    if (ho != d->ho || wo != d->wo) { set_error("%s: ho/wo do not match the geometry", who); return SUHPE_EINVAL; }
    return SUHPE_OK;
}

static DwGeom dw_geom(const suhpe_conv_t* d, int mode, int max_blocks, int cw_cap = 256) {
    DwGeom g;
    g.N = d->n; g.C = d->cin; g.R = d->r; g.S = d->s; g.stride = d->stride; g.pad = d->pad; g.mode = mode;
    if (mode == 0) { g.GH = d->h; g.GW = d->w; g.OH = d->ho; g.OW = d->wo; }
    else { g.GH = d->ho; g.GW = d->wo; g.OH = d->h; g.OW = d->w; }
    const int cv = g.C / 4;
    g.cw = cv < cw_cap ? cv : cw_cap;
    g.rp = 256 / g.cw;
    g.P = (int64_t)g.N * g.OH * g.OW;
    int64_t gx = ceil_div(g.P, (int64_t)g.rp * 4);
    if (gx > max_blocks) gx = max_blocks;
    if (gx < 1) gx = 1;
    g.pix_per_block = ceil_div(g.P, gx);
    return g;
}

static bool dw_fast(const suhpe_conv_t* d) { return d->r == 3 && d->s == 3 && d->pad == 1; }

// grid for the row-walking kernels: thread = (channel vector, output row); rows are strided over the grid
static void dw_row_grid(int C, int rows, int max_gx, int* cw, int* rp, int* gx, int* ny, int cw_cap = 256) {
    const int cv = C / 4;
    *cw = cv < cw_cap ? cv : cw_cap;
    *rp = 256 / *cw;
    *ny = (cv + *cw - 1) / *cw;
    int64_t g = ceil_div(rows, *rp);
    if (g > max_gx) g = max_gx;
    if (g < 1) g = 1;
    *gx = (int)g;
}

}  // namespace suhpe

using namespace suhpe;

// the input transform is a template flag: the plain kernels carry none of its registers or code
#define DW_LAUNCH(kern, grid, block, lds, st, ...)                                                       \
    do {                                                                                                 \
        if (DW_XF_ON) hipLaunchKernelGGL((kern<true>), grid, block, lds, st, __VA_ARGS__);               \
        else hipLaunchKernelGGL((kern<false>), grid, block, lds, st, __VA_ARGS__);                       \
    } while (0)

// ---- the image-tile kernel's descriptors and grid: 3x3 / pad 1 / stride 1 on 14 x 14 or 7 x 7 maps (SUHPE_DW_TILE=0: off)
static const bool kDwTile = !(getenv("SUHPE_DW_TILE") && atoi(getenv("SUHPE_DW_TILE")) == 0);
static int dw_tile_hw(const suhpe_conv_t* d) {
    if (!kDwTile || !dw_fast(d) || d->stride != 1 || d->h != d->w || !(d->h == 14 || d->h == 7)) return 0;
    return d->h;
}
// persistent workgroups: as many as stay resident at once (64 KB / 41 KB of LDS each), every one the same number of image groups
static void dw_tile_grid(const suhpe_conv_t* d, int hw, int* gx, int* ny, int* ngroups) {
    const int imgs = kTileSlots / hw;
    *ngroups = (d->n + imgs - 1) / imgs;
    *ny = (d->cin / 4 + 15) / 16;
    const int64_t slots = 256 * (hw == 14 ? 2 : 3);
    const int64_t iters = ceil_div((int64_t)*ngroups * *ny, slots);
    *gx = (int)ceil_div((int64_t)*ngroups, iters < 1 ? 1 : iters);
}
template <int HW, bool XF, bool STATS, bool BNB>
static void launch_dw_tile(const suhpe_conv_t* d, const float* src, const float* w, float* dst, int flip, const DwXform& xf, float* stats, int seg_n,
                           const DwBnb& bnb, int act, hipStream_t st) {
    int gx, ny, ngroups;
    dw_tile_grid(d, HW, &gx, &ny, &ngroups);
    auto go = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, DwTile<HW>::lds_bytes);
        hipLaunchKernelGGL(kern, dim3(gx, ny), dim3(256), DwTile<HW>::lds_bytes, st, src, w, dst, d->n, d->cin, flip, ngroups, xf, stats, seg_n, bnb);
    };
    if ((XF || BNB) && act == SUHPE_ACT_SILU) go(dw3x3_tile_kernel<HW, XF, STATS, BNB, SUHPE_ACT_SILU>);
    else go(dw3x3_tile_kernel<HW, XF, STATS, BNB, -1>);
}
static void launch_dw_wgrad_tile(int hw, const suhpe_conv_t* d, const float* x, const float* dy, float* part, int gx, int ny, int ngroups,
                                 const DwXform& xf, hipStream_t st) {
    auto go = [&](auto kern, int lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, dim3(gx, ny), dim3(256), lds, st, x, dy, part, d->n, d->cin, ngroups, xf);
    };
    const bool on = xf.mean != nullptr, silu = on && xf.act == SUHPE_ACT_SILU;
    if (hw == 14) {
        if (silu) go(dw3x3_wgrad_tile_kernel<14, true, SUHPE_ACT_SILU>, DwTile<14>::lds_bytes);
        else if (on) go(dw3x3_wgrad_tile_kernel<14, true, -1>, DwTile<14>::lds_bytes);
        else go(dw3x3_wgrad_tile_kernel<14, false, -1>, DwTile<14>::lds_bytes);
    } else {
        if (silu) go(dw3x3_wgrad_tile_kernel<7, true, SUHPE_ACT_SILU>, DwTile<7>::lds_bytes);
        else if (on) go(dw3x3_wgrad_tile_kernel<7, true, -1>, DwTile<7>::lds_bytes);
        else go(dw3x3_wgrad_tile_kernel<7, false, -1>, DwTile<7>::lds_bytes);
    }
}
template <bool XF, bool STATS, bool BNB>
static void launch_dw_tile_hw(int hw, const suhpe_conv_t* d, const float* src, const float* w, float* dst, int flip, const DwXform& xf, float* stats,
                              int seg_n, const DwBnb& bnb, int act, hipStream_t st) {
    if (hw == 14) launch_dw_tile<14, XF, STATS, BNB>(d, src, w, dst, flip, xf, stats, seg_n, bnb, act, st);
    else launch_dw_tile<7, XF, STATS, BNB>(d, src, w, dst, flip, xf, stats, seg_n, bnb, act, st);
}

extern "C" {

// grid of the fast 3x3 forward kernels (shared by the launch and by suhpe_dwconv_fwd_stats_blocks)
static bool dw_fwd_fast_grid(const suhpe_conv_t* d, bool* s1, int* cw, int* rp, int* gx, int* ny) {
    if (!(dw_fast(d) && (d->stride == 1 || d->stride == 2))) return false;
    // the four-row kernel needs enough row groups to fill the chip; small batches keep one output row per thread
    *s1 = d->stride == 1 && (int64_t)d->n * d->h * d->w * d->cin * 4 < ((int64_t)1 << 31) &&
          (int64_t)d->n * ((d->h + kDwRows - 1) / kDwRows) * (d->cin / 4) >= 65536;
    dw_row_grid(d->cin, *s1 ? d->n * ((d->h + kDwRows - 1) / kDwRows) : d->n * d->ho, 4096, cw, rp, gx, ny);
    return true;
}

static int dwconv_fwd_impl(const suhpe_conv_t* d, const float* x, const float* w, float* y, const DwXform& xf, suhpe_stream_t stream,
                           float* stats = nullptr, int seg_n = 0) {
    int rc = dw_check(d, "suhpe_dwconv_fwd");
    if (rc) return rc;
    SUHPE_REQUIRE(x && w && y, "suhpe_dwconv_fwd: null pointer");
    const bool DW_XF_ON = xf.mean != nullptr;
    bool s1;
    int cw, rp, gx, ny;
    if (const int thw = dw_tile_hw(d)) {
        const DwBnb nb{};
        if (DW_XF_ON) {
            if (stats) launch_dw_tile_hw<true, true, false>(thw, d, x, w, y, 0, xf, stats, seg_n, nb, xf.act, as_stream(stream));
            else launch_dw_tile_hw<true, false, false>(thw, d, x, w, y, 0, xf, stats, seg_n, nb, xf.act, as_stream(stream));
        } else {
            if (stats) launch_dw_tile_hw<false, true, false>(thw, d, x, w, y, 0, xf, stats, seg_n, nb, 0, as_stream(stream));
            else launch_dw_tile_hw<false, false, false>(thw, d, x, w, y, 0, xf, stats, seg_n, nb, 0, as_stream(stream));
        }
        return check_launch(stats ? "suhpe_dwconv_fwd_stats" : "suhpe_dwconv_fwd");
    }
    if (dw_fwd_fast_grid(d, &s1, &cw, &rp, &gx, &ny)) {
        if (stats) {
#define DW_LAUNCH_STATS(kern, ...)                                                                                          \
    do {                                                                                                                    \
        if (DW_XF_ON) hipLaunchKernelGGL((kern<true, true>), dim3(gx, ny), dim3(256), 0, as_stream(stream), __VA_ARGS__);  \
        else hipLaunchKernelGGL((kern<false, true>), dim3(gx, ny), dim3(256), 0, as_stream(stream), __VA_ARGS__);          \
    } while (0)
            if (s1)
                DW_LAUNCH_STATS(dw3x3_s1_kernel, x, (int64_t)d->n * d->h * d->w * d->cin * 4, w, y, d->n, d->h, d->w, d->cin, 0, cw, rp, xf, stats, seg_n);
            else
                DW_LAUNCH_STATS(dw3x3_row_kernel, x, w, y, d->n, d->h, d->w, d->ho, d->wo, d->cin, d->stride, 0, cw, rp, xf, stats, seg_n);
#undef DW_LAUNCH_STATS
            return check_launch("suhpe_dwconv_fwd_stats");
        }
        if (s1)
            DW_LAUNCH(dw3x3_s1_kernel, dim3(gx, ny), dim3(256), 0, as_stream(stream), x, (int64_t)d->n * d->h * d->w * d->cin * 4, w, y,
                               d->n, d->h, d->w, d->cin, 0, cw, rp, xf);
        else
            DW_LAUNCH(dw3x3_row_kernel, dim3(gx, ny), dim3(256), 0, as_stream(stream), x, w, y, d->n, d->h, d->w, d->ho, d->wo, d->cin,
                               d->stride, 0, cw, rp, xf);
        return check_launch("suhpe_dwconv_fwd");
    }
    SUHPE_REQUIRE(!stats, "suhpe_dwconv_fwd_stats: only the 3x3 / pad 1 / stride 1-2 kernels emit statistics (suhpe_dwconv_fwd_stats_blocks returns 0 otherwise)");
    const DwGeom g = dw_geom(d, 0, 2048);
    const int gxg = (int)ceil_div(g.P, g.pix_per_block), nyg = (int)ceil_div(g.C / 4, g.cw);
    DW_LAUNCH(dw_conv_kernel, dim3(gxg, nyg), dim3(256), 0, as_stream(stream), g, x, w, y, xf);
    return check_launch("suhpe_dwconv_fwd");
}

static const DwXform kNoXform{nullptr, nullptr, nullptr, nullptr, 0, 0, 0};

int suhpe_dwconv_fwd(const suhpe_conv_t* d, const float* x, const float* w, float* y, suhpe_stream_t stream) {
    return dwconv_fwd_impl(d, x, w, y, kNoXform, stream);
}

int suhpe_dwconv_fwd_stats_blocks(const suhpe_conv_t* d) {
    if (!d || dw_check(d, "suhpe_dwconv_fwd_stats_blocks")) return 0;
    bool s1;
    int cw, rp, gx, ny;
    if (const int thw = dw_tile_hw(d)) {
        int ngroups;
        dw_tile_grid(d, thw, &gx, &ny, &ngroups);
        return gx;
    }
    return dw_fwd_fast_grid(d, &s1, &cw, &rp, &gx, &ny) ? gx : 0;
}

int suhpe_dwconv_fwd_stats(const suhpe_conv_t* d, const float* x, const float* w, float* y, float* stats, int seg_n, suhpe_stream_t stream) {
    SUHPE_REQUIRE(stats && d && seg_n >= 0 && seg_n < d->n, "suhpe_dwconv_fwd_stats: stats required, 0 <= seg_n < n");
    return dwconv_fwd_impl(d, x, w, y, kNoXform, stream, stats, seg_n);
}

int suhpe_dwconv_bnact_fwd(const suhpe_conv_t* d, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           int act, int seg_n, const float* w, float* y, float* stats, suhpe_stream_t stream) {
    SUHPE_REQUIRE(d && mean && rstd, "suhpe_dwconv_bnact_fwd: mean / rstd required");
    SUHPE_REQUIRE(seg_n >= 0 && seg_n < d->n, "suhpe_dwconv_bnact_fwd: 0 <= seg_n < n");
    SUHPE_REQUIRE(!stats || suhpe_dwconv_fwd_stats_blocks(d) > 0, "suhpe_dwconv_bnact_fwd: this shape has no statistics epilogue (suhpe_dwconv_fwd_stats_blocks)");
    return dwconv_fwd_impl(d, x, w, y, DwXform{mean, rstd, gamma, beta, act, seg_n, d->cin}, stream, stats, seg_n);
}

int suhpe_dwconv_dgrad(const suhpe_conv_t* d, const float* dy, const float* w, float* dx, suhpe_stream_t stream) {
    int rc = dw_check(d, "suhpe_dwconv_dgrad");
    if (rc) return rc;
    SUHPE_REQUIRE(dy && w && dx, "suhpe_dwconv_dgrad: null pointer");
    const bool DW_XF_ON = false;
    if (const int thw = dw_tile_hw(d)) {
        launch_dw_tile_hw<false, false, false>(thw, d, dy, w, dx, 1, kNoXform, nullptr, 0, DwBnb{}, 0, as_stream(stream));
        return check_launch("suhpe_dwconv_dgrad");
    }
    if (dw_fast(d) && d->stride == 1) {             // same conv on dy with the 180-degree rotated filter
        int cw, rp, gx, ny;
        const bool s1 = (int64_t)d->n * d->h * d->w * d->cin * 4 < ((int64_t)1 << 31) &&
                        (int64_t)d->n * ((d->h + kDwRows - 1) / kDwRows) * (d->cin / 4) >= 65536;
        dw_row_grid(d->cin, s1 ? d->n * ((d->h + kDwRows - 1) / kDwRows) : d->n * d->h, 4096, &cw, &rp, &gx, &ny);
        if (s1)
            DW_LAUNCH(dw3x3_s1_kernel, dim3(gx, ny), dim3(256), 0, as_stream(stream), dy, (int64_t)d->n * d->h * d->w * d->cin * 4, w, dx,
                               d->n, d->h, d->w, d->cin, 1, cw, rp, kNoXform);
        else
            DW_LAUNCH(dw3x3_row_kernel, dim3(gx, ny), dim3(256), 0, as_stream(stream), dy, w, dx, d->n, d->ho, d->wo, d->h, d->w, d->cin, 1, 1,
                               cw, rp, kNoXform);
        return check_launch("suhpe_dwconv_dgrad");
    }
    const DwGeom g = dw_geom(d, 1, 2048);
    const int gx = (int)ceil_div(g.P, g.pix_per_block), ny = (int)ceil_div(g.C / 4, g.cw);
    DW_LAUNCH(dw_conv_kernel, dim3(gx, ny), dim3(256), 0, as_stream(stream), g, dy, w, dx, kNoXform);
    return check_launch("suhpe_dwconv_dgrad");
}

// which descriptors suhpe_dwconv_dgrad_bnsums serves: the sliding-window stride-1 kernel's (3x3 / pad 1 / stride 1, large tensors)
static bool dw_bnsums_ok(const suhpe_conv_t* d) {
    if (d && d->cin % 4 == 0 && dw_tile_hw(d)) return true;
    return d && dw_fast(d) && d->stride == 1 && d->cin % 4 == 0 && (int64_t)d->n * d->h * d->w * d->cin * 4 < ((int64_t)1 << 31) &&
           (int64_t)d->n * ((d->h + kDwRows - 1) / kDwRows) * (d->cin / 4) >= 65536;
}

int suhpe_dwconv_dgrad_bnsums_ok(const suhpe_conv_t* d) { return dw_bnsums_ok(d) ? 1 : 0; }

int64_t suhpe_dwconv_dgrad_bnsums_workspace_bytes(const suhpe_conv_t* d) {
    if (!dw_bnsums_ok(d)) return 0;
    int cw, rp, gx, ny;
    if (const int thw = dw_tile_hw(d)) {
        int ngroups;
        dw_tile_grid(d, thw, &gx, &ny, &ngroups);
    } else {
        dw_row_grid(d->cin, d->n * ((d->h + kDwRows - 1) / kDwRows), 4096, &cw, &rp, &gx, &ny);
    }
    return (int64_t)2 * gx * d->cin * 2 * sizeof(float);
}

int suhpe_dwconv_dgrad_bnsums(const suhpe_conv_t* d, const float* dy, const float* w, float* dx, const float* bn_x, const float* mean,
                              const float* rstd, const float* gamma, const float* beta, int act, int seg_n, float* sums, float* dgamma,
                              float* dbeta, int accumulate, void* workspace, int64_t workspace_bytes, suhpe_stream_t stream) {
    int rc = dw_check(d, "suhpe_dwconv_dgrad_bnsums");
    if (rc) return rc;
    SUHPE_REQUIRE(dw_bnsums_ok(d), "suhpe_dwconv_dgrad_bnsums: 3x3 / pad 1 / stride 1 depthwise convs of the sliding-window kernel only (see _ok)");
    SUHPE_REQUIRE(dy && w && dx && bn_x && mean && rstd && sums && workspace && seg_n >= 0 && seg_n < d->n, "suhpe_dwconv_dgrad_bnsums: bad arguments");
    SUHPE_REQUIRE(act >= SUHPE_ACT_NONE && act <= SUHPE_ACT_SIGMOID, "suhpe_dwconv_dgrad_bnsums: unknown activation %d", act);
    if (workspace_bytes < suhpe_dwconv_dgrad_bnsums_workspace_bytes(d)) {
        set_error("suhpe_dwconv_dgrad_bnsums: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)suhpe_dwconv_dgrad_bnsums_workspace_bytes(d));
        return SUHPE_EWORKSPACE;
    }
    int cw, rp, gx, ny;
    float* part = static_cast<float*>(workspace);
    const DwBnb bnb{bn_x, mean, rstd, gamma, beta, act, seg_n, part};
    if (const int thw = dw_tile_hw(d)) {
        int ngroups;
        dw_tile_grid(d, thw, &gx, &ny, &ngroups);
        launch_dw_tile_hw<false, false, true>(thw, d, dy, w, dx, 1, kNoXform, nullptr, 0, bnb, act, as_stream(stream));
    } else {
        dw_row_grid(d->cin, d->n * ((d->h + kDwRows - 1) / kDwRows), 4096, &cw, &rp, &gx, &ny);
        hipLaunchKernelGGL((dw3x3_s1_kernel<false, false, true>), dim3(gx, ny), dim3(256), 0, as_stream(stream), dy, (int64_t)d->n * d->h * d->w * d->cin * 4,
                           w, dx, d->n, d->h, d->w, d->cin, 1, cw, rp, kNoXform, nullptr, 0, bnb);
    }
    // finish: (sum g, sum g xhat) per channel and segment -> sums [c][2] / [2][c][2], dgamma / dbeta (nn_elem.hip's finalisers)
    launch_bn_bwd_finalize(part, seg_n > 0 ? gx : 0, seg_n > 0 ? 2 * gx : gx, d->cin, sums, dgamma, dbeta, accumulate, as_stream(stream));
    return check_launch("suhpe_dwconv_dgrad_bnsums");
}

// ---- the depthwise conv's whole backward in one launch (dw3x3_bwd_tile_kernel): 3x3 / pad 1 / stride 1 on 14 x 14 or 7 x 7 maps
static void dw_bwd_tile_grid(const suhpe_conv_t* d, int hw, int* gx, int* ny, int* ngroups) {
    const int imgs = kTileSlots / hw;
    *ngroups = (d->n + imgs - 1) / imgs;
    *ny = (d->cin / 4 + 15) / 16;
    const int64_t slots = 256 * 2;                          // two resident workgroups per compute unit (register budget of the fused kernel)
    const int64_t iters = ceil_div((int64_t)*ngroups * *ny, slots);
    *gx = (int)ceil_div((int64_t)*ngroups, iters < 1 ? 1 : iters);
}
static const bool kDwBwdFused = !(getenv("SUHPE_DW_BWD_FUSED") && atoi(getenv("SUHPE_DW_BWD_FUSED")) == 0);
int suhpe_dwconv_bwd_fused_ok(const suhpe_conv_t* d) {
    return (kDwBwdFused && d && d->groups == d->cin && d->cin == d->cout && d->cin % 4 == 0 && dw_tile_hw(d)) ? 1 : 0;
}
int64_t suhpe_dwconv_bwd_fused_workspace_bytes(const suhpe_conv_t* d) {
    if (!suhpe_dwconv_bwd_fused_ok(d)) return 0;
    int gx, ny, ngroups;
    dw_bwd_tile_grid(d, dw_tile_hw(d), &gx, &ny, &ngroups);
    return ((int64_t)2 * gx * d->cin * 2 + (int64_t)gx * 9 * d->cin) * sizeof(float);
}
int suhpe_dwconv_bwd_fused(const suhpe_conv_t* d, const float* dy, const float* w, float* dx, const float* bn_x, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int act, int seg_n, float* sums, float* dgamma, float* dbeta, int accumulate, float* dw,
                           int dw_accumulate, const float* mid_y, const float* mid_s, const float* mid_dpool, const float* mid_chan, const float* mid_mean,
                           const float* mid_rstd, const float* mid_gamma, const float* mid_beta, int mid_act, int mid_train_stats, void* workspace,
                           int64_t workspace_bytes, suhpe_stream_t stream) {
    int rc = dw_check(d, "suhpe_dwconv_bwd_fused");
    if (rc) return rc;
    SUHPE_REQUIRE(suhpe_dwconv_bwd_fused_ok(d), "suhpe_dwconv_bwd_fused: 3x3 / pad 1 / stride 1 depthwise convs on 14 x 14 or 7 x 7 maps only (see _ok)");
    SUHPE_REQUIRE(dy && w && dx && bn_x && mean && rstd && sums && dw && workspace && seg_n >= 0 && seg_n < d->n, "suhpe_dwconv_bwd_fused: bad arguments");
    SUHPE_REQUIRE(act >= SUHPE_ACT_NONE && act <= SUHPE_ACT_SIGMOID && mid_act >= SUHPE_ACT_NONE && mid_act <= SUHPE_ACT_SIGMOID, "suhpe_dwconv_bwd_fused: unknown activation");
    SUHPE_REQUIRE(!mid_y || (mid_s && mid_dpool && mid_mean && mid_rstd && (mid_chan || !mid_train_stats)),
                  "suhpe_dwconv_bwd_fused: the section form (mid_y) needs s, dpool, mean, rstd and, in train mode, the channel sums");
    if (workspace_bytes < suhpe_dwconv_bwd_fused_workspace_bytes(d)) {
        set_error("suhpe_dwconv_bwd_fused: workspace %lld < %lld bytes", (long long)workspace_bytes, (long long)suhpe_dwconv_bwd_fused_workspace_bytes(d));
        return SUHPE_EWORKSPACE;
    }
    const int hw = dw_tile_hw(d);
    int gx, ny, ngroups;
    dw_bwd_tile_grid(d, hw, &gx, &ny, &ngroups);
    float* part = static_cast<float*>(workspace);
    float* wpart = part + (int64_t)2 * gx * d->cin * 2;
    const DwBnb bnb{bn_x, mean, rstd, gamma, beta, act, seg_n, part};
    const int64_t rows = (int64_t)d->n * hw * hw, rows_a = (int64_t)seg_n * hw * hw;
    const DwMid mid{mid_y, mid_s, mid_dpool, mid_chan, mid_mean, mid_rstd, mid_gamma, mid_beta, mid_act, seg_n, mid_train_stats,
                    1.0f / (float)(rows_a > 0 ? rows_a : rows), 1.0f / (float)(rows - rows_a), 1.0f / (float)(hw * hw)};
    hipStream_t st = as_stream(stream);
    auto go = [&](auto kern, int lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, dim3(gx, ny), dim3(256), lds, st, dy, w, dx, d->n, d->cin, ngroups, bnb, mid, wpart);
    };
    const bool silu = act == SUHPE_ACT_SILU && (!mid_y || mid_act == SUHPE_ACT_SILU);
    constexpr int kTab = (2 * 4 * 16 * 4 + 9 * 64) * (int)sizeof(float);      // BatchNorm_d constants of both segments + the flipped filter
    if (hw == 14) {
        if (mid_y) { if (silu) go(dw3x3_bwd_tile_kernel<14, true, SUHPE_ACT_SILU>, DwTile<14>::lds_bytes + kTab); else go(dw3x3_bwd_tile_kernel<14, true, -1>, DwTile<14>::lds_bytes + kTab); }
        else { if (silu) go(dw3x3_bwd_tile_kernel<14, false, SUHPE_ACT_SILU>, DwTile<14>::lds_bytes + kTab); else go(dw3x3_bwd_tile_kernel<14, false, -1>, DwTile<14>::lds_bytes + kTab); }
    } else {
        if (mid_y) { if (silu) go(dw3x3_bwd_tile_kernel<7, true, SUHPE_ACT_SILU>, DwTile<7>::lds_bytes + kTab); else go(dw3x3_bwd_tile_kernel<7, true, -1>, DwTile<7>::lds_bytes + kTab); }
        else { if (silu) go(dw3x3_bwd_tile_kernel<7, false, SUHPE_ACT_SILU>, DwTile<7>::lds_bytes + kTab); else go(dw3x3_bwd_tile_kernel<7, false, -1>, DwTile<7>::lds_bytes + kTab); }
    }
    // finish: (sum g, sum g xhat) per channel and segment -> sums, dgamma / dbeta; filter-gradient records -> dw (fp64, fixed order)
    launch_bn_bwd_finalize(part, seg_n > 0 ? gx : 0, seg_n > 0 ? 2 * gx : gx, d->cin, sums, dgamma, dbeta, accumulate, st);
    hipLaunchKernelGGL(dw_wgrad_finalize, dim3((9 * d->cin + 63) / 64), dim3(256), 0, st, wpart, gx, 9, d->cin, dw, dw_accumulate);
    return check_launch("suhpe_dwconv_bwd_fused");
}

int64_t suhpe_dwconv_wgrad_workspace_bytes(const suhpe_conv_t* d) {
    if (!d) return 0;
    return (int64_t)kDwMaxRowBlocks * d->r * d->s * d->cin * sizeof(float);
}

static int dwconv_wgrad_impl(const suhpe_conv_t* d, const float* x, const float* dy, float* dw, int accumulate, void* workspace,
                             int64_t workspace_bytes, const DwXform& xf, suhpe_stream_t stream) {
    int rc = dw_check(d, "suhpe_dwconv_wgrad");
    if (rc) return rc;
    SUHPE_REQUIRE(x && dy && dw && workspace, "suhpe_dwconv_wgrad: null pointer");
    const bool DW_XF_ON = xf.mean != nullptr;
    if (workspace_bytes < suhpe_dwconv_wgrad_workspace_bytes(d)) { set_error("suhpe_dwconv_wgrad: workspace too small"); return SUHPE_EWORKSPACE; }
    float* part = static_cast<float*>(workspace);
    const int taps = d->r * d->s;
    if (const int thw = dw_tile_hw(d)) {
        int gx, ny, ngroups;
        dw_tile_grid(d, thw, &gx, &ny, &ngroups);
        if (gx > kDwMaxRowBlocks) gx = kDwMaxRowBlocks;                  // the workspace holds that many record sets
        launch_dw_wgrad_tile(thw, d, x, dy, part, gx, ny, ngroups, xf, as_stream(stream));
        hipLaunchKernelGGL(dw_wgrad_finalize, dim3((taps * d->cin + 63) / 64), dim3(256), 0, as_stream(stream), part, gx, taps, d->cin, dw, accumulate);
        return check_launch("suhpe_dwconv_wgrad");
    }
    if (dw_fast(d) && (d->stride == 1 || d->stride == 2)) {
        int cw, rp, gx, ny;
        dw_row_grid(d->cin, d->n * d->ho, kDwMaxRowBlocks, &cw, &rp, &gx, &ny, 32);
        DW_LAUNCH(dw3x3_wgrad_row_kernel, dim3(gx, ny), dim3(256), 0, as_stream(stream), x, dy, part, d->n, d->h, d->w, d->ho, d->wo,
                           d->cin, d->stride, cw, rp, xf);
        hipLaunchKernelGGL(dw_wgrad_finalize, dim3((taps * d->cin + 63) / 64), dim3(256), 0, as_stream(stream), part, gx, taps, d->cin, dw, accumulate);
        return check_launch("suhpe_dwconv_wgrad");
    }
    const DwGeom g = dw_geom(d, 0, kDwMaxRowBlocks, 32);      // narrow channel chunks: more workgroups per slab
    const int gx = (int)ceil_div(g.P, g.pix_per_block), ny = (int)ceil_div(g.C / 4, g.cw);
    DW_LAUNCH(dw_wgrad_partial, dim3(gx, ny), dim3(256), 0, as_stream(stream), g, x, dy, part, xf);
    hipLaunchKernelGGL(dw_wgrad_finalize, dim3((taps * g.C + 63) / 64), dim3(256), 0, as_stream(stream), part, gx, taps, g.C, dw, accumulate);
    return check_launch("suhpe_dwconv_wgrad");
}

int suhpe_dwconv_wgrad(const suhpe_conv_t* d, const float* x, const float* dy, float* dw, int accumulate, void* workspace,
                       int64_t workspace_bytes, suhpe_stream_t stream) {
    return dwconv_wgrad_impl(d, x, dy, dw, accumulate, workspace, workspace_bytes, kNoXform, stream);
}

int suhpe_dwconv_bnact_wgrad(const suhpe_conv_t* d, const float* x, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, int act, int seg_n, const float* dy, float* dw, int accumulate, void* workspace,
                             int64_t workspace_bytes, suhpe_stream_t stream) {
    SUHPE_REQUIRE(d && mean && rstd, "suhpe_dwconv_bnact_wgrad: mean / rstd required");
    SUHPE_REQUIRE(seg_n >= 0 && seg_n < d->n, "suhpe_dwconv_bnact_wgrad: 0 <= seg_n < n");
    return dwconv_wgrad_impl(d, x, dy, dw, accumulate, workspace, workspace_bytes, DwXform{mean, rstd, gamma, beta, act, seg_n, d->cin}, stream);
}

}  // extern "C"
