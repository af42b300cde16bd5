// HBM-bound layer kernels of the backbone on gfx950: BatchNorm (train/eval) fused with the activation,
// LayerNorm, pooling, squeeze-excite scaling, residual/stochastic-depth, dropout masks, loss reductions,
// Adam and the EMA teacher update.  All tensors are fp32, activations are pixel-major (rows = n*h*w, c fastest),
// every access is a 16-byte vector per lane when c % 4 == 0 (scalar fallback otherwise), reductions are
// per-thread fp32 partials -> LDS -> fixed-order fp64 finalisation (deterministic, no atomics).
//
// Reference ops replaced: nn.BatchNorm2d/1d, nn.SiLU/ReLU/ReLU6/GELU/Sigmoid, nn.AdaptiveAvgPool2d,
// torchvision SqueezeExcitation / StochasticDepth("row"), nn.Dropout, nn.LayerNorm, optim.Adam (src/agent.py:37),
// the EMA loop (src/agent.py:298-299).
#include "common.h"
#include <cstdlib>

namespace suhpe {

constexpr int kRowBlockBudget = 2048;              // row blocks x column chunks of one reduction launch (8 blocks per CU)
constexpr int64_t kPartFloats = 2048LL * 256 * 4;  // record workspace: gx*c <= 2048*256 records of up to 4 floats

// ------------------------------------------------------------------------------------------ column reductions
// Geometry shared by every per-channel reduction over a (rows, c) matrix.  A block owns `cw` consecutive
// columns (vec-columns when VEC=4) and a contiguous slab of rows; thread t -> (col = t % cw, rsub = t / cw).
struct ColGeom {
    int cv;        // columns in vector units
    int cw;        // columns per block (<=64)
    int ny;        // column chunks
    int rp;        // rows per pass = 256 / cw
    int gx;        // row blocks
    int64_t rows_per_block;
};

static ColGeom col_geom(int64_t rows, int c, int vec) {
    ColGeom g;
    g.cv = c / vec;
    g.cw = g.cv < 64 ? g.cv : 64;
    g.ny = (g.cv + g.cw - 1) / g.cw;
    g.rp = 256 / g.cw;
    int64_t want = ceil_div(rows, (int64_t)g.rp * 8);          // >= 8 passes per block
    int64_t cap = kRowBlockBudget / g.ny > 1 ? kRowBlockBudget / g.ny : 1;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    g.gx = (int)want;
    g.rows_per_block = ceil_div(rows, g.gx);
    return g;
}

// Geometry of the row/column elementwise kernels: thread -> (fixed vector column, rows strided by rp); no per-element
// integer division, per-channel constants live in registers.
struct EwGeom { int cw, ny, rp, gx; int rows_per_block; };
static EwGeom ew_geom(int64_t rows, int c, int vec) {
    EwGeom g;
    const int cv = c / vec;
    g.cw = cv < 64 ? cv : 64;
    g.ny = (cv + g.cw - 1) / g.cw;
    g.rp = 256 / g.cw;
    int64_t gx = ceil_div(rows, (int64_t)g.rp * 8);
    const int64_t cap = 4096 / g.ny > 1 ? 4096 / g.ny : 1;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    g.gx = (int)gx;
    g.rows_per_block = (int)ceil_div(rows, gx);
    return g;
}
#define EW_BX blockIdx.x
#define EW_THREAD_SETUP(VEC)                                                              \
    const int t = threadIdx.x, col = t % cw, rsub = t / cw;                               \
    const int cv = c / VEC, vc = blockIdx.y * cw + col;                                   \
    if (rsub >= rp || vc >= cv) return;                                                   \
    const int ch = vc * VEC; (void)ch;                                                    \
    const int r0 = EW_BX * rows_per_block;                                                \
    int r1 = r0 + rows_per_block;                                                         \
    if (r1 > rows) r1 = rows;

// Two batch segments in one launch.  Row blocks [0, nb_a) cover rows [0, rows_a) with the first statistics set, the blocks
// after them the remaining rows with the second set: every per-channel array of statistics (mean, rstd, backward sums) is
// then [2][c], and the reduction records of the two segments never mix.  nb_a == 0: ordinary single-segment launch.
struct SegArgs { int64_t rows_a; int nb_a; int rev; };     // rev: walk the row blocks from the last to the first (see kRevSecondPass)
// -> the block's index inside its segment; `rows` becomes the segment's row count, row_off its first row, set its index
__device__ __forceinline__ int seg_select(const SegArgs& sg, int64_t* rows, int64_t* row_off, int* set) {
    int bx = sg.rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    *row_off = 0;
    *set = 0;
    if (sg.nb_a > 0) {
        if (bx >= sg.nb_a) { bx -= sg.nb_a; *row_off = sg.rows_a; *rows -= sg.rows_a; *set = 1; }
        else *rows = sg.rows_a;
    }
    return bx;
}

// The second pass of a two-pass op (BatchNorm backward: sums, then apply) re-reads what the first pass just streamed.  Walking its row
// blocks from the last to the first meets the most recently touched lines first, i.e. whatever part of the tensors is still in
// L2 / Infinity Cache; in the same order as the first pass the cached tail is evicted before it is reached.  SUHPE_REV_PASS=0: off.
static const int kRevSecondPass = !(getenv("SUHPE_REV_PASS") && atoi(getenv("SUHPE_REV_PASS")) == 0);

template <int VEC> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<1> { using type = float; };

template <int VEC> __device__ __forceinline__ void vload(const float* p, float* o) {
    if (VEC == 4) { const float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    else o[0] = p[0];
}
template <int VEC> __device__ __forceinline__ void vstore(float* p, const float* o) {
    if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    else p[0] = o[0];
}

// Block-level combine of per-thread partial sums that share a column: LDS [256][2*VEC] then the first cw
// threads add the rp partials of their column in fixed order and write one record per channel to part[bx][channel]
// (a block's records are contiguous: coalesced writes).  The finalisers give a 1024-thread block to 8 channels:
// thread = (channel, one of 128 row-block lanes), line-sized contiguous reads, fp64 accumulation, fixed-order reduction.
constexpr int kFinCh = 8, kFinThreads = 1024, kFinSub = kFinThreads / kFinCh;
// thread = (channel chl = t & 7, row-block lane sub = t >> 3): adds the 128 row-block lanes of each channel in a fixed order
// (xor-shuffles inside the wave, then 16 wave slots through LDS).  True for the thread that ends up owning the totals.
__device__ __forceinline__ bool fin_reduce(double* a, double* b, int ch, int c) {
    __shared__ double red[kFinThreads / 64][kFinCh][2];
    double x = *a, y = *b;
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) { x += __shfl_xor(x, m, 64); y += __shfl_xor(y, m, 64); }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (lane < kFinCh) { red[wave][lane][0] = x; red[wave][lane][1] = y; }
    __syncthreads();
    if (t >= kFinCh || ch >= c) return false;
    x = red[0][t][0]; y = red[0][t][1];
#pragma unroll
    for (int w = 1; w < kFinThreads / 64; ++w) { x += red[w][t][0]; y += red[w][t][1]; }
    *a = x;
    *b = y;
    return true;
}
__device__ __forceinline__ bool block_channel_sums(const float* __restrict__ part, int gx, int c, double* s, double* q) {
    const int chl = threadIdx.x & (kFinCh - 1), sub = threadIdx.x / kFinCh;
    const int ch = blockIdx.x * kFinCh + chl;
    double a = 0.0, b = 0.0;
    if (ch < c) {
        const float2* p = reinterpret_cast<const float2*>(part) + ch;
#pragma unroll 8
        for (int i = sub; i < gx; i += kFinSub) { const float2 v = p[(int64_t)i * c]; a += (double)v.x; b += (double)v.y; }
    }
    *s = a;
    *q = b;
    return fin_reduce(s, q, ch, c);
}
// shift != nullptr selects the 4-float statistics record (S, Q, k, n) instead of the plain pair.
template <int VEC>
__device__ __forceinline__ void block_combine(const float* s0, const float* s1, int cw, int rp, int col0, int c,
                                              int bx, float* __restrict__ part, float* lds, const float* shift = nullptr,
                                              float count = 0.f) {
    const int t = threadIdx.x;
#pragma unroll
    for (int v = 0; v < VEC; ++v) { lds[(t * 2 + 0) * VEC + v] = s0[v]; lds[(t * 2 + 1) * VEC + v] = s1[v]; }
    __syncthreads();
    if (t < cw && (col0 + t) * VEC < c) {
        float a0[VEC], a1[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { a0[v] = 0.f; a1[v] = 0.f; }
        for (int r = 0; r < rp; ++r) {
            const int tt = r * cw + t;
#pragma unroll
            for (int v = 0; v < VEC; ++v) { a0[v] += lds[(tt * 2 + 0) * VEC + v]; a1[v] += lds[(tt * 2 + 1) * VEC + v]; }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int ch = (col0 + t) * VEC + v;
            if (shift) {
                reinterpret_cast<float4*>(part)[(int64_t)bx * c + ch] = make_float4(a0[v], a1[v], shift[v], count);
            } else {
                reinterpret_cast<float2*>(part)[(int64_t)bx * c + ch] = make_float2(a0[v], a1[v]);      // [row block][channel]
            }
        }
    }
}

// BN statistics, pass 1: per-block record (S, Q, k, n) per channel: sums of (x-k) and (x-k)^2 over the block's n rows,
// shifted by k = the block's first-row value, so the variance is never the difference of two large sums.
template <int VEC>
__global__ __launch_bounds__(256) void bn_stats_partial(const float* __restrict__ x, int64_t rows, int c, int cw, int rp,
                                                        int64_t rows_per_block, float* __restrict__ part, const SegArgs sg = SegArgs{0, 0, 0}) {
    __shared__ float lds[256 * 2 * VEC];
    int64_t row_off;
    int set;
    const int bx = seg_select(sg, &rows, &row_off, &set);
    x += row_off * c;
    const int t = threadIdx.x, col = t % cw, rsub = t / cw;
    const int col0 = blockIdx.y * cw;
    const int cv = c / VEC;
    float s0[VEC], s1[VEC], k[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { s0[v] = 0.f; s1[v] = 0.f; k[v] = 0.f; }
    const int64_t r0 = (int64_t)bx * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    if (rsub < rp && col0 + col < cv && r0 < rows) {
        vload<VEC>(x + (r0 * cv + col0 + col) * VEC, k);
        int64_t r = r0 + rsub;
        // eight independent row streams per thread (8 x 16 B in flight; the pass is pure HBM latency otherwise); rows past
        // the slab read a clamped address and are masked, so no branch separates the loads
        for (; r < r1; r += 8 * (int64_t)rp) {
            float xv[8][VEC];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t rr = r + u * (int64_t)rp;
                vload<VEC>(x + ((rr < r1 ? rr : r1 - 1) * cv + col0 + col) * VEC, xv[u]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = r + u * (int64_t)rp < r1;
#pragma unroll
                for (int v = 0; v < VEC; ++v) { const float a = ok ? xv[u][v] - k[v] : 0.f; s0[v] += a; s1[v] += a * a; }
            }
        }
    }
    block_combine<VEC>(s0, s1, cw, rp, col0, c, blockIdx.x, part, lds, k, r1 > r0 ? (float)(r1 - r0) : 0.f);
}

// pass 2: 8 channels per 1024-thread block; running-stat update as nn.BatchNorm does.  Records are [row block][channel] float4.
__device__ __forceinline__ void bn_stats_finish(const float* __restrict__ part, int gx, int c, int64_t rows, float eps, float momentum,
                                                float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                                                float* __restrict__ running_var, const float* __restrict__ extra) {
    const int chl = threadIdx.x & (kFinCh - 1), sub = threadIdx.x / kFinCh;
    const int ch = blockIdx.x * kFinCh + chl;
    // re-centre every record on zero in fp64 (exact to 1e-16) and add: sum x = S + n k, sum x^2 = Q + 2 k S + n k^2
    double a = 0.0, b = 0.0, k0 = 0.0;
    if (ch < c) {
        const float4* pr = reinterpret_cast<const float4*>(part) + ch;
        const float4* ex = extra ? reinterpret_cast<const float4*>(extra) + ch : nullptr;      // one more record (segment boundary tile)
        k0 = gx > 0 ? (double)pr[0].z : (double)ex[0].z;
#pragma unroll 8
        for (int i = sub; i < gx; i += kFinSub) {
            const float4 v = pr[(int64_t)i * c];
            const double S = v.x, Q = v.y, k = v.z, n = v.w;
            a += S + n * k;
            b += Q + 2.0 * k * S + n * k * k;
        }
        if (ex && sub == 0) {
            const float4 v = ex[0];
            const double S = v.x, Q = v.y, k = v.z, n = v.w;
            a += S + n * k;
            b += Q + 2.0 * k * S + n * k * k;
        }
    }
    if (!fin_reduce(&a, &b, ch, c)) return;
    const double s = a, q = b;
    // centre on the first record's shift before the final subtraction: var = E[(x-k0)^2] - (E[x]-k0)^2
    const double m = s / (double)rows;
    double var = (q - 2.0 * k0 * s + (double)rows * k0 * k0) / (double)rows - (m - k0) * (m - k0);
    if (var < 0.0) var = 0.0;
    mean[ch] = (float)m;
    rstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
        running_mean[ch] = (float)((1.0 - momentum) * (double)running_mean[ch] + momentum * m);
        running_var[ch] = (float)((1.0 - momentum) * (double)running_var[ch] + momentum * unb);
    }
}

__global__ __launch_bounds__(kFinThreads) void bn_stats_finalize(const float* __restrict__ part, int gx, int c, int64_t rows, float eps,
                                                         float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                                         int64_t* __restrict__ nbt, const float* __restrict__ extra = nullptr) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
    bn_stats_finish(part, gx, c, rows, eps, momentum, mean, rstd, running_mean, running_var, extra);
}

// two segments: records [0, gx_a) then [gx_a, gx_a + gx_b) (+ the boundary record of a conv epilogue, which belongs to the
// second segment); statistics to mean/rstd[0..c) and [c..2c); the running statistics take the two momentum updates in order
// (the same thread owns a channel in both rounds).
__global__ __launch_bounds__(kFinThreads) void bn_stats_finalize2(const float* __restrict__ part, int gx_a, int gx_b, int c, int64_t rows_a,
                                                          int64_t rows_b, float eps, float momentum, float* __restrict__ mean,
                                                          float* __restrict__ rstd, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, int64_t* __restrict__ nbt,
                                                          const float* __restrict__ extra_b) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 2;
    bn_stats_finish(part, gx_a, c, rows_a, eps, momentum, mean, rstd, running_mean, running_var, nullptr);
    __syncthreads();                                     // the reduction scratch is reused
    bn_stats_finish(part + (int64_t)gx_a * c * 4, gx_b, c, rows_b, eps, momentum, mean + c, rstd + c, running_mean, running_var, extra_b);
}

// ---- long record lists (the 112 x 112 and 56 x 56 layers of a 160-image batch leave 4 000 - 16 000 records per channel): the finalizers above
// run ceil(c / 8) workgroups, so a 24-channel layer reduced 6 MB of records on THREE compute units (62 us; rocprofv3 kernel trace of the C2 step).
// bn_records_fold spreads the same reduction over the chip: workgroup (channel group, run) adds the records of one run, re-centred on zero in fp64
// as bn_stats_finish does, and leaves (sum x, sum x^2) as two doubles IN the run's first record (16 bytes either way; the run's other records are
// not read again, and no other workgroup reads this run).  bn_stats_finalize_folded then adds <= kFoldRuns pairs per channel and segment.
constexpr int kFoldRuns = 32;
struct FoldSeg { int first, count, run; };               // records [first, first + count) in runs of `run`
__global__ __launch_bounds__(kFinThreads) void bn_records_fold(float* __restrict__ part, FoldSeg sa, FoldSeg sb, int runs_a, int c) {
    const bool second = (int)blockIdx.y >= runs_a;
    const FoldSeg sg = second ? sb : sa;
    const int f = second ? blockIdx.y - runs_a : blockIdx.y;
    const int lo = sg.first + f * sg.run, hi = min(sg.first + sg.count, lo + sg.run);
    const int chl = threadIdx.x & (kFinCh - 1), sub = threadIdx.x / kFinCh;
    const int ch = blockIdx.x * kFinCh + chl;
    double a = 0.0, b = 0.0;
    if (ch < c) {
        const float4* pr = reinterpret_cast<const float4*>(part) + ch;
#pragma unroll 4
        for (int i = lo + sub; i < hi; i += kFinSub) {
            const float4 v = pr[(int64_t)i * c];
            const double S = v.x, Q = v.y, k = v.z, n = v.w;
            a += S + n * k;
            b += Q + 2.0 * k * S + n * k * k;
        }
    }
    if (!fin_reduce(&a, &b, ch, c)) return;              // (behind its barrier: every read of the run has been made)
    reinterpret_cast<double2*>(part)[(int64_t)lo * c + ch] = make_double2(a, b);
}
__device__ __forceinline__ void bn_stats_finish_folded(const float* __restrict__ part, FoldSeg sg, int c, int64_t rows, float eps, float momentum,
                                                       float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, const float* __restrict__ extra) {
    const int chl = threadIdx.x & (kFinCh - 1), sub = threadIdx.x / kFinCh;
    const int ch = blockIdx.x * kFinCh + chl;
    const int runs = (sg.count + sg.run - 1) / sg.run;
    double a = 0.0, b = 0.0;
    if (ch < c) {
        if (sub < runs) {
            const double2 v = reinterpret_cast<const double2*>(part)[(int64_t)(sg.first + sub * sg.run) * c + ch];
            a = v.x;
            b = v.y;
        }
        if (extra && sub == kFinSub - 1) {               // the boundary record of a conv epilogue (never a folded one)
            const float4 v = reinterpret_cast<const float4*>(extra)[ch];
            const double S = v.x, Q = v.y, k = v.z, n = v.w;
            a += S + n * k;
            b += Q + 2.0 * k * S + n * k * k;
        }
    }
    if (!fin_reduce(&a, &b, ch, c)) return;
    const double m = a / (double)rows;
    double var = b / (double)rows - m * m;               // fp64 sums of values of O(1): bn_stats_finish's extra centring buys nothing here
    if (var < 0.0) var = 0.0;
    mean[ch] = (float)m;
    rstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
        running_mean[ch] = (float)((1.0 - momentum) * (double)running_mean[ch] + momentum * m);
        running_var[ch] = (float)((1.0 - momentum) * (double)running_var[ch] + momentum * unb);
    }
}
__global__ __launch_bounds__(kFinThreads) void bn_stats_finalize_folded(const float* __restrict__ part, FoldSeg sa, FoldSeg sb, int c, int64_t rows_a,
                                                                        int64_t rows_b, float eps, float momentum, float* __restrict__ mean,
                                                                        float* __restrict__ rstd, float* __restrict__ running_mean,
                                                                        float* __restrict__ running_var, int64_t* __restrict__ nbt,
                                                                        const float* __restrict__ extra_b) {
    const bool two = rows_b > 0;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += two ? 2 : 1;
    bn_stats_finish_folded(part, sa, c, rows_a, eps, momentum, mean, rstd, running_mean, running_var, nullptr);
    if (!two) return;
    __syncthreads();
    bn_stats_finish_folded(part, sb, c, rows_b, eps, momentum, mean + c, rstd + c, running_mean, running_var, extra_b);
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ rm, const float* __restrict__ rv, int c, float eps,
                                     float* __restrict__ mean, float* __restrict__ rstd) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    mean[ch] = rm[ch];
    rstd[ch] = 1.f / sqrtf(rv[ch] + eps);
}

// SyncBN (opt-in global-batch statistics, layers.BatchNorm2d._fwd_sync): this rank's (mean, rstd) as the pair (E[x], E[x^2]) that averages across
// ranks with equally many rows each -- moments[0][c] = mean, moments[1][c] = var + mean^2 with var = 1 / rstd^2 - eps
__global__ void syncbn_moments_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, int c, float eps, float* __restrict__ moments) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const float m = mean[ch], r = rstd[ch];
    moments[ch] = m;
    moments[c + ch] = 1.f / (r * r) - eps + m * m;
}

// ... and back, from the moments averaged over the ranks: mean, rstd of the GLOBAL batch of n rows, and the running statistics' momentum update
// (unbiased variance n / (n - 1), nn.BatchNorm semantics); thread 0 counts the batch
__global__ void syncbn_finish_kernel(const float* __restrict__ moments, int c, float eps, float momentum, float unbias, float* __restrict__ mean,
                                     float* __restrict__ rstd, float* __restrict__ running_mean, float* __restrict__ running_var,
                                     long* __restrict__ num_batches_tracked) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (ch >= c) return;
    const float m = moments[ch];
    const float var = fmaxf(moments[c + ch] - m * m, 0.f);
    mean[ch] = m;
    rstd[ch] = 1.f / sqrtf(var + eps);
    if (running_mean) running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * m;
    if (running_var) running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbias * var;
}

// eval-mode BatchNorm as a per-channel affine map: scale = gamma*rstd, shift = beta - mean*scale
__global__ void bn_eval_affine_kernel(const float* __restrict__ rm, const float* __restrict__ rv, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, int c, float eps, float* __restrict__ scale,
                                      float* __restrict__ shift) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const float sc = (gamma ? gamma[ch] : 1.f) / sqrtf(rv[ch] + eps);
    scale[ch] = sc;
    shift[ch] = (beta ? beta[ch] : 0.f) - rm[ch] * sc;
}

// train-mode BatchNorm as a per-channel affine map for a consumer that applies it while loading (suhpe_conv2d_fwd_stats_xf):
// scale[s][c] = gamma[c] rstd[s][c], shift[s][c] = beta[c] - mean[s][c] scale[s][c] for each of `segs` statistics sets
__global__ void bn_affine_from_stats_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, int segs, int c, float* __restrict__ scale, float* __restrict__ shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= segs * c) return;
    const int ch = i % c;
    const float sc = (gamma ? gamma[ch] : 1.f) * rstd[i];
    scale[i] = sc;
    shift[i] = (beta ? beta[ch] : 0.f) - mean[i] * sc;
}

// RepVGG structural re-parameterisation for eval (src/repvgg.py:104-142, get_equivalent_kernel_bias): the 3x3 branch, the
// 1x1 branch and the identity BatchNorm collapse into one 3x3 filter and a bias,
//   W[co][r][s][ci] = W3[co][r][s][ci] t3[co] + [r = s = 1] (W1[co][ci] t1[co] + [ci = co mod Cg] tid[co]),   t = gamma / sqrt(var + eps)
//   b[co] = sum over branches of beta - mean t.        Filters are in the physical layout [Cout][R][S][Cin/g].
struct BnRef { const float* gamma; const float* beta; const float* mean; const float* var; float eps; };
__global__ __launch_bounds__(256) void repvgg_fuse_kernel(const float* __restrict__ w3, const float* __restrict__ w1, BnRef b3, BnRef b1,
                                                          BnRef bid, int has_id, int cout, int cg, float* __restrict__ w_eq,
                                                          float* __restrict__ ones, float* __restrict__ b_eq) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)cout * 9 * cg;
    if (idx >= total) return;
    const int co = (int)(idx / (9 * cg)), rem = (int)(idx - (int64_t)co * 9 * cg), tap = rem / cg, ci = rem - tap * cg;
    const float t3 = b3.gamma[co] / sqrtf(b3.var[co] + b3.eps);
    const float t1 = b1.gamma[co] / sqrtf(b1.var[co] + b1.eps);
    const float tid = has_id ? bid.gamma[co] / sqrtf(bid.var[co] + bid.eps) : 0.f;
    float v = w3[idx] * t3;
    if (tap == 4) {
        v += w1[(int64_t)co * cg + ci] * t1;
        if (has_id && ci == co % cg) v += tid;
    }
    w_eq[idx] = v;
    if (rem == 0) {
        float b = (b3.beta[co] - b3.mean[co] * t3) + (b1.beta[co] - b1.mean[co] * t1);
        if (has_id) b += bid.beta[co] - bid.mean[co] * tid;
        b_eq[co] = b;
        ones[co] = 1.f;
    }
}

// y = act(gamma*(x-mean)*rstd + beta + residual)                       (post == 0: pre-activation residual)
// y = residual + rowscale[row / hw] * act(gamma*(x-mean)*rstd + beta)   (post == 1: the block's skip connection with
//                                                                        StochasticDepth('row') folded in; rowscale may be NULL)
template <int VEC, int ACTC = -1>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ x, int rows, int c, int cw, int rp, int rows_per_block,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int act, const float* __restrict__ residual, int post,
                                                         const float* __restrict__ rowscale, int hw,
                                                         float* __restrict__ y, const SegArgs sg = SegArgs{0, 0, 0}) {
    if (ACTC >= 0) act = ACTC;                         // compile-time activation: the switch folds, the row loop is one basic block
    int64_t rows_s = rows, row_off;
    int set;
    const int bx = seg_select(sg, &rows_s, &row_off, &set);
    rows = (int)rows_s;
    x += row_off * c; y += row_off * c;
    if (residual) residual += row_off * c;
    if (rowscale) rowscale += row_off / hw;
    mean += set * c; rstd += set * c;
#undef EW_BX
#define EW_BX bx
    EW_THREAD_SETUP(VEC)
#undef EW_BX
#define EW_BX blockIdx.x
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const float g = gamma ? gamma[ch + v] : 1.f, b = beta ? beta[ch + v] : 0.f;
        sc[v] = rstd[ch + v] * g;
        sh[v] = b - mean[ch + v] * sc[v];
    }
    // four rows per round: every load of the round is issued before the first use (one row at a time the loop waits for
    // each load right behind it and only occupancy hides the latency); rows past the slab read a clamped address
    constexpr int U = 4;
    for (int r = r0 + rsub; r < r1; r += U * rp) {
        float xv[U][VEC], rv[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * rp < r1 ? r + u * rp : r1 - 1;
            const int64_t off = ((int64_t)rr * cv + vc) * VEC;
            vload<VEC>(x + off, xv[u]);
            if (residual) vload<VEC>(residual + off, rv[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * rp;
            if (rr >= r1) break;
            const float rsc = (post && rowscale) ? rowscale[rr / hw] : 1.f;
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                float uu = xv[u][v] * sc[v] + sh[v];
                if (residual && !post) uu += rv[u][v];
                o[v] = act_apply(uu, act);
                if (post) o[v] = rv[u][v] + rsc * o[v];
            }
            vstore<VEC>(y + ((int64_t)rr * cv + vc) * VEC, o);
        }
    }
}

// y = act(BN(x)) and, from the same pass, pool[sample][channel] = mean over the sample's hw rows of y (the squeeze of the
// SqueezeExcitation block that follows the depthwise conv's BatchNorm: no second pass over y).  One row block = one sample,
// so the pooled value is a fixed-order sum inside one workgroup (no atomics).
template <int VEC, int ACTC = -1>
__global__ __launch_bounds__(256) void bn_act_pool_fwd_kernel(const float* __restrict__ x, int rows, int c, int cw, int rp, int hw,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                              float* __restrict__ y, float* __restrict__ pool, const SegArgs sg) {
    if (ACTC >= 0) act = ACTC;                         // compile-time activation: the switch folds, the row loop is one basic block
    __shared__ float pl[256 * VEC];
    int64_t rows_s = rows, row_off;
    int set;
    const int bx = seg_select(sg, &rows_s, &row_off, &set);
    x += row_off * c;
    if (y) y += row_off * c;
    mean += set * c; rstd += set * c;
    const int t = threadIdx.x, col = t % cw, rsub = t / cw;
    const int cv = c / VEC, vc = blockIdx.y * cw + col;
    const bool active = rsub < rp && vc < cv;
    const int ch = vc * VEC;
    const int r0 = bx * hw, r1 = r0 + hw;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    if (active) {
        float sc[VEC], sh[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float g = gamma ? gamma[ch + v] : 1.f, b = beta ? beta[ch + v] : 0.f;
            sc[v] = rstd[ch + v] * g;
            sh[v] = b - mean[ch + v] * sc[v];
        }
        // one workgroup per (sample, 32 vector columns): a 7x7 sample is ONE round of 8 rows per thread, all loads issued before the
        // first use -- with 64 columns and 4 rows per round the grid was 2.5-4 workgroups per CU with 4 KB in flight each (2.3 TB/s)
        constexpr int U = 8;
        for (int r = r0 + rsub; r < r1; r += U * rp) {
            float xv[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r + u * rp < r1 ? r + u * rp : r1 - 1;
                vload<VEC>(x + ((int64_t)rr * cv + vc) * VEC, xv[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r + u * rp;
                if (rr >= r1) break;
                float o[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) { o[v] = act_apply(xv[u][v] * sc[v] + sh[v], act); acc[v] += o[v]; }
                if (y) vstore<VEC>(y + ((int64_t)rr * cv + vc) * VEC, o);           // y == NULL: the squeeze alone (no tensor written)
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) pl[t * VEC + v] = acc[v];
    __syncthreads();
    if (active && rsub == 0) {
        const float inv = 1.f / (float)hw;
        const int64_t sample = row_off / hw + bx;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float a = 0.f;
            for (int r = 0; r < rp; ++r) a += pl[(r * cw + col) * VEC + v];
            pool[sample * c + ch + v] = a * inv;
        }
    }
}

// backward pass 1: partial sums of g and g*xhat, g = dy * act'(u)
template <int VEC, int ACTC = -1>
__global__ __launch_bounds__(256) void bn_bwd_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ yout, int64_t rows, int c, int cw, int rp,
                                                      int64_t rows_per_block, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int act, const float* __restrict__ rowscale,
                                                      int hw, float* __restrict__ part, const SegArgs sg = SegArgs{0, 0, 0}) {
    if (ACTC >= 0) act = ACTC;                         // compile-time activation: the switch folds, the row loop is one basic block
    __shared__ float lds[256 * 2 * VEC];
    int64_t row_off;
    int set;
    const int bx = seg_select(sg, &rows, &row_off, &set);
    x += row_off * c; dy += row_off * c;
    if (yout) yout += row_off * c;
    if (rowscale) rowscale += row_off / hw;
    mean += set * c; rstd += set * c;
    const int t = threadIdx.x, col = t % cw, rsub = t / cw;
    const int col0 = blockIdx.y * cw;
    const int cv = c / VEC;
    float s0[VEC], s1[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { s0[v] = 0.f; s1[v] = 0.f; }
    if (rsub < rp && col0 + col < cv) {
        const int ch = (col0 + col) * VEC;
        float mu[VEC], rs[VEC], ga[VEC], be[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            mu[v] = mean[ch + v]; rs[v] = rstd[ch + v];
            ga[v] = gamma ? gamma[ch + v] : 1.f; be[v] = beta ? beta[ch + v] : 0.f;
        }
        const int64_t r0 = (int64_t)bx * rows_per_block;
        int64_t r1 = r0 + rows_per_block;
        if (r1 > rows) r1 = rows;
        // eight rows per round: 16-24 independent 16-byte loads in flight per thread, clamped + masked instead of branched
        for (int64_t r = r0 + rsub; r < r1; r += 8 * (int64_t)rp) {
            float xv[8][VEC], dv[8][VEC], yv[8][VEC];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t rr = r + u * (int64_t)rp;
                const int64_t off = ((rr < r1 ? rr : r1 - 1) * cv + col0 + col) * VEC;
                vload<VEC>(x + off, xv[u]);
                vload<VEC>(dy + off, dv[u]);
                if (yout) vload<VEC>(yout + off, yv[u]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t rr = r + u * (int64_t)rp;
                const bool ok = rr < r1;
                const float rsc = (rowscale && ok) ? rowscale[rr / hw] : 1.f;     // StochasticDepth scale of the sample
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float xh = (xv[u][v] - mu[v]) * rs[v];
                    float g;
                    if (yout) g = yv[u][v] > 0.f ? dv[u][v] : 0.f;       // ReLU applied after a pre-activation residual
                    else g = dv[u][v] * rsc * act_grad(xh * ga[v] + be[v], act);
                    g = ok ? g : 0.f;
                    s0[v] += g;
                    s1[v] += g * xh;
                }
            }
        }
    }
    block_combine<VEC>(s0, s1, cw, rp, col0, c, blockIdx.x, part, lds);
}

// pass 2: dbeta = sum g, dgamma = sum g*xhat (one wavefront per channel); leaves them in sums[c][2] for pass 3
__global__ __launch_bounds__(kFinThreads) void bn_bwd_finalize(const float* __restrict__ part, int gx, int c, float* __restrict__ sums,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    double s, q;
    if (!block_channel_sums(part, gx, c, &s, &q)) return;
    const int ch = blockIdx.x * kFinCh + (threadIdx.x & (kFinCh - 1));
    sums[2 * ch] = (float)s;
    sums[2 * ch + 1] = (float)q;
    if (dbeta) dbeta[ch] = accumulate ? dbeta[ch] + (float)s : (float)s;
    if (dgamma) dgamma[ch] = accumulate ? dgamma[ch] + (float)q : (float)q;
}

// two segments: records [0, gx_a) and [gx_a, gx); sums[0..2c) / [2c..4c) for pass 3, one accumulation into dgamma / dbeta
__global__ __launch_bounds__(kFinThreads) void bn_bwd_finalize2(const float* __restrict__ part, int gx_a, int gx, int c, float* __restrict__ sums,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    double s0, q0, s1, q1;
    const bool own0 = block_channel_sums(part, gx_a, c, &s0, &q0);
    __syncthreads();                                     // the reduction scratch is reused
    const bool own1 = block_channel_sums(part + (int64_t)gx_a * c * 2, gx - gx_a, c, &s1, &q1);
    if (!(own0 && own1)) return;
    const int ch = blockIdx.x * kFinCh + (threadIdx.x & (kFinCh - 1));
    sums[2 * ch] = (float)s0;
    sums[2 * ch + 1] = (float)q0;
    sums[2 * c + 2 * ch] = (float)s1;
    sums[2 * c + 2 * ch + 1] = (float)q1;
    if (dbeta) dbeta[ch] = (accumulate ? dbeta[ch] : 0.f) + (float)s0 + (float)s1;
    if (dgamma) dgamma[ch] = (accumulate ? dgamma[ch] : 0.f) + (float)q0 + (float)q1;
}

// pass 3: dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat))   [train]   or gamma*rstd*g [eval]
template <int VEC, int ACTC = -1>
__global__ __launch_bounds__(256) void bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ dy,
                                                    const float* __restrict__ yout, int rows, int c, int cw, int rp, int rows_per_block,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                    const float* __restrict__ sums, float inv_rows, int train_stats,
                                                    const float* __restrict__ rowscale, int hw, float* __restrict__ dx,
                                                    const SegArgs sg = SegArgs{0, 0, 0}, float inv_rows_b = 0.f) {
    if (ACTC >= 0) act = ACTC;                         // compile-time activation: the switch folds, the row loop is one basic block
    int64_t rows_s = rows, row_off;
    int set;
    const int bx = seg_select(sg, &rows_s, &row_off, &set);
    rows = (int)rows_s;
    x += row_off * c; dy += row_off * c; dx += row_off * c;
    if (yout) yout += row_off * c;
    if (rowscale) rowscale += row_off / hw;
    mean += set * c; rstd += set * c; sums += set * 2 * c;
    if (set) inv_rows = inv_rows_b;
#undef EW_BX
#define EW_BX bx
    EW_THREAD_SETUP(VEC)
#undef EW_BX
#define EW_BX blockIdx.x
    float mu[VEC], rs[VEC], ga[VEC], be[VEC], m0[VEC], m1[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        mu[v] = mean[ch + v]; rs[v] = rstd[ch + v];
        ga[v] = gamma ? gamma[ch + v] : 1.f; be[v] = beta ? beta[ch + v] : 0.f;
        m0[v] = train_stats ? sums[2 * (ch + v)] * inv_rows : 0.f;
        m1[v] = train_stats ? sums[2 * (ch + v) + 1] * inv_rows : 0.f;
    }
    constexpr int U = 4;                                 // four rows of loads in flight per round (see bn_act_fwd_kernel)
    for (int r = r0 + rsub; r < r1; r += U * rp) {
        float xv[U][VEC], dv[U][VEC], yv[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * rp < r1 ? r + u * rp : r1 - 1;
            const int64_t off = ((int64_t)rr * cv + vc) * VEC;
            vload<VEC>(x + off, xv[u]);
            vload<VEC>(dy + off, dv[u]);
            if (yout) vload<VEC>(yout + off, yv[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * rp;
            if (rr >= r1) break;
            const float rsc = rowscale ? rowscale[rr / hw] : 1.f;
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float xh = (xv[u][v] - mu[v]) * rs[v];
                float g;
                if (yout) g = yv[u][v] > 0.f ? dv[u][v] : 0.f;
                else g = dv[u][v] * rsc * act_grad(xh * ga[v] + be[v], act);
                g = g - m0[v] - xh * m1[v];
                o[v] = ga[v] * rs[v] * g;
            }
            vstore<VEC>(dx + ((int64_t)rr * cv + vc) * VEC, o);
        }
    }
}

// ------------------------------------------------------------------------------------------ RepVGG block tail
// y = relu(BN_a(x_a) + BN_b(x_b) [+ BN_c(x_c)])  (src/repvgg.py:67-76: the 3x3 branch, the 1x1 branch and the identity
// BatchNorm, train-mode statistics per branch) as ONE pass: 3-4 streams instead of the 8 of three chained apply kernels.
// The backward shares g = dout * (y > 0) between the branches: one partial pass (sum g, sum g xhat_j), one finaliser, one
// apply pass writing the three input gradients.  Segments as in the other BatchNorm kernels (statistics [2][c]).
struct Bn3 {
    const float* x[3];
    const float* mean[3];
    const float* rstd[3];
    const float* gamma[3];
    const float* beta[3];
    int nb;                                              // 2 or 3 branches
};

// The block's pre-activation a = sum_j BN_j(x_j) as one affine form, written with explicit fmaf so that the forward kernel and the two backward
// kernels (which recompute the ReLU mask a > 0 from the branch inputs instead of reading the output tensor: one stream of five / eight less)
// round identically whatever the compiler contracts elsewhere:
//   sc_j = rstd_j gamma_j,  sh = sum_j (beta_j - mean_j sc_j)  (j ascending),  a = fma(x_2, sc_2, fma(x_1, sc_1, fma(x_0, sc_0, sh)))
template <int VEC>
__device__ __forceinline__ void bn3_affine(const Bn3& b, int set, int c, int ch, float (&sc)[3][VEC], float (&sh)[VEC]) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) sh[v] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            if (j < b.nb) {
                sc[j][v] = b.rstd[j][set * c + ch + v] * b.gamma[j][ch + v];
                sh[v] = sh[v] + fmaf(-b.mean[j][set * c + ch + v], sc[j][v], b.beta[j][ch + v]);
            } else {
                sc[j][v] = 0.f;
            }
        }
    }
}
__device__ __forceinline__ float bn3_pre(int nb, float x0, float x1, float x2, float sc0, float sc1, float sc2, float sh) {
    float a = fmaf(x1, sc1, fmaf(x0, sc0, sh));
    if (nb == 3) a = fmaf(x2, sc2, a);
    return a;
}

// STATS (round 5): the kernel also emits the (S, Q, k, n) statistics records of its OUTPUT per (row block, channel) -- bn_stats_partial's format, the
// shift k = the output at the block's first row, computed by every thread of a channel column from the same three loads -- so that the NEXT RepVGG
// block's identity BatchNorm (whose input is this output) needs no statistics pass of its own over the tensor (bn_stats_partial: 33 us x 50 per C3 step).
template <int VEC, bool STATS = false>
__global__ __launch_bounds__(256) void bn3_act_fwd_kernel(Bn3 b, int rows, int c, int cw, int rp, int rows_per_block, float* __restrict__ y,
                                                          const SegArgs sg, float* __restrict__ part = nullptr) {
    __shared__ float lds[STATS ? 256 * 2 * VEC : 1];
    int64_t rows_s = rows, row_off;
    int set;
    const int bx = seg_select(sg, &rows_s, &row_off, &set);
    rows = (int)rows_s;
    y += row_off * c;
    const int t = threadIdx.x, col = t % cw, rsub = t / cw;
    const int cv = c / VEC, vc = blockIdx.y * cw + col;
    const bool active = rsub < rp && vc < cv;
    if (!STATS && !active) return;
    const int ch = active ? vc * VEC : 0;
    const int r0 = bx * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float sc[3][VEC], sh[VEC];
    bn3_affine<VEC>(b, set, c, ch, sc, sh);
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (j < b.nb) b.x[j] += row_off * c;
    float s0[VEC], s1[VEC], k[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { s0[v] = 0.f; s1[v] = 0.f; k[v] = 0.f; }
    if (STATS && active && r0 < r1) {                        // the shift: this channel's output at the block's first row
        float xk[3][VEC];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < b.nb) vload<VEC>(b.x[j] + ((int64_t)r0 * cv + vc) * VEC, xk[j]);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float a = bn3_pre(b.nb, xk[0][v], xk[1][v], b.nb == 3 ? xk[2][v] : 0.f, sc[0][v], sc[1][v], sc[2][v], sh[v]);
            k[v] = a > 0.f ? a : 0.f;
        }
    }
    constexpr int U = 4;
    for (int r = r0 + rsub; active && r < r1; r += U * rp) {
        float xv[3][U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * rp < r1 ? r + u * rp : r1 - 1;
            const int64_t off = ((int64_t)rr * cv + vc) * VEC;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < b.nb) vload<VEC>(b.x[j] + off, xv[j][u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * rp;
            if (rr >= r1) break;
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float a = bn3_pre(b.nb, xv[0][u][v], xv[1][u][v], b.nb == 3 ? xv[2][u][v] : 0.f, sc[0][v], sc[1][v], sc[2][v], sh[v]);
                o[v] = a > 0.f ? a : 0.f;
                if (STATS) { const float d = o[v] - k[v]; s0[v] += d; s1[v] += d * d; }
            }
            vstore<VEC>(y + ((int64_t)rr * cv + vc) * VEC, o);
        }
    }
    if (STATS) block_combine<VEC>(s0, s1, cw, rp, blockIdx.y * cw, c, blockIdx.x, part, lds, k, r1 > r0 ? (float)(r1 - r0) : 0.f);
}

// records per (row block, channel): (sum g, sum g xhat_a, sum g xhat_b, sum g xhat_c)
// REMASK: the ReLU mask is recomputed from the branch inputs (bn3_pre > 0, the forward's own arithmetic) -- yout is not read
template <int VEC, bool REMASK = false>
__global__ __launch_bounds__(256) void bn3_bwd_partial(Bn3 b, const float* __restrict__ dout, const float* __restrict__ yout, int64_t rows, int c,
                                                       int cw, int rp, int64_t rows_per_block, float* __restrict__ part, const SegArgs sg) {
    __shared__ float lds[256 * 4 * VEC];
    int64_t row_off;
    int set;
    const int bx = seg_select(sg, &rows, &row_off, &set);
    dout += row_off * c; yout += row_off * c;
    const int t = threadIdx.x, col = t % cw, rsub = t / cw;
    const int col0 = blockIdx.y * cw;
    const int cv = c / VEC;
    float s[4][VEC];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) s[k][v] = 0.f;
    if (rsub < rp && col0 + col < cv) {
        const int ch = (col0 + col) * VEC;
        float mu[3][VEC], rs[3][VEC], sc[3][VEC], sh[VEC];
        if (REMASK) bn3_affine<VEC>(b, set, c, ch, sc, sh);
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < b.nb) {
                b.x[j] += row_off * c;
#pragma unroll
                for (int v = 0; v < VEC; ++v) { mu[j][v] = b.mean[j][set * c + ch + v]; rs[j][v] = b.rstd[j][set * c + ch + v]; }
            }
        const int64_t r0 = (int64_t)bx * rows_per_block;
        int64_t r1 = r0 + rows_per_block;
        if (r1 > rows) r1 = rows;
        constexpr int U = 4;
        for (int64_t r = r0 + rsub; r < r1; r += U * (int64_t)rp) {
            float xv[3][U][VEC], dv[U][VEC], yv[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t rr = r + u * (int64_t)rp;
                const int64_t off = ((rr < r1 ? rr : r1 - 1) * cv + col0 + col) * VEC;
                vload<VEC>(dout + off, dv[u]);
                if (!REMASK) vload<VEC>(yout + off, yv[u]);
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (j < b.nb) vload<VEC>(b.x[j] + off, xv[j][u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = r + u * (int64_t)rp < r1;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    if (REMASK) yv[u][v] = bn3_pre(b.nb, xv[0][u][v], xv[1][u][v], b.nb == 3 ? xv[2][u][v] : 0.f, sc[0][v], sc[1][v], sc[2][v], sh[v]);
                    const float g = (ok && yv[u][v] > 0.f) ? dv[u][v] : 0.f;
                    s[0][v] += g;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (j < b.nb) s[j + 1][v] += g * (xv[j][u][v] - mu[j][v]) * rs[j][v];
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) lds[(t * 4 + k) * VEC + v] = s[k][v];
    __syncthreads();
    if (t < cw && (col0 + t) * VEC < c) {
        float a[4][VEC];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v) a[k][v] = 0.f;
        for (int r = 0; r < rp; ++r) {
            const int tt = r * cw + t;
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int v = 0; v < VEC; ++v) a[k][v] += lds[(tt * 4 + k) * VEC + v];
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            reinterpret_cast<float4*>(part)[(int64_t)blockIdx.x * c + (col0 + t) * VEC + v] = make_float4(a[0][v], a[1][v], a[2][v], a[3][v]);
    }
}

// sums[seg][k][c] (k = g, g xhat_a, g xhat_b, g xhat_c); dgamma_j += sum over segments of sums[.][j+1], dbeta_j += sums[.][0]
struct Bn3Grads { float* dgamma[3]; float* dbeta[3]; int nb; };
__global__ __launch_bounds__(kFinThreads) void bn3_bwd_finalize(const float* __restrict__ part, int gx_a, int gx, int c, float* __restrict__ sums,
                                                        const Bn3Grads gr) {
    const int chl = threadIdx.x & (kFinCh - 1), sub = threadIdx.x / kFinCh;
    const int ch = blockIdx.x * kFinCh + chl;
    double tot[2][4];
    bool own = false;
    for (int seg = 0; seg < 2; ++seg) {
        const int first = seg == 0 ? 0 : gx_a, count = seg == 0 ? gx_a : gx - gx_a;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        if (ch < c) {
            const float4* p = reinterpret_cast<const float4*>(part) + (int64_t)first * c + ch;
#pragma unroll 4
            for (int i = sub; i < count; i += kFinSub) {
                const float4 v = p[(int64_t)i * c];
                a[0] += (double)v.x; a[1] += (double)v.y; a[2] += (double)v.z; a[3] += (double)v.w;
            }
        }
        __syncthreads();
        own = fin_reduce(&a[0], &a[1], ch, c);
        __syncthreads();
        fin_reduce(&a[2], &a[3], ch, c);
#pragma unroll
        for (int k = 0; k < 4; ++k) tot[seg][k] = a[k];
    }
    if (!own) return;
#pragma unroll
    for (int seg = 0; seg < 2; ++seg)
#pragma unroll
        for (int k = 0; k < 4; ++k) sums[((int64_t)seg * 4 + k) * c + ch] = (float)tot[seg][k];
    for (int j = 0; j < gr.nb; ++j) {
        gr.dbeta[j][ch] += (float)tot[0][0] + (float)tot[1][0];
        gr.dgamma[j][ch] += (float)tot[0][j + 1] + (float)tot[1][j + 1];
    }
}

struct Bn3Out { float* dx[3]; };
template <int VEC, bool REMASK = false>
__global__ __launch_bounds__(256) void bn3_bwd_apply(Bn3 b, const float* __restrict__ dout, const float* __restrict__ yout, int rows, int c, int cw,
                                                     int rp, int rows_per_block, const float* __restrict__ sums, float inv_rows_a, float inv_rows_b,
                                                     Bn3Out o, const SegArgs sg) {
    int64_t rows_s = rows, row_off;
    int set;
    const int bx = seg_select(sg, &rows_s, &row_off, &set);
    rows = (int)rows_s;
    dout += row_off * c; yout += row_off * c;
    const float inv_rows = set ? inv_rows_b : inv_rows_a;
#undef EW_BX
#define EW_BX bx
    EW_THREAD_SETUP(VEC)
#undef EW_BX
#define EW_BX blockIdx.x
    float mu[3][VEC], rs[3][VEC], ga[3][VEC], m1[3][VEC], m0[VEC], sc[3][VEC], sh[VEC];
    if (REMASK) bn3_affine<VEC>(b, set, c, ch, sc, sh);
#pragma unroll
    for (int v = 0; v < VEC; ++v) m0[v] = sums[((int64_t)set * 4 + 0) * c + ch + v] * inv_rows;
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (j < b.nb) {
            b.x[j] += row_off * c;
            o.dx[j] += row_off * c;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                mu[j][v] = b.mean[j][set * c + ch + v];
                rs[j][v] = b.rstd[j][set * c + ch + v];
                ga[j][v] = b.gamma[j][ch + v] * rs[j][v];
                m1[j][v] = sums[((int64_t)set * 4 + j + 1) * c + ch + v] * inv_rows;
            }
        }
    constexpr int U = 4;
    for (int r = r0 + rsub; r < r1; r += U * rp) {
        float xv[3][U][VEC], dv[U][VEC], yv[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * rp < r1 ? r + u * rp : r1 - 1;
            const int64_t off = ((int64_t)rr * cv + vc) * VEC;
            vload<VEC>(dout + off, dv[u]);
            if (!REMASK) vload<VEC>(yout + off, yv[u]);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < b.nb) vload<VEC>(b.x[j] + off, xv[j][u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * rp;
            if (rr >= r1) break;
            if (REMASK) {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    yv[u][v] = bn3_pre(b.nb, xv[0][u][v], xv[1][u][v], b.nb == 3 ? xv[2][u][v] : 0.f, sc[0][v], sc[1][v], sc[2][v], sh[v]);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < b.nb) {
                    float d[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const float g = yv[u][v] > 0.f ? dv[u][v] : 0.f;
                        const float xh = (xv[j][u][v] - mu[j][v]) * rs[j][v];
                        d[v] = ga[j][v] * (g - m0[v] - xh * m1[j][v]);
                    }
                    vstore<VEC>(o.dx[j] + ((int64_t)rr * cv + vc) * VEC, d);
                }
        }
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm (one wave per row)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int64_t rows, int c, float eps,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ y, float* __restrict__ mean,
                                                            float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * c;
    float s = 0.f;
    for (int j = lane; j < c; j += 64) s += xr[j];
    const float m = wave_sum(s) / c;
    float q = 0.f;
    for (int j = lane; j < c; j += 64) { const float d = xr[j] - m; q += d * d; }
    const float rs = 1.f / sqrtf(wave_sum(q) / c + eps);
    for (int j = lane; j < c; j += 64) y[row * c + j] = (xr[j] - m) * rs * (gamma ? gamma[j] : 1.f) + (beta ? beta[j] : 0.f);
    if (lane == 0) { mean[row] = m; rstd[row] = rs; }
}

// 16-byte form (c % 4 == 0, c <= 256 NV): the row lives in registers between the statistics and the normalisation -- one pass over
// memory, NV independent float4 loads per lane issued together (the scalar kernel above reads the row three times from L1 / L2)
template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_v4(const float* __restrict__ x, int64_t rows, int c, float eps, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean,
                                                        float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63, cv = c / 4;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* xr = reinterpret_cast<const float4*>(x + row * c);
    float4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = lane + 64 * k < cv ? xr[lane + 64 * k] : make_float4(0.f, 0.f, 0.f, 0.f);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    const float m = wave_sum(s) / c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (lane + 64 * k < cv) {
            const float a = v[k].x - m, b = v[k].y - m, cc = v[k].z - m, d = v[k].w - m;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    const float rs = 1.f / sqrtf(wave_sum(q) / c + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = lane + 64 * k;
        if (j >= cv) continue;
        const float4 ga = gamma ? reinterpret_cast<const float4*>(gamma)[j] : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 be = beta ? reinterpret_cast<const float4*>(beta)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(y + row * c)[j] = make_float4((v[k].x - m) * rs * ga.x + be.x, (v[k].y - m) * rs * ga.y + be.y,
                                                                (v[k].z - m) * rs * ga.z + be.z, (v[k].w - m) * rs * ga.w + be.w);
    }
    if (lane == 0) { mean[row] = m; rstd[row] = rs; }
}

// dx = rstd (g - mean(g) - xhat mean(g xhat)) [+ skip]: the gradient that bypassed the sub-block through its residual connection
// (TinyViTBlock: x + drop_path(attn(x)), x + drop_path(mlp(x)), src/tiny_vit.py:366,378) is added here instead of by a pass of its own
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_dx_v4(const float* __restrict__ x, const float* __restrict__ dy, int64_t rows, int c,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ skip, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63, cv = c / 4;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float m = mean[row], rs = rstd[row];
    float g[NV][4], xh[NV][4], sk[NV][4];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = lane + 64 * k;
        const bool ok = j < cv;
        const float4 xv = ok ? reinterpret_cast<const float4*>(x + row * c)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 dv = ok ? reinterpret_cast<const float4*>(dy + row * c)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 ga = (ok && gamma) ? reinterpret_cast<const float4*>(gamma)[j] : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 sv = (ok && skip) ? reinterpret_cast<const float4*>(skip + row * c)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        g[k][0] = dv.x * ga.x; g[k][1] = dv.y * ga.y; g[k][2] = dv.z * ga.z; g[k][3] = dv.w * ga.w;
        xh[k][0] = (xv.x - m) * rs; xh[k][1] = (xv.y - m) * rs; xh[k][2] = (xv.z - m) * rs; xh[k][3] = (xv.w - m) * rs;
        sk[k][0] = sv.x; sk[k][1] = sv.y; sk[k][2] = sv.z; sk[k][3] = sv.w;
        if (ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { s0 += g[k][e]; s1 += g[k][e] * xh[k][e]; }
        }
    }
    s0 = wave_sum(s0) / c;
    s1 = wave_sum(s1) / c;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = lane + 64 * k;
        if (j >= cv) continue;
        reinterpret_cast<float4*>(dx + row * c)[j] = make_float4(rs * (g[k][0] - s0 - xh[k][0] * s1) + sk[k][0], rs * (g[k][1] - s0 - xh[k][1] * s1) + sk[k][1],
                                                                 rs * (g[k][2] - s0 - xh[k][2] * s1) + sk[k][2], rs * (g[k][3] - s0 - xh[k][3] * s1) + sk[k][3]);
    }
}

// dx per row; dgamma/dbeta partials via the column-reduction machinery (VEC=1 form, g = dy, xhat from saved stats)
__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               int64_t rows, int c, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ skip, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float m = mean[row], rs = rstd[row];
    float s0 = 0.f, s1 = 0.f;
    for (int j = lane; j < c; j += 64) {
        const float g = dy[row * c + j] * (gamma ? gamma[j] : 1.f);
        const float xh = (x[row * c + j] - m) * rs;
        s0 += g;
        s1 += g * xh;
    }
    s0 = wave_sum(s0) / c;
    s1 = wave_sum(s1) / c;
    for (int j = lane; j < c; j += 64) {
        const float g = dy[row * c + j] * (gamma ? gamma[j] : 1.f);
        const float xh = (x[row * c + j] - m) * rs;
        dx[row * c + j] = rs * (g - s0 - xh * s1) + (skip ? skip[row * c + j] : 0.f);
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void layernorm_bwd_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                                             int64_t rows, int c, int cw, int rp, int64_t rows_per_block,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float* __restrict__ part) {
    __shared__ float lds[256 * 2 * VEC];
    const int t = threadIdx.x, col = t % cw, rsub = t / cw;
    const int col0 = blockIdx.y * cw;
    const int cv = c / VEC;
    float s0[VEC], s1[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { s0[v] = 0.f; s1[v] = 0.f; }
    if (rsub < rp && col0 + col < cv) {
        const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
        int64_t r1 = r0 + rows_per_block;
        if (r1 > rows) r1 = rows;
        for (int64_t r = r0 + rsub; r < r1; r += rp) {
            float xv[VEC], dv[VEC];
            const int64_t off = (r * cv + col0 + col) * VEC;
            vload<VEC>(x + off, xv);
            vload<VEC>(dy + off, dv);
            const float m = mean[r], rs = rstd[r];
#pragma unroll
            for (int v = 0; v < VEC; ++v) { s0[v] += dv[v]; s1[v] += dv[v] * (xv[v] - m) * rs; }
        }
    }
    block_combine<VEC>(s0, s1, cw, rp, col0, c, blockIdx.x, part, lds);
}

// ------------------------------------------------------------------------------------------ plain activation (+bias)
template <int VEC>
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, int rows, int c, int cw, int rp, int rows_per_block,
                                                      const float* __restrict__ bias, int act, float* __restrict__ y) {
    EW_THREAD_SETUP(VEC)
    float bv[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) bv[v] = bias ? bias[ch + v] : 0.f;
    constexpr int U = 4;
    for (int r = r0 + rsub; r < r1; r += U * rp) {
        float xv[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) vload<VEC>(x + ((int64_t)(r + u * rp < r1 ? r + u * rp : r1 - 1) * cv + vc) * VEC, xv[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u * rp >= r1) break;
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) o[v] = act_apply(xv[u][v] + bv[v], act);
            vstore<VEC>(y + ((int64_t)(r + u * rp) * cv + vc) * VEC, o);
        }
    }
}
template <int VEC>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int rows, int c, int cw,
                                                      int rp, int rows_per_block, const float* __restrict__ bias, int act,
                                                      float* __restrict__ dx) {
    EW_THREAD_SETUP(VEC)
    float bv[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) bv[v] = bias ? bias[ch + v] : 0.f;
    constexpr int U = 4;
    for (int r = r0 + rsub; r < r1; r += U * rp) {
        float xv[U][VEC], dv[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t off = ((int64_t)(r + u * rp < r1 ? r + u * rp : r1 - 1) * cv + vc) * VEC;
            vload<VEC>(x + off, xv[u]);
            vload<VEC>(dy + off, dv[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u * rp >= r1) break;
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) o[v] = dv[u][v] * act_grad(xv[u][v] + bv[v], act);
            vstore<VEC>(dx + ((int64_t)(r + u * rp) * cv + vc) * VEC, o);
        }
    }
}

// ------------------------------------------------------------------------------------------ max pooling (ResNet stem)
// nn.MaxPool2d(k, stride, pad) on (N, H, W, C): the window is scanned kh, kw ascending and a value replaces the running
// maximum only when it is greater (or NaN), so ties go to the FIRST position, as torch's kernels do; the position inside the
// window (0 .. k*k-1) is kept for the backward.  One thread per (output pixel, channel vector).
template <int VEC>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, int n, int h, int w, int c, int k, int stride, int pad,
                                                          int oh, int ow, float* __restrict__ y, uint8_t* __restrict__ idx) {
    const int cv = c / VEC;
    const int64_t total = (int64_t)n * oh * ow * cv;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int vc = (int)(e % cv);
        const int64_t pix = e / cv;
        const int x0 = (int)(pix % ow), y0 = (int)((pix / ow) % oh), b = (int)(pix / ((int64_t)ow * oh));
        float best[VEC];
        int arg[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { best[v] = -INFINITY; arg[v] = 0; }
        bool first = true;
        for (int kh = 0; kh < k; ++kh) {
            const int ih = y0 * stride - pad + kh;
            if (ih < 0 || ih >= h) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int iw = x0 * stride - pad + kw;
                if (iw < 0 || iw >= w) continue;
                float xv[VEC];
                vload<VEC>(x + (((int64_t)b * h + ih) * w + iw) * c + vc * VEC, xv);
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (first || xv[v] > best[v] || xv[v] != xv[v]) { best[v] = xv[v]; arg[v] = kh * k + kw; }
                first = false;
            }
        }
        vstore<VEC>(y + pix * c + vc * VEC, best);
#pragma unroll
        for (int v = 0; v < VEC; ++v) idx[pix * c + vc * VEC + v] = (uint8_t)arg[v];
    }
}

// dx[b, ih, iw, c] = sum of dy over the output pixels whose window holds (ih, iw) at their recorded position (gather form:
// deterministic, no atomics)
template <int VEC>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, int n, int h, int w, int c,
                                                          int k, int stride, int pad, int oh, int ow, float* __restrict__ dx) {
    const int cv = c / VEC;
    const int64_t total = (int64_t)n * h * w * cv;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int vc = (int)(e % cv);
        const int64_t pix = e / cv;
        const int iw = (int)(pix % w), ih = (int)((pix / w) % h), b = (int)(pix / ((int64_t)w * h));
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        // window positions descending = output pixels ascending: the terms are added in the order torch's scatter visits them
        for (int kh = k - 1; kh >= 0; --kh) {
            const int num = ih + pad - kh;
            if (num < 0 || num % stride) continue;
            const int y0 = num / stride;
            if (y0 >= oh) continue;
            for (int kw = k - 1; kw >= 0; --kw) {
                const int nw = iw + pad - kw;
                if (nw < 0 || nw % stride) continue;
                const int x0 = nw / stride;
                if (x0 >= ow) continue;
                const int64_t o = (((int64_t)b * oh + y0) * ow + x0) * c + vc * VEC;
                float dv[VEC];
                vload<VEC>(dy + o, dv);
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (idx[o + v] == kh * k + kw) acc[v] += dv[v];
            }
        }
        vstore<VEC>(dx + pix * c + vc * VEC, acc);
    }
}

// ------------------------------------------------------------------------------------------ pooling / SE
// One block per (sample, column chunk): y[n,c] = scale * sum_p x[n,p,c] * (w ? w[n,p,c] : 1)
template <int VEC>
__global__ __launch_bounds__(256) void pool_sum_kernel(const float* __restrict__ x, const float* __restrict__ w, int hw, int c,
                                                       int cw, int rp, float scale, float* __restrict__ y) {
    __shared__ float lds[256 * VEC];
    const int t = threadIdx.x, col = t % cw, rsub = t / cw;
    const int col0 = blockIdx.y * cw, cv = c / VEC;
    const int64_t n = blockIdx.x;
    float s[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[v] = 0.f;
    if (rsub < rp && col0 + col < cv) {
        constexpr int U = 4;                             // four pixels of loads in flight per round
        for (int p = rsub; p < hw; p += U * rp) {
            float xv[U][VEC], wv[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pp = p + u * rp < hw ? p + u * rp : hw - 1;
                const int64_t off = ((n * hw + pp) * cv + col0 + col) * VEC;
                vload<VEC>(x + off, xv[u]);
                if (w) vload<VEC>(w + off, wv[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float m = p + u * rp < hw ? 1.f : 0.f;
#pragma unroll
                for (int v = 0; v < VEC; ++v) s[v] += m * (w ? xv[u][v] * wv[u][v] : xv[u][v]);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) lds[t * VEC + v] = s[v];
    __syncthreads();
    if (t < cw && col0 + t < cv) {
        float a[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) a[v] = 0.f;
        for (int r = 0; r < rp; ++r)
#pragma unroll
            for (int v = 0; v < VEC; ++v) a[v] += lds[(r * cw + t) * VEC + v];
#pragma unroll
        for (int v = 0; v < VEC; ++v) a[v] *= scale;
        vstore<VEC>(y + (n * cv + col0 + t) * VEC, a);
    }
}

// mode 0: y[n,p,c] = x[n,p,c]*s[n,c] + dpool[n,c]*inv_hw   (SE scale fwd / bwd-x)
// mode 1: y[n,p,c] = s[n,c]*inv_hw                          (average-pool backward; x unused)
// grid (row blocks within a sample, column chunks, n)
template <int VEC>
__global__ __launch_bounds__(256) void per_sample_scale_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                               const float* __restrict__ dpool, int hw, int c, int cw, int rp,
                                                               int rows_per_block, float inv_hw, int mode, float* __restrict__ y) {
    const int rows = hw;
    EW_THREAD_SETUP(VEC)
    const int64_t n = blockIdx.z;
    float sv[VEC], pv[VEC];
    vload<VEC>(s + (n * cv + vc) * VEC, sv);
#pragma unroll
    for (int v = 0; v < VEC; ++v) pv[v] = 0.f;
    if (dpool) vload<VEC>(dpool + (n * cv + vc) * VEC, pv);
#pragma unroll
    for (int v = 0; v < VEC; ++v) { pv[v] *= inv_hw; if (mode == 1) sv[v] *= inv_hw; }
    constexpr int U = 4;
    for (int r = r0 + rsub; r < r1; r += U * rp) {
        float xv[U][VEC];
        if (mode == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) vload<VEC>(x + ((n * hw + (r + u * rp < r1 ? r + u * rp : r1 - 1)) * cv + vc) * VEC, xv[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u * rp >= r1) break;
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) o[v] = mode == 0 ? xv[u][v] * sv[v] + pv[v] : sv[v];
            vstore<VEC>(y + ((n * hw + r + u * rp) * cv + vc) * VEC, o);
        }
    }
}

// ---- MBConv middle section fused around the depthwise conv's BatchNorm (torchvision MBConv: dw conv -> BatchNorm -> SiLU ->
// SqueezeExcitation -> project conv).  With y = the depthwise conv's raw output, z = act(BN(y)), s = the excitation (n, c):
//   forward   pool[n,c] = mean_hw z            (bn_act_pool_fwd_kernel with y == NULL: one read, nothing written)
//             out = z * s[n,c]                  (bn_act_scale_fwd_kernel: z itself is never stored)
//   backward  given dout = d loss / d out:   ds[n,c] = sum_hw dout z,   dz = dout s + dpool[n,c]/hw,   g = dz act'(u),
//             and the BatchNorm backward needs sum g and sum g xhat over a whole segment.  Both are linear in (s, dpool):
//               sum_hw g      = s A2 + dpool/hw A4,      sum_hw g xhat = s A3 + dpool/hw A5
//             with the per-sample sums A1 = sum dout z, A2 = sum dout a', A3 = sum dout a' xhat, A4 = sum a', A5 = sum a' xhat
//             (a' = act'(u)), so ONE pass over (y, dout) produces everything the excitation MLP's backward and the BatchNorm
//             backward need (se_bn_bwd_reduce_kernel), a tiny kernel folds them per channel once dpool is known
//             (se_bn_bwd_combine_kernel), and ONE more pass writes dy (se_bn_bwd_apply_kernel): 4 reads + 1 write of the
//             expanded tensor instead of the 7 + 2 of scale-backward (two kernels) + BatchNorm backward (two passes).
// Grids are per sample (a workgroup row never straddles samples), so the sums are fixed-order and s / dpool are per-thread
// constants; batch segments only select the statistics set by the sample index.
template <int VEC, int ACTC = -1>
__global__ __launch_bounds__(256) void bn_act_scale_fwd_kernel(const float* __restrict__ x, int hw, int c, int cw, int rp, int rows_per_block,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                               const float* __restrict__ s, int n_a, float* __restrict__ y) {
    if (ACTC >= 0) act = ACTC;                         // compile-time activation: the switch folds, the row loop is one basic block
    const int rows = hw;
    EW_THREAD_SETUP(VEC)
    const int64_t n = blockIdx.z;
    const int set = (n_a > 0 && n >= n_a) ? 1 : 0;
    float sc[VEC], sh[VEC], sv[VEC];
    vload<VEC>(s + (n * cv + vc) * VEC, sv);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const float g = gamma ? gamma[ch + v] : 1.f, b = beta ? beta[ch + v] : 0.f;
        sc[v] = rstd[set * c + ch + v] * g;
        sh[v] = b - mean[set * c + ch + v] * sc[v];
    }
    constexpr int U = 4;
    for (int r = r0 + rsub; r < r1; r += U * rp) {
        float xv[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) vload<VEC>(x + ((n * hw + (r + u * rp < r1 ? r + u * rp : r1 - 1)) * cv + vc) * VEC, xv[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u * rp >= r1) break;
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) o[v] = act_apply(xv[u][v] * sc[v] + sh[v], act) * sv[v];
            vstore<VEC>(y + ((n * hw + r + u * rp) * cv + vc) * VEC, o);
        }
    }
}

// sums[k][n][c], k = 0..4 (A1..A5 above); one workgroup per (sample, column chunk)
template <int VEC, int ACTC = -1>
__global__ __launch_bounds__(256) void se_bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dout, int nsamp, int hw, int c,
                                                               int cw, int rp, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, int act, int n_a,
                                                               float* __restrict__ sums) {
    if (ACTC >= 0) act = ACTC;                         // compile-time activation: the switch folds, the row loop is one basic block
    __shared__ float lds[256 * 5 * VEC];
    const int t = threadIdx.x, col = t % cw, rsub = t / cw;
    const int col0 = blockIdx.y * cw, cv = c / VEC;
    const int64_t n = blockIdx.x;
    const int set = (n_a > 0 && n >= n_a) ? 1 : 0;
    float a[5][VEC];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) a[k][v] = 0.f;
    if (rsub < rp && col0 + col < cv) {
        const int ch = (col0 + col) * VEC;
        float mu[VEC], rs[VEC], ga[VEC], be[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            mu[v] = mean[set * c + ch + v]; rs[v] = rstd[set * c + ch + v];
            ga[v] = gamma ? gamma[ch + v] : 1.f; be[v] = beta ? beta[ch + v] : 0.f;
        }
        constexpr int U = 4;
        for (int p = rsub; p < hw; p += U * rp) {
            float xv[U][VEC], dv[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pp = p + u * rp < hw ? p + u * rp : hw - 1;
                const int64_t off = ((n * hw + pp) * cv + col0 + col) * VEC;
                vload<VEC>(x + off, xv[u]);
                vload<VEC>(dout + off, dv[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = p + u * rp < hw;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float xh = (xv[u][v] - mu[v]) * rs[v], uu = xh * ga[v] + be[v];
                    float z, ap;
                    if (act == SUHPE_ACT_SILU) {                     // one sigmoid serves both: z = u s, act' = s + z (1 - s)
                        const float sg = fast_sigmoid(uu);
                        z = uu * sg;
                        ap = sg + z - z * sg;
                    } else {
                        z = act_apply(uu, act);
                        ap = act_grad(uu, act);
                    }
                    ap = ok ? ap : 0.f;                              // rows past the sample: no contribution (d = 0 silences z)
                    const float d = ok ? dv[u][v] : 0.f, dap = d * ap;
                    a[0][v] += d * z;
                    a[1][v] += dap;
                    a[2][v] += dap * xh;
                    a[3][v] += ap;
                    a[4][v] += ap * xh;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) lds[(t * 5 + k) * VEC + v] = a[k][v];
    __syncthreads();
    if (t < cw && col0 + t < cv) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) o[v] = 0.f;
            for (int r = 0; r < rp; ++r)
#pragma unroll
                for (int v = 0; v < VEC; ++v) o[v] += lds[((r * cw + t) * 5 + k) * VEC + v];
            vstore<VEC>(sums + (((int64_t)k * nsamp + n) * cv + col0 + t) * VEC, o);
        }
    }
}

// chan[set][2c] = (sum g, sum g xhat) over the samples of each segment; dbeta / dgamma accumulate both segments.
// A 1024-thread block owns 64 channels: thread = (channel, one of 16 sample lanes); each lane adds its samples (n = lane, lane + 16, ...)
// in fp64, the 16 lane sums of a channel meet in LDS in a fixed order (a single thread per channel walking all samples took 58 us).
constexpr int kCombCh = 64, kCombLanes = 16;
__global__ __launch_bounds__(kCombCh * kCombLanes) void se_bn_bwd_combine_kernel(const float* __restrict__ sums, const float* __restrict__ s,
                                                                                 const float* __restrict__ dpool, int nsamp, int n_a, int c, float inv_hw,
                                                                                 float* __restrict__ chan, float* __restrict__ dgamma,
                                                                                 float* __restrict__ dbeta, int accumulate) {
    __shared__ double red[2][kCombLanes][kCombCh][2];
    const int chl = threadIdx.x % kCombCh, lane = threadIdx.x / kCombCh;
    const int ch = blockIdx.x * kCombCh + chl;
    const int64_t plane = (int64_t)nsamp * c;
    const int nsets = n_a > 0 ? 2 : 1;
    for (int set = 0; set < nsets; ++set) {
        const int lo = set == 0 ? 0 : n_a, hi = (n_a > 0 && set == 0) ? n_a : nsamp;
        double g = 0.0, q = 0.0;
        if (ch < c) {
            for (int n = lo + lane; n < hi; n += kCombLanes) {
                const int64_t i = (int64_t)n * c + ch;
                const double sv = s[i], dp = (double)dpool[i] * inv_hw;
                g += sv * sums[plane + i] + dp * sums[3 * plane + i];
                q += sv * sums[2 * plane + i] + dp * sums[4 * plane + i];
            }
        }
        red[set][lane][chl][0] = g;
        red[set][lane][chl][1] = q;
    }
    __syncthreads();
    if (lane == 0 && ch < c) {
        double tot_g = 0.0, tot_q = 0.0;
        for (int set = 0; set < nsets; ++set) {
            double g = 0.0, q = 0.0;
            for (int l = 0; l < kCombLanes; ++l) { g += red[set][l][chl][0]; q += red[set][l][chl][1]; }
            chan[set * 2 * c + 2 * ch] = (float)g;
            chan[set * 2 * c + 2 * ch + 1] = (float)q;
            tot_g += (double)(float)g;
            tot_q += (double)(float)q;
        }
        if (dbeta) dbeta[ch] = (accumulate ? dbeta[ch] : 0.f) + (float)tot_g;
        if (dgamma) dgamma[ch] = (accumulate ? dgamma[ch] : 0.f) + (float)tot_q;
    }
}

// dy = gamma rstd (g - mean(g) - xhat mean(g xhat)),  g = (dout s + dpool/hw) act'(u);   grid (row blocks of a sample, column chunks, n)
template <int VEC, int ACTC = -1>
__global__ __launch_bounds__(256) void se_bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dout, int hw, int c, int cw, int rp,
                                                              int rows_per_block, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                              const float* __restrict__ s, const float* __restrict__ dpool,
                                                              const float* __restrict__ chan, int n_a, float inv_rows_a, float inv_rows_b,
                                                              int train_stats, float* __restrict__ dy) {
    if (ACTC >= 0) act = ACTC;                         // compile-time activation: the switch folds, the row loop is one basic block
    const int rows = hw;
    EW_THREAD_SETUP(VEC)
    const int64_t n = blockIdx.z;
    const int set = (n_a > 0 && n >= n_a) ? 1 : 0;
    const float inv_rows = set ? inv_rows_b : inv_rows_a, inv_hw = 1.f / (float)hw;
    float mu[VEC], rs[VEC], ga[VEC], be[VEC], m0[VEC], m1[VEC], sv[VEC], pv[VEC];
    vload<VEC>(s + (n * cv + vc) * VEC, sv);
    vload<VEC>(dpool + (n * cv + vc) * VEC, pv);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        mu[v] = mean[set * c + ch + v]; rs[v] = rstd[set * c + ch + v];
        ga[v] = gamma ? gamma[ch + v] : 1.f; be[v] = beta ? beta[ch + v] : 0.f;
        m0[v] = train_stats ? chan[set * 2 * c + 2 * (ch + v)] * inv_rows : 0.f;
        m1[v] = train_stats ? chan[set * 2 * c + 2 * (ch + v) + 1] * inv_rows : 0.f;
        pv[v] *= inv_hw;
    }
    constexpr int U = 4;
    for (int r = r0 + rsub; r < r1; r += U * rp) {
        float xv[U][VEC], dv[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t off = ((n * hw + (r + u * rp < r1 ? r + u * rp : r1 - 1)) * cv + vc) * VEC;
            vload<VEC>(x + off, xv[u]);
            vload<VEC>(dout + off, dv[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u * rp >= r1) break;
            float o[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float xh = (xv[u][v] - mu[v]) * rs[v];
                float g = (dv[u][v] * sv[v] + pv[v]) * act_grad(xh * ga[v] + be[v], act);
                g = g - m0[v] - xh * m1[v];
                o[v] = ga[v] * rs[v] * g;
            }
            vstore<VEC>(dy + ((n * hw + r + u * rp) * cv + vc) * VEC, o);
        }
    }
}

// y = a + b*rowscale[n]   (a may be NULL -> y = b*rowscale[n]); grid (chunks of a sample, n)
__global__ __launch_bounds__(256) void add_rowscale_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ rowscale, int per_vec, float* __restrict__ y) {
    const int64_t base = (int64_t)blockIdx.y * per_vec;
    const float sc = rowscale ? rowscale[blockIdx.y] : 1.f;
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per_vec; i += 4 * stride) {
        float4 bv[4], av[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ii = i + u * stride < per_vec ? i + u * stride : per_vec - 1;
            bv[u] = reinterpret_cast<const float4*>(b)[base + ii];
            if (a) av[u] = reinterpret_cast<const float4*>(a)[base + ii];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u * stride >= per_vec) break;
            float4 o = make_float4(bv[u].x * sc, bv[u].y * sc, bv[u].z * sc, bv[u].w * sc);
            if (a) { o.x += av[u].x; o.y += av[u].y; o.z += av[u].z; o.w += av[u].w; }
            reinterpret_cast<float4*>(y)[base + i + u * stride] = o;
        }
    }
}

// binary elementwise: op 0 = add, 1 = mul
__global__ __launch_bounds__(256) void binary_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t count,
                                                     int op, float* __restrict__ y) {
    const int64_t nvec = count / 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += 4 * stride) {
        float4 av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t ii = i + u * stride < nvec ? i + u * stride : nvec - 1;
            av[u] = reinterpret_cast<const float4*>(a)[ii];
            bv[u] = reinterpret_cast<const float4*>(b)[ii];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u * stride >= nvec) break;
            reinterpret_cast<float4*>(y)[i + u * stride] = op == 0 ? make_float4(av[u].x + bv[u].x, av[u].y + bv[u].y, av[u].z + bv[u].z, av[u].w + bv[u].w)
                                                                   : make_float4(av[u].x * bv[u].x, av[u].y * bv[u].y, av[u].z * bv[u].z, av[u].w * bv[u].w);
        }
    }
    for (int64_t i = nvec * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = op == 0 ? a[i] + b[i] : a[i] * b[i];
}

// y = (a + b) + c: the three input-gradient paths of a RepVGG block in one pass (same rounding order as two binary adds)
__global__ __launch_bounds__(256) void add3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c3,
                                                   int64_t count, float* __restrict__ y) {
    const int64_t nvec = count / 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += 4 * stride) {
        float4 av[4], bv[4], cv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t ii = i + u * stride < nvec ? i + u * stride : nvec - 1;
            av[u] = reinterpret_cast<const float4*>(a)[ii];
            bv[u] = reinterpret_cast<const float4*>(b)[ii];
            cv[u] = reinterpret_cast<const float4*>(c3)[ii];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u * stride >= nvec) break;
            reinterpret_cast<float4*>(y)[i + u * stride] = make_float4((av[u].x + bv[u].x) + cv[u].x, (av[u].y + bv[u].y) + cv[u].y,
                                                                       (av[u].z + bv[u].z) + cv[u].z, (av[u].w + bv[u].w) + cv[u].w);
        }
    }
    for (int64_t i = nvec * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = (a[i] + b[i]) + c3[i];
}

// ------------------------------------------------------------------------------------------ dropout mask (Philox4x32-10)
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(const uint64_t* __restrict__ counter, uint64_t extra, float keep, int64_t count,
                                                           float* __restrict__ mask) {
    const uint64_t seed = counter[0], offset = counter[1] + extra;
    const float inv = keep > 0.f ? 1.f / keep : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i * 4 < count; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t ctr = offset + (uint64_t)i;
        uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x5355u, c3 = 0x4850u;
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        const uint32_t rr[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (i * 4 + v < count) mask[i * 4 + v] = ((rr[v] >> 8) * (1.0f / 16777216.0f)) < keep ? inv : 0.f;
    }
}
__global__ void advance_counter_kernel(uint64_t* counter, uint64_t by) { counter[1] += by; }
// the masks of a whole forward pass from one launch: request k = blockIdx.y draws count[k] values at stream position + rel[k] into
// out + out_off[k]; with rel[k] = the positions a sequence of suhpe_dropout_mask calls would have used, the values are the same
__global__ __launch_bounds__(256) void dropout_mask_batch_kernel(const uint64_t* __restrict__ counter, uint64_t extra, const int64_t* __restrict__ rel,
                                                                 const float* __restrict__ keep_k, const int64_t* __restrict__ count_k,
                                                                 const int64_t* __restrict__ out_off, float* __restrict__ out) {
    const int k = blockIdx.y;
    const uint64_t seed = counter[0], offset = counter[1] + extra + (uint64_t)rel[k];
    const float keep = keep_k[k], inv = keep > 0.f ? 1.f / keep : 0.f;
    const int64_t count = count_k[k];
    float* mask = out + out_off[k];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i * 4 < count; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t ctr = offset + (uint64_t)i;
        uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x5355u, c3 = 0x4850u;
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        const uint32_t rr[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (i * 4 + v < count) mask[i * 4 + v] = ((rr[v] >> 8) * (1.0f / 16777216.0f)) < keep ? inv : 0.f;
    }
}

// ------------------------------------------------------------------------------------------ loss reductions
__global__ __launch_bounds__(256) void weighted_sum_kernel(const float* __restrict__ x, const float* __restrict__ w, int64_t count,
                                                           float scale, float* __restrict__ out) {
    __shared__ double part[4];
    double s = 0.0;
    // a zero weight SELECTS the sample out (x may be nan / inf there: the reference compacts with pred[mask] and never
    // evaluates those samples, src/agent.py:152-157)
    for (int64_t i = threadIdx.x; i < count; i += blockDim.x) {
        if (!w) s += (double)x[i];
        else if (w[i] != 0.f) s += (double)x[i] * (double)w[i];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((part[0] + part[1] + part[2] + part[3]) * (double)scale);
}

// out[0] = sum(mask*x) / sum(mask)   (nan when nothing is selected, like the mean of an empty tensor)
__global__ __launch_bounds__(256) void masked_mean_kernel(const float* __restrict__ x, const float* __restrict__ mask, int64_t count,
                                                          float* __restrict__ out) {
    __shared__ double part[8];
    double s = 0.0, c = 0.0;
    for (int64_t i = threadIdx.x; i < count; i += blockDim.x) {
        if (mask[i] != 0.f) { s += (double)x[i] * (double)mask[i]; c += (double)mask[i]; }      // select, not multiply (nan * 0)
    }
    s = wave_sum(s);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = s; part[4 + (threadIdx.x >> 6)] = c; }
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)((part[0] + part[1] + part[2] + part[3]) / (part[4] + part[5] + part[6] + part[7]));
}

__global__ void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ g, int64_t rows, int c, float scale,
                                  float* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * c) return;
    const float gv = g ? g[i / c] : 1.f;
    y[i] = gv != 0.f ? x[i] * gv * scale : 0.f;      // a zero row gradient selects the row out (x may be nan / inf in unselected samples)
}

// out[i] = src[0]*scale (broadcast of a device scalar, e.g. d(mean)/dx) ; out = alpha*a + beta*b on short vectors
__global__ void broadcast_scale_kernel(const float* __restrict__ src, float scale, int64_t count, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = src[0] * scale;
}
__global__ void axpby_kernel(const float* __restrict__ a, float alpha, const float* __restrict__ b, float beta, int64_t count,
                             float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}

// column sums of dy (rows, c): partial + finalize (bias gradient)
template <int VEC>
__global__ __launch_bounds__(256) void colsum_partial(const float* __restrict__ x, int64_t rows, int c, int cw, int rp,
                                                      int64_t rows_per_block, float* __restrict__ part) {
    __shared__ float lds[256 * 2 * VEC];
    const int t = threadIdx.x, col = t % cw, rsub = t / cw;
    const int col0 = blockIdx.y * cw, cv = c / VEC;
    float s0[VEC], s1[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { s0[v] = 0.f; s1[v] = 0.f; }
    if (rsub < rp && col0 + col < cv) {
        const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
        int64_t r1 = r0 + rows_per_block;
        if (r1 > rows) r1 = rows;
        for (int64_t r = r0 + rsub; r < r1; r += rp) {
            float xv[VEC];
            vload<VEC>(x + (r * cv + col0 + col) * VEC, xv);
#pragma unroll
            for (int v = 0; v < VEC; ++v) s0[v] += xv[v];
        }
    }
    block_combine<VEC>(s0, s1, cw, rp, col0, c, blockIdx.x, part, lds);
}
__global__ __launch_bounds__(kFinThreads) void colsum_finalize(const float* __restrict__ part, int gx, int c, float* __restrict__ out, int accumulate) {
    double s, q;
    if (!block_channel_sums(part, gx, c, &s, &q)) return;
    const int ch = blockIdx.x * kFinCh + (threadIdx.x & (kFinCh - 1));
    out[ch] = accumulate ? out[ch] + (float)s : (float)s;
}

// ------------------------------------------------------------------------------------------ optimiser / EMA
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t count, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt) {
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ e, const float* __restrict__ p, int64_t count, float alpha) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        e[i] = e[i] * alpha + (1.f - alpha) * p[i];
}
__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, int64_t count, float value) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) p[i] = value;
}

static inline bool vec4_ok(int c, const void* a, const void* b = nullptr, const void* d = nullptr) {
    auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return (c % 4 == 0) && al(a) && al(b) && al(d);
}

}  // namespace suhpe

using namespace suhpe;

#define LAUNCH(kern, grid, block, stream, ...) hipLaunchKernelGGL(kern, grid, block, 0, as_stream(stream), __VA_ARGS__)
// kernels with the activation as a template constant: SiLU (every BatchNorm of EfficientNetV2) and GELU (TinyViT) get their own instantiations
#define LAUNCH_ACT(kern, VEC, act, grid, block, stream, ...)                                                                       \
    do {                                                                                                                            \
        if ((act) == SUHPE_ACT_SILU) hipLaunchKernelGGL((kern<VEC, SUHPE_ACT_SILU>), grid, block, 0, as_stream(stream), __VA_ARGS__); \
        else if ((act) == SUHPE_ACT_GELU) hipLaunchKernelGGL((kern<VEC, SUHPE_ACT_GELU>), grid, block, 0, as_stream(stream), __VA_ARGS__); \
        else hipLaunchKernelGGL((kern<VEC, -1>), grid, block, 0, as_stream(stream), __VA_ARGS__);                                   \
    } while (0)

namespace suhpe {
void launch_bn_bwd_finalize(const float* part, int gx_a, int gx, int c, float* sums, float* dgamma, float* dbeta, int accumulate, hipStream_t st) {
    if (gx_a > 0)
        hipLaunchKernelGGL(bn_bwd_finalize2, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), 0, st, part, gx_a, gx, c, sums, dgamma, dbeta, accumulate);
    else
        hipLaunchKernelGGL(bn_bwd_finalize, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), 0, st, part, gx, c, sums, dgamma, dbeta, accumulate);
}
}  // namespace suhpe

extern "C" {

// records (one spare row block for the two-segment launches) + per-channel backward sums of up to two segments
int64_t suhpe_bn_workspace_bytes(int c) { return (kPartFloats + (int64_t)c * 4 + (int64_t)c * 8) * sizeof(float); }
static float* bn_sums_of(void* workspace, int c) { return static_cast<float*>(workspace) + kPartFloats + (int64_t)c * 4; }

int suhpe_bn_stats(const float* x, int64_t rows, int c, float eps, float momentum, float* mean, float* rstd,
                   float* running_mean, float* running_var, int64_t* nbt, void* workspace, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && mean && rstd && workspace && rows > 0 && c > 0, "suhpe_bn_stats: bad arguments (rows=%lld c=%d)", (long long)rows, c);
    float* part = static_cast<float*>(workspace);
    if (vec4_ok(c, x)) {
        const ColGeom g = col_geom(rows, c, 4);
        LAUNCH(bn_stats_partial<4>, dim3(g.gx, g.ny), dim3(256), stream, x, rows, c, g.cw, g.rp, g.rows_per_block, part);
        LAUNCH(bn_stats_finalize, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, g.gx, c, rows, eps, momentum, mean, rstd,
               running_mean, running_var, nbt);
    } else {
        const ColGeom g = col_geom(rows, c, 1);
        LAUNCH(bn_stats_partial<1>, dim3(g.gx, g.ny), dim3(256), stream, x, rows, c, g.cw, g.rp, g.rows_per_block, part);
        LAUNCH(bn_stats_finalize, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, g.gx, c, rows, eps, momentum, mean, rstd,
               running_mean, running_var, nbt);
    }
    return check_launch("suhpe_bn_stats");
}

int suhpe_bn_stats_from_partials(const float* part, int gx, int64_t rows, int c, float eps, float momentum, float* mean, float* rstd,
                                 float* running_mean, float* running_var, int64_t* nbt, suhpe_stream_t stream) {
    SUHPE_REQUIRE(part && mean && rstd && gx > 0 && rows > 0 && c > 0, "suhpe_bn_stats_from_partials: bad arguments");
    LAUNCH(bn_stats_finalize, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, gx, c, rows, eps, momentum, mean, rstd, running_mean, running_var, nbt);
    return check_launch("suhpe_bn_stats_from_partials");
}

int suhpe_bn_stats_from_partials_range(const float* part, int first, int count, const float* extra, int64_t rows, int c, float eps,
                                       float momentum, float* mean, float* rstd, float* running_mean, float* running_var, int64_t* nbt,
                                       suhpe_stream_t stream) {
    SUHPE_REQUIRE(part && mean && rstd && first >= 0 && count >= 0 && (count > 0 || extra) && rows > 0 && c > 0, "suhpe_bn_stats_from_partials_range: bad arguments");
    LAUNCH(bn_stats_finalize, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part + (int64_t)first * c * 4, count, c, rows, eps, momentum, mean,
           rstd, running_mean, running_var, nbt, extra);
    return check_launch("suhpe_bn_stats_from_partials_range");
}

int suhpe_bn_eval_stats(const float* rm, const float* rv, int c, float eps, float* mean, float* rstd, suhpe_stream_t stream) {
    SUHPE_REQUIRE(rm && rv && mean && rstd && c > 0, "suhpe_bn_eval_stats: bad arguments");
    LAUNCH(bn_eval_stats_kernel, dim3((c + 255) / 256), dim3(256), stream, rm, rv, c, eps, mean, rstd);
    return check_launch("suhpe_bn_eval_stats");
}

int suhpe_syncbn_moments(const float* mean, const float* rstd, int c, float eps, float* moments, suhpe_stream_t stream) {
    SUHPE_REQUIRE(mean && rstd && moments && c > 0, "suhpe_syncbn_moments: bad arguments");
    LAUNCH(syncbn_moments_kernel, dim3((c + 255) / 256), dim3(256), stream, mean, rstd, c, eps, moments);
    return check_launch("suhpe_syncbn_moments");
}

int suhpe_syncbn_finish(const float* moments, int c, float eps, float momentum, int64_t rows_global, float* mean, float* rstd, float* running_mean,
                        float* running_var, int64_t* num_batches_tracked, suhpe_stream_t stream) {
    SUHPE_REQUIRE(moments && mean && rstd && c > 0 && rows_global > 0, "suhpe_syncbn_finish: bad arguments");
    const float unbias = rows_global > 1 ? (float)((double)rows_global / (double)(rows_global - 1)) : 1.f;
    LAUNCH(syncbn_finish_kernel, dim3((c + 255) / 256), dim3(256), stream, moments, c, eps, momentum, unbias, mean, rstd, running_mean, running_var,
           reinterpret_cast<long*>(num_batches_tracked));
    return check_launch("suhpe_syncbn_finish");
}

int suhpe_repvgg_fuse(const float* w3, const float* w1, const float* const* bn3, const float* const* bn1, const float* const* bnid,
                      float eps, int cout, int cin_per_group, float* w_eq, float* ones, float* b_eq, suhpe_stream_t stream) {
    SUHPE_REQUIRE(w3 && w1 && bn3 && bn1 && w_eq && ones && b_eq && cout > 0 && cin_per_group > 0, "suhpe_repvgg_fuse: bad arguments");
    for (int k = 0; k < 4; ++k) SUHPE_REQUIRE(bn3[k] && bn1[k] && (!bnid || bnid[k]), "suhpe_repvgg_fuse: BatchNorm arrays are (gamma, beta, running_mean, running_var)");
    const BnRef b3{bn3[0], bn3[1], bn3[2], bn3[3], eps}, b1{bn1[0], bn1[1], bn1[2], bn1[3], eps};
    const BnRef bi = bnid ? BnRef{bnid[0], bnid[1], bnid[2], bnid[3], eps} : BnRef{nullptr, nullptr, nullptr, nullptr, eps};
    const int64_t total = (int64_t)cout * 9 * cin_per_group;
    LAUNCH(repvgg_fuse_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), stream, w3, w1, b3, b1, bi, bnid ? 1 : 0, cout, cin_per_group, w_eq,
           ones, b_eq);
    return check_launch("suhpe_repvgg_fuse");
}

int suhpe_bn_affine_from_stats(const float* mean, const float* rstd, const float* gamma, const float* beta, int segments, int c, float* scale,
                               float* shift, suhpe_stream_t stream) {
    SUHPE_REQUIRE(mean && rstd && scale && shift && c > 0 && (segments == 1 || segments == 2), "suhpe_bn_affine_from_stats: bad arguments");
    LAUNCH(bn_affine_from_stats_kernel, dim3((segments * c + 255) / 256), dim3(256), stream, mean, rstd, gamma, beta, segments, c, scale, shift);
    return check_launch("suhpe_bn_affine_from_stats");
}

int suhpe_bn_eval_affine(const float* rm, const float* rv, const float* gamma, const float* beta, int c, float eps, float* scale,
                         float* shift, suhpe_stream_t stream) {
    SUHPE_REQUIRE(rm && rv && scale && shift && c > 0, "suhpe_bn_eval_affine: bad arguments");
    LAUNCH(bn_eval_affine_kernel, dim3((c + 255) / 256), dim3(256), stream, rm, rv, gamma, beta, c, eps, scale, shift);
    return check_launch("suhpe_bn_eval_affine");
}

static int bn_act_fwd_impl(const float* x, int64_t rows, int c, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, int act, const float* residual, int post, const float* rowscale, int hw, float* y,
                           suhpe_stream_t stream, const char* who) {
    if (!(x && y && mean && rstd && rows > 0 && rows < (1LL << 31) && c > 0)) { set_error("%s: bad arguments", who); return SUHPE_EINVAL; }
    if (post && !(residual && hw > 0 && rows % hw == 0)) { set_error("%s: the skip form needs a skip tensor and rows %% hw == 0", who); return SUHPE_EINVAL; }
    if (vec4_ok(c, x, y, residual)) {
        const EwGeom g = ew_geom(rows, c, 4);
        LAUNCH_ACT(bn_act_fwd_kernel, 4, act, dim3(g.gx, g.ny), dim3(256), stream, x, (int)rows, c, g.cw, g.rp, g.rows_per_block, mean, rstd, gamma, beta,
               act, residual, post, rowscale, hw, y);
    } else {
        const EwGeom g = ew_geom(rows, c, 1);
        LAUNCH_ACT(bn_act_fwd_kernel, 1, act, dim3(g.gx, g.ny), dim3(256), stream, x, (int)rows, c, g.cw, g.rp, g.rows_per_block, mean, rstd, gamma, beta,
               act, residual, post, rowscale, hw, y);
    }
    return check_launch(who);
}

int suhpe_bn_act_fwd(const float* x, int64_t rows, int c, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, int act, const float* residual, float* y, suhpe_stream_t stream) {
    return bn_act_fwd_impl(x, rows, c, mean, rstd, gamma, beta, act, residual, 0, nullptr, 1, y, stream, "suhpe_bn_act_fwd");
}

int suhpe_bn_act_skip_fwd(const float* x, int64_t rows, int c, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int act, const float* skip, const float* rowscale, int hw, float* y,
                          suhpe_stream_t stream) {
    return bn_act_fwd_impl(x, rows, c, mean, rstd, gamma, beta, act, skip, 1, rowscale, hw, y, stream, "suhpe_bn_act_skip_fwd");
}

static int bn_act_bwd_impl(const float* x, const float* dy, const float* y_out, int64_t rows, int c, const float* mean,
                           const float* rstd, const float* gamma, const float* beta, int act, int train_stats, const float* rowscale,
                           int hw, float* dx, float* dgamma, float* dbeta, int accumulate, void* workspace, suhpe_stream_t stream,
                           const char* who) {
    if (!(x && dy && dx && mean && rstd && workspace && rows > 0 && rows < (1LL << 31) && c > 0)) { set_error("%s: bad arguments", who); return SUHPE_EINVAL; }
    if (y_out && act != SUHPE_ACT_RELU) { set_error("%s: y_out (pre-activation residual) needs act=RELU", who); return SUHPE_EINVAL; }
    if (rowscale && !(hw > 0 && rows % hw == 0 && !y_out)) { set_error("%s: rowscale needs rows %% hw == 0 and no y_out", who); return SUHPE_EINVAL; }
    float* part = static_cast<float*>(workspace);
    float* sums = bn_sums_of(workspace, c);
    const float inv_rows = 1.0f / (float)rows;
    if (vec4_ok(c, x, dy, dx) && vec4_ok(c, y_out)) {
        const ColGeom g = col_geom(rows, c, 4);
        LAUNCH_ACT(bn_bwd_partial, 4, act, dim3(g.gx, g.ny), dim3(256), stream, x, dy, y_out, rows, c, g.cw, g.rp, g.rows_per_block, mean,
               rstd, gamma, beta, act, rowscale, hw, part);
        LAUNCH(bn_bwd_finalize, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, g.gx, c, sums, dgamma, dbeta, accumulate);
        const EwGeom e = ew_geom(rows, c, 4);
        LAUNCH_ACT(bn_bwd_apply, 4, act, dim3(e.gx, e.ny), dim3(256), stream, x, dy, y_out, (int)rows, c, e.cw, e.rp, e.rows_per_block, mean, rstd,
               gamma, beta, act, sums, inv_rows, train_stats, rowscale, hw, dx, SegArgs{0, 0, kRevSecondPass}, 0.f);
    } else {
        const ColGeom g = col_geom(rows, c, 1);
        LAUNCH_ACT(bn_bwd_partial, 1, act, dim3(g.gx, g.ny), dim3(256), stream, x, dy, y_out, rows, c, g.cw, g.rp, g.rows_per_block, mean,
               rstd, gamma, beta, act, rowscale, hw, part);
        LAUNCH(bn_bwd_finalize, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, g.gx, c, sums, dgamma, dbeta, accumulate);
        const EwGeom e = ew_geom(rows, c, 1);
        LAUNCH_ACT(bn_bwd_apply, 1, act, dim3(e.gx, e.ny), dim3(256), stream, x, dy, y_out, (int)rows, c, e.cw, e.rp, e.rows_per_block, mean, rstd,
               gamma, beta, act, sums, inv_rows, train_stats, rowscale, hw, dx, SegArgs{0, 0, kRevSecondPass}, 0.f);
    }
    return check_launch(who);
}

int suhpe_bn_act_bwd(const float* x, const float* dy, const float* y_out, int64_t rows, int c, const float* mean,
                     const float* rstd, const float* gamma, const float* beta, int act, int train_stats, float* dx,
                     float* dgamma, float* dbeta, int accumulate, void* workspace, suhpe_stream_t stream) {
    return bn_act_bwd_impl(x, dy, y_out, rows, c, mean, rstd, gamma, beta, act, train_stats, nullptr, 1, dx, dgamma, dbeta, accumulate,
                           workspace, stream, "suhpe_bn_act_bwd");
}

int suhpe_bn_act_skip_bwd(const float* x, const float* dy, int64_t rows, int c, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, int act, int train_stats, const float* rowscale, int hw,
                          float* dx, float* dgamma, float* dbeta, int accumulate, void* workspace, suhpe_stream_t stream) {
    return bn_act_bwd_impl(x, dy, nullptr, rows, c, mean, rstd, gamma, beta, act, train_stats, rowscale, hw, dx, dgamma, dbeta, accumulate,
                           workspace, stream, "suhpe_bn_act_skip_bwd");
}

// ---- two batch segments per launch (rows [0, rows_a) and [rows_a, rows)); mean / rstd are [2][c]
static bool seg_ok(int64_t rows, int64_t rows_a) { return rows_a > 0 && rows_a < rows; }

int suhpe_bn_stats_seg(const float* x, int64_t rows, int64_t rows_a, int c, float eps, float momentum, float* mean, float* rstd,
                       float* running_mean, float* running_var, int64_t* nbt, void* workspace, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && mean && rstd && workspace && c > 0 && seg_ok(rows, rows_a), "suhpe_bn_stats_seg: bad arguments (rows=%lld rows_a=%lld c=%d)",
                  (long long)rows, (long long)rows_a, c);
    float* part = static_cast<float*>(workspace);
    const int vec = vec4_ok(c, x) ? 4 : 1;
    const ColGeom g = col_geom(rows, c, vec);
    const int gx_a = (int)ceil_div(rows_a, g.rows_per_block), gx_b = (int)ceil_div(rows - rows_a, g.rows_per_block);
    const SegArgs sg{rows_a, gx_a, 0};
    if (vec == 4) LAUNCH(bn_stats_partial<4>, dim3(gx_a + gx_b, g.ny), dim3(256), stream, x, rows, c, g.cw, g.rp, g.rows_per_block, part, sg);
    else LAUNCH(bn_stats_partial<1>, dim3(gx_a + gx_b, g.ny), dim3(256), stream, x, rows, c, g.cw, g.rp, g.rows_per_block, part, sg);
    LAUNCH(bn_stats_finalize2, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, gx_a, gx_b, c, rows_a, rows - rows_a, eps, momentum,
           mean, rstd, running_mean, running_var, nbt, static_cast<const float*>(nullptr));
    return check_launch("suhpe_bn_stats_seg");
}

int suhpe_bn_stats_from_partials_seg(const float* part, int tiles, int first_records, const float* boundary, int64_t rows, int64_t rows_a,
                                     int c, float eps, float momentum, float* mean, float* rstd, float* running_mean,
                                     float* running_var, int64_t* nbt, suhpe_stream_t stream) {
    SUHPE_REQUIRE(part && mean && rstd && c > 0 && seg_ok(rows, rows_a) && first_records > 0 && first_records <= tiles &&
                      (first_records < tiles || boundary),
                  "suhpe_bn_stats_from_partials_seg: bad arguments");
    LAUNCH(bn_stats_finalize2, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, first_records, tiles - first_records, c, rows_a,
           rows - rows_a, eps, momentum, mean, rstd, running_mean, running_var, nbt, boundary);
    return check_launch("suhpe_bn_stats_from_partials_seg");
}

// the two entry points above for LONG record lists, spread over the chip (bn_records_fold); the records are overwritten
int suhpe_bn_stats_from_partials_consume(float* part, int tiles, int first_records, const float* boundary, int64_t rows, int64_t rows_a, int c,
                                         float eps, float momentum, float* mean, float* rstd, float* running_mean, float* running_var,
                                         int64_t* nbt, suhpe_stream_t stream) {
    const bool two = rows_a < rows;
    SUHPE_REQUIRE(part && mean && rstd && c > 0 && rows > 0 && rows_a > 0 && rows_a <= rows && tiles > 0 && first_records > 0 && first_records <= tiles,
                  "suhpe_bn_stats_from_partials_consume: bad arguments");
    SUHPE_REQUIRE(two ? (first_records < tiles || boundary) : (first_records == tiles && !boundary),
                  "suhpe_bn_stats_from_partials_consume: one segment takes every record and no boundary record; two need records or a boundary record for the second");
    auto seg_of = [](int first, int count) {
        FoldSeg s{first, count, 1};
        s.run = count > 0 ? (count + kFoldRuns - 1) / kFoldRuns : 1;
        return s;
    };
    const FoldSeg sa = seg_of(0, first_records), sb = seg_of(first_records, tiles - first_records);
    const int runs_a = (sa.count + sa.run - 1) / sa.run, runs_b = sb.count > 0 ? (sb.count + sb.run - 1) / sb.run : 0;
    static_assert(kFoldRuns <= kFinSub - 1, "one pair per reducing thread, the last one takes the boundary record");
    LAUNCH(bn_records_fold, dim3((c + kFinCh - 1) / kFinCh, runs_a + runs_b), dim3(kFinThreads), stream, part, sa, sb, runs_a, c);
    LAUNCH(bn_stats_finalize_folded, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, sa, sb, c, rows_a, rows - rows_a, eps, momentum,
           mean, rstd, running_mean, running_var, nbt, boundary);
    return check_launch("suhpe_bn_stats_from_partials_consume");
}

int suhpe_bn_act_fwd_seg(const float* x, int64_t rows, int64_t rows_a, int c, const float* mean, const float* rstd, const float* gamma,
                         const float* beta, int act, const float* residual, int post, const float* rowscale, int hw, float* y,
                         suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && y && mean && rstd && rows < (1LL << 31) && c > 0 && seg_ok(rows, rows_a), "suhpe_bn_act_fwd_seg: bad arguments");
    SUHPE_REQUIRE(!post || (residual && hw > 0 && rows % hw == 0 && rows_a % hw == 0), "suhpe_bn_act_fwd_seg: the skip form needs a skip tensor and whole samples per segment");
    const int vec = vec4_ok(c, x, y, residual) ? 4 : 1;
    const EwGeom g = ew_geom(rows, c, vec);
    const int gx_a = (int)ceil_div(rows_a, (int64_t)g.rows_per_block), gx_b = (int)ceil_div(rows - rows_a, (int64_t)g.rows_per_block);
    const SegArgs sg{rows_a, gx_a, 0};
    if (vec == 4)
        LAUNCH_ACT(bn_act_fwd_kernel, 4, act, dim3(gx_a + gx_b, g.ny), dim3(256), stream, x, (int)rows, c, g.cw, g.rp, g.rows_per_block, mean, rstd, gamma, beta, act,
               residual, post, rowscale, hw, y, sg);
    else
        LAUNCH_ACT(bn_act_fwd_kernel, 1, act, dim3(gx_a + gx_b, g.ny), dim3(256), stream, x, (int)rows, c, g.cw, g.rp, g.rows_per_block, mean, rstd, gamma, beta, act,
               residual, post, rowscale, hw, y, sg);
    return check_launch("suhpe_bn_act_fwd_seg");
}

// ---- the two halves of the BatchNorm backward as separate calls, so that a data-parallel caller can all-reduce the per-channel
// sums in between (SyncBN: statistics and gradient means over the global batch)
int suhpe_bn_act_bwd_sums(const float* x, const float* dy, const float* y_out, int64_t rows, int c, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, int act, const float* rowscale, int hw, float* sums, float* dgamma,
                          float* dbeta, int accumulate, void* workspace, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && dy && sums && mean && rstd && workspace && rows > 0 && rows < (1LL << 31) && c > 0, "suhpe_bn_act_bwd_sums: bad arguments");
    SUHPE_REQUIRE(!y_out || act == SUHPE_ACT_RELU, "suhpe_bn_act_bwd_sums: y_out (pre-activation residual) needs act=RELU");
    SUHPE_REQUIRE(!rowscale || (hw > 0 && rows % hw == 0 && !y_out), "suhpe_bn_act_bwd_sums: rowscale needs rows %% hw == 0 and no y_out");
    float* part = static_cast<float*>(workspace);
    const int vec = (vec4_ok(c, x, dy) && vec4_ok(c, y_out)) ? 4 : 1;
    const ColGeom g = col_geom(rows, c, vec);
    if (vec == 4)
        LAUNCH_ACT(bn_bwd_partial, 4, act, dim3(g.gx, g.ny), dim3(256), stream, x, dy, y_out, rows, c, g.cw, g.rp, g.rows_per_block, mean, rstd, gamma, beta, act, rowscale,
               hw, part);
    else
        LAUNCH_ACT(bn_bwd_partial, 1, act, dim3(g.gx, g.ny), dim3(256), stream, x, dy, y_out, rows, c, g.cw, g.rp, g.rows_per_block, mean, rstd, gamma, beta, act, rowscale,
               hw, part);
    LAUNCH(bn_bwd_finalize, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, g.gx, c, sums, dgamma, dbeta, accumulate);
    return check_launch("suhpe_bn_act_bwd_sums");
}

int suhpe_bn_act_bwd_apply(const float* x, const float* dy, const float* y_out, int64_t rows, int64_t total_rows, int c, const float* mean,
                           const float* rstd, const float* gamma, const float* beta, int act, const float* sums, const float* rowscale, int hw,
                           float* dx, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && dy && dx && sums && mean && rstd && rows > 0 && rows < (1LL << 31) && total_rows >= rows && c > 0, "suhpe_bn_act_bwd_apply: bad arguments");
    const int vec = (vec4_ok(c, x, dy, dx) && vec4_ok(c, y_out)) ? 4 : 1;
    const EwGeom e = ew_geom(rows, c, vec);
    const float inv = 1.0f / (float)total_rows;
    if (vec == 4)
        LAUNCH_ACT(bn_bwd_apply, 4, act, dim3(e.gx, e.ny), dim3(256), stream, x, dy, y_out, (int)rows, c, e.cw, e.rp, e.rows_per_block, mean, rstd, gamma, beta, act, sums,
               inv, 1, rowscale, hw, dx, SegArgs{0, 0, kRevSecondPass}, 0.f);
    else
        LAUNCH_ACT(bn_bwd_apply, 1, act, dim3(e.gx, e.ny), dim3(256), stream, x, dy, y_out, (int)rows, c, e.cw, e.rp, e.rows_per_block, mean, rstd, gamma, beta, act, sums,
               inv, 1, rowscale, hw, dx, SegArgs{0, 0, kRevSecondPass}, 0.f);
    return check_launch("suhpe_bn_act_bwd_apply");
}

/* the apply pass alone with two batch segments: sums [2][c][2] from a producer that reduced them (suhpe_dwconv_dgrad_bnsums) */
int suhpe_bn_act_bwd_apply_seg(const float* x, const float* dy, int64_t rows, int64_t rows_a, int c, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, int act, const float* sums, float* dx, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && dy && dx && sums && mean && rstd && rows > 0 && rows < (1LL << 31) && c > 0 && seg_ok(rows, rows_a), "suhpe_bn_act_bwd_apply_seg: bad arguments");
    const int vec = vec4_ok(c, x, dy, dx) ? 4 : 1;
    const EwGeom e = ew_geom(rows, c, vec);
    const int ex_a = (int)ceil_div(rows_a, (int64_t)e.rows_per_block), ex_b = (int)ceil_div(rows - rows_a, (int64_t)e.rows_per_block);
    const SegArgs se{rows_a, ex_a, kRevSecondPass};
    const float inv_a = 1.0f / (float)rows_a, inv_b = 1.0f / (float)(rows - rows_a);
    if (vec == 4)
        LAUNCH_ACT(bn_bwd_apply, 4, act, dim3(ex_a + ex_b, e.ny), dim3(256), stream, x, dy, (const float*)nullptr, (int)rows, c, e.cw, e.rp, e.rows_per_block, mean, rstd,
               gamma, beta, act, sums, inv_a, 1, (const float*)nullptr, 1, dx, se, inv_b);
    else
        LAUNCH_ACT(bn_bwd_apply, 1, act, dim3(ex_a + ex_b, e.ny), dim3(256), stream, x, dy, (const float*)nullptr, (int)rows, c, e.cw, e.rp, e.rows_per_block, mean, rstd,
               gamma, beta, act, sums, inv_a, 1, (const float*)nullptr, 1, dx, se, inv_b);
    return check_launch("suhpe_bn_act_bwd_apply_seg");
}

int suhpe_bn_act_pool_fwd(const float* x, int64_t rows, int64_t rows_a, int c, int hw, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int act, float* y, float* pool, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && pool && mean && rstd && rows > 0 && rows < (1LL << 31) && c > 0 && hw > 0 && rows % hw == 0 && rows_a >= 0 && rows_a < rows &&
                      rows_a % hw == 0,
                  "suhpe_bn_act_pool_fwd: bad arguments (whole samples per segment: rows %% hw == 0, rows_a %% hw == 0)");
    const int vec = (vec4_ok(c, x, y) && vec4_ok(c, pool)) ? 4 : 1;
    const int cv = c / vec, cw = cv < 32 ? cv : 32, ny = (cv + cw - 1) / cw, rp = 256 / cw;
    const SegArgs sg{rows_a, (int)(rows_a / hw), 0};
    const dim3 grid((unsigned)(rows / hw), ny);
    if (vec == 4) LAUNCH_ACT(bn_act_pool_fwd_kernel, 4, act, grid, dim3(256), stream, x, (int)rows, c, cw, rp, hw, mean, rstd, gamma, beta, act, y, pool, sg);
    else LAUNCH_ACT(bn_act_pool_fwd_kernel, 1, act, grid, dim3(256), stream, x, (int)rows, c, cw, rp, hw, mean, rstd, gamma, beta, act, y, pool, sg);
    return check_launch("suhpe_bn_act_pool_fwd");
}

static bool se_bn_args_ok(const float* x, int64_t rows, int64_t rows_a, int c, int hw) {
    return x && rows > 0 && rows < (1LL << 31) && c > 0 && hw > 0 && rows % hw == 0 && rows_a >= 0 && rows_a < rows && rows_a % hw == 0;
}

int suhpe_bn_act_scale_fwd(const float* x, int64_t rows, int64_t rows_a, int c, int hw, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, int act, const float* s, float* y, suhpe_stream_t stream) {
    SUHPE_REQUIRE(se_bn_args_ok(x, rows, rows_a, c, hw) && mean && rstd && s && y, "suhpe_bn_act_scale_fwd: bad arguments (whole samples per segment)");
    const int n = (int)(rows / hw), n_a = (int)(rows_a / hw);
    const bool v4 = vec4_ok(c, x, s, y);
    const EwGeom g = ew_geom(hw, c, v4 ? 4 : 1);
    if (v4) LAUNCH_ACT(bn_act_scale_fwd_kernel, 4, act, dim3(g.gx, g.ny, n), dim3(256), stream, x, hw, c, g.cw, g.rp, g.rows_per_block, mean, rstd, gamma, beta, act, s, n_a, y);
    else LAUNCH_ACT(bn_act_scale_fwd_kernel, 1, act, dim3(g.gx, g.ny, n), dim3(256), stream, x, hw, c, g.cw, g.rp, g.rows_per_block, mean, rstd, gamma, beta, act, s, n_a, y);
    return check_launch("suhpe_bn_act_scale_fwd");
}

int suhpe_se_bn_bwd_reduce(const float* x, const float* dout, int64_t rows, int64_t rows_a, int c, int hw, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int act, float* sums, suhpe_stream_t stream) {
    SUHPE_REQUIRE(se_bn_args_ok(x, rows, rows_a, c, hw) && dout && mean && rstd && sums, "suhpe_se_bn_bwd_reduce: bad arguments (whole samples per segment)");
    const int n = (int)(rows / hw), n_a = (int)(rows_a / hw);
    const int vec = (vec4_ok(c, x, dout, sums)) ? 4 : 1;
    const int cv = c / vec, cw = cv < 32 ? cv : 32, ny = (cv + cw - 1) / cw, rp = 256 / cw;
    if (vec == 4) LAUNCH_ACT(se_bn_bwd_reduce_kernel, 4, act, dim3(n, ny), dim3(256), stream, x, dout, n, hw, c, cw, rp, mean, rstd, gamma, beta, act, n_a, sums);
    else LAUNCH_ACT(se_bn_bwd_reduce_kernel, 1, act, dim3(n, ny), dim3(256), stream, x, dout, n, hw, c, cw, rp, mean, rstd, gamma, beta, act, n_a, sums);
    return check_launch("suhpe_se_bn_bwd_reduce");
}

int suhpe_se_bn_bwd_combine(const float* sums, const float* s, const float* dpool, int n, int n_a, int c, int hw, float* chan, float* dgamma,
                            float* dbeta, int accumulate, suhpe_stream_t stream) {
    SUHPE_REQUIRE(sums && s && dpool && chan && n > 0 && n_a >= 0 && n_a < n && c > 0 && hw > 0, "suhpe_se_bn_bwd_combine: bad arguments");
    LAUNCH(se_bn_bwd_combine_kernel, dim3((c + kCombCh - 1) / kCombCh), dim3(kCombCh * kCombLanes), stream, sums, s, dpool, n, n_a, c, 1.0f / (float)hw, chan, dgamma,
           dbeta, accumulate);
    return check_launch("suhpe_se_bn_bwd_combine");
}

int suhpe_se_bn_bwd_apply(const float* x, const float* dout, int64_t rows, int64_t rows_a, int c, int hw, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, int act, const float* s, const float* dpool, const float* chan, int train_stats,
                          float* dy, suhpe_stream_t stream) {
    SUHPE_REQUIRE(se_bn_args_ok(x, rows, rows_a, c, hw) && dout && mean && rstd && s && dpool && chan && dy, "suhpe_se_bn_bwd_apply: bad arguments");
    const int n = (int)(rows / hw), n_a = (int)(rows_a / hw);
    const bool v4 = vec4_ok(c, x, dout, dy) && vec4_ok(c, s, dpool);
    const EwGeom g = ew_geom(hw, c, v4 ? 4 : 1);
    const float inv_a = 1.0f / (float)(rows_a > 0 ? rows_a : rows), inv_b = 1.0f / (float)(rows - rows_a);
    if (v4) LAUNCH_ACT(se_bn_bwd_apply_kernel, 4, act, dim3(g.gx, g.ny, n), dim3(256), stream, x, dout, hw, c, g.cw, g.rp, g.rows_per_block, mean, rstd, gamma, beta, act,
                   s, dpool, chan, n_a, inv_a, inv_b, train_stats, dy);
    else LAUNCH_ACT(se_bn_bwd_apply_kernel, 1, act, dim3(g.gx, g.ny, n), dim3(256), stream, x, dout, hw, c, g.cw, g.rp, g.rows_per_block, mean, rstd, gamma, beta, act,
                s, dpool, chan, n_a, inv_a, inv_b, train_stats, dy);
    return check_launch("suhpe_se_bn_bwd_apply");
}

int suhpe_bn_act_bwd_seg(const float* x, const float* dy, const float* y_out, int64_t rows, int64_t rows_a, int c, const float* mean,
                         const float* rstd, const float* gamma, const float* beta, int act, const float* rowscale, int hw, float* dx,
                         float* dgamma, float* dbeta, int accumulate, void* workspace, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && dy && dx && mean && rstd && workspace && rows < (1LL << 31) && c > 0 && seg_ok(rows, rows_a), "suhpe_bn_act_bwd_seg: bad arguments");
    SUHPE_REQUIRE(!y_out || act == SUHPE_ACT_RELU, "suhpe_bn_act_bwd_seg: y_out (pre-activation residual) needs act=RELU");
    SUHPE_REQUIRE(!rowscale || (hw > 0 && rows % hw == 0 && rows_a % hw == 0 && !y_out), "suhpe_bn_act_bwd_seg: rowscale needs whole samples per segment and no y_out");
    float* part = static_cast<float*>(workspace);
    float* sums = bn_sums_of(workspace, c);
    const int vec = (vec4_ok(c, x, dy, dx) && vec4_ok(c, y_out)) ? 4 : 1;
    const ColGeom g = col_geom(rows, c, vec);
    const int gx_a = (int)ceil_div(rows_a, g.rows_per_block), gx_b = (int)ceil_div(rows - rows_a, g.rows_per_block);
    const SegArgs sg{rows_a, gx_a, 0};
    const EwGeom e = ew_geom(rows, c, vec);
    const int ex_a = (int)ceil_div(rows_a, (int64_t)e.rows_per_block), ex_b = (int)ceil_div(rows - rows_a, (int64_t)e.rows_per_block);
    const SegArgs se{rows_a, ex_a, kRevSecondPass};
    const float inv_a = 1.0f / (float)rows_a, inv_b = 1.0f / (float)(rows - rows_a);
    if (vec == 4) {
        LAUNCH_ACT(bn_bwd_partial, 4, act, dim3(gx_a + gx_b, g.ny), dim3(256), stream, x, dy, y_out, rows, c, g.cw, g.rp, g.rows_per_block, mean, rstd, gamma, beta,
               act, rowscale, hw, part, sg);
        LAUNCH(bn_bwd_finalize2, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, gx_a, gx_a + gx_b, c, sums, dgamma, dbeta, accumulate);
        LAUNCH_ACT(bn_bwd_apply, 4, act, dim3(ex_a + ex_b, e.ny), dim3(256), stream, x, dy, y_out, (int)rows, c, e.cw, e.rp, e.rows_per_block, mean, rstd, gamma,
               beta, act, sums, inv_a, 1, rowscale, hw, dx, se, inv_b);
    } else {
        LAUNCH_ACT(bn_bwd_partial, 1, act, dim3(gx_a + gx_b, g.ny), dim3(256), stream, x, dy, y_out, rows, c, g.cw, g.rp, g.rows_per_block, mean, rstd, gamma, beta,
               act, rowscale, hw, part, sg);
        LAUNCH(bn_bwd_finalize2, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, gx_a, gx_a + gx_b, c, sums, dgamma, dbeta, accumulate);
        LAUNCH_ACT(bn_bwd_apply, 1, act, dim3(ex_a + ex_b, e.ny), dim3(256), stream, x, dy, y_out, (int)rows, c, e.cw, e.rp, e.rows_per_block, mean, rstd, gamma,
               beta, act, sums, inv_a, 1, rowscale, hw, dx, se, inv_b);
    }
    return check_launch("suhpe_bn_act_bwd_seg");
}

// RepVGG block tail (train mode): x / mean / rstd / gamma / beta are arrays of nb (2 or 3) pointers, branch order as the sum is
// written in the reference: 3x3 branch, 1x1 branch, identity BatchNorm; statistics [c] (rows_a = 0) or [2][c].
static bool bn3_fill(Bn3* b, const float* const* x, const float* const* mean, const float* const* rstd, const float* const* gamma,
                     const float* const* beta, int nb) {
    if (!(x && mean && rstd && gamma && (nb == 2 || nb == 3))) return false;
    b->nb = nb;
    for (int j = 0; j < 3; ++j) {
        const bool on = j < nb;
        b->x[j] = on ? x[j] : nullptr; b->mean[j] = on ? mean[j] : nullptr; b->rstd[j] = on ? rstd[j] : nullptr;
        b->gamma[j] = on ? gamma[j] : nullptr; b->beta[j] = (on && beta) ? beta[j] : nullptr;
        if (on && !(b->x[j] && b->mean[j] && b->rstd[j] && b->gamma[j] && (!beta || b->beta[j]))) return false;
    }
    return true;
}

int suhpe_repvgg_bn_fwd(const float* const* x, const float* const* mean, const float* const* rstd, const float* const* gamma,
                        const float* const* beta, int nb, int64_t rows, int64_t rows_a, int c, float* y, suhpe_stream_t stream) {
    Bn3 b;
    SUHPE_REQUIRE(bn3_fill(&b, x, mean, rstd, gamma, beta, nb) && beta && y && rows > 0 && rows < (1LL << 31) && c > 0 && rows_a >= 0 && rows_a < rows,
                  "suhpe_repvgg_bn_fwd: bad arguments");
    const int vec = (vec4_ok(c, b.x[0], b.x[1], y) && vec4_ok(c, b.x[2])) ? 4 : 1;
    const EwGeom g = ew_geom(rows, c, vec);
    int gx = g.gx;
    SegArgs sg{0, 0, 0};
    if (rows_a > 0) {
        const int gx_a = (int)ceil_div(rows_a, (int64_t)g.rows_per_block);
        gx = gx_a + (int)ceil_div(rows - rows_a, (int64_t)g.rows_per_block);
        sg = SegArgs{rows_a, gx_a, 0};
    }
    if (vec == 4) LAUNCH(bn3_act_fwd_kernel<4>, dim3(gx, g.ny), dim3(256), stream, b, (int)rows, c, g.cw, g.rp, g.rows_per_block, y, sg);
    else LAUNCH(bn3_act_fwd_kernel<1>, dim3(gx, g.ny), dim3(256), stream, b, (int)rows, c, g.cw, g.rp, g.rows_per_block, y, sg);
    return check_launch("suhpe_repvgg_bn_fwd");
}

// the statistics-emitting form: `stats` receives suhpe_repvgg_bn_fwd_stats_blocks() records of (S, Q, k, n) per channel ([record][c] float4), the first
// *first_records of them covering the first batch segment (all of them when rows_a == 0): what suhpe_bn_stats_from_partials[_seg] finalises
static void bn3_stats_grid(int64_t rows, int64_t rows_a, int c, EwGeom* g, int* gx, int* gx_a) {
    *g = ew_geom(rows, c, c % 4 == 0 ? 4 : 1);
    *gx = g->gx; *gx_a = g->gx;
    if (rows_a > 0) {
        *gx_a = (int)ceil_div(rows_a, (int64_t)g->rows_per_block);
        *gx = *gx_a + (int)ceil_div(rows - rows_a, (int64_t)g->rows_per_block);
    }
}

int suhpe_repvgg_bn_fwd_stats_blocks(int64_t rows, int64_t rows_a, int c, int32_t* first_records) {
    if (rows <= 0 || rows >= (1LL << 31) || c <= 0 || rows_a < 0 || rows_a >= rows) return 0;
    EwGeom g;
    int gx, gx_a;
    bn3_stats_grid(rows, rows_a, c, &g, &gx, &gx_a);
    if (first_records) *first_records = gx_a;
    return gx;
}

int suhpe_repvgg_bn_fwd_stats(const float* const* x, const float* const* mean, const float* const* rstd, const float* const* gamma,
                              const float* const* beta, int nb, int64_t rows, int64_t rows_a, int c, float* y, float* stats, suhpe_stream_t stream) {
    Bn3 b;
    SUHPE_REQUIRE(bn3_fill(&b, x, mean, rstd, gamma, beta, nb) && beta && y && stats && rows > 0 && rows < (1LL << 31) && c > 0 && rows_a >= 0 &&
                      rows_a < rows, "suhpe_repvgg_bn_fwd_stats: bad arguments");
    const int vec = c % 4 == 0 ? 4 : 1;
    SUHPE_REQUIRE(vec == 1 || (vec4_ok(c, b.x[0], b.x[1], y) && vec4_ok(c, b.x[2]) && (reinterpret_cast<uintptr_t>(stats) & 15) == 0), "suhpe_repvgg_bn_fwd_stats: 16-byte aligned tensors");
    EwGeom g;
    int gx, gx_a;
    bn3_stats_grid(rows, rows_a, c, &g, &gx, &gx_a);
    SegArgs sg{0, 0, 0};
    if (rows_a > 0) sg = SegArgs{rows_a, gx_a, 0};
    if (vec == 4) LAUNCH((bn3_act_fwd_kernel<4, true>), dim3(gx, g.ny), dim3(256), stream, b, (int)rows, c, g.cw, g.rp, g.rows_per_block, y, sg, stats);
    else LAUNCH((bn3_act_fwd_kernel<1, true>), dim3(gx, g.ny), dim3(256), stream, b, (int)rows, c, g.cw, g.rp, g.rows_per_block, y, sg, stats);
    return check_launch("suhpe_repvgg_bn_fwd_stats");
}

// beta != nullptr: the ReLU mask is recomputed from the branch inputs (y is not read and may be null); beta == nullptr: the mask is y > 0
static int repvgg_bn_bwd_impl(const float* const* x, const float* const* mean, const float* const* rstd, const float* const* gamma,
                              const float* const* beta, int nb, const float* dout, const float* y, int64_t rows, int64_t rows_a, int c,
                              float* const* dx, float* const* dgamma, float* const* dbeta, void* workspace, suhpe_stream_t stream) {
    Bn3 b;
    const bool remask = beta != nullptr;
    SUHPE_REQUIRE(bn3_fill(&b, x, mean, rstd, gamma, beta, nb) && dout && (y || remask) && dx && dgamma && dbeta && workspace && rows > 0 &&
                      rows < (1LL << 31) && c > 0 && rows_a >= 0 && rows_a < rows,
                  "suhpe_repvgg_bn_bwd: bad arguments");
    Bn3Grads gr;
    Bn3Out out;
    gr.nb = nb;
    for (int j = 0; j < 3; ++j) {
        gr.dgamma[j] = j < nb ? dgamma[j] : nullptr; gr.dbeta[j] = j < nb ? dbeta[j] : nullptr; out.dx[j] = j < nb ? dx[j] : nullptr;
        SUHPE_REQUIRE(j >= nb || (gr.dgamma[j] && gr.dbeta[j] && out.dx[j]), "suhpe_repvgg_bn_bwd: null gradient pointer");
    }
    float* part = static_cast<float*>(workspace);
    float* sums = bn_sums_of(workspace, c);
    const int vec = (vec4_ok(c, b.x[0], b.x[1], dout) && vec4_ok(c, b.x[2], remask ? dout : y) && vec4_ok(c, out.dx[0], out.dx[1], out.dx[2])) ? 4 : 1;
    const ColGeom g = col_geom(rows, c, vec);
    const EwGeom e = ew_geom(rows, c, vec);
    int gx_a = g.gx, gx = g.gx, ex = e.gx;
    SegArgs sg{0, 0, 0}, se{0, 0, 0};
    float inv_a = 1.0f / (float)rows, inv_b = 0.f;
    if (rows_a > 0) {
        gx_a = (int)ceil_div(rows_a, g.rows_per_block);
        gx = gx_a + (int)ceil_div(rows - rows_a, g.rows_per_block);
        const int ex_a = (int)ceil_div(rows_a, (int64_t)e.rows_per_block);
        ex = ex_a + (int)ceil_div(rows - rows_a, (int64_t)e.rows_per_block);
        sg = SegArgs{rows_a, gx_a, 0};
        se = SegArgs{rows_a, ex_a, 0};
        inv_a = 1.0f / (float)rows_a;
        inv_b = 1.0f / (float)(rows - rows_a);
    }
#define BN3_BWD(VEC, RM)                                                                                                                             \
    do {                                                                                                                                             \
        LAUNCH((bn3_bwd_partial<VEC, RM>), dim3(gx, g.ny), dim3(256), stream, b, dout, y, rows, c, g.cw, g.rp, g.rows_per_block, part, sg);           \
        LAUNCH(bn3_bwd_finalize, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, gx_a, gx, c, sums, gr);                           \
        LAUNCH((bn3_bwd_apply<VEC, RM>), dim3(ex, e.ny), dim3(256), stream, b, dout, y, (int)rows, c, e.cw, e.rp, e.rows_per_block, sums, inv_a,      \
               inv_b, out, se);                                                                                                                      \
    } while (0)
    if (vec == 4) { if (remask) BN3_BWD(4, true); else BN3_BWD(4, false); }
    else { if (remask) BN3_BWD(1, true); else BN3_BWD(1, false); }
#undef BN3_BWD
    return check_launch("suhpe_repvgg_bn_bwd");
}

int suhpe_repvgg_bn_bwd(const float* const* x, const float* const* mean, const float* const* rstd, const float* const* gamma, int nb,
                        const float* dout, const float* y, int64_t rows, int64_t rows_a, int c, float* const* dx, float* const* dgamma,
                        float* const* dbeta, void* workspace, suhpe_stream_t stream) {
    SUHPE_REQUIRE(y, "suhpe_repvgg_bn_bwd: y is required (suhpe_repvgg_bn_bwd_remask recomputes the mask instead)");
    return repvgg_bn_bwd_impl(x, mean, rstd, gamma, nullptr, nb, dout, y, rows, rows_a, c, dx, dgamma, dbeta, workspace, stream);
}

int suhpe_repvgg_bn_bwd_remask(const float* const* x, const float* const* mean, const float* const* rstd, const float* const* gamma,
                               const float* const* beta, int nb, const float* dout, int64_t rows, int64_t rows_a, int c, float* const* dx,
                               float* const* dgamma, float* const* dbeta, void* workspace, suhpe_stream_t stream) {
    SUHPE_REQUIRE(beta, "suhpe_repvgg_bn_bwd_remask: beta is required");
    return repvgg_bn_bwd_impl(x, mean, rstd, gamma, beta, nb, dout, nullptr, rows, rows_a, c, dx, dgamma, dbeta, workspace, stream);
}

int suhpe_layernorm_fwd(const float* x, int64_t rows, int c, float eps, const float* gamma, const float* beta, float* y,
                        float* mean, float* rstd, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && y && mean && rstd && rows > 0 && c > 0, "suhpe_layernorm_fwd: bad arguments");
    const dim3 grid((unsigned)ceil_div(rows, 4));
    if (vec4_ok(c, x, y) && (!gamma || vec4_ok(c, gamma)) && (!beta || vec4_ok(c, beta)) && c <= 768) {
        if (c <= 256) LAUNCH(layernorm_fwd_v4<1>, grid, dim3(256), stream, x, rows, c, eps, gamma, beta, y, mean, rstd);
        else if (c <= 512) LAUNCH(layernorm_fwd_v4<2>, grid, dim3(256), stream, x, rows, c, eps, gamma, beta, y, mean, rstd);
        else LAUNCH(layernorm_fwd_v4<3>, grid, dim3(256), stream, x, rows, c, eps, gamma, beta, y, mean, rstd);
        return check_launch("suhpe_layernorm_fwd");
    }
    LAUNCH(layernorm_fwd_kernel, grid, dim3(256), stream, x, rows, c, eps, gamma, beta, y, mean, rstd);
    return check_launch("suhpe_layernorm_fwd");
}

int suhpe_layernorm_bwd(const float* x, const float* dy, int64_t rows, int c, const float* mean, const float* rstd,
                        const float* gamma, float* dx, float* dgamma, float* dbeta, int accumulate, void* workspace,
                        suhpe_stream_t stream) {
    return suhpe_layernorm_bwd_skip(x, dy, rows, c, mean, rstd, gamma, nullptr, dx, dgamma, dbeta, accumulate, workspace, stream);
}

int suhpe_layernorm_bwd_skip(const float* x, const float* dy, int64_t rows, int c, const float* mean, const float* rstd, const float* gamma,
                             const float* skip, float* dx, float* dgamma, float* dbeta, int accumulate, void* workspace, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && dy && dx && mean && rstd && workspace && rows > 0 && c > 0, "suhpe_layernorm_bwd: bad arguments");
    const dim3 grid((unsigned)ceil_div(rows, 4));
    if (vec4_ok(c, x, dy) && vec4_ok(c, dx) && (!gamma || vec4_ok(c, gamma)) && (!skip || vec4_ok(c, skip)) && c <= 768) {
        if (c <= 256) LAUNCH(layernorm_bwd_dx_v4<1>, grid, dim3(256), stream, x, dy, rows, c, mean, rstd, gamma, skip, dx);
        else if (c <= 512) LAUNCH(layernorm_bwd_dx_v4<2>, grid, dim3(256), stream, x, dy, rows, c, mean, rstd, gamma, skip, dx);
        else LAUNCH(layernorm_bwd_dx_v4<3>, grid, dim3(256), stream, x, dy, rows, c, mean, rstd, gamma, skip, dx);
    } else {
        LAUNCH(layernorm_bwd_dx_kernel, grid, dim3(256), stream, x, dy, rows, c, mean, rstd, gamma, skip, dx);
    }
    if (dgamma || dbeta) {
        float* part = static_cast<float*>(workspace);
        float* sums = bn_sums_of(workspace, c);
        if (vec4_ok(c, x, dy)) {
            const ColGeom g = col_geom(rows, c, 4);
            LAUNCH(layernorm_bwd_partial<4>, dim3(g.gx, g.ny), dim3(256), stream, x, dy, rows, c, g.cw, g.rp, g.rows_per_block, mean, rstd, part);
            LAUNCH(bn_bwd_finalize, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, g.gx, c, sums, dgamma, dbeta, accumulate);
        } else {
            const ColGeom g = col_geom(rows, c, 1);
            LAUNCH(layernorm_bwd_partial<1>, dim3(g.gx, g.ny), dim3(256), stream, x, dy, rows, c, g.cw, g.rp, g.rows_per_block, mean, rstd, part);
            LAUNCH(bn_bwd_finalize, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, g.gx, c, sums, dgamma, dbeta, accumulate);
        }
    }
    return check_launch("suhpe_layernorm_bwd");
}

int suhpe_act_fwd(const float* x, int64_t rows, int c, const float* bias, int act, float* y, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && y && rows > 0 && c > 0, "suhpe_act_fwd: bad arguments");
    SUHPE_REQUIRE(rows < (1LL << 31), "suhpe_act_fwd: too many rows");
    if (vec4_ok(c, x, y)) { const EwGeom g = ew_geom(rows, c, 4); LAUNCH(act_fwd_kernel<4>, dim3(g.gx, g.ny), dim3(256), stream, x, (int)rows, c, g.cw, g.rp, g.rows_per_block, bias, act, y); }
    else { const EwGeom g = ew_geom(rows, c, 1); LAUNCH(act_fwd_kernel<1>, dim3(g.gx, g.ny), dim3(256), stream, x, (int)rows, c, g.cw, g.rp, g.rows_per_block, bias, act, y); }
    return check_launch("suhpe_act_fwd");
}

int suhpe_act_bwd(const float* x, const float* dy, int64_t rows, int c, const float* bias, int act, float* dx, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && dy && dx && rows > 0 && c > 0, "suhpe_act_bwd: bad arguments");
    SUHPE_REQUIRE(rows < (1LL << 31), "suhpe_act_bwd: too many rows");
    if (vec4_ok(c, x, dy, dx)) { const EwGeom g = ew_geom(rows, c, 4); LAUNCH(act_bwd_kernel<4>, dim3(g.gx, g.ny), dim3(256), stream, x, dy, (int)rows, c, g.cw, g.rp, g.rows_per_block, bias, act, dx); }
    else { const EwGeom g = ew_geom(rows, c, 1); LAUNCH(act_bwd_kernel<1>, dim3(g.gx, g.ny), dim3(256), stream, x, dy, (int)rows, c, g.cw, g.rp, g.rows_per_block, bias, act, dx); }
    return check_launch("suhpe_act_bwd");
}

static int pool_sum(const float* x, const float* w, int n, int hw, int c, float scale, float* y, suhpe_stream_t stream, const char* who) {
    if (!(x && y && n > 0 && hw > 0 && c > 0)) { set_error("%s: bad arguments", who); return SUHPE_EINVAL; }
    const int vec = vec4_ok(c, x, w, y) ? 4 : 1;
    const int cv = c / vec, cw = cv < 32 ? cv : 32, ny = (cv + cw - 1) / cw, rp = 256 / cw;
    if (vec == 4) LAUNCH(pool_sum_kernel<4>, dim3(n, ny), dim3(256), stream, x, w, hw, c, cw, rp, scale, y);
    else LAUNCH(pool_sum_kernel<1>, dim3(n, ny), dim3(256), stream, x, w, hw, c, cw, rp, scale, y);
    return check_launch(who);
}

static int per_sample_scale(const float* x, const float* s, const float* dpool, int n, int hw, int c, float inv_hw, int mode, float* y,
                            suhpe_stream_t stream, const char* who) {
    const bool v4 = vec4_ok(c, x, s, y) && vec4_ok(c, dpool);
    const EwGeom g = ew_geom(hw, c, v4 ? 4 : 1);
    if (v4) LAUNCH(per_sample_scale_kernel<4>, dim3(g.gx, g.ny, n), dim3(256), stream, x, s, dpool, hw, c, g.cw, g.rp, g.rows_per_block, inv_hw, mode, y);
    else LAUNCH(per_sample_scale_kernel<1>, dim3(g.gx, g.ny, n), dim3(256), stream, x, s, dpool, hw, c, g.cw, g.rp, g.rows_per_block, inv_hw, mode, y);
    return check_launch(who);
}

int suhpe_maxpool_fwd(const float* x, int n, int h, int w, int c, int k, int stride, int pad, float* y, uint8_t* idx, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && y && idx && n > 0 && h > 0 && w > 0 && c > 0 && k > 0 && k <= 15 && stride > 0 && pad >= 0 && 2 * pad <= k, "suhpe_maxpool_fwd: bad arguments");
    const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
    SUHPE_REQUIRE(oh > 0 && ow > 0, "suhpe_maxpool_fwd: empty output");
    const int vec = vec4_ok(c, x, y) ? 4 : 1;
    const int64_t total = (int64_t)n * oh * ow * (c / vec);
    if (vec == 4) LAUNCH(maxpool_fwd_kernel<4>, dim3(grid_for(total, 256)), dim3(256), stream, x, n, h, w, c, k, stride, pad, oh, ow, y, idx);
    else LAUNCH(maxpool_fwd_kernel<1>, dim3(grid_for(total, 256)), dim3(256), stream, x, n, h, w, c, k, stride, pad, oh, ow, y, idx);
    return check_launch("suhpe_maxpool_fwd");
}

int suhpe_maxpool_bwd(const float* dy, const uint8_t* idx, int n, int h, int w, int c, int k, int stride, int pad, float* dx, suhpe_stream_t stream) {
    SUHPE_REQUIRE(dy && dx && idx && n > 0 && h > 0 && w > 0 && c > 0 && k > 0 && k <= 15 && stride > 0 && pad >= 0 && 2 * pad <= k, "suhpe_maxpool_bwd: bad arguments");
    const int oh = (h + 2 * pad - k) / stride + 1, ow = (w + 2 * pad - k) / stride + 1;
    const int vec = vec4_ok(c, dy, dx) ? 4 : 1;
    const int64_t total = (int64_t)n * h * w * (c / vec);
    if (vec == 4) LAUNCH(maxpool_bwd_kernel<4>, dim3(grid_for(total, 256)), dim3(256), stream, dy, idx, n, h, w, c, k, stride, pad, oh, ow, dx);
    else LAUNCH(maxpool_bwd_kernel<1>, dim3(grid_for(total, 256)), dim3(256), stream, dy, idx, n, h, w, c, k, stride, pad, oh, ow, dx);
    return check_launch("suhpe_maxpool_bwd");
}

int suhpe_avgpool_fwd(const float* x, int n, int hw, int c, float* y, suhpe_stream_t stream) {
    return pool_sum(x, nullptr, n, hw, c, 1.0f / (float)hw, y, stream, "suhpe_avgpool_fwd");
}

int suhpe_avgpool_bwd(const float* dy, int n, int hw, int c, float* dx, suhpe_stream_t stream) {
    SUHPE_REQUIRE(dy && dx && n > 0 && hw > 0 && c > 0, "suhpe_avgpool_bwd: bad arguments");
    return per_sample_scale(nullptr, dy, nullptr, n, hw, c, 1.0f / hw, 1, dx, stream, "suhpe_avgpool_bwd");
}

int suhpe_se_scale_fwd(const float* x, const float* s, int n, int hw, int c, float* y, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && s && y && n > 0 && hw > 0 && c > 0, "suhpe_se_scale_fwd: bad arguments");
    return per_sample_scale(x, s, nullptr, n, hw, c, 0.f, 0, y, stream, "suhpe_se_scale_fwd");
}

int suhpe_se_scale_bwd_s(const float* dy, const float* x, int n, int hw, int c, float* ds, suhpe_stream_t stream) {
    SUHPE_REQUIRE(dy && x && ds, "suhpe_se_scale_bwd_s: null pointer");
    return pool_sum(dy, x, n, hw, c, 1.0f, ds, stream, "suhpe_se_scale_bwd_s");
}

int suhpe_se_scale_bwd_x(const float* dy, const float* s, const float* dpool, int n, int hw, int c, float* dx, suhpe_stream_t stream) {
    SUHPE_REQUIRE(dy && s && dx && n > 0 && hw > 0 && c > 0, "suhpe_se_scale_bwd_x: bad arguments");
    return per_sample_scale(dy, s, dpool, n, hw, c, 1.0f / hw, 0, dx, stream, "suhpe_se_scale_bwd_x");
}

int suhpe_add_rowscale(const float* a, const float* b, const float* rowscale, int n, int64_t per, float* y, suhpe_stream_t stream) {
    SUHPE_REQUIRE(b && y && n > 0 && per > 0 && per % 4 == 0, "suhpe_add_rowscale: bad arguments (per must be a multiple of 4)");
    SUHPE_REQUIRE(vec4_ok(4, a, b, y), "suhpe_add_rowscale: pointers must be 16-byte aligned");
    SUHPE_REQUIRE(per / 4 < (1LL << 31), "suhpe_add_rowscale: sample too large");
    const int per_vec = (int)(per / 4);
    int gx = (per_vec + 256 * 4 - 1) / (256 * 4);
    if (gx > 64) gx = 64;
    LAUNCH(add_rowscale_kernel, dim3(gx, n), dim3(256), stream, a, b, rowscale, per_vec, y);
    return check_launch("suhpe_add_rowscale");
}

int suhpe_mul_rowscale(const float* x, const float* rowscale, int n, int64_t per, float* y, suhpe_stream_t stream) {
    return suhpe_add_rowscale(nullptr, x, rowscale, n, per, y, stream);
}

int suhpe_mul(const float* x, const float* m, int64_t count, float* y, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && m && y && count > 0 && vec4_ok(4, x, m, y), "suhpe_mul: bad arguments");
    LAUNCH(binary_kernel, dim3(grid_for(count / 4 + 1, 256)), dim3(256), stream, x, m, count, 1, y);
    return check_launch("suhpe_mul");
}

int suhpe_add(const float* a, const float* b, int64_t count, float* y, suhpe_stream_t stream) {
    SUHPE_REQUIRE(a && b && y && count > 0 && vec4_ok(4, a, b, y), "suhpe_add: bad arguments");
    LAUNCH(binary_kernel, dim3(grid_for(count / 4 + 1, 256)), dim3(256), stream, a, b, count, 0, y);
    return check_launch("suhpe_add");
}

int suhpe_add3(const float* a, const float* b, const float* c, int64_t count, float* y, suhpe_stream_t stream) {
    SUHPE_REQUIRE(a && b && c && y && count > 0 && vec4_ok(4, a, b, y) && vec4_ok(4, c), "suhpe_add3: bad arguments");
    LAUNCH(add3_kernel, dim3(grid_for(count / 4 + 1, 256)), dim3(256), stream, a, b, c, count, y);
    return check_launch("suhpe_add3");
}

int suhpe_dropout_mask(uint64_t* counter, uint64_t host_offset, int advance, float keep, int64_t count, float* mask, suhpe_stream_t stream) {
    SUHPE_REQUIRE(counter && mask && count > 0 && keep >= 0.f && keep <= 1.f, "suhpe_dropout_mask: bad arguments");
    const int64_t n4 = ceil_div(count, 4);
    LAUNCH(dropout_mask_kernel, dim3(grid_for(n4, 256)), dim3(256), stream, counter, host_offset, keep, count, mask);
    if (advance) LAUNCH(advance_counter_kernel, dim3(1), dim3(1), stream, counter, (uint64_t)n4);
    return check_launch("suhpe_dropout_mask");
}

int suhpe_dropout_mask_batch(const uint64_t* counter, uint64_t host_offset, const int64_t* rel, const float* keep, const int64_t* count,
                             const int64_t* out_off, int nreq, int64_t max_count, float* out, suhpe_stream_t stream) {
    SUHPE_REQUIRE(counter && rel && keep && count && out_off && out && nreq > 0 && max_count > 0, "suhpe_dropout_mask_batch: bad arguments");
    const int64_t n4 = ceil_div(max_count, 4);
    int64_t gx = ceil_div(n4, 256);
    if (gx > 64) gx = 64;
    LAUNCH(dropout_mask_batch_kernel, dim3((unsigned)gx, (unsigned)nreq), dim3(256), stream, counter, host_offset, rel, keep, count, out_off, out);
    return check_launch("suhpe_dropout_mask_batch");
}

int suhpe_weighted_sum(const float* x, const float* w, int64_t count, float scale, float* out, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && out && count >= 0, "suhpe_weighted_sum: bad arguments");
    LAUNCH(weighted_sum_kernel, dim3(1), dim3(256), stream, x, w, count, scale, out);
    return check_launch("suhpe_weighted_sum");
}

int suhpe_masked_mean(const float* x, const float* mask, int64_t count, float* out, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && mask && out && count >= 0, "suhpe_masked_mean: bad arguments");
    LAUNCH(masked_mean_kernel, dim3(1), dim3(256), stream, x, mask, count, out);
    return check_launch("suhpe_masked_mean");
}

int suhpe_scale_rows(const float* x, const float* g, int64_t rows, int c, float scale, float* y, suhpe_stream_t stream) {
    SUHPE_REQUIRE(x && y && rows >= 0 && c > 0, "suhpe_scale_rows: bad arguments");
    if (rows == 0) return SUHPE_OK;
    LAUNCH(scale_rows_kernel, dim3((unsigned)ceil_div(rows * c, 256)), dim3(256), stream, x, g, rows, c, scale, y);
    return check_launch("suhpe_scale_rows");
}

int suhpe_broadcast_scale(const float* src, float scale, int64_t count, float* out, suhpe_stream_t stream) {
    SUHPE_REQUIRE(src && out && count >= 0, "suhpe_broadcast_scale: bad arguments");
    if (count == 0) return SUHPE_OK;
    LAUNCH(broadcast_scale_kernel, dim3((unsigned)ceil_div(count, 256)), dim3(256), stream, src, scale, count, out);
    return check_launch("suhpe_broadcast_scale");
}

int suhpe_axpby(const float* a, float alpha, const float* b, float beta, int64_t count, float* out, suhpe_stream_t stream) {
    SUHPE_REQUIRE(a && out && count >= 0, "suhpe_axpby: bad arguments");
    if (count == 0) return SUHPE_OK;
    LAUNCH(axpby_kernel, dim3((unsigned)ceil_div(count, 256)), dim3(256), stream, a, alpha, b, beta, count, out);
    return check_launch("suhpe_axpby");
}

int suhpe_bias_grad(const float* dy, int64_t rows, int c, float* db, int accumulate, void* workspace, suhpe_stream_t stream) {
    SUHPE_REQUIRE(dy && db && workspace && rows > 0 && c > 0, "suhpe_bias_grad: bad arguments");
    float* part = static_cast<float*>(workspace);
    if (vec4_ok(c, dy)) {
        const ColGeom g = col_geom(rows, c, 4);
        LAUNCH(colsum_partial<4>, dim3(g.gx, g.ny), dim3(256), stream, dy, rows, c, g.cw, g.rp, g.rows_per_block, part);
        LAUNCH(colsum_finalize, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, g.gx, c, db, accumulate);
    } else {
        const ColGeom g = col_geom(rows, c, 1);
        LAUNCH(colsum_partial<1>, dim3(g.gx, g.ny), dim3(256), stream, dy, rows, c, g.cw, g.rp, g.rows_per_block, part);
        LAUNCH(colsum_finalize, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinThreads), stream, part, g.gx, c, db, accumulate);
    }
    return check_launch("suhpe_bias_grad");
}

int suhpe_adam_step(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int64_t step, suhpe_stream_t stream) {
    SUHPE_REQUIRE(p && g && m && v && count > 0 && step >= 1, "suhpe_adam_step: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    LAUNCH(adam_kernel, dim3(grid_for(count, 256)), dim3(256), stream, p, g, m, v, count, lr, beta1, beta2, eps, weight_decay,
           (float)bc1, (float)sqrt(bc2));
    return check_launch("suhpe_adam_step");
}

int suhpe_ema_update(float* ema, const float* p, int64_t count, float alpha, suhpe_stream_t stream) {
    SUHPE_REQUIRE(ema && p && count > 0, "suhpe_ema_update: bad arguments");
    LAUNCH(ema_kernel, dim3(grid_for(count, 256)), dim3(256), stream, ema, p, count, alpha);
    return check_launch("suhpe_ema_update");
}

int suhpe_fill(float* p, int64_t count, float value, suhpe_stream_t stream) {
    SUHPE_REQUIRE(p && count >= 0, "suhpe_fill: bad arguments");
    if (count == 0) return SUHPE_OK;
    LAUNCH(fill_kernel, dim3(grid_for(count, 256)), dim3(256), stream, p, count, value);
    return check_launch("suhpe_fill");
}

}  // extern "C"
