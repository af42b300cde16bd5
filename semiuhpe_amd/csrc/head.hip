// Matrix-Fisher rotation head on gfx950: one 64-lane wavefront per sample.
//
// The reference moves every head call to the CPU (torch.svd + 512-node quadrature, src/fisher/fisher_utils.py:
// 27,35,40,47) — 8-10 device->host->device round trips per SSL step.  Here each sample is one
// wavefront: the 3x3 proper SVD is computed redundantly by all lanes (wave-uniform, no divergence), the four
// 512-node trapezoid integrals are split 8 nodes per lane and combined with wave shuffles, lanes 0..8 write the
// 3x3 outputs.  Everything stays on the launch stream; nothing syncs.  Math: head_math.h (fp64 internally).
#include "common.h"
#include "head_math.h"

namespace suhpe {

__device__ __forceinline__ void load_mat(const float* p, double* o) {
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = (double)p[i];
}

// full-wave quadrature: value part L and gradient g of the log normaliser for proper singular values s
__device__ __forceinline__ void wave_quadrature(const double s[3], int lane, double* L, double g[3]) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    constexpr int per = kQuadNodes / kWave;   // 8
    quad_partial(s, lane * per, lane * per + per, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = wave_sum(acc[i]);
    quad_finish(acc, L, g);
}

enum { kNll = 0, kProject = 1, kEntropy = 2, kLogPdf = 3 };

template <int MODE>
__global__ __launch_bounds__(256) void head_single_kernel(const float* __restrict__ A, const float* __restrict__ R,
                                                          int n, float overreg, float* __restrict__ out0,
                                                          float* __restrict__ out1, float* __restrict__ out2) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;                      // wave-uniform
    double a[9];
    load_mat(A + 9 * (int64_t)b, a);
    Svd3 d;
    svd3_proper(a, &d);
    if (MODE == kProject) {
        double r[9];
        svd3_rotation(d, r);
        if (lane < 9) out0[9 * (int64_t)b + lane] = (float)r[lane];
        return;
    }
    double L, g[3];
    wave_quadrature(d.s, lane, &L, g);
    if (MODE == kEntropy) {
        if (lane == 0) out0[b] = (float)entropy_from_parts(d, L, g);
    } else if (MODE == kLogPdf) {
        double rg[9], dot = 0;
        load_mat(R + 9 * (int64_t)b, rg);
#pragma unroll
        for (int i = 0; i < 9; ++i) dot += a[i] * rg[i];
        if (lane == 0) out0[b] = (float)(dot - (L + d.s[0] + d.s[1] + d.s[2]));
    } else {   // kNll: out0 loss, out1 dA, out2 Rest
        double rg[9], loss, da[9];
        load_mat(R + 9 * (int64_t)b, rg);
        nll_from_parts(a, rg, d, L, g, (double)overreg, &loss, da);
        if (out0 && lane == 0) out0[b] = (float)loss;
        if (out1 && lane < 9) out1[9 * (int64_t)b + lane] = (float)da[lane];
        if (out2) {
            double r[9];
            svd3_rotation(d, r);
            if (lane < 9) out2[9 * (int64_t)b + lane] = (float)r[lane];
        }
    }
}

__global__ __launch_bounds__(256) void head_ce_kernel(const float* __restrict__ A1, const float* __restrict__ A2,
                                                      const float* __restrict__ mask, int n, float* __restrict__ ce,
                                                      float* __restrict__ dA2) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;
    if (mask && mask[b] == 0.0f) {           // wave-uniform; outputs exactly zero for unselected samples
        if (lane == 0) ce[b] = 0.0f;
        if (dA2 && lane < 9) dA2[9 * (int64_t)b + lane] = 0.0f;
        return;
    }
    double a1[9], a2[9];
    load_mat(A1 + 9 * (int64_t)b, a1);
    load_mat(A2 + 9 * (int64_t)b, a2);
    Svd3 t, s;
    svd3_proper(a1, &t);
    svd3_proper(a2, &s);
    double L1, g1[3], L2, g2[3];
    wave_quadrature(t.s, lane, &L1, g1);
    wave_quadrature(s.s, lane, &L2, g2);
    double c, da[9];
    ce_from_parts(t, g1, s, L2, g2, &c, da);
    if (lane == 0) ce[b] = (float)c;
    if (dA2 && lane < 9) dA2[9 * (int64_t)b + lane] = (float)da[lane];
}

// one thread per sample: tiny 3x3 products
__global__ void rot_adjust_kernel(const float* __restrict__ A, const float* __restrict__ aug, int n, int mode,
                                  float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    float a[9], g[9], o[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { a[i] = A[9 * (int64_t)b + i]; g[i] = aug[9 * (int64_t)b + i]; }
    if (mode == 0) {   // DAD3DHeads: aug * A   (agent.py:113-114)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) o[3 * i + j] = g[3 * i] * a[j] + g[3 * i + 1] * a[3 + j] + g[3 * i + 2] * a[6 + j];
    } else {           // 300WLP: (D aug D A^T)^T, D = diag(1,-1,-1)   (agent.py:115-120)
        const float dsg[3] = {1.f, -1.f, -1.f};
        float m[9];    // m = D * aug * D
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) m[3 * i + j] = dsg[i] * g[3 * i + j] * dsg[j];
        // out = (m A^T)^T = A m^T
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) o[3 * i + j] = a[3 * i] * m[3 * j] + a[3 * i + 1] * m[3 * j + 1] + a[3 * i + 2] * m[3 * j + 2];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) out[9 * (int64_t)b + i] = o[i];
}

__global__ void euler_kernel(const float* __restrict__ R, int n, int full_range, float* __restrict__ e) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    double r[9], o[3];
    load_mat(R + 9 * (int64_t)b, r);
    euler_from_rot(r, full_range, o);
    for (int i = 0; i < 3; ++i) e[3 * (int64_t)b + i] = (float)o[i];
}

__global__ void pose_err_kernel(const float* __restrict__ P, const float* __restrict__ G,
                                const float* __restrict__ gt_euler, int n, float* __restrict__ err) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    double p[9];
    load_mat(P + 9 * (int64_t)b, p);
    if (gt_euler) {
        double o[3], acc = 0;
        euler_from_rot(p, 0, o);
        for (int i = 0; i < 3; ++i)
            acc += fabs((double)((float)(o[i] * 180.0 / 3.141592653589793)) - (double)gt_euler[3 * (int64_t)b + i]);
        err[b] = (float)(acc / 3.0);
    } else {
        double g[9];
        load_mat(G + 9 * (int64_t)b, g);
        err[b] = (float)(geodesic_rad(p, g) * 57.29577951308232);
    }
}

__global__ void entropy_mask_kernel(const float* __restrict__ H, const float* __restrict__ thres, int n,
                                    float* __restrict__ mask, float* __restrict__ count) {
    // single block
    __shared__ float part[4];
    const float t = thres[0];
    float c = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float m = H[i] < t ? 1.f : 0.f;
        mask[i] = m;
        c += m;
    }
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0 && count) count[0] = part[0] + part[1] + part[2] + part[3];
}

// Rotation-Laplace NLL (src/laplace/rotation_laplace.py:24-173): one workgroup per sample, the 4608 grid rotations
// split over 256 threads in two passes (max, then the normaliser and the gradient sums), fp64 inside.
//   p_g = -sqrt(max(S - <A,X_g>, 1e-8)),  S = s1+s2+s3 (s3 signed),  logF = c + log(mean_g e^{p_g-c}/(-p_g))
//   nll = logF - p_x + log(-p_x);   dS/dA = U V^T (proper) =: Rhat,  dp/dA = (Rhat - X)/(2p) where not clamped
__device__ __forceinline__ double block_reduce(double v, double* sh, int op) {   // op 0 sum, 1 max; result to all threads
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o, 64); v = op ? fmax(v, t) : v + t; }
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return op ? fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3])) : (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void laplace_nll_kernel(const float* __restrict__ A, const float* __restrict__ R, const float* __restrict__ grids,
                                                          int ngrid, float* __restrict__ loss, float* __restrict__ dA, float* __restrict__ mode) {
    __shared__ double sh[4];
    const int b = blockIdx.x;
    double a[9], x[9], rh[9];
    load_mat(A + 9 * (int64_t)b, a);
    load_mat(R + 9 * (int64_t)b, x);
    Svd3 d;
    svd3_proper(a, &d);
    svd3_rotation(d, rh);
    const double S = d.s[0] + d.s[1] + d.s[2];
    const double eps = 1e-8;
    double mx = -1e300;
    for (int g = threadIdx.x; g < ngrid; g += 256) {
        double tr = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) tr += a[i] * (double)grids[9 * (int64_t)g + i];
        mx = fmax(mx, -sqrt(fmax(S - tr, eps)));
    }
    const double c = block_reduce(mx, sh, 1);
    double sw = 0, sc = 0, sx[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int g = threadIdx.x; g < ngrid; g += 256) {
        double xg[9], tr = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) { xg[i] = (double)grids[9 * (int64_t)g + i]; tr += a[i] * xg[i]; }
        const double q = S - tr;
        const double p = -sqrt(fmax(q, eps));
        const double w = exp(p - c) / (-p);
        sw += w;
        if (q > eps) {                                   // clamp_min passes no gradient below the floor
            const double coef = w * (1.0 - 1.0 / p) / (2.0 * p);
            sc += coef;
#pragma unroll
            for (int i = 0; i < 9; ++i) sx[i] += coef * xg[i];
        }
    }
    sw = block_reduce(sw, sh, 0);
    sc = block_reduce(sc, sh, 0);
#pragma unroll
    for (int i = 0; i < 9; ++i) sx[i] = block_reduce(sx[i], sh, 0);
    if (threadIdx.x == 0) {
        const double logF = c + log(sw / (double)ngrid);
        double trx = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) trx += a[i] * x[i];
        const double qx = S - trx, px = -sqrt(fmax(qx, eps));
        if (loss) loss[b] = (float)(logF - px + log(-px));
        const double cx = qx > eps ? (-1.0 + 1.0 / px) / (2.0 * px) : 0.0;
        for (int i = 0; i < 9; ++i) {
            if (dA) dA[9 * (int64_t)b + i] = (float)((rh[i] * sc - sx[i]) / sw + cx * (rh[i] - x[i]));
            if (mode) mode[9 * (int64_t)b + i] = (float)rh[i];
        }
    }
}

}  // namespace suhpe

using namespace suhpe;

extern "C" {

int suhpe_fisher_nll(const float* A, const float* R, int n, float overreg, float* loss, float* dA, float* Rest,
                     suhpe_stream_t stream) {
    SUHPE_REQUIRE(A && R && n >= 0, "suhpe_fisher_nll: null input or negative n");
    if (n == 0) return SUHPE_OK;
    hipLaunchKernelGGL(head_single_kernel<kNll>, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), A, R, n, overreg,
                       loss, dA, Rest);
    return check_launch("suhpe_fisher_nll");
}

int suhpe_fisher_svd_project(const float* A, int n, float* R, suhpe_stream_t stream) {
    SUHPE_REQUIRE(A && R && n >= 0, "suhpe_fisher_svd_project: null pointer or negative n");
    if (n == 0) return SUHPE_OK;
    hipLaunchKernelGGL(head_single_kernel<kProject>, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), A, nullptr, n,
                       0.f, R, nullptr, nullptr);
    return check_launch("suhpe_fisher_svd_project");
}

int suhpe_fisher_entropy(const float* A, int n, float* H, suhpe_stream_t stream) {
    SUHPE_REQUIRE(A && H && n >= 0, "suhpe_fisher_entropy: null pointer or negative n");
    if (n == 0) return SUHPE_OK;
    hipLaunchKernelGGL(head_single_kernel<kEntropy>, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), A, nullptr, n,
                       0.f, H, nullptr, nullptr);
    return check_launch("suhpe_fisher_entropy");
}

int suhpe_fisher_log_pdf(const float* A, const float* R, int n, float* logp, suhpe_stream_t stream) {
    SUHPE_REQUIRE(A && R && logp && n >= 0, "suhpe_fisher_log_pdf: null pointer or negative n");
    if (n == 0) return SUHPE_OK;
    hipLaunchKernelGGL(head_single_kernel<kLogPdf>, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), A, R, n, 0.f,
                       logp, nullptr, nullptr);
    return check_launch("suhpe_fisher_log_pdf");
}

int suhpe_fisher_ce(const float* A1, const float* A2, const float* mask, int n, float* ce, float* dA2,
                    suhpe_stream_t stream) {
    SUHPE_REQUIRE(A1 && A2 && ce && n >= 0, "suhpe_fisher_ce: null pointer or negative n");
    if (n == 0) return SUHPE_OK;
    hipLaunchKernelGGL(head_ce_kernel, dim3((n + 3) / 4), dim3(256), 0, as_stream(stream), A1, A2, mask, n, ce, dA2);
    return check_launch("suhpe_fisher_ce");
}

int suhpe_laplace_nll(const float* A, const float* R, const float* grids, int n, int ngrid, float* loss, float* dA, float* mode,
                      suhpe_stream_t stream) {
    SUHPE_REQUIRE(A && R && grids && n >= 0 && ngrid > 0, "suhpe_laplace_nll: null pointer or bad sizes");
    if (n == 0) return SUHPE_OK;
    hipLaunchKernelGGL(laplace_nll_kernel, dim3(n), dim3(256), 0, as_stream(stream), A, R, grids, ngrid, loss, dA, mode);
    return check_launch("suhpe_laplace_nll");
}

int suhpe_rot_adjust(const float* A, const float* aug, int n, int mode, float* out, suhpe_stream_t stream) {
    SUHPE_REQUIRE(A && aug && out && n >= 0, "suhpe_rot_adjust: null pointer or negative n");
    SUHPE_REQUIRE(mode == 0 || mode == 1, "suhpe_rot_adjust: mode must be 0 (DAD3DHeads) or 1 (300WLP)");
    if (n == 0) return SUHPE_OK;
    hipLaunchKernelGGL(rot_adjust_kernel, dim3((n + 63) / 64), dim3(64), 0, as_stream(stream), A, aug, n, mode, out);
    return check_launch("suhpe_rot_adjust");
}

int suhpe_euler_from_rot(const float* R, int n, int full_range, float* euler, suhpe_stream_t stream) {
    SUHPE_REQUIRE(R && euler && n >= 0, "suhpe_euler_from_rot: null pointer or negative n");
    if (n == 0) return SUHPE_OK;
    hipLaunchKernelGGL(euler_kernel, dim3((n + 63) / 64), dim3(64), 0, as_stream(stream), R, n, full_range, euler);
    return check_launch("suhpe_euler_from_rot");
}

int suhpe_pose_err_deg(const float* pred, const float* gt, const float* gt_euler_deg, int n, float* err,
                       suhpe_stream_t stream) {
    SUHPE_REQUIRE(pred && err && (gt || gt_euler_deg) && n >= 0, "suhpe_pose_err_deg: null pointer or negative n");
    if (n == 0) return SUHPE_OK;
    hipLaunchKernelGGL(pose_err_kernel, dim3((n + 63) / 64), dim3(64), 0, as_stream(stream), pred, gt, gt_euler_deg, n,
                       err);
    return check_launch("suhpe_pose_err_deg");
}

int suhpe_entropy_mask(const float* H, const float* thres, int n, float* mask, float* count, suhpe_stream_t stream) {
    SUHPE_REQUIRE(H && thres && mask && n >= 0, "suhpe_entropy_mask: null pointer or negative n");
    hipLaunchKernelGGL(entropy_mask_kernel, dim3(1), dim3(256), 0, as_stream(stream), H, thres, n, mask, count);
    return check_launch("suhpe_entropy_mask");
}

}  // extern "C"
