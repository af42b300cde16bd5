// Scalar math of the matrix-Fisher rotation head, written once for host and device.
//
// Compiled by hipcc into the gfx950 kernels of head.hip, and by g++ into the host-side unit-test
// library (tests/host_build) so the math is checked against the reference's golden vectors on CPU.
// All arithmetic is fp64 internally (MI355X has full-rate-ish fp64 VALU; the head is O(100) samples),
// inputs/outputs are fp32 like the reference.
//
// Reference behaviour restated here (file:line in /root/reference):
//   src/fisher/torch_norm_factor.py:4-92     log normaliser by 512-node trapezoid + its gradient
//   src/fisher/fisher_utils.py:14-99         NLL, nearest rotation, entropy, cross entropy
//   src/fisher/between_bingham_fisher.py:63-152, src/fisher/bingham_utils.py:5-73
// using the quaternion-free identities of SURVEY.md Appendix C.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define SUHPE_HD __host__ __device__ __forceinline__
#else
#define SUHPE_HD inline
#endif

namespace suhpe {

constexpr int kQuadNodes = 512;   // torch_norm_factor.py:69

// exp(-|x|) I0(x), Abramowitz & Stegun 9.8.1 / 9.8.2 (torch_norm_factor.py:10-19)
SUHPE_HD double bessel_i0e(double x) {
    const double ax = fabs(x);
    if (ax <= 3.75) {
        const double t = (ax / 3.75) * (ax / 3.75);
        const double p = 1.0 + t * (3.5156229 + t * (3.0899424 + t * (1.2067492 + t * (0.2659732 +
                         t * (0.360768e-1 + t * 0.45813e-2)))));
        return p * exp(-ax);
    }
    const double u = 3.75 / ax;
    const double p = 0.39894228 + u * (0.1328592e-1 + u * (0.225319e-2 + u * (-0.157565e-2 +
                     u * (0.916281e-2 + u * (-0.2057706e-1 + u * (0.2635537e-1 + u * (-0.1647633e-1 +
                     u * 0.392377e-2)))))));
    return p / sqrt(ax);
}

// Partial trapezoid sums over nodes [j0, j1) of the four integrands needed by value + gradient:
//   acc[0] += w f(x; s0,s1,s2)            (integrand_CF  :33-47)
//   acc[1+i] += w x f(x; rot_i(s))        (integrand_Cdiff :50-63, cyclic rotation i of s)
// with x_j = -1 + 2j/(N-1), w = 2/(N-1) (half at both ends).
SUHPE_HD double quad_f(double x, double a, double hi, double lo) {
    return bessel_i0e(0.5 * (hi - lo) * (1.0 - x)) * bessel_i0e(0.5 * (hi + lo) * (1.0 + x)) *
           exp((a + lo) * (x - 1.0));
}

SUHPE_HD void quad_partial(const double s[3], int j0, int j1, double acc[4]) {
    const double h = 2.0 / (kQuadNodes - 1);
    for (int j = j0; j < j1; ++j) {
        const double x = j * h - 1.0;
        const double w = (j == 0 || j == kQuadNodes - 1) ? 0.5 * h : h;
        acc[0] += w * quad_f(x, s[0], s[1], s[2]);
        for (int i = 0; i < 3; ++i) {
            const double a = s[i], b = s[(i + 1) % 3], c = s[(i + 2) % 3];
            const double hi = b > c ? b : c, lo = b > c ? c : b;
            acc[1 + i] += w * x * quad_f(x, a, hi, lo);
        }
    }
}

// From the four full integrals: L = log(1/2 I0) (= logC_F - sum s), g_i = dlogC_F/ds_i.
SUHPE_HD void quad_finish(const double acc[4], double* L, double g[3]) {
    const double factor = 0.5 * acc[0];
    *L = log(factor);
    for (int i = 0; i < 3; ++i) g[i] = 0.5 * acc[1 + i] / factor;
}

struct Svd3 {
    double U[9], V[9], s[3];   // row-major 3x3; proper: det U = det V = +1, s0>=s1>=|s2|, s2 signed
};

SUHPE_HD void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// Proper SVD of a 3x3 by one-sided (Hestenes) Jacobi on the columns of A:  A V = G, columns of G
// orthogonal.  Equivalent to between_bingham_fisher.py:63-82 (torch.svd + determinant fixes).
SUHPE_HD void svd3_proper(const double A[9], Svd3* out) {
    double G[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; ++i) G[i] = A[i];
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0;
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            double al = 0, be = 0, ga = 0;
            for (int r = 0; r < 3; ++r) {
                al += G[3 * r + p] * G[3 * r + p];
                be += G[3 * r + q] * G[3 * r + q];
                ga += G[3 * r + p] * G[3 * r + q];
            }
            const double lim = 1e-17 * sqrt(al * be);
            if (fabs(ga) > lim && ga != 0.0) {
                off += fabs(ga) / sqrt(al * be);
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int r = 0; r < 3; ++r) {
                    const double gp = G[3 * r + p], gq = G[3 * r + q];
                    G[3 * r + p] = c * gp - sn * gq;
                    G[3 * r + q] = sn * gp + c * gq;
                    const double vp = V[3 * r + p], vq = V[3 * r + q];
                    V[3 * r + p] = c * vp - sn * vq;
                    V[3 * r + q] = sn * vp + c * vq;
                }
            }
        }
        if (off < 1e-15) break;
    }
    double n2[3];
    for (int c = 0; c < 3; ++c) n2[c] = G[c] * G[c] + G[3 + c] * G[3 + c] + G[6 + c] * G[6 + c];
    int o0 = 0, o1 = 1, o2 = 2, tmp;     // sort column order by norm, descending
    if (n2[o0] < n2[o1]) { tmp = o0; o0 = o1; o1 = tmp; }
    if (n2[o1] < n2[o2]) { tmp = o1; o1 = o2; o2 = tmp; }
    if (n2[o0] < n2[o1]) { tmp = o0; o0 = o1; o1 = tmp; }
    const int ord[3] = {o0, o1, o2};
    double v[3][3], g[3][3];
    for (int k = 0; k < 3; ++k)
        for (int r = 0; r < 3; ++r) { v[k][r] = V[3 * r + ord[k]]; g[k][r] = G[3 * r + ord[k]]; }
    // make V proper: v2 = v0 x v1 (flip v2/g2 together if the permutation made det V = -1)
    double vx[3];
    cross3(v[0], v[1], vx);
    if (vx[0] * v[2][0] + vx[1] * v[2][1] + vx[2] * v[2][2] < 0)
        for (int r = 0; r < 3; ++r) { v[2][r] = -v[2][r]; g[2][r] = -g[2][r]; }
    double u[3][3];
    const double s0 = sqrt(n2[o0]), s1 = sqrt(n2[o1]);
    if (s0 > 0) { for (int r = 0; r < 3; ++r) u[0][r] = g[0][r] / s0; }
    else { u[0][0] = 1; u[0][1] = 0; u[0][2] = 0; }
    if (s1 > 1e-300 && s1 > 1e-14 * s0) {
        // re-orthogonalise against u0 (exact up to rounding already)
        double d = 0;
        for (int r = 0; r < 3; ++r) { u[1][r] = g[1][r] / s1; d += u[1][r] * u[0][r]; }
        double nn = 0;
        for (int r = 0; r < 3; ++r) { u[1][r] -= d * u[0][r]; nn += u[1][r] * u[1][r]; }
        nn = sqrt(nn);
        for (int r = 0; r < 3; ++r) u[1][r] /= nn;
    } else {
        // rank <= 1: any unit vector orthogonal to u0
        int k = 0;
        if (fabs(u[0][1]) < fabs(u[0][k])) k = 1;
        if (fabs(u[0][2]) < fabs(u[0][k])) k = 2;
        double e[3] = {0, 0, 0};
        e[k] = 1;
        cross3(u[0], e, u[1]);
        const double nn = sqrt(u[1][0] * u[1][0] + u[1][1] * u[1][1] + u[1][2] * u[1][2]);
        for (int r = 0; r < 3; ++r) u[1][r] /= nn;
    }
    cross3(u[0], u[1], u[2]);            // det U = +1 by construction
    out->s[0] = s0;
    out->s[1] = s1;
    out->s[2] = u[2][0] * g[2][0] + u[2][1] * g[2][1] + u[2][2] * g[2][2];   // signed: u2 . (A v2)
    for (int k = 0; k < 3; ++k)
        for (int r = 0; r < 3; ++r) { out->U[3 * r + k] = u[k][r]; out->V[3 * r + k] = v[k][r]; }
}

// R = U V^T (fisher_utils.py:39-48 with the proper factors)
SUHPE_HD void svd3_rotation(const Svd3& d, double R[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0;
            for (int k = 0; k < 3; ++k) a += d.U[3 * i + k] * d.V[3 * j + k];
            R[3 * i + j] = a;
        }
}

// NLL value and gradient (fisher_utils.py:21-36):  loss = -<A,R> + overreg (L + sum s),
// dA = -R + overreg U diag(g) V^T.
SUHPE_HD void nll_from_parts(const double A[9], const double Rgt[9], const Svd3& d, double L, const double g[3],
                             double overreg, double* loss, double dA[9]) {
    double dot = 0;
    for (int i = 0; i < 9; ++i) dot += A[i] * Rgt[i];
    *loss = -dot + overreg * (L + d.s[0] + d.s[1] + d.s[2]);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double a = 0;
            for (int k = 0; k < 3; ++k) a += d.U[3 * i + k] * g[k] * d.V[3 * j + k];
            dA[3 * i + j] = -Rgt[3 * i + j] + overreg * a;
        }
}

// Entropy H = logC_F - sum s_i g_i (fisher_utils.py:70-81)
SUHPE_HD double entropy_from_parts(const Svd3& d, double L, const double g[3]) {
    return L + d.s[0] + d.s[1] + d.s[2] - (d.s[0] * g[0] + d.s[1] * g[1] + d.s[2] * g[2]);
}

// Cross entropy teacher(1) -> student(2) as written in bingham_utils.py:5-32 (row-index quirk kept),
// value and gradient w.r.t. the student's A.  g1: teacher's dlogC/ds; (L2,g2): student's.
//   CE = L2 - sum_i lam_i [ beta_i + sum_k (alpha_ik - beta_i) w_k ]
SUHPE_HD void ce_from_parts(const Svd3& t, const double g1[3], const Svd3& st, double L2, const double g2[3],
                            double* ce, double dA2[9]) {
    const double w[3] = {0.25 * (g1[0] - g1[1] - g1[2]) + 0.25, 0.25 * (-g1[0] + g1[1] - g1[2]) + 0.25,
                         0.25 * (-g1[0] - g1[1] + g1[2]) + 0.25};
    const double W = w[0] + w[1] + w[2];
    double P1[9], P2[9], M1[3][9], M2[3][9];
    svd3_rotation(t, P1);
    svd3_rotation(st, P2);
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                M1[k][3 * i + j] = 2.0 * t.U[3 * i + k] * t.V[3 * j + k] - P1[3 * i + j];
                M2[k][3 * i + j] = 2.0 * st.U[3 * i + k] * st.V[3 * j + k] - P2[3 * i + j];
            }
    const double lam[3] = {-2.0 * (st.s[1] + st.s[2]), -2.0 * (st.s[0] + st.s[2]), -2.0 * (st.s[0] + st.s[1])};
    double inner[3], second = 0;
    for (int i = 0; i < 3; ++i) {
        double b = 0;
        for (int e = 0; e < 9; ++e) b += P1[e] * M2[i][e];
        b = 0.25 * (1.0 + b);
        double acc = b * (1.0 - W);
        for (int k = 0; k < 3; ++k) {
            double a = 0;
            for (int e = 0; e < 9; ++e) a += M1[i][e] * M2[k][e];
            acc += 0.25 * (1.0 + a) * w[k];
        }
        inner[i] = acc;
        second += lam[i] * acc;
    }
    *ce = L2 - second;
    // ---- gradient.  H_k = d(second)/dM2_k = [(1-W) lam_k P1 + w_k Q]/4,  Q = sum_i lam_i M1_i
    double Q[9];
    for (int e = 0; e < 9; ++e) Q[e] = lam[0] * M1[0][e] + lam[1] * M1[1][e] + lam[2] * M1[2][e];
    double H[3][9], Hs[9];
    for (int e = 0; e < 9; ++e) {
        Hs[e] = 0;
        for (int k = 0; k < 3; ++k) { H[k][e] = 0.25 * ((1.0 - W) * lam[k] * P1[e] + w[k] * Q[e]); Hs[e] += H[k][e]; }
    }
    // d f / d(u_k v_k^T) = G_k = -(2 H_k... ) : f = L2 - second, M2_k = 2 N_k - sum_m N_m
    //   => G_k = -(2 H_k - Hs) = Hs - 2 H_k ... with the minus sign of "- second": G_k = -(2H_k - Hs)
    // T^k = U^T G_k V
    double T[3][9];
    for (int k = 0; k < 3; ++k) {
        double Gk[9], tmp[9];
        for (int e = 0; e < 9; ++e) Gk[e] = Hs[e] - 2.0 * H[k][e];
        for (int a = 0; a < 3; ++a)
            for (int j = 0; j < 3; ++j) {
                double x = 0;
                for (int i = 0; i < 3; ++i) x += st.U[3 * i + a] * Gk[3 * i + j];
                tmp[3 * a + j] = x;
            }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double x = 0;
                for (int j = 0; j < 3; ++j) x += tmp[3 * a + j] * st.V[3 * j + b];
                T[k][3 * a + b] = x;
            }
    }
    double Pb[9];
    const double* s = st.s;
    Pb[0] = (g2[0] - 1.0) + 2.0 * (inner[1] + inner[2]);
    Pb[4] = (g2[1] - 1.0) + 2.0 * (inner[0] + inner[2]);
    Pb[8] = (g2[2] - 1.0) + 2.0 * (inner[0] + inner[1]);
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j) {
            const double D = 1.0 / (s[j] * s[j] - s[i] * s[i]);
            const double a = T[j][3 * i + j] - T[i][3 * j + i];     // X_ij - X_ji
            const double b = T[j][3 * j + i] - T[i][3 * i + j];     // Y_ij - Y_ji
            Pb[3 * i + j] = D * (a * s[j] + b * s[i]);
            Pb[3 * j + i] = D * (a * s[i] + b * s[j]);
        }
    // dA = U Pb V^T
    double tmp[9];
    for (int i = 0; i < 3; ++i)
        for (int b = 0; b < 3; ++b) {
            double x = 0;
            for (int a = 0; a < 3; ++a) x += st.U[3 * i + a] * Pb[3 * a + b];
            tmp[3 * i + b] = x;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double x = 0;
            for (int b = 0; b < 3; ++b) x += tmp[3 * i + b] * st.V[3 * j + b];
            dA2[3 * i + j] = x;
        }
}

// Euler angles [pitch(x), yaw(y), roll(z)] in radians (src/utils.py:232-260)
SUHPE_HD void euler_from_rot(const double R[9], int full_range, double e[3]) {
    double sy = sqrt(R[0] * R[0] + R[3] * R[3]);
    const int singular = sy < 1e-6;
    if (full_range && R[0] < 0) sy = -sy;
    e[1] = atan2(-R[6], sy);
    if (!singular) { e[0] = atan2(R[7], R[8]); e[2] = atan2(R[3], R[0]); }
    else { e[0] = atan2(-R[5], R[4]); e[2] = 0.0; }
}

// pytorch3d-0.7.2 so3_relative_angle with cos_bound = 1e-4 (acos linearly extrapolated outside the bound)
SUHPE_HD double geodesic_rad(const double R1[9], const double R2[9]) {
    double tr = 0;
    for (int i = 0; i < 9; ++i) tr += R1[i] * R2[i];
    const double c = 0.5 * (tr - 1.0), b = 1.0 - 1e-4;
    const double slope = -1.0 / sqrt(1.0 - b * b);
    if (c >= b) return (c - b) * slope + acos(b);
    if (c <= -b) return (c + b) * slope + acos(-b);
    return acos(c);
}

}  // namespace suhpe
