// Host build of semiuhpe_amd/csrc/head_math.h for CPU unit tests (tests/test_head_math_host.py).
// NOT part of the product: the shipped path runs the same header inside the gfx950 kernels of head.hip.
#include "../../semiuhpe_amd/csrc/head_math.h"

using namespace suhpe;

static void load9(const float* p, double* o) { for (int i = 0; i < 9; ++i) o[i] = p[i]; }

static void parts(const Svd3& d, double* L, double g[3]) {
    double acc[4] = {0, 0, 0, 0};
    quad_partial(d.s, 0, kQuadNodes, acc);
    quad_finish(acc, L, g);
}

extern "C" {

void host_svd(const float* A, int n, float* U, float* S, float* V, float* R) {
    for (int b = 0; b < n; ++b) {
        double a[9], r[9];
        load9(A + 9 * b, a);
        Svd3 d;
        svd3_proper(a, &d);
        svd3_rotation(d, r);
        for (int i = 0; i < 9; ++i) { U[9 * b + i] = (float)d.U[i]; V[9 * b + i] = (float)d.V[i]; R[9 * b + i] = (float)r[i]; }
        for (int i = 0; i < 3; ++i) S[3 * b + i] = (float)d.s[i];
    }
}

void host_nll(const float* A, const float* Rgt, int n, double overreg, float* loss, float* dA, float* Rest) {
    for (int b = 0; b < n; ++b) {
        double a[9], rg[9], r[9], L, g[3], l, da[9];
        load9(A + 9 * b, a);
        load9(Rgt + 9 * b, rg);
        Svd3 d;
        svd3_proper(a, &d);
        parts(d, &L, g);
        nll_from_parts(a, rg, d, L, g, overreg, &l, da);
        svd3_rotation(d, r);
        loss[b] = (float)l;
        for (int i = 0; i < 9; ++i) { dA[9 * b + i] = (float)da[i]; Rest[9 * b + i] = (float)r[i]; }
    }
}

void host_entropy(const float* A, int n, float* H) {
    for (int b = 0; b < n; ++b) {
        double a[9], L, g[3];
        load9(A + 9 * b, a);
        Svd3 d;
        svd3_proper(a, &d);
        parts(d, &L, g);
        H[b] = (float)entropy_from_parts(d, L, g);
    }
}

void host_ce(const float* A1, const float* A2, int n, float* ce, float* dA2) {
    for (int b = 0; b < n; ++b) {
        double a1[9], a2[9], L1, g1[3], L2, g2[3], c, da[9];
        load9(A1 + 9 * b, a1);
        load9(A2 + 9 * b, a2);
        Svd3 t, s;
        svd3_proper(a1, &t);
        svd3_proper(a2, &s);
        parts(t, &L1, g1);
        parts(s, &L2, g2);
        ce_from_parts(t, g1, s, L2, g2, &c, da);
        ce[b] = (float)c;
        for (int i = 0; i < 9; ++i) dA2[9 * b + i] = (float)da[i];
    }
}

void host_euler(const float* R, int n, int full_range, float* e) {
    for (int b = 0; b < n; ++b) {
        double r[9], o[3];
        load9(R + 9 * b, r);
        euler_from_rot(r, full_range, o);
        for (int i = 0; i < 3; ++i) e[3 * b + i] = (float)o[i];
    }
}

void host_geodesic_deg(const float* R1, const float* R2, int n, float* out) {
    for (int b = 0; b < n; ++b) {
        double r1[9], r2[9];
        load9(R1 + 9 * b, r1);
        load9(R2 + 9 * b, r2);
        out[b] = (float)(geodesic_rad(r1, r2) * 57.29577951308232);
    }
}
}
