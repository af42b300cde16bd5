// AddressSanitizer / UndefinedBehaviorSanitizer pass over the scalar head math (semiuhpe_amd/csrc/head_math.h as built for the
// host by head_host.cpp): random, degenerate (zero, rank-one, repeated singular values, det < 0) and extreme-scale matrices
// through every entry point; g++ -fsanitize=address,undefined -fno-sanitize-recover=all (tests/test_head_math_host.py).
// GPU sanitizers are not available on the MI355X pool, so the kernels' shared math is sanitized on the CPU build.
#include "head_host.cpp"
#include <cmath>
#include <cstdio>
#include <vector>

int main() {
    const int n = 512;
    std::vector<float> A(n * 9), A1(n * 9), R(n * 9), U(n * 9), S(n * 3), V(n * 9), Ro(n * 9), loss(n), dA(n * 9), H(n), ce(n), dA2(n * 9);
    unsigned long long s = 99;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
    const float scales[] = {1e-6f, 1e-3f, 1.f, 30.f, 150.f};
    for (int b = 0; b < n; ++b) {
        const float sc = scales[b % 5];
        for (int i = 0; i < 9; ++i) { A[b * 9 + i] = rnd() * sc; A1[b * 9 + i] = rnd() * sc; }
        if (b % 16 == 0) for (int i = 0; i < 9; ++i) A[b * 9 + i] = 0.f;                                  // zero matrix
        if (b % 16 == 1) for (int i = 3; i < 9; ++i) A[b * 9 + i] = 0.f;                                  // rank one
        if (b % 16 == 2) for (int i = 0; i < 9; ++i) A[b * 9 + i] = (i % 4 == 0) ? sc : 0.f;              // repeated singular values
        if (b % 16 == 3) for (int i = 0; i < 9; ++i) A[b * 9 + i] = (i % 4 == 0) ? (i == 8 ? -sc : sc) : 0.f;   // det < 0
    }
    host_svd(A.data(), n, U.data(), S.data(), V.data(), Ro.data());
    for (int b = 0; b < n; ++b)
        for (int i = 0; i < 9; ++i) R[b * 9 + i] = Ro[b * 9 + i];                                        // proper rotations as ground truth
    host_nll(A1.data(), R.data(), n, 1.025, loss.data(), dA.data(), Ro.data());
    host_entropy(A1.data(), n, H.data());
    host_ce(A.data(), A1.data(), n, ce.data(), dA2.data());
    int bad = 0;
    for (int b = 0; b < n; ++b) bad += !std::isfinite(loss[b]) || !std::isfinite(H[b]);
    std::printf("HEAD_SANITIZE_OK non-finite %d\n", bad);
    return bad ? 2 : 0;
}
