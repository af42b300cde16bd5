// Laboratory for the HBM-bound BatchNorm kernels (round 3): how much of a streaming BN + SiLU kernel's time is VALU work
// (accurate expf / IEEE division) and how much is the row-loop structure (loads in flight, occupancy)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/proto/ew_lab.hip -o tools/proto/bin/ew_lab && tools/proto/bin/ew_lab
// Kernels are stand-alone copies of the shapes of bn_act_fwd_kernel / bn_act_pool_fwd_kernel / bn_bwd_partial / bn_bwd_apply (nn_elem.hip).
// MATH: 0 = expf + IEEE division (the library's act_apply / act_grad), 1 = v_exp_f32 + v_rcp_f32, 2 = no activation (pure BN).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MATH> __device__ __forceinline__ float silu(float u) {
    if (MATH == 0) return u / (1.f + expf(-u));
    if (MATH == 1) return u * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * u));
    return u;
}
template <int MATH> __device__ __forceinline__ float silu_grad(float u) {
    if (MATH == 0) { const float s = 1.f / (1.f + expf(-u)); return s * (1.f + u * (1.f - s)); }
    if (MATH == 1) { const float s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * u)); return s * (1.f + u * (1.f - s)); }
    return 1.f;
}

struct Geom { int cw, ny, rp, gx, rpb; };
typedef float nfloat4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(float4 v, float4* dst) {
    nfloat4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<nfloat4*>(dst));
}

// y = silu(x*sc + sh): thread -> (vector column, rows strided by rp), U rows of loads in flight
template <int MATH, int U, bool NT>
__global__ __launch_bounds__(256) void apply_fwd(const float* __restrict__ x, int rows, int c, Geom g, const float* __restrict__ scale,
                                                 const float* __restrict__ shift, float* __restrict__ y) {
    const int t = threadIdx.x, col = t % g.cw, rsub = t / g.cw, cv = c / 4, vc = blockIdx.y * g.cw + col;
    if (rsub >= g.rp || vc >= cv) return;
    const int r0 = blockIdx.x * g.rpb;
    int r1 = r0 + g.rpb;
    if (r1 > rows) r1 = rows;
    const float4 sc = reinterpret_cast<const float4*>(scale)[vc], sh = reinterpret_cast<const float4*>(shift)[vc];
    for (int r = r0 + rsub; r < r1; r += U * g.rp) {
        float4 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * g.rp < r1 ? r + u * g.rp : r1 - 1;
            xv[u] = reinterpret_cast<const float4*>(x)[(int64_t)rr * cv + vc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * g.rp;
            if (rr >= r1) break;
            float4 o;
            o.x = silu<MATH>(xv[u].x * sc.x + sh.x); o.y = silu<MATH>(xv[u].y * sc.y + sh.y);
            o.z = silu<MATH>(xv[u].z * sc.z + sh.z); o.w = silu<MATH>(xv[u].w * sc.w + sh.w);
            float4* dst = reinterpret_cast<float4*>(y) + (int64_t)rr * cv + vc;
            if (NT) nt_store(o, dst); else *dst = o;
        }
    }
}

// sum over rows of silu(x*sc+sh) per channel -> one record per (row block, channel)   (the squeeze pass)
template <int MATH, int U>
__global__ __launch_bounds__(256) void reduce_fwd(const float* __restrict__ x, int rows, int c, Geom g, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, float* __restrict__ part) {
    __shared__ float4 lds[256];
    const int t = threadIdx.x, col = t % g.cw, rsub = t / g.cw, cv = c / 4, vc = blockIdx.y * g.cw + col;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rsub < g.rp && vc < cv) {
        const int r0 = blockIdx.x * g.rpb;
        int r1 = r0 + g.rpb;
        if (r1 > rows) r1 = rows;
        const float4 sc = reinterpret_cast<const float4*>(scale)[vc], sh = reinterpret_cast<const float4*>(shift)[vc];
        for (int r = r0 + rsub; r < r1; r += U * g.rp) {
            float4 xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r + u * g.rp < r1 ? r + u * g.rp : r1 - 1;
                xv[u] = reinterpret_cast<const float4*>(x)[(int64_t)rr * cv + vc];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = r + u * g.rp < r1;
                const float m = ok ? 1.f : 0.f;
                acc.x += m * silu<MATH>(xv[u].x * sc.x + sh.x); acc.y += m * silu<MATH>(xv[u].y * sc.y + sh.y);
                acc.z += m * silu<MATH>(xv[u].z * sc.z + sh.z); acc.w += m * silu<MATH>(xv[u].w * sc.w + sh.w);
            }
        }
    }
    lds[t] = acc;
    __syncthreads();
    if (t < g.cw && vc < cv) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < g.rp; ++r) { const float4 v = lds[r * g.cw + t]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        reinterpret_cast<float4*>(part)[(int64_t)blockIdx.x * cv + vc] = a;
    }
}

// sums of g = dy*silu'(xhat*ga+be) and g*xhat per channel (BatchNorm backward, pass 1)
template <int MATH, int U>
__global__ __launch_bounds__(256) void reduce_bwd(const float* __restrict__ x, const float* __restrict__ dy, int rows, int c, Geom g,
                                                  const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float* __restrict__ part) {
    __shared__ float4 lds[512];
    const int t = threadIdx.x, col = t % g.cw, rsub = t / g.cw, cv = c / 4, vc = blockIdx.y * g.cw + col;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (rsub < g.rp && vc < cv) {
        const int r0 = blockIdx.x * g.rpb;
        int r1 = r0 + g.rpb;
        if (r1 > rows) r1 = rows;
        float mu[4], rs[4], ga[4], be[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) { mu[v] = mean[vc * 4 + v]; rs[v] = rstd[vc * 4 + v]; ga[v] = gamma[vc * 4 + v]; be[v] = beta[vc * 4 + v]; }
        for (int r = r0 + rsub; r < r1; r += U * g.rp) {
            float4 xv[U], dv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int rr = r + u * g.rp < r1 ? r + u * g.rp : r1 - 1;
                xv[u] = reinterpret_cast<const float4*>(x)[(int64_t)rr * cv + vc];
                dv[u] = reinterpret_cast<const float4*>(dy)[(int64_t)rr * cv + vc];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = r + u * g.rp < r1;
                const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, ds[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float xh = (xs[v] - mu[v]) * rs[v];
                    float gg = ds[v] * silu_grad<MATH>(xh * ga[v] + be[v]);
                    gg = ok ? gg : 0.f;
                    s0[v] += gg;
                    s1[v] += gg * xh;
                }
            }
        }
    }
    lds[t * 2] = make_float4(s0[0], s0[1], s0[2], s0[3]);
    lds[t * 2 + 1] = make_float4(s1[0], s1[1], s1[2], s1[3]);
    __syncthreads();
    if (t < g.cw && vc < cv) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        for (int r = 0; r < g.rp; ++r) {
            const float4 v = lds[(r * g.cw + t) * 2], w = lds[(r * g.cw + t) * 2 + 1];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; b.x += w.x; b.y += w.y; b.z += w.z; b.w += w.w;
        }
        reinterpret_cast<float4*>(part)[((int64_t)blockIdx.x * cv + vc) * 2] = a;
        reinterpret_cast<float4*>(part)[((int64_t)blockIdx.x * cv + vc) * 2 + 1] = b;
    }
}

// dx = ga*rs*(g - c1 - xhat*c2), g = dy*silu'(.)   (BatchNorm backward, pass 3)
template <int MATH, int U, bool NT>
__global__ __launch_bounds__(256) void apply_bwd(const float* __restrict__ x, const float* __restrict__ dy, int rows, int c, Geom g,
                                                 const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float* __restrict__ dx) {
    const int t = threadIdx.x, col = t % g.cw, rsub = t / g.cw, cv = c / 4, vc = blockIdx.y * g.cw + col;
    if (rsub >= g.rp || vc >= cv) return;
    const int r0 = blockIdx.x * g.rpb;
    int r1 = r0 + g.rpb;
    if (r1 > rows) r1 = rows;
    float mu[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) { mu[v] = mean[vc * 4 + v]; rs[v] = rstd[vc * 4 + v]; ga[v] = gamma[vc * 4 + v]; be[v] = beta[vc * 4 + v]; }
    for (int r = r0 + rsub; r < r1; r += U * g.rp) {
        float4 xv[U], dv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * g.rp < r1 ? r + u * g.rp : r1 - 1;
            xv[u] = reinterpret_cast<const float4*>(x)[(int64_t)rr * cv + vc];
            dv[u] = reinterpret_cast<const float4*>(dy)[(int64_t)rr * cv + vc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = r + u * g.rp;
            if (rr >= r1) break;
            const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, ds[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
            float o[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float xh = (xs[v] - mu[v]) * rs[v];
                const float gg = ds[v] * silu_grad<MATH>(xh * ga[v] + be[v]);
                o[v] = ga[v] * rs[v] * (gg - 0.01f - xh * 0.02f);
            }
            float4* dst = reinterpret_cast<float4*>(dx) + (int64_t)rr * cv + vc;
            const float4 ov = make_float4(o[0], o[1], o[2], o[3]);
            if (NT) nt_store(ov, dst); else *dst = ov;
        }
    }
}

static Geom geom(int64_t rows, int c, int passes, int cap_blocks) {
    Geom g;
    const int cv = c / 4;
    g.cw = cv < 64 ? cv : 64;
    g.ny = (cv + g.cw - 1) / g.cw;
    g.rp = 256 / g.cw;
    int64_t gx = (rows + (int64_t)g.rp * passes - 1) / ((int64_t)g.rp * passes);
    const int64_t cap = cap_blocks / g.ny > 1 ? cap_blocks / g.ny : 1;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    g.gx = (int)gx;
    g.rpb = (int)((rows + gx - 1) / gx);
    return g;
}

static float* g_flush;
static size_t g_flush_bytes = 1ull << 30;
static float *g_px, *g_pdy;           // the timed kernel's inputs (re-written by the "producer" of protocol 1)
static size_t g_pn;
static int g_protocol = 0;            // 0: caches flushed by a READ of 1 GiB (clean lines), 1: inputs just written by a producer kernel (as in the step)

__global__ void flush_read(const float4* __restrict__ p, size_t n4, float* sink) {
    float a = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 12345.678f) *sink = a;
}
__global__ void producer(float4* __restrict__ x, float4* __restrict__ dy, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float f = (float)((i * 2654435761u) >> 8 & 0xffff) * (1.f / 32768.f) - 1.f;
        x[i] = make_float4(f, -f, 0.5f * f, f + 0.25f);
        dy[i] = make_float4(-f, f, 0.25f * f, f - 0.5f);
    }
}

template <class F> static double time_cold(F launch, int reps = 7) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        flush_read<<<2048, 256>>>(reinterpret_cast<const float4*>(g_flush), g_flush_bytes / 16, g_flush);
        if (g_protocol == 1) producer<<<2048, 256>>>(reinterpret_cast<float4*>(g_px), reinterpret_cast<float4*>(g_pdy), g_pn / 4);
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2] * 1e3;   // us
}

template <class F> static void run(const char* label, double bytes, F f) {
    const double us = time_cold(f);
    printf("  %-46s %8.1f us %7.0f GB/s\n", label, us, bytes / us / 1e3);
}

int main() {
    CK(hipMalloc(&g_flush, g_flush_bytes)); CK(hipMemset(g_flush, 0, g_flush_bytes));
    const int shapes[][2] = {{31360, 960}, {7840, 1536}, {31360, 160}, {125440, 256}, {501760, 192}, {25088, 960}};
    for (auto& s : shapes) {
        const int64_t rows = s[0];
        const int c = s[1];
        const size_t n = (size_t)rows * c;
        float *x, *dy, *y, *par, *part;
        CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&dy, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&par, c * 4 * 4)); CK(hipMalloc(&part, 64 << 20));
        std::vector<float> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 32768.f - 1.f;
        CK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, h.data(), n * 4, hipMemcpyHostToDevice));
        std::vector<float> hp(c * 4, 1.f);
        CK(hipMemcpy(par, hp.data(), c * 16, hipMemcpyHostToDevice));
        const float *mean = par, *rstd = par + c, *gamma = par + 2 * c, *beta = par + 3 * c;
        const double mb = n * 4 / 1e6;
        g_px = x; g_pdy = dy; g_pn = n;
        for (g_protocol = 0; g_protocol < 2; ++g_protocol) {
        printf("== rows %lld c %d (%.1f MB per tensor) %s\n", (long long)rows, c, mb, g_protocol ? "inputs just written (producer -> consumer)" : "clean cold caches");
        const Geom ge = geom(rows, c, 8, 4096), gr = geom(rows, c, 8, 2048), gr4 = geom(rows, c, 8, 4096), gr8 = geom(rows, c, 16, 8192);
        run("apply_fwd  accurate U4", 2 * mb * 1e6, [&] { apply_fwd<0, 4, false><<<dim3(ge.gx, ge.ny), dim3(256), 0, 0>>>(x, (int)rows, c, ge, gamma, beta, y); });
        run("apply_fwd  fast     U4", 2 * mb * 1e6, [&] { apply_fwd<1, 4, false><<<dim3(ge.gx, ge.ny), dim3(256), 0, 0>>>(x, (int)rows, c, ge, gamma, beta, y); });
        run("apply_fwd  none     U4", 2 * mb * 1e6, [&] { apply_fwd<2, 4, false><<<dim3(ge.gx, ge.ny), dim3(256), 0, 0>>>(x, (int)rows, c, ge, gamma, beta, y); });
        run("apply_fwd  fast     U4 nt-store", 2 * mb * 1e6, [&] { apply_fwd<1, 4, true><<<dim3(ge.gx, ge.ny), dim3(256), 0, 0>>>(x, (int)rows, c, ge, gamma, beta, y); });
        run("apply_fwd  fast     U8", 2 * mb * 1e6, [&] { apply_fwd<1, 8, false><<<dim3(ge.gx, ge.ny), dim3(256), 0, 0>>>(x, (int)rows, c, ge, gamma, beta, y); });
        run("reduce_fwd accurate U4 (2048 blocks)", mb * 1e6, [&] { reduce_fwd<0, 4><<<dim3(gr.gx, gr.ny), dim3(256), 0, 0>>>(x, (int)rows, c, gr, gamma, beta, part); });
        run("reduce_fwd fast     U4 (2048 blocks)", mb * 1e6, [&] { reduce_fwd<1, 4><<<dim3(gr.gx, gr.ny), dim3(256), 0, 0>>>(x, (int)rows, c, gr, gamma, beta, part); });
        run("reduce_fwd none     U4 (2048 blocks)", mb * 1e6, [&] { reduce_fwd<2, 4><<<dim3(gr.gx, gr.ny), dim3(256), 0, 0>>>(x, (int)rows, c, gr, gamma, beta, part); });
        run("reduce_fwd fast     U8 (2048 blocks)", mb * 1e6, [&] { reduce_fwd<1, 8><<<dim3(gr.gx, gr.ny), dim3(256), 0, 0>>>(x, (int)rows, c, gr, gamma, beta, part); });
        run("reduce_fwd fast     U4 (4096 blocks)", mb * 1e6, [&] { reduce_fwd<1, 4><<<dim3(gr4.gx, gr4.ny), dim3(256), 0, 0>>>(x, (int)rows, c, gr4, gamma, beta, part); });
        run("reduce_bwd accurate U8 (2048 blocks)", 2 * mb * 1e6, [&] { reduce_bwd<0, 8><<<dim3(gr.gx, gr.ny), dim3(256), 0, 0>>>(x, dy, (int)rows, c, gr, mean, rstd, gamma, beta, part); });
        run("reduce_bwd fast     U8 (2048 blocks)", 2 * mb * 1e6, [&] { reduce_bwd<1, 8><<<dim3(gr.gx, gr.ny), dim3(256), 0, 0>>>(x, dy, (int)rows, c, gr, mean, rstd, gamma, beta, part); });
        run("reduce_bwd fast     U4 (2048 blocks)", 2 * mb * 1e6, [&] { reduce_bwd<1, 4><<<dim3(gr.gx, gr.ny), dim3(256), 0, 0>>>(x, dy, (int)rows, c, gr, mean, rstd, gamma, beta, part); });
        run("reduce_bwd fast     U4 (4096 blocks)", 2 * mb * 1e6, [&] { reduce_bwd<1, 4><<<dim3(gr4.gx, gr4.ny), dim3(256), 0, 0>>>(x, dy, (int)rows, c, gr4, mean, rstd, gamma, beta, part); });
        run("reduce_bwd fast     U2 (8192 blocks)", 2 * mb * 1e6, [&] { reduce_bwd<1, 2><<<dim3(gr8.gx, gr8.ny), dim3(256), 0, 0>>>(x, dy, (int)rows, c, gr8, mean, rstd, gamma, beta, part); });
        run("reduce_bwd none     U4 (2048 blocks)", 2 * mb * 1e6, [&] { reduce_bwd<2, 4><<<dim3(gr.gx, gr.ny), dim3(256), 0, 0>>>(x, dy, (int)rows, c, gr, mean, rstd, gamma, beta, part); });
        run("apply_bwd  accurate U4", 3 * mb * 1e6, [&] { apply_bwd<0, 4, false><<<dim3(ge.gx, ge.ny), dim3(256), 0, 0>>>(x, dy, (int)rows, c, ge, mean, rstd, gamma, beta, y); });
        run("apply_bwd  fast     U4", 3 * mb * 1e6, [&] { apply_bwd<1, 4, false><<<dim3(ge.gx, ge.ny), dim3(256), 0, 0>>>(x, dy, (int)rows, c, ge, mean, rstd, gamma, beta, y); });
        run("apply_bwd  fast     U4 nt-store", 3 * mb * 1e6, [&] { apply_bwd<1, 4, true><<<dim3(ge.gx, ge.ny), dim3(256), 0, 0>>>(x, dy, (int)rows, c, ge, mean, rstd, gamma, beta, y); });
        run("apply_bwd  fast     U2", 3 * mb * 1e6, [&] { apply_bwd<1, 2, false><<<dim3(ge.gx, ge.ny), dim3(256), 0, 0>>>(x, dy, (int)rows, c, ge, mean, rstd, gamma, beta, y); });
        run("apply_bwd  none     U4", 3 * mb * 1e6, [&] { apply_bwd<2, 4, false><<<dim3(ge.gx, ge.ny), dim3(256), 0, 0>>>(x, dy, (int)rows, c, ge, mean, rstd, gamma, beta, y); });
        }
        CK(hipFree(x)); CK(hipFree(dy)); CK(hipFree(y)); CK(hipFree(par)); CK(hipFree(part));
    }
    return 0;
}
