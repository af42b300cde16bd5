// Shared helpers for the gfx950 kernels of libsuhpe.so (error reporting, launch checks, wave reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/suhpe.h"

namespace suhpe {

constexpr int kWave = 64;   // CDNA4 wavefront

void set_error(const char* fmt, ...);
// the dense-conv launchers report which kernel they ACTUALLY launched on this thread (suhpe_last_kernel_name: per-kernel profiles key on it;
// suhpe_conv2d_kernel_name predicts from the descriptor alone and cannot see run-time fallbacks such as a bias on a register-resident Winograd shape)
void note_kernel(const char* name);
const char* last_kernel();

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SUHPE_ELAUNCH;
    }
    return SUHPE_OK;
}

#define SUHPE_REQUIRE(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            ::suhpe::set_error(__VA_ARGS__); \
            return SUHPE_EINVAL;            \
        }                                   \
    } while (0)

inline hipStream_t as_stream(suhpe_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grid for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8, capped
inline int grid_for(int64_t work_items, int block) {
    int64_t g = ceil_div(work_items, block);
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

// nn_elem.hip: finish BatchNorm-backward partial records [row block][channel] float2 (sum g, sum g xhat) into sums / dgamma / dbeta;
// gx_a > 0: two batch segments, records [0, gx_a) and [gx_a, gx)
void launch_bn_bwd_finalize(const float* part, int gx_a, int gx, int c, float* sums, float* dgamma, float* dbeta, int accumulate, hipStream_t st);

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Predicated loads go through buffer descriptors: a lane that must read zero (padding tap, row beyond M, column beyond N)
// gets an out-of-range offset and the hardware bounds check returns 0.  With a plain load + select the select sits right
// behind the load, and the compiler then waits for the data there (before the MFMAs of the current slab) instead of at
// the LDS store one or two slabs later: that wait alone cost the pointwise GEMM a third of its rate.
using BufRsrc = __amdgpu_buffer_rsrc_t;
constexpr unsigned kOob = 0x80000000u;          // beyond any descriptor (tensors are < 2 GiB, checked by the host)
__device__ __forceinline__ BufRsrc make_rsrc(const float* base, int64_t bytes) {
    // readfirstlane makes the descriptor inputs provably wave-uniform (otherwise every buffer op gets a waterfall loop)
    const uint64_t addr = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr), hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
    const int nb = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffff ? 0x7fffffff : bytes));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)hi << 32) | lo), 0, nb, 0x00020000);
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int VEC> __device__ __forceinline__ void buf_load(BufRsrc r, unsigned byte_off, float* dst) {
    if constexpr (VEC == 4) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
        dst[0] = __uint_as_float(v.x); dst[1] = __uint_as_float(v.y); dst[2] = __uint_as_float(v.z); dst[3] = __uint_as_float(v.w);
    } else {
        dst[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
    }
}

// ------------------------------------------------------------------------------------------ activations
// sigmoid through the hardware transcendentals: v_exp_f32 (2^x) and v_rcp_f32, 1 ulp each, instead of expf's range reduction and an
// IEEE division (~25 VALU instructions per element).  The read-only BatchNorm / SqueezeExcitation reductions were VALU-bound on the
// accurate forms (tools/proto/ew_lab.hip: the squeeze pass over 31 360 x 960 31.0 -> 21.9 us, the BatchNorm backward sums 44.6 -> 38.8 us);
// the error stays below 4e-7 relative for |u| <= 5 and below fp32 resolution of the result beyond.
__device__ __forceinline__ float fast_sigmoid(float u) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * u));
}
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e. fp32 resolution of the GELU it feeds) on the same two
// transcendentals; *e receives exp(-x^2), which GELU' needs as well (libm's erff is ~30 VALU instructions + an expf for the density).
__device__ __forceinline__ float fast_erf(float x, float* e) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    *e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    return copysignf(1.f - p * t * *e, x);
}
__device__ __forceinline__ float act_apply(float u, int act) {
    switch (act) {
        case SUHPE_ACT_RELU: return u > 0.f ? u : 0.f;
        case SUHPE_ACT_RELU6: return fminf(fmaxf(u, 0.f), 6.f);
        case SUHPE_ACT_SILU: return u * fast_sigmoid(u);
        case SUHPE_ACT_GELU: { float e; return 0.5f * u * (1.f + fast_erf(u * 0.70710678118654752f, &e)); }
        case SUHPE_ACT_SIGMOID: return fast_sigmoid(u);
        default: return u;
    }
}
// d act(u) / du
__device__ __forceinline__ float act_grad(float u, int act) {
    switch (act) {
        case SUHPE_ACT_RELU: return u > 0.f ? 1.f : 0.f;
        case SUHPE_ACT_RELU6: return (u > 0.f && u < 6.f) ? 1.f : 0.f;
        case SUHPE_ACT_SILU: {
            const float s = fast_sigmoid(u);
            return s * (1.f + u * (1.f - s));
        }
        case SUHPE_ACT_GELU: {
            float e;                                     // exp(-(u / sqrt 2)^2) = exp(-u^2 / 2): the density's exponential
            const float cdf = 0.5f * (1.f + fast_erf(u * 0.70710678118654752f, &e));
            return cdf + u * 0.3989422804014327f * e;
        }
        case SUHPE_ACT_SIGMOID: {
            const float s = fast_sigmoid(u);
            return s * (1.f - s);
        }
        default: return 1.f;
    }
}

}  // namespace suhpe
