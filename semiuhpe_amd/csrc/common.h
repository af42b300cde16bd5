// Shared helpers for the gfx950 kernels of libsuhpe.so (error reporting, launch checks, wave reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/suhpe.h"

namespace suhpe {

constexpr int kWave = 64;   // CDNA4 wavefront

void set_error(const char* fmt, ...);

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

// ------------------------------------------------------------------------------------------ activations
__device__ __forceinline__ float act_apply(float u, int act) {
    switch (act) {
        case SUHPE_ACT_RELU: return u > 0.f ? u : 0.f;
        case SUHPE_ACT_RELU6: return fminf(fmaxf(u, 0.f), 6.f);
        case SUHPE_ACT_SILU: return u / (1.f + expf(-u));
        case SUHPE_ACT_GELU: return 0.5f * u * (1.f + erff(u * 0.70710678118654752f));
        case SUHPE_ACT_SIGMOID: return 1.f / (1.f + expf(-u));
        default: return u;
    }
}
// d act(u) / du
__device__ __forceinline__ float act_grad(float u, int act) {
    switch (act) {
        case SUHPE_ACT_RELU: return u > 0.f ? 1.f : 0.f;
        case SUHPE_ACT_RELU6: return (u > 0.f && u < 6.f) ? 1.f : 0.f;
        case SUHPE_ACT_SILU: {
            const float s = 1.f / (1.f + expf(-u));
            return s * (1.f + u * (1.f - s));
        }
        case SUHPE_ACT_GELU: {
            const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752f));
            return cdf + u * 0.3989422804014327f * expf(-0.5f * u * u);
        }
        case SUHPE_ACT_SIGMOID: {
            const float s = 1.f / (1.f + expf(-u));
            return s * (1.f - s);
        }
        default: return 1.f;
    }
}

}  // namespace suhpe
