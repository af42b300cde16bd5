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

}  // namespace suhpe
