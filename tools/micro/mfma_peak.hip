// Register-only MFMA loop: the practical fp32 matrix-pipe ceiling at a given occupancy (clock under load included).
#include <hip/hip_runtime.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a, float b) {
    floatx16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = (float)(threadIdx.x + c);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16 / CHAINS; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int mfma_peak(int chains, int blocks, int iters, float* out, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (chains == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 0.5f);
    else if (chains == 2) hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 0.5f);
    else hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 0.5f);
    return (int)hipGetLastError();
}
