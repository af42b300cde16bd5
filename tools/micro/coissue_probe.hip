// Do vector-ALU instructions issue in the shadow of fp32 MFMAs on gfx950?  (1) one wavefront: a loop of 16 independent v_mfma_f32_16x16x4_f32, NV
// independent v_fma_f32 behind each; (2) two wavefronts per SIMD, one running the MFMA loop and the other a pure v_fma loop (role = workgroup parity).
#include <hip/hip_runtime.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ __launch_bounds__(256) void mix_loop(float* out, int iters, float a, float b, int role_split) {
    floatx4 acc[16];
    float v[8];
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[c][e] = (float)(threadIdx.x + c);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (float)(threadIdx.x * 3 + k);
    const int role = role_split ? (blockIdx.x & 1) + 1 : 0;      // 0: both, 1: MFMA only, 2: VALU only
    if (role == 0) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < NV; ++k) v[k & 7] = fmaf(v[k & 7], a, b);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
            }
        }
    } else if (role == 1) {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    } else {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int c = 0; c < 16; ++c)
#pragma unroll
                for (int k = 0; k < NV; ++k) v[k & 7] = fmaf(v[k & 7], a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += acc[c][e];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int coissue(int nv, int blocks, int iters, int role_split, float* out, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (nv) {
        case 0: hipLaunchKernelGGL(mix_loop<0>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 0.5f, role_split); break;
        case 2: hipLaunchKernelGGL(mix_loop<2>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 0.5f, role_split); break;
        case 4: hipLaunchKernelGGL(mix_loop<4>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 0.5f, role_split); break;
        case 6: hipLaunchKernelGGL(mix_loop<6>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 0.5f, role_split); break;
        case 8: hipLaunchKernelGGL(mix_loop<8>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 0.5f, role_split); break;
        default: hipLaunchKernelGGL(mix_loop<12>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f, 0.5f, role_split); break;
    }
    return (int)hipGetLastError();
}
