// Does the fp32 MFMA shape change what the chip sustains?  Same 64x64 accumulator tile per wave, operands re-read from LDS
// (random data) every step, v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32: equal cycles per FLOP on paper, so any difference in
// TFLOP/s is the clock the chip holds (MI355X_MICROARCH.md, DVFS give-back item 7, measured there for the bf16 shapes).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape_probe.hip -o tools/proto/bin/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int LDK = 36;

template <int SHAPE>
__global__ __launch_bounds__(256) void loop_kernel(const float* __restrict__ init, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) float As[128 * LDK], Bs[128 * LDK];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = t; i < 128 * LDK; i += 256) { As[i] = init[i]; Bs[i] = init[128 * LDK + i]; }
    __syncthreads();
    float s = 0.f;
    if (SHAPE == 32) {
        floatx16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int r32 = lane & 31, h = lane >> 5;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                float4 af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const float4*>(&As[(wm * 64 + i * 32 + r32) * LDK + kg * 8 + h * 4]);
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + j * 32 + r32) * LDK + kg * 8 + h * 4]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    } else {
        floatx4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        const int r16 = lane & 15, kq = lane >> 4;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kg = 0; kg < 2; ++kg) {
                float4 af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const float4*>(&As[(wm * 64 + i * 16 + r16) * LDK + kg * 16 + kq * 4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + j * 16 + r16) * LDK + kg * 16 + kq * 4]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    }
    out[blockIdx.x * 256 + t] = s;
}

int main() {
    std::vector<float> h(2 * 128 * LDK);
    unsigned long long s = 7;
    for (auto& v : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0) * 1e-3f; }
    float *init, *out;
    hipMalloc(&init, h.size() * 4); hipMalloc(&out, 4096 * 256 * 4);
    hipMemcpy(init, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024}) {
        for (int shape : {32, 16, 32, 16}) {
            const int iters = 4000;
            auto run = [&]() {
                if (shape == 32) hipLaunchKernelGGL(loop_kernel<32>, dim3(blocks), dim3(256), 0, 0, init, out, iters);
                else hipLaunchKernelGGL(loop_kernel<16>, dim3(blocks), dim3(256), 0, 0, init, out, iters);
            };
            run(); hipDeviceSynchronize();
            hipEventRecord(e0); for (int r = 0; r < 5; ++r) run(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            const double flop = (double)blocks * 4 /*waves*/ * iters * 32 /*k per iter*/ * 64 * 64 * 2;
            printf("blocks %4d  mfma %s : %.3f ms  %.1f TF/s\n", blocks, shape == 32 ? "32x32x2 " : "16x16x4 ", ms, flop / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
