// Store-pattern microbenchmark for the GEMM epilogue: a (rows x cols) fp32 matrix written by 512-thread blocks that own
// 128x128 tiles, with the three candidate lane->address mappings.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC
#include <hip/hip_runtime.h>
#include <cstdint>

// (a) MFMA 32x32 C layout as the kernel stores it today: per instruction a wave writes 2 rows x 128 B, one dword per lane
__global__ __launch_bounds__(512) void store_mfma_layout(float* __restrict__ y, int rows, int cols, int col_tiles) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 2, wn = wave & 3;
    const int b = blockIdx.x, row_tile = b / col_tiles, col_tile = b % col_tiles;
    const int r32 = lane & 31, h = lane >> 5;
    const int col = col_tile * 128 + wn * 32 + r32;
    if (col >= cols) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t row = (int64_t)row_tile * 128 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (row < rows) y[row * cols + col] = (float)e;
        }
}
// (b) wave-private transpose: per instruction 8 rows x 128 B, 16 B per lane
__global__ __launch_bounds__(512) void store_wave_f4(float* __restrict__ y, int rows, int cols, int col_tiles) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 2, wn = wave & 3;
    const int b = blockIdx.x, row_tile = b / col_tiles, col_tile = b % col_tiles;
    const int col = col_tile * 128 + wn * 32 + (lane & 7) * 4;
    if (col >= cols) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t row = (int64_t)row_tile * 128 + wm * 64 + k * 8 + (lane >> 3);
        if (row < rows) *reinterpret_cast<float4*>(y + row * cols + col) = make_float4(1.f, 2.f, 3.f, (float)k);
    }
}
// (c) block-wide rows: per instruction 2 rows x 512 B, 16 B per lane
__global__ __launch_bounds__(512) void store_block_f4(float* __restrict__ y, int rows, int cols, int col_tiles) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.x, row_tile = b / col_tiles, col_tile = b % col_tiles;
    const int col = col_tile * 128 + (lane & 31) * 4;
    if (col >= cols) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t row = (int64_t)row_tile * 128 + wave * 16 + k * 2 + (lane >> 5);
        if (row < rows) *reinterpret_cast<float4*>(y + row * cols + col) = make_float4(1.f, 2.f, 3.f, (float)k);
    }
}

extern "C" int store_pattern(int which, float* y, int rows, int cols, void* stream) {
    const int col_tiles = (cols + 127) / 128, row_tiles = (rows + 127) / 128;
    const dim3 grid(row_tiles * col_tiles), block(512);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (which == 0) hipLaunchKernelGGL(store_mfma_layout, grid, block, 0, st, y, rows, cols, col_tiles);
    else if (which == 1) hipLaunchKernelGGL(store_wave_f4, grid, block, 0, st, y, rows, cols, col_tiles);
    else hipLaunchKernelGGL(store_block_f4, grid, block, 0, st, y, rows, cols, col_tiles);
    return (int)hipGetLastError();
}
