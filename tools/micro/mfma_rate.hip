// Developer micro-benchmark: issue rate of the f32 MFMAs on gfx950 as a function of independent accumulator chains
// and waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(256) void k16(float *out, int iters, float a, float b)
{
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float a, float b)
{
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int j = 0; j < 16; ++j) acc[c][j] = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the group body of srf_spconv_gs_k: 64 MFMAs on 32 distinct A and 64 distinct B registers, two chains
__global__ __launch_bounds__(256) void k16_regs(float *out, int iters, const float *src)
{
    float a[32], b[64];
    for (int i = 0; i < 32; ++i) a[i] = src[threadIdx.x + 256 * i];
    for (int i = 0; i < 64; ++i) b[i] = src[threadIdx.x + 256 * (32 + i)];
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[32 + u], acc[1], 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][3];
}

template <typename F>
static void run(const char *name, F launch, double flop_per_wave_iter, int ch, int wgs_per_cu)
{
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch(out, 10, wgs_per_cu);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(out, iters, wgs_per_cu);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = flop_per_wave_iter * 16 * ch * iters * 4.0 * 256 * wgs_per_cu;
    printf("%-10s chains=%d waves/SIMD=%d  %8.3f ms  %7.1f TFLOP/s\n", name, ch, wgs_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main()
{
    for (int w = 1; w <= 2; ++w) {
        run("16x16x4", [](float *o, int it, int w) { hipLaunchKernelGGL(k16<1>, dim3(256 * w), dim3(256), 0, 0, o, it, 1.f, 1.f); }, 2048, 1, w);
        run("16x16x4", [](float *o, int it, int w) { hipLaunchKernelGGL(k16<2>, dim3(256 * w), dim3(256), 0, 0, o, it, 1.f, 1.f); }, 2048, 2, w);
        run("16x16x4", [](float *o, int it, int w) { hipLaunchKernelGGL(k16<4>, dim3(256 * w), dim3(256), 0, 0, o, it, 1.f, 1.f); }, 2048, 4, w);
        run("32x32x2", [](float *o, int it, int w) { hipLaunchKernelGGL(k32<1>, dim3(256 * w), dim3(256), 0, 0, o, it, 1.f, 1.f); }, 4096, 1, w);
        run("32x32x2", [](float *o, int it, int w) { hipLaunchKernelGGL(k32<2>, dim3(256 * w), dim3(256), 0, 0, o, it, 1.f, 1.f); }, 4096, 2, w);
    }
    float *src;
    hipMalloc(&src, 256 * 96 * sizeof(float));
    hipMemset(src, 0, 256 * 96 * sizeof(float));
    for (int w = 1; w <= 2; ++w)
        run("16x16x4 regs", [src](float *o, int it, int w) { hipLaunchKernelGGL(k16_regs, dim3(256 * w), dim3(256), 0, 0, o, it, src); }, 2048 * 2, 2, w);
    return 0;
}
