// Developer micro-benchmark (gfx950): cycles per v_mfma_f32_32x32x2_f32 of the 64-MFMA block of srf_conv1x1_nhwc_k
// (4 x 4 accumulators = 256 AGPRs, operands = 4 A + 4 B float4 fragments) under increasingly realistic conditions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float *out, long long *cyc, int iters, const float *src)
{
    __shared__ f32x4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = f32x4{src[i & 1023], src[(i + 1) & 1023], src[(i + 2) & 1023], src[(i + 3) & 1023]};
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    f32x16 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    f32x4 fa[2][4], fb[2][4];
    for (int s = 0; s < 2; ++s)
        for (int i = 0; i < 4; ++i) {
            fa[s][i] = lds[lh * 256 + i * 32 + li + s * 512];
            fb[s][i] = lds[2048 + lh * 256 + i * 32 + li + s * 512];
        }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (MODE >= 2) {  // re-read the OTHER set's fragments (as the kernel does, one block ahead)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[s ^ 1][i] = lds[lh * 256 + i * 32 + li + ((it + s) & 3) * 512];
                    fb[s ^ 1][i] = lds[2048 + lh * 256 + i * 32 + li + ((it + s) & 3) * 512];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(fa[s][i]));
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(fb[s][j]));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (MODE == 0)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, 0.5f, acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s][i][ks], fb[s][j][ks], acc[i][j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][15];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
static void run(const char *name, const float *src)
{
    float *out;
    long long *cyc, h = 0;
    (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    (void)hipMalloc(&cyc, 8);
    const int iters = 500;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, cyc, 10, src);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, cyc, iters, src);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6.1f cycles/MFMA  %.3f ms  %.1f TFLOP/s\n", name, (double)h / (128.0 * iters), ms, 4096.0 * 128 * iters * 1024 / ms / 1e9);
}

int main()
{
    float *src, h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.f - 0.5f;
    (void)hipMalloc(&src, sizeof(h));
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>("constant operands, 16 accumulators", src);
    run<1>("register fragments (random data)", src);
    run<2>("+ 8 ds_read_b128 per 64 MFMAs", src);
    return 0;
}
