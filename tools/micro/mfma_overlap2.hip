// Developer micro-benchmark (gfx950): does VALU work of ANOTHER wave on the same SIMD overlap with v_mfma_f32_32x32x2_f32?
// 512-thread workgroups, one per CU: waves 0-3 issue only MFMAs (one per SIMD), waves 4-7 issue only v_fma_f32 / v_pk_fma_f32
// (the partner wave of each SIMD).  Reports the MFMA wave's cycles per MFMA and the VALU wave's cycles per VALU instruction,
// alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>  // 1 = MFMA waves only work, 2 = VALU waves only, 3 = both
__global__ __launch_bounds__(512, 2) void k(float *out, long long *cyc, int iters, float a, float b)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (MODE & 1) {
            f32x16 acc[2];
            for (int c = 0; c < 2; ++c)
                for (int j = 0; j < 16; ++j) acc[c][j] = 0;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 1], 0, 0, 0);
            }
            s = acc[0][0] + acc[1][15];
        }
    } else {
        if (MODE & 2) {
            float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[u & 7]) : "v"(v[(u + 3) & 7]));
            }
            for (int j = 0; j < 8; ++j) s += v[j];
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
}

template <int MODE>
static void run(const char *name)
{
    float *out;
    long long *cyc, h[2] = {0, 0};
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&cyc, 16);
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, cyc, 10, 1.f, 0.5f);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1.f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-28s MFMA wave: %7.1f cycles / MFMA    VALU wave: %6.2f cycles / v_fma_f32\n", name, (double)h[0] / (16.0 * iters),
           (double)h[1] / (64.0 * iters));
}

int main()
{
    run<1>("MFMA waves alone");
    run<2>("VALU waves alone");
    run<3>("both (partner waves per SIMD)");
    return 0;
}
