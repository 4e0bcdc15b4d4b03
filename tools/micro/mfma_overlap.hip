// Developer micro-benchmark (gfx950): how much independent VALU / LDS work hides behind v_mfma_f32_32x32x2_f32 issued by
// the SAME wave (one wave per SIMD, the regime of srf_wino3x3_k).  Per MFMA: NV independent v_fma_f32 (or v_pk_fma_f32)
// and NL ds_read_b128.  Reports cycles per MFMA (s_memtime) -- 64 = the MFMA alone.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_overlap.hip -o /tmp/mfma_overlap && /tmp/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV, int NL, int PK, int NW>
__global__ __launch_bounds__(256, 1) void k(float *out, long long *cyc, int iters, float a, float b)
{
    __shared__ f32x4 lds[1024];
    lds[threadIdx.x] = f32x4{a, b, a, b};
    __syncthreads();
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int j = 0; j < 16; ++j) acc[c][j] = 0;
    float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
    f32x2 pv[4] = {{a, b}, {b, a}, {a + 1, b}, {b + 1, a}};
    f32x4 l = {0, 0, 0, 0};
    const int li = threadIdx.x & 255;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            f32x16 c = acc[u & 3];
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
            asm volatile("" : "+a"(c));
            acc[u & 3] = c;
            if (PK) {
#pragma unroll
                for (int j = 0; j < NV; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[j & 3]) : "v"(pv[(j + 1) & 3]));
            } else {
#pragma unroll
                for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j & 7]) : "v"(v[(j + 3) & 7]));
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                f32x4 t = lds[(li + 64 * j + 16 * u) & 1023];
                asm volatile("" ::"v"(t));
            }
#pragma unroll
            for (int j = 0; j < NW; ++j) lds[(li + 256 * j) & 1023] = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = l[0];
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][15];
    for (int j = 0; j < 8; ++j) s += v[j];
    for (int j = 0; j < 4; ++j) s += pv[j][0] + pv[j][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NV, int NL, int PK, int NW>
static void run()
{
    float *out;
    long long *cyc, h = 0;
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<NV, NL, PK, NW>), dim3(256), dim3(256), 0, 0, out, cyc, 10, 1.f, 0.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NL, PK, NW>), dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("per MFMA: %2d %s, %d ds_read_b128, %d ds_write_b128 -> %6.1f cycles/MFMA  (%.3f ms, %.1f TFLOP/s)\n", NV, PK ? "v_pk_fma_f32" : "v_fma_f32   ", NL, NW,
           (double)h / (16.0 * iters), ms, 4096.0 * 16 * iters * 1024 / ms / 1e9);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    run<0, 0, 0, 0>();
    run<2, 0, 0, 0>();
    run<4, 0, 0, 0>();
    run<8, 0, 0, 0>();
    run<12, 0, 0, 0>();
    run<16, 0, 0, 0>();
    run<4, 0, 1, 0>();
    run<8, 0, 1, 0>();
    run<0, 1, 0, 0>();
    run<0, 2, 0, 0>();
    run<0, 4, 0, 0>();
    run<0, 0, 0, 1>();
    run<0, 0, 0, 2>();
    run<4, 1, 0, 0>();
    run<4, 1, 0, 1>();
    return 0;
}
