// Developer micro-benchmark (gfx950): would splitting the 16 Winograd frequencies over TWO co-resident workgroups per CU
// (8 frequencies = 128 accumulators per wave, two waves per SIMD with independent barriers) keep the MFMA pipe busier than
// the current kernel (16 frequencies = 256 accumulators, one wave per SIMD)?
// Every wave does the instruction mix of one 8-channel chunk of srf_wino3x3_k (same wave does loader, transform and MFMA
// work, as in the real kernel), scaled by the split:
//   FULL : 64 MFMA 32x32x2, 32 ds_read_b128 (fragments), 12 ds_read_b128 (patch), 64 VALU, 16+8+3 ds_write_b128, 11 loads
//   HALF : 32 MFMA,         16 fragment reads,            9 patch reads,            40 VALU, 8+4+3 writes,         7 loads
// one barrier per chunk.  Prints ns per chunk and the MFMA-equivalent rate of a CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NF /* frequencies per wave: 16 or 8 */, int WPC>
__global__ __launch_bounds__(256, WPC) void k(float *out, int iters, const float *src, size_t src_elems)
{
    constexpr int NG = NF / 2;                    // groups of 2 frequencies
    constexpr int NPR = NF == 16 ? 12 : 9;        // patch reads
    constexpr int NVW = NF == 16 ? 16 : 8;        // V writes
    constexpr int NUW = NF == 16 ? 8 : 4;         // U loads / writes
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];  // V[2][NF*128] | U[2][NF*128] | RAW[924 (x2 when NF == 16)]
    f32x4 *sV = lds, *sU = lds + 2 * NF * 128, *sR = sU + 2 * NF * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * NF * 128 + 924; i += 256) lds[i] = f32x4{src[i & 4095], 0.f, 1.f, 2.f};
    __syncthreads();
    const int th = wave & 1, chh = wave >> 1, li = lane & 31, lh = lane >> 5;
    const int a_off = lh * 64 + th * 32 + li, b_off = lh * 64 + chh * 32 + li;
    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    f32x4 gr[3], ur[NUW], t[NVW];
#pragma unroll
    for (int j = 0; j < 3; ++j) gr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NUW; ++j) ur[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NVW; ++j) t[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        const f32x4 *v = sV + (it & 1) * NF * 128, *u = sU + (it & 1) * NF * 128;
        f32x4 *vw = sV + ((it + 1) & 1) * NF * 128, *uw = sU + ((it + 1) & 1) * NF * 128;
        f32x4 fa[2], fb[2];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                fa[e] = v[a_off + (g * 2 + e) * 128];
                fb[e] = u[b_off + (g * 2 + e) * 128];
            }
            // the non-MFMA work of the chunk, spread over the groups as in the kernel
            if (g == 0) {
#pragma unroll
                for (int j = 0; j < 3; ++j) sR[(j * 256 + tid) % 924] = gr[j];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const size_t off = ((size_t)blockIdx.x * 8192 + (size_t)(it & 63) * 3 * 1024 + j * 1024 + tid * 4) % src_elems;
                    gr[j] = *reinterpret_cast<const f32x4 *>(src + off);
                }
            }
            if (g == 1 || g == 3) {
#pragma unroll
                for (int j = 0; j < NVW / 2; ++j) vw[((g >> 1) * (NVW / 2) + j) * 256 + tid] = t[(g >> 1) * (NVW / 2) + j];
            }
            if (g == 2) {
#pragma unroll
                for (int j = 0; j < NUW; ++j) uw[j * 256 + tid] = ur[j];
#pragma unroll
                for (int j = 0; j < NUW; ++j) ur[j] = *reinterpret_cast<const f32x4 *>(src + ((size_t)(it & 31) * 8192 + j * 1024 + tid * 4));
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    acc[g * 2 + e] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e][ks], fb[e][ks], acc[g * 2 + e], 0, 0, 0);
        }
        __syncthreads();
        // patch -> registers -> transform (VALU), results kept for the next chunk's V writes
        f32x4 p[NPR];
#pragma unroll
        for (int j = 0; j < NPR; ++j) p[j] = sR[(tid * 3 + j * 37) % 924];
#pragma unroll
        for (int j = 0; j < NVW; ++j) t[j] = (p[j % NPR] - p[(j + 2) % NPR]) + (p[(j + 1) % NPR] + t[j] * 1e-30f);
    }
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) s += acc[f][0] + acc[f][15];
    out[blockIdx.x * 256 + tid] = s + t[0][0] + ur[0][0] + gr[0][0];
}

template <int NF, int WPC>
static void run(const char *name, const float *src, size_t n)
{
    float *out;
    const int blocks = 256 * WPC;
    (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    const int iters = 400;
    const size_t sh = (size_t)(4 * NF * 128 + 924 * (NF == 16 ? 2 : 1)) * 16;
    (void)hipFuncSetAttribute((const void *)k<NF, WPC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL((k<NF, WPC>), dim3(blocks), dim3(256), sh, 0, out, 20, src, n);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NF, WPC>), dim3(blocks), dim3(256), sh, 0, out, iters, src, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // MFMAs per CU per chunk-iteration: WPC workgroups x 4 waves x 4 NF; FLOPs per MFMA 4096
    const double tf = 256.0 * WPC * 4 * 4 * NF * 4096.0 * iters / (ms * 1e-3) / 1e12;
    printf("%-40s LDS %6zu B x %d  %8.1f us  %7.1f ns per chunk  %6.1f TFLOP/s on the MFMA  err=%s\n", name, sh, WPC, ms * 1e3,
           ms * 1e6 / iters, tf, hipGetErrorString(hipGetLastError()));
    (void)hipFree(out);
}

int main()
{
    const size_t n = 64u << 20;
    float *src;
    (void)hipMalloc(&src, n * 4);
    (void)hipMemset(src, 0, n * 4);
    for (int i = 0; i < 2; ++i) run<16, 1>("warm-up", src, n);
    run<16, 1>("16 frequencies, 1 workgroup per CU", src, n);
    run<8, 1>("8 frequencies, 1 workgroup per CU", src, n);
    run<8, 2>("8 frequencies, 2 workgroups per CU", src, n);
    return 0;
}
