// Developer micro-benchmark (gfx950): would a wave-specialised Winograd kernel keep the MFMA pipe busier?
// One workgroup per CU, 8 waves: waves 0-3 only read fragments from LDS and issue MFMAs (v_mfma_f32_16x16x4_f32, wave tile
// 32 tiles x 16 channels x 16 frequencies = 128 accumulators), waves 4-7 do the loader's work of one 8-channel chunk
// (global loads, ~70 VALU of transform, LDS stores); one barrier per chunk.  Each SIMD then holds one wave of each kind.
// MODE 0: MFMA waves alone (loader waves only join the barrier); 1: + loader LDS/VALU work; 2: + global loads.
// Prints cycles per chunk seen by an MFMA wave (ideal: 64 MFMAs x 32 cycles = 2048).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int ROLE>
__global__ __launch_bounds__(512, 1) void k(float *out, long long *cyc, int iters, const float *src, size_t src_elems)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];  // V 2 x 8192 floats, U 2 x 4096 floats, raw 2 x 2048
    float *sV = lds, *sU = lds + 2 * 8192, *sR = sU + 2 * 4096;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ROLE 0: waves 0-3 are the MFMA waves; 1: the even waves; 2: waves 0,1,4,5
    const bool is_mma = ROLE == 0 ? wv < 4 : (ROLE == 1 ? !(wv & 1) : !(wv & 2));
    const int wave = ROLE == 0 ? (wv & 3) : (ROLE == 1 ? (wv >> 1) : ((wv & 1) | ((wv >> 2) << 1)));
    for (int i = tid; i < 2 * 8192 + 2 * 4096 + 2 * 2048; i += 512) lds[i] = src[i & 4095];
    __syncthreads();
    long long t0 = 0, t1 = 0;
    if (is_mma) {
        // ---- MFMA wave: tile half th = wave & 1 (32 tiles), channel half chh = wave >> 1 (16 channels)
        const int th = wave & 1, chh = wave >> 1;
        const int li = lane & 15, g = lane >> 4;
        f32x4 acc[16][2];
#pragma unroll
        for (int f = 0; f < 16; ++f)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[f][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            const float *v = sV + (it & 1) * 8192, *u = sU + (it & 1) * 4096;
            // V[f][g 4][tile 64][2] (k = g, 4 + g), U[f][g][channel 32][2]: a wave instruction reads 4 runs of 128 contiguous bytes
            f32x2 a0 = *reinterpret_cast<const f32x2 *>(v + ((0 * 4 + g) * 64 + th * 32 + li) * 2);
            f32x2 a1 = *reinterpret_cast<const f32x2 *>(v + ((0 * 4 + g) * 64 + th * 32 + 16 + li) * 2);
            f32x2 b = *reinterpret_cast<const f32x2 *>(u + ((0 * 4 + g) * 32 + chh * 16 + li) * 2);
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                f32x2 na0 = a0, na1 = a1, nb = b;
                if (f + 1 < 16 && MODE != 3 && MODE != 5) {
                    na0 = *reinterpret_cast<const f32x2 *>(v + (((f + 1) * 4 + g) * 64 + th * 32 + li) * 2);
                    na1 = *reinterpret_cast<const f32x2 *>(v + (((f + 1) * 4 + g) * 64 + th * 32 + 16 + li) * 2);
                    nb = *reinterpret_cast<const f32x2 *>(u + (((f + 1) * 4 + g) * 32 + chh * 16 + li) * 2);
                }
                acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], b[0], acc[f][0], 0, 0, 0);
                acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[0], b[0], acc[f][1], 0, 0, 0);
                acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[1], b[1], acc[f][0], 0, 0, 0);
                acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[1], b[1], acc[f][1], 0, 0, 0);
                a0 = na0, a1 = na1, b = nb;
            }
            if (MODE < 4) __syncthreads();
        }
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
#pragma unroll
        for (int f = 0; f < 16; ++f) s += acc[f][0][0] + acc[f][1][3];
        out[blockIdx.x * 512 + tid] = s;
    } else {
        // ---- loader wave (256 threads): per chunk 11 global loads (b128), 12 LDS reads, 64 VALU, 15 LDS b128 writes
        const int lt = wave * 64 + lane;
        f32x4 gr[11];
#pragma unroll
        for (int j = 0; j < 11; ++j) gr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        float carry = 0.f;
        for (int it = 0; it < (MODE >= 4 ? 0 : iters); ++it) {
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 11; ++j) {
                    const size_t off = ((size_t)blockIdx.x * 65536 + (size_t)it * 11 * 1024 + j * 1024 + lt * 4) % src_elems;
                    gr[j] = *reinterpret_cast<const f32x4 *>(src + off);
                }
            }
            if (MODE == 1 || MODE == 2) {
                float *v = sV + ((it + 1) & 1) * 8192, *u = sU + ((it + 1) & 1) * 4096, *r = sR + (it & 1) * 2048;
                // raw patch: 3 stores, 12 reads
#pragma unroll
                for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4 *>(r + (j * 256 + lt) * 4) = gr[j];
                f32x4 p[12];
#pragma unroll
                for (int j = 0; j < 12; ++j) p[j] = *reinterpret_cast<const f32x4 *>(sR + ((it + 1) & 1) * 2048 + ((lt * 5 + j * 37) & 511) * 4);
                // transform: 64 VALU (adds), results in 8 float4
                f32x4 t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    t[j] = (p[j] - p[(j + 2) % 12]) + (p[(j + 1) % 12] + p[(j + 3) % 12]);
                    t[j] = t[j] + carry;
                }
                carry = t[0][0] * 1e-30f;
#pragma unroll
                for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4 *>(v + (j * 256 + lt) * 4) = t[j];
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4 *>(u + (j * 256 + lt) * 4) = gr[3 + j] + gr[7 + j];
            }
            __syncthreads();
        }
        out[blockIdx.x * 512 + tid] = carry + gr[0][0];
    }
    if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int ROLE>
static void run(const char *name, const float *src, size_t n)
{
    float *out;
    long long *cyc, h = 0;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    (void)hipMalloc(&cyc, 8);
    const int iters = 400;
    const size_t sh = (2 * 8192 + 2 * 4096 + 2 * 2048) * 4;
    (void)hipFuncSetAttribute((const void *)k<MODE, ROLE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL((k<MODE, ROLE>), dim3(256), dim3(512), sh, 0, out, cyc, 20, src, n);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, ROLE>), dim3(256), dim3(512), sh, 0, out, cyc, iters, src, n);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz: convert with the event time instead
    printf("%-44s %8.1f us  %7.1f ns per chunk  (memtime ticks %lld)  err=%s\n", name, ms * 1e3, ms * 1e6 / iters, h,
           hipGetErrorString(hipGetLastError()));
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main()
{
    const size_t n = 64u << 20;
    float *src;
    (void)hipMalloc(&src, n * 4);
    (void)hipMemset(src, 0, n * 4);
    // clock warm-up
    for (int i = 0; i < 3; ++i) run<0, 0>("warm-up", src, n);
    run<0, 0>("MFMA waves 0-3 alone", src, n);
    run<1, 0>("+ loader LDS / VALU", src, n);
    run<2, 0>("+ loader global loads", src, n);
    run<3, 0>("MFMA waves alone, no LDS reads", src, n);
    run<4, 0>("MFMA waves alone, no barrier", src, n);
    run<5, 0>("MFMA waves alone, no LDS reads, no barrier", src, n);
    run<0, 1>("MFMA = even waves, alone", src, n);
    run<1, 1>("+ loader LDS / VALU", src, n);
    run<2, 1>("+ loader global loads", src, n);
    run<0, 2>("MFMA = waves 0,1,4,5, alone", src, n);
    run<1, 2>("+ loader LDS / VALU", src, n);
    run<2, 2>("+ loader global loads", src, n);
    printf("ideal: 64 MFMAs x 32 cycles = 2048 cycles = %.0f ns at 2.1 GHz; the current kernel spends 5812 cycles per chunk on twice the channels\n",
           2048 / 2.1);
    return 0;
}
