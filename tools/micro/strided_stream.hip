// Developer micro-benchmark (gfx950): how fast can the A-operand access pattern of srf_conv1x1_nhwc_k stream from memory?
// Workgroup = 256 threads owns 128 rows of an (M, K) f32 matrix; per chunk of 32 columns thread (row = tid >> 3 + 32 j,
// quad = tid & 7) loads one float4 -- 8 lanes read one 128-byte line, a wave instruction 8 lines K*4 bytes apart.
// MODE 0: that pattern, chunk after chunk (DEPTH chunks in flight); MODE 1: the same bytes, but every thread walks ITS row
// contiguously (a wave reads 8 rows x 128 B as well, only the order of chunks differs: none); MODE 2: fully contiguous tile
// (the matrix stored tile-major): the plain streaming rate for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256, 3) void k(const float *__restrict__ A, long long M, int K, float *out)
{
    const int tid = threadIdx.x;
    const long long p0 = (long long)blockIdx.x * 128;
    const int nchunk = K / 32;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 r[DEPTH][4];
    auto addr = [&](int c, int j) -> const f32x4 * {
        if (MODE == 2) return reinterpret_cast<const f32x4 *>(A + ((long long)blockIdx.x * nchunk + c) * 4096 + (j * 256 + tid) * 4);
        return reinterpret_cast<const f32x4 *>(A + (p0 + (tid >> 3) + 32 * j) * K + c * 32 + (tid & 7) * 4);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) r[d][j] = *addr(d < nchunk ? d : nchunk - 1, j);
    for (int c = 0; c < nchunk; c += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += r[d][j];
            const int cn = c + d + DEPTH < nchunk ? c + d + DEPTH : nchunk - 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[d][j] = *addr(cn, j);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[blockIdx.x] = acc[0];
}

template <int MODE, int DEPTH>
static void run(const char *name, const float *A, long long M, int K, float *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int blocks = (int)(M / 128);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(blocks), dim3(256), 0, 0, A, M, K, out);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(blocks), dim3(256), 0, 0, A, M, K, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s M=%lld K=%d: %8.1f us  %7.1f GB/s  %s\n", name, M, K, ms * 1e3 / 5, (double)M * K * 4 / (ms / 5 * 1e-3) / 1e9,
           hipGetErrorString(hipGetLastError()));
}

int main()
{
    const long long M = 556800;
    float *A, *out;
    (void)hipMalloc(&A, (size_t)M * 1728 * 4);
    (void)hipMalloc(&out, 1 << 20);
    (void)hipMemset(A, 0, (size_t)M * 1728 * 4);
    for (int K : {256, 576, 768, 1728}) {
        run<0, 1>("row-strided lines, 1 chunk in flight", A, M, K, out);
        run<0, 2>("row-strided lines, 2 chunks in flight", A, M, K, out);
        run<0, 4>("row-strided lines, 4 chunks in flight", A, M, K, out);
        run<2, 2>("tile-major (contiguous 16 KB per chunk), 2 in flight", A, M, K, out);
        run<2, 4>("tile-major (contiguous 16 KB per chunk), 4 in flight", A, M, K, out);
    }
    return 0;
}
