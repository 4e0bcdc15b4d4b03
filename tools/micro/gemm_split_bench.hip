// gemm_split_bench.hip -- developer micro-benchmark: is an f32 GEMM through the bf16 MFMA worth building?
//
// Every f32 value is the EXACT sum of three bf16 values (8 + 8 + 8 significant bits, round-to-nearest at each step):
// x = xh + xm + xl.  A product a b is then the sum of nine bf16 x bf16 products, each exact in f32; the six of magnitude
// >= 2^-16 |a b| (hh, hm, mh, mm, hl, lh) are accumulated on v_mfma_f32_32x32x16_bf16 (f32 accumulate), the three dropped
// ones sum to < 2^-23 |a b|, the size of one f32 rounding.  The bf16 MFMA does 16x the FLOPs per clock of the f32 MFMA, so
// six products cost 6/16 of the f32 MFMA's cycles -- IF the operands can be fed.  This file measures exactly that for the
// 1x1-convolution GEMMs of the camera branch (y[p][co] = sum_k x[p][k] W[co][k], vovnet.py:222-223): A split on the fly from
// the f32 activations while it is staged to LDS, B pre-split once per layer; and the error against float64 next to the f32
// MFMA chain's.  Build: hipcc --offload-arch=gfx950 -O3 -o gemm_split_bench.bin gemm_split_bench.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cstring>

typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned pk_bf16(float a, float b)
{
    f2 v = {a, b};
    bf2 h = __builtin_convertvector(v, bf2);
    return *reinterpret_cast<unsigned *>(&h);
}

__device__ __forceinline__ float sub1(float a, float b)   // one v_sub_f32: keeps the compiler from SLP-packing into v_pk_add_f32
{
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int VAR>
__device__ __forceinline__ void split2v(float x0, float x1, unsigned &h, unsigned &m, unsigned &l)
{
    if (VAR == 0) {
        h = pk_bf16(x0, x1);
        const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
        m = pk_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
        l = pk_bf16(s0, s1);
    } else if (VAR == 4) {   // ablation: no split arithmetic (wrong results)
        h = __float_as_uint(x0);
        m = __float_as_uint(x1);
        l = h ^ m;
    } else {
        h = pk_bf16(x0, x1);
        const float r0 = sub1(x0, __uint_as_float(h << 16)), r1 = sub1(x1, __uint_as_float(h & 0xffff0000u));
        m = pk_bf16(r0, r1);
        const float s0 = sub1(r0, __uint_as_float(m << 16)), s1 = sub1(r1, __uint_as_float(m & 0xffff0000u));
        l = pk_bf16(s0, s1);
    }
}

// (x0, x1) -> packed (h, m, l) pairs
__device__ __forceinline__ void split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l)
{
    h = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = pk_bf16(s0, s1);
}

// B image per (chunk of 32 k, column tile of 128): [plane 3][col 128][slot 4][8 bf16], slot s of column n holds oct s ^ ((n >> 2) & 3)
__global__ __launch_bounds__(256) void pack_b(const float *__restrict__ W, int Cout, int K, int nct, unsigned short *__restrict__ P, long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int e = (int)(t & 7), s = (int)((t >> 3) & 3), n = (int)((t >> 5) & 127);
    long long rest = t >> 12;
    const int p = (int)(rest % 3);
    rest /= 3;
    const int ct = (int)(rest % nct), c = (int)(rest / nct);
    const int oct = s ^ ((n >> 2) & 3);
    const int k = c * 32 + oct * 8 + e, co = ct * 128 + n;
    const float x = co < Cout ? W[(size_t)co * K + k] : 0.f;
    unsigned h, m, l;
    split2(x, 0.f, h, m, l);
    P[t] = (unsigned short)((p == 0 ? h : p == 1 ? m : l) & 0xffffu);
}

template <int WPE, int VAR>
__global__ __launch_bounds__(256, WPE) void gemm_split(const float *__restrict__ x, long long M, int K, long long x_ld, const unsigned short *__restrict__ Bp,
                                                       int Cout, float *__restrict__ y, long long y_ld, int nct)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 24576];   // A planes | B planes
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int ct = jq % nct;
    const long long mblocks = (M + 127) / 128;
    const long long mb = (long long)(jq / nct) * 8 + xcd;
    if (mb >= mblocks) return;
    const long long p0 = mb * 128;
    const long long rows_here = M - p0 < 128 ? M - p0 : 128;
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x) + p0 * x_ld, 0, (int)(rows_here * x_ld * 4), 0x00020000);
    const int nchunk = K / 32;
    const size_t chunk_stride = (size_t)nct * 24576;   // bytes
    const unsigned char *bsrc = reinterpret_cast<const unsigned char *>(Bp) + (size_t)ct * 24576 + (size_t)tid * 16;

    const int q = tid & 7, r0 = tid >> 3;
    const unsigned aoff0 = (unsigned)((r0 * x_ld + q * 4) * 4), aoff_step = (unsigned)(32 * x_ld * 4);
    f4 araw[4];
    u4 braw[6];
    unsigned sp[3][4][2];
#define LOAD(C)                                                                                                            \
    do {                                                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                 \
            auto v_ = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(aoff0 + j_ * aoff_step), (C) * 128, 0);              \
            araw[j_] = *reinterpret_cast<f4 *>(&v_);                                                                       \
        }                                                                                                                  \
        const u4 *bb_ = reinterpret_cast<const u4 *>(bsrc + (size_t)(C) * chunk_stride);                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) braw[i_] = bb_[i_ * 256];                                         \
    } while (0)
#define SPLIT()                                                                                                            \
    do {                                                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                 \
            split2v<VAR>(araw[j_][0], araw[j_][1], sp[0][j_][0], sp[1][j_][0], sp[2][j_][0]);                              \
            split2v<VAR>(araw[j_][2], araw[j_][3], sp[0][j_][1], sp[1][j_][1], sp[2][j_][1]);                              \
        }                                                                                                                  \
    } while (0)
    // A slot of (row, oct) = oct ^ ((row >> 2) & 3); this thread's 4 floats are half `q & 1` of oct `q >> 1`
#define STORE()                                                                                                            \
    do {                                                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                 \
            const int row_ = r0 + 32 * j_;                                                                                 \
            const int off_ = row_ * 64 + (((q >> 1) ^ ((row_ >> 2) & 3)) << 4) + (q & 1) * 8;                              \
            _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) { const u2 w_ = {sp[p_][j_][0], sp[p_][j_][1]}; *reinterpret_cast<u2 *>(lds + p_ * 8192 + off_) = w_; } \
        }                                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) *reinterpret_cast<u4 *>(lds + 24576 + (tid + i_ * 256) * 16) = braw[i_]; \
    } while (0)

    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    int a_off[2], b_off[2], swz_a[2], swz_b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + li, col = wn * 64 + i * 32 + li;
        a_off[i] = row * 64;
        b_off[i] = 24576 + col * 64;
        swz_a[i] = (row >> 2) & 3;
        swz_b[i] = (col >> 2) & 3;
    }
    f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf8 fa[3][2], fb[3][2];
#define READ(S)                                                                                                            \
    do {                                                                                                                   \
        _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_)                                                                   \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                             \
                fa[p_][i_] = *reinterpret_cast<const bf8 *>(lds + p_ * 8192 + a_off[i_] + (((2 * (S) + lh) ^ swz_a[i_]) << 4)); \
                fb[p_][i_] = *reinterpret_cast<const bf8 *>(lds + p_ * 8192 + b_off[i_] + (((2 * (S) + lh) ^ swz_b[i_]) << 4)); \
            }                                                                                                              \
    } while (0)
#define MM(PA, PB)                                                                                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                   \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][i_], fb[PB][j_], acc[i_][j_], 0, 0, 0)
    // smallest terms first: l h, h l, m m, then m h, h m, then h h
#define MFMA6() do { MM(2, 0); MM(0, 2); MM(1, 1); MM(1, 0); MM(0, 1); MM(0, 0); } while (0)

    const int last = nchunk - 1;
    LOAD(0);
    SPLIT();
    STORE();
    LOAD(last < 1 ? last : 1);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int c2 = c + 2 < nchunk ? c + 2 : last;
        if (VAR != 5 || c == 0) READ(0);
        MFMA6();
        SPLIT();
        if (VAR != 5) READ(1);
        if (VAR == 2 || VAR == 3) {   // phase 1: the 12 fragment reads, then every MFMA followed by four of the split's vector instructions
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
            for (int i_ = 0; i_ < 24; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        MFMA6();
        if (VAR == 6) continue;
        if (VAR != 12 && VAR != 13) __syncthreads();
        if (VAR != 11) STORE();
        if (VAR != 10) LOAD(c2);
        if (VAR == 3) {   // phase 2: MFMAs spread over the stores and the loads
#pragma unroll
            for (int i_ = 0; i_ < 18; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 1);
            }
#pragma unroll
            for (int i_ = 0; i_ < 6; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x20, 2, 1);
            }
        }
        if (VAR != 12) __syncthreads();
    }
    // epilogue: lane = channel li of block j; accumulator register r = row (r & 3) + 8 (r >> 2) + 4 lh of block i
    const int co0 = ct * 128 + wn * 64 + li;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = p0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = co0 + j * 32;
                if (row < M && co < Cout) y[row * y_ld + co] = acc[i][j][r];
            }
        }
}

// ---- v5 (round 5): workgroup tile 128 x 256 -- wave tiles 64 x 128 (2 x 4 accumulator tiles = 128 registers), the A
// image split ONCE for 256 output channels (half the split arithmetic, the A loads and the A stores per MFMA; A re-reads of the launch
// halved), B = two adjacent 128-column images of the same packed layout (Cout % 256 == 0).  72 KB of LDS, ~290 registers: one
// workgroup per CU (launch bounds 256, 1).  Same k order and product order per output as the baseline: identical bits.
__global__ __launch_bounds__(256, 1) void gemm_split5(const float *__restrict__ x, long long M, int K, long long x_ld, const unsigned short *__restrict__ Bp,
                                                      int Cout, float *__restrict__ y, long long y_ld, int nct /* 128-column tiles */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // A planes 24 KB | B planes of two column tiles 48 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nct2 = nct / 2;
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int ct = jq % nct2;
    const long long mblocks = (M + 127) / 128;
    const long long mb = (long long)(jq / nct2) * 8 + xcd;
    if (mb >= mblocks) return;
    const long long p0 = mb * 128;
    const long long rows_here = M - p0 < 128 ? M - p0 : 128;
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x) + p0 * x_ld, 0, (int)(rows_here * x_ld * 4), 0x00020000);
    const int nchunk = K / 32;
    const size_t chunk_stride = (size_t)nct * 24576;
    const unsigned char *bsrc = reinterpret_cast<const unsigned char *>(Bp) + (size_t)(2 * ct) * 24576 + (size_t)tid * 16;
    const int q = tid & 7, r0 = tid >> 3;
    const unsigned aoff0 = (unsigned)((r0 * x_ld + q * 4) * 4), aoff_step = (unsigned)(32 * x_ld * 4);
    f4 araw[4];
    u4 braw[12];
    unsigned sp[3][4][2];
#define LOAD5(C)                                                                                                           \
    do {                                                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                 \
            auto v_ = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(aoff0 + j_ * aoff_step), (C) * 128, 0);              \
            araw[j_] = *reinterpret_cast<f4 *>(&v_);                                                                       \
        }                                                                                                                  \
        const u4 *bb_ = reinterpret_cast<const u4 *>(bsrc + (size_t)(C) * chunk_stride);                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) braw[i_] = bb_[i_ * 256];                                        \
    } while (0)
#define SPLIT5()                                                                                                           \
    do {                                                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                 \
            split2(araw[j_][0], araw[j_][1], sp[0][j_][0], sp[1][j_][0], sp[2][j_][0]);                                    \
            split2(araw[j_][2], araw[j_][3], sp[0][j_][1], sp[1][j_][1], sp[2][j_][1]);                                    \
        }                                                                                                                  \
    } while (0)
#define STORE5()                                                                                                           \
    do {                                                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                 \
            const int row_ = r0 + 32 * j_;                                                                                 \
            const int off_ = row_ * 64 + (((q >> 1) ^ ((row_ >> 2) & 3)) << 4) + (q & 1) * 8;                              \
            _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) { const u2 w_ = {sp[p_][j_][0], sp[p_][j_][1]}; *reinterpret_cast<u2 *>(lds + p_ * 8192 + off_) = w_; } \
        }                                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) *reinterpret_cast<u4 *>(lds + 24576 + (tid + i_ * 256) * 16) = braw[i_]; \
    } while (0)
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    int a_off[2], b_off[4], swz_a[2], swz_b[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + li;
        a_off[i] = row * 64;
        swz_a[i] = (row >> 2) & 3;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = wn * 128 + j * 32 + li;      // column of the 256-wide tile: image (col >> 7), column (col & 127) of it
        b_off[j] = 24576 + (col >> 7) * 24576 + (col & 127) * 64;
        swz_b[j] = (col >> 2) & 3;
    }
    f16v acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf8 fa[3][2], fb[3][4];
#define READ5(S)                                                                                                           \
    do {                                                                                                                   \
        _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) {                                                                 \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                               \
                fa[p_][i_] = *reinterpret_cast<const bf8 *>(lds + p_ * 8192 + a_off[i_] + (((2 * (S) + lh) ^ swz_a[i_]) << 4)); \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                               \
                fb[p_][j_] = *reinterpret_cast<const bf8 *>(lds + p_ * 8192 + b_off[j_] + (((2 * (S) + lh) ^ swz_b[j_]) << 4)); \
        }                                                                                                                  \
    } while (0)
#define MM5(PA, PB)                                                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                   \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][i_], fb[PB][j_], acc[i_][j_], 0, 0, 0)
#define MFMA65() do { MM5(2, 0); MM5(0, 2); MM5(1, 1); MM5(1, 0); MM5(0, 1); MM5(0, 0); } while (0)
    const int last = nchunk - 1;
    LOAD5(0);
    SPLIT5();
    STORE5();
    LOAD5(last < 1 ? last : 1);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int c2 = c + 2 < nchunk ? c + 2 : last;
        READ5(0);
        MFMA65();
        SPLIT5();
        READ5(1);
        MFMA65();
        __syncthreads();
        STORE5();
        LOAD5(c2);
        __syncthreads();
    }
    const int co0 = ct * 256 + wn * 128 + li;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = p0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + j * 32;
                if (row < M && co < Cout) y[row * y_ld + co] = acc[i][j][r];
            }
        }
}

// ---- v2: B never touches LDS (fragment-ordered image, L2 -> registers, reloaded in place one step ahead, as srf_gemm_direct_k);
// A split once per workgroup into a DOUBLE-buffered LDS image: one barrier per chunk, the stores of chunk c + 1 anywhere in chunk c
// B image per (chunk, column tile): [wn 2][step 2][j 2][plane 3][lane 64][8 bf16]
__global__ __launch_bounds__(256) void pack_b2(const float *__restrict__ W, int Cout, int K, int nct, unsigned short *__restrict__ P, long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int e = (int)(t & 7), lane = (int)((t >> 3) & 63);
    long long rest = t >> 9;
    const int p = (int)(rest % 3);
    rest /= 3;
    const int j = (int)(rest & 1), st = (int)((rest >> 1) & 1), wn = (int)((rest >> 2) & 1);
    rest >>= 3;
    const int ct = (int)(rest % nct), c = (int)(rest / nct);
    const int co = ct * 128 + wn * 64 + j * 32 + (lane & 31), k = c * 32 + st * 16 + (lane >> 5) * 8 + e;
    const float x = co < Cout ? W[(size_t)co * K + k] : 0.f;
    unsigned h, m, l;
    split2(x, 0.f, h, m, l);
    P[t] = (unsigned short)((p == 0 ? h : p == 1 ? m : l) & 0xffffu);
}

template <int VAR>
__global__ __launch_bounds__(256, 3) void gemm_split2(const float *__restrict__ x, long long M, int K, long long x_ld, const unsigned short *__restrict__ Bp,
                                                      int Cout, float *__restrict__ y, long long y_ld, int nct)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 24576];   // A planes, two stages
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int ct = jq % nct;
    const long long mblocks = (M + 127) / 128;
    const long long mb = (long long)(jq / nct) * 8 + xcd;
    if (mb >= mblocks) return;
    const long long p0 = mb * 128;
    const long long rows_here = M - p0 < 128 ? M - p0 : 128;
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x) + p0 * x_ld, 0, (int)(rows_here * x_ld * 4), 0x00020000);
    const int nchunk = K / 32;
    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const size_t chunk_stride = (size_t)nct * 24576;   // bytes
    __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(reinterpret_cast<const unsigned char *>(Bp)) + (size_t)ct * 24576 + (size_t)wn * 12288, 0,
        (int)((size_t)nchunk * chunk_stride - (size_t)ct * 24576 - (size_t)wn * 12288), 0x00020000);
    const int boff = lane * 16;

    const int q = tid & 7, r0 = tid >> 3;
    const unsigned aoff0 = (unsigned)((r0 * x_ld + q * 4) * 4), aoff_step = (unsigned)(32 * x_ld * 4);
    f4 araw[4];
    unsigned sp[3][4][2];
    bf8 fb[2][2][3];   // [step][j][plane]
#define LOADA(C)                                                                                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                     \
        auto v_ = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(aoff0 + j_ * aoff_step), (C) * 128, 0);                  \
        araw[j_] = *reinterpret_cast<f4 *>(&v_);                                                                           \
    }
#define LOADB(S, C)                                                                                                        \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                       \
        _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) {                                                                 \
            auto v_ = __builtin_amdgcn_raw_buffer_load_b128(br, boff, (int)((C) * chunk_stride) + (((S) * 2 + j_) * 3 + p_) * 1024, 0); \
            fb[S][j_][p_] = *reinterpret_cast<bf8 *>(&v_);                                                                 \
        }
#define SPLIT2()                                                                                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                     \
        split2v<VAR>(araw[j_][0], araw[j_][1], sp[0][j_][0], sp[1][j_][0], sp[2][j_][0]);                                  \
        split2v<VAR>(araw[j_][2], araw[j_][3], sp[0][j_][1], sp[1][j_][1], sp[2][j_][1]);                                  \
    }
#define STOREA(BUF)                                                                                                        \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                     \
        const int row_ = r0 + 32 * j_;                                                                                     \
        const int off_ = (BUF) * 24576 + row_ * 64 + (((q >> 1) ^ ((row_ >> 2) & 3)) << 4) + (q & 1) * 8;                  \
        _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) { const u2 w_ = {sp[p_][j_][0], sp[p_][j_][1]}; *reinterpret_cast<u2 *>(lds + p_ * 8192 + off_) = w_; } \
    }
    int a_off[2], swz_a[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + li;
        a_off[i] = row * 64;
        swz_a[i] = (row >> 2) & 3;
    }
    f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf8 fa[3][2];
#define READA(BUF, S)                                                                                                      \
    _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_)                                                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                                   \
            fa[p_][i_] = *reinterpret_cast<const bf8 *>(lds + (BUF) * 24576 + p_ * 8192 + a_off[i_] + (((2 * (S) + lh) ^ swz_a[i_]) << 4));
#define MM2(S, PA, PB)                                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                   \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][i_], fb[S][j_][PB], acc[i_][j_], 0, 0, 0)
#define MFMA6B(S) do { MM2(S, 2, 0); MM2(S, 0, 2); MM2(S, 1, 1); MM2(S, 1, 0); MM2(S, 0, 1); MM2(S, 0, 0); } while (0)

    const int last = nchunk - 1;
    LOADA(0);
    LOADB(0, 0);
    LOADB(1, 0);
    SPLIT2();
    STOREA(0);
    LOADA(last < 1 ? last : 1);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
        const int c1 = c + 1 < nchunk ? c + 1 : last, c2 = c + 2 < nchunk ? c + 2 : last;
        READA(cur, 0);
        MFMA6B(0);
        LOADB(0, c1);
        __builtin_amdgcn_sched_barrier(0);
        SPLIT2();
        READA(cur, 1);
        MFMA6B(1);
        LOADB(1, c1);
        STOREA(cur ^ 1);
        LOADA(c2);
        __syncthreads();
    }
    const int co0 = ct * 128 + wn * 64 + li;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = p0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = co0 + j * 32;
                if (row < M && co < Cout) y[row * y_ld + co] = acc[i][j][r];
            }
        }
}

// ---- v3: A never touches LDS.  A wave owns 32 rows x all 128 columns of the tile (1 x 4 accumulator tiles), so no other wave needs its A
// rows: lane (row, k half) loads its 8 floats of a k-step straight from the activation (two dwordx4) and splits them in registers -- every
// element is split exactly once per workgroup, as before, but the 24 KB A image, its 12 ds_write_b64 per thread and a third of the
// fragment reads are gone.  B as in v1 (pre-split image copied to LDS), double buffered: one barrier per chunk.
template <int WPE>
__global__ __launch_bounds__(256, WPE) void gemm_split3(const float *__restrict__ x, long long M, int K, long long x_ld, const unsigned short *__restrict__ Bp,
                                                        int Cout, float *__restrict__ y, long long y_ld, int nct)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 24576];   // B planes, two stages
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int ct = jq % nct;
    const long long mblocks = (M + 127) / 128;
    const long long mb = (long long)(jq / nct) * 8 + xcd;
    if (mb >= mblocks) return;
    const long long p0 = mb * 128;
    const long long rows_here = M - p0 < 128 ? M - p0 : 128;
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x) + p0 * x_ld, 0, (int)(rows_here * x_ld * 4), 0x00020000);
    const int nchunk = K / 32;
    const size_t chunk_stride = (size_t)nct * 24576;   // bytes
    const unsigned char *bsrc = reinterpret_cast<const unsigned char *>(Bp) + (size_t)ct * 24576 + (size_t)tid * 16;
    const int li = lane & 31, lh = lane >> 5;
    const unsigned aoff = (unsigned)(((wave * 32 + li) * x_ld + lh * 8) * 4);   // + chunk * 128 + step * 64 (+ 16)
    f4 araw[2][2];
    u4 braw[6];
#define LOADA3(C)                                                                                                          \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                                       \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                                 \
            auto v_ = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)aoff, (C) * 128 + s_ * 64 + h_ * 16, 0);              \
            araw[s_][h_] = *reinterpret_cast<f4 *>(&v_);                                                                   \
        }
#define LOADB3(C)                                                                                                          \
    do {                                                                                                                   \
        const u4 *bb_ = reinterpret_cast<const u4 *>(bsrc + (size_t)(C) * chunk_stride);                                   \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) braw[i_] = bb_[i_ * 256];                                         \
    } while (0)
#define STOREB3(BUF) _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) *reinterpret_cast<u4 *>(lds + (BUF) * 24576 + (tid + i_ * 256) * 16) = braw[i_];
    int b_off[4], swz_b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = j * 32 + li;
        b_off[j] = col * 64;
        swz_b[j] = (col >> 2) & 3;
    }
    f16v acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf8 pa[2][3];   // [step][plane]
#define SPLITA3()                                                                                                          \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                                     \
        unsigned h_[4], m_[4], l_[4];                                                                                      \
        split2v<1>(araw[s_][0][0], araw[s_][0][1], h_[0], m_[0], l_[0]);                                                   \
        split2v<1>(araw[s_][0][2], araw[s_][0][3], h_[1], m_[1], l_[1]);                                                   \
        split2v<1>(araw[s_][1][0], araw[s_][1][1], h_[2], m_[2], l_[2]);                                                   \
        split2v<1>(araw[s_][1][2], araw[s_][1][3], h_[3], m_[3], l_[3]);                                                   \
        const u4 vh_ = {h_[0], h_[1], h_[2], h_[3]}, vm_ = {m_[0], m_[1], m_[2], m_[3]}, vl_ = {l_[0], l_[1], l_[2], l_[3]}; \
        pa[s_][0] = *reinterpret_cast<const bf8 *>(&vh_);                                                                  \
        pa[s_][1] = *reinterpret_cast<const bf8 *>(&vm_);                                                                  \
        pa[s_][2] = *reinterpret_cast<const bf8 *>(&vl_);                                                                  \
    }
    const int last = nchunk - 1;
    LOADA3(0);
    LOADB3(0);
    STOREB3(0);
    SPLITA3();
    LOADA3(last < 1 ? last : 1);
    LOADB3(last < 1 ? last : 1);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
        const int c2 = c + 2 < nchunk ? c + 2 : last;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf8 fbj[3];
#pragma unroll
                for (int p_ = 0; p_ < 3; ++p_)
                    fbj[p_] = *reinterpret_cast<const bf8 *>(lds + cur * 24576 + p_ * 8192 + b_off[j] + (((2 * s + lh) ^ swz_b[j]) << 4));
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s][2], fbj[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s][0], fbj[2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s][1], fbj[1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s][1], fbj[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s][0], fbj[1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s][0], fbj[0], acc[j], 0, 0, 0);
            }
        STOREB3(cur ^ 1);     // chunk c + 1 (its last readers passed the barrier that ended chunk c - 1)
        SPLITA3();            // chunk c + 1's A planes (the MFMAs above have consumed chunk c's)
        LOADA3(c2);
        LOADB3(c2);
        __syncthreads();
    }
    const int co0 = ct * 128 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long row = p0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = co0 + j * 32;
            if (row < M && co < Cout) y[row * y_ld + co] = acc[j][r];
        }
    }
}

// ---- v4: v3 with the A loads TWO chunks ahead (two register sets, loop unrolled by two; two workgroups per CU for the registers)
__global__ __launch_bounds__(256, 2) void gemm_split4(const float *__restrict__ x, long long M, int K, long long x_ld, const unsigned short *__restrict__ Bp,
                                                     int Cout, float *__restrict__ y, long long y_ld, int nct)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 24576];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int ct = jq % nct;
    const long long mblocks = (M + 127) / 128;
    const long long mb = (long long)(jq / nct) * 8 + xcd;
    if (mb >= mblocks) return;
    const long long p0 = mb * 128;
    const long long rows_here = M - p0 < 128 ? M - p0 : 128;
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x) + p0 * x_ld, 0, (int)(rows_here * x_ld * 4), 0x00020000);
    const int nchunk = K / 32;
    const size_t chunk_stride = (size_t)nct * 24576;
    const unsigned char *bsrc = reinterpret_cast<const unsigned char *>(Bp) + (size_t)ct * 24576 + (size_t)tid * 16;
    const int li = lane & 31, lh = lane >> 5;
    const unsigned aoff = (unsigned)(((wave * 32 + li) * x_ld + lh * 8) * 4);
    f4 ara[2][2], arb[2][2];
    u4 braw[6];
#define LOADA4(R, C)                                                                                                       \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                                       \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                                 \
            auto v_ = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)aoff, (C) * 128 + s_ * 64 + h_ * 16, 0);              \
            R[s_][h_] = *reinterpret_cast<f4 *>(&v_);                                                                      \
        }
    int b_off[4], swz_b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = j * 32 + li;
        b_off[j] = col * 64;
        swz_b[j] = (col >> 2) & 3;
    }
    f16v acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf8 pa[2][3];
#define SPLITA4(R)                                                                                                         \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                                     \
        unsigned h_[4], m_[4], l_[4];                                                                                      \
        split2v<1>(R[s_][0][0], R[s_][0][1], h_[0], m_[0], l_[0]);                                                         \
        split2v<1>(R[s_][0][2], R[s_][0][3], h_[1], m_[1], l_[1]);                                                         \
        split2v<1>(R[s_][1][0], R[s_][1][1], h_[2], m_[2], l_[2]);                                                         \
        split2v<1>(R[s_][1][2], R[s_][1][3], h_[3], m_[3], l_[3]);                                                         \
        const u4 vh_ = {h_[0], h_[1], h_[2], h_[3]}, vm_ = {m_[0], m_[1], m_[2], m_[3]}, vl_ = {l_[0], l_[1], l_[2], l_[3]}; \
        pa[s_][0] = *reinterpret_cast<const bf8 *>(&vh_);                                                                  \
        pa[s_][1] = *reinterpret_cast<const bf8 *>(&vm_);                                                                  \
        pa[s_][2] = *reinterpret_cast<const bf8 *>(&vl_);                                                                  \
    }
#define MULT4(CUR)                                                                                                         \
    _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
            bf8 fbj[3];                                                                                                    \
            _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_)                                                               \
                fbj[p_] = *reinterpret_cast<const bf8 *>(lds + (CUR) * 24576 + p_ * 8192 + b_off[j] + (((2 * s + lh) ^ swz_b[j]) << 4)); \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s][2], fbj[0], acc[j], 0, 0, 0);                           \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s][0], fbj[2], acc[j], 0, 0, 0);                           \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s][1], fbj[1], acc[j], 0, 0, 0);                           \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s][1], fbj[0], acc[j], 0, 0, 0);                           \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s][0], fbj[1], acc[j], 0, 0, 0);                           \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s][0], fbj[0], acc[j], 0, 0, 0);                           \
        }
    const int last = nchunk - 1;
#define CL(C) ((C) < nchunk ? (C) : last)
    LOADA4(ara, 0);
    LOADB3(0);
    STOREB3(0);
    SPLITA4(ara);              // planes of chunk 0
    LOADA4(ara, CL(1));        // ara <- chunk 1
    LOADA4(arb, CL(2));        // arb <- chunk 2
    LOADB3(CL(1));
    __syncthreads();
    for (int c = 0; c < nchunk; c += 2) {
        // even chunk c: planes hold c; ara = c + 1, arb = c + 2; braw = B(c + 1)
        MULT4(0);
        STOREB3(1);
        SPLITA4(ara);          // planes <- c + 1
        LOADA4(ara, CL(c + 3));
        LOADB3(CL(c + 2));
        __syncthreads();
        if (c + 1 >= nchunk) break;
        // odd chunk c + 1: planes hold c + 1; arb = c + 2, ara = c + 3; braw = B(c + 2)
        MULT4(1);
        STOREB3(0);
        SPLITA4(arb);          // planes <- c + 2
        LOADA4(arb, CL(c + 4));
        LOADB3(CL(c + 3));
        __syncthreads();
    }
    const int co0 = ct * 128 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long row = p0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = co0 + j * 32;
            if (row < M && co < Cout) y[row * y_ld + co] = acc[j][r];
        }
    }
}

// the f32 MFMA chain for comparison of the error only (one wave per 32 x 32 outputs, operands from global memory)
__global__ __launch_bounds__(64) void gemm_f32_chain(const float *__restrict__ x, int K, long long x_ld, const float *__restrict__ W, float *__restrict__ y,
                                                     long long y_ld)
{
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    const long long r0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = 0; k < K; k += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[(r0 + li) * x_ld + k + lh], W[(size_t)(c0 + li) * K + k + lh], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) y[(r0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * y_ld + c0 + li] = acc[r];
}

int main(int argc, char **argv)
{
    struct Shape { long long M; int K, N; const char *what; };
    const Shape shapes[] = {{556800, 768, 256, "stage 2 concat (6 x 232 x 400)"}, {139200, 1312, 512, "stage 3 concat"},
                            {34800, 1728, 768, "stage 4 concat"}, {8700, 2144, 1024, "stage 5 concat"},
                            {556800, 256, 256, "finest FPN lateral"}};
    for (const Shape &s : shapes) {
        const long long M = s.M;
        const int K = s.K, N = s.N, nct = (N + 127) / 128;
        std::vector<float> hx((size_t)M * K), hw((size_t)N * K);
        unsigned seed = 12345u;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 65536.0f; };
        for (auto &v : hx) { const float u = rnd(); v = u < 0.45f ? 0.f : (u - 0.45f) * 3.7f; }   // post-ReLU-like: 45 % zeros
        for (auto &v : hw) v = (rnd() - 0.5f) * 0.08f;
        float *dx, *dw, *dy, *dy32;
        unsigned short *dp, *dp2;
        const size_t pbytes = (size_t)(K / 32) * nct * 24576;
        CK(hipMalloc(&dx, hx.size() * 4));
        CK(hipMalloc(&dw, hw.size() * 4));
        CK(hipMalloc(&dy, (size_t)M * N * 4));
        CK(hipMalloc(&dy32, (size_t)1024 * N * 4));
        CK(hipMalloc(&dp, pbytes));
        CK(hipMalloc(&dp2, pbytes));
        CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        const long long ptotal = (long long)(pbytes / 2);
        hipLaunchKernelGGL(pack_b, dim3((unsigned)((ptotal + 255) / 256)), dim3(256), 0, 0, dw, N, K, nct, dp, ptotal);
        hipLaunchKernelGGL(pack_b2, dim3((unsigned)((ptotal + 255) / 256)), dim3(256), 0, 0, dw, N, K, nct, dp2, ptotal);
        const long long mblocks = (M + 127) / 128;
        const unsigned grid = (unsigned)(((mblocks + 7) / 8) * 8 * nct);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const int NV = 17;
        double best[NV], med[NV][5];
        for (int v = 0; v < NV; ++v) best[v] = 1e30;
        auto launch = [&](int var) {
#define LV(V) hipLaunchKernelGGL((gemm_split<3, V>), dim3(grid), dim3(256), 0, 0, dx, M, K, (long long)K, dp, N, dy, (long long)N, nct)
            switch (var) {
            case 0: LV(0); break;
            case 1: LV(1); break;
            case 2: LV(2); break;
            case 3: LV(3); break;
            case 4: LV(4); break;
            case 5: LV(5); break;
            case 6: LV(6); break;
            case 13: hipLaunchKernelGGL((gemm_split3<3>), dim3(grid), dim3(256), 0, 0, dx, M, K, (long long)K, dp, N, dy, (long long)N, nct); break;
            case 14: hipLaunchKernelGGL((gemm_split3<2>), dim3(grid), dim3(256), 0, 0, dx, M, K, (long long)K, dp, N, dy, (long long)N, nct); break;
            case 15: hipLaunchKernelGGL(gemm_split4, dim3(grid), dim3(256), 0, 0, dx, M, K, (long long)K, dp, N, dy, (long long)N, nct); break;
            case 9: LV(10); break;
            case 10: LV(11); break;
            case 11: LV(12); break;
            case 12: LV(13); break;
            case 7: hipLaunchKernelGGL((gemm_split2<0>), dim3(grid), dim3(256), 0, 0, dx, M, K, (long long)K, dp2, N, dy, (long long)N, nct); break;
            case 8: hipLaunchKernelGGL((gemm_split2<1>), dim3(grid), dim3(256), 0, 0, dx, M, K, (long long)K, dp2, N, dy, (long long)N, nct); break;
            case 16: hipLaunchKernelGGL(gemm_split5, dim3((unsigned)(((mblocks + 7) / 8) * 8 * (nct / 2))), dim3(256), 3 * 24576, 0, dx, M, K, (long long)K, dp, N, dy, (long long)N, nct); break;
            }
        };
        CK(hipFuncSetAttribute((const void *)gemm_split5, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 24576));
        {   // v5 against the baseline: same chains, same bits
            std::vector<float> y0((size_t)2048 * N), y5((size_t)2048 * N);
            launch(0);
            CK(hipMemcpy(y0.data(), dy + (size_t)(M - 2048) * N, y0.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemset(dy, 0xff, (size_t)M * N * 4));
            launch(16);
            CK(hipMemcpy(y5.data(), dy + (size_t)(M - 2048) * N, y5.size() * 4, hipMemcpyDeviceToHost));
            printf("    v5 == baseline on the last 2048 rows: %s\n", memcmp(y0.data(), y5.data(), y0.size() * 4) == 0 ? "bitwise equal" : "DIFFERENT");
        }
        for (int i = 0; i < 6; ++i) launch(1);   // clocks settle
        for (int round = 0; round < 5; ++round)
            for (int var = 0; var < NV; ++var) {
                launch(var);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                const int reps = 4;
                for (int i = 0; i < reps; ++i) launch(var);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= reps;
                med[var][round] = ms;
                if (ms < best[var]) best[var] = ms;
            }
        const double fl = 2.0 * M * K * N;
        const char *names[NV] = {"baseline", "v_sub asm", "sched groups ph1", "sched groups ph1+2", "ABL no split", "ABL no frag reads", "ABL no stage", "v2 B direct, A 2-stage", "v2 + v_sub asm", "ABL no global loads", "ABL no LDS stores", "ABL no barriers", "ABL one barrier", "v3 A direct, B 2-stage (3/CU)", "v3 (2/CU)", "v4 = v3 + A two chunks ahead", "v5 tile 128 x 256, 1 workgroup / CU"};
        for (int var = 0; var < NV; ++var) {
            double m5[5];
            for (int i = 0; i < 5; ++i) m5[i] = med[var][i];
            for (int i = 0; i < 5; ++i) for (int j = i + 1; j < 5; ++j) if (m5[j] < m5[i]) { double t = m5[i]; m5[i] = m5[j]; m5[j] = t; }
            printf("%-28s K %5d N %5d  %-20s min %8.1f med %8.1f us  %7.1f TFLOP/s f32-eq (%.0f bf16 issued)\n", s.what, K, N, names[var], best[var] * 1e3,
                   m5[2] * 1e3, fl / m5[2] / 1e9, 6 * fl / m5[2] / 1e9);
        }
        launch(15);   // the error check below reads v4's output
        // error against float64 on the first 1024 rows, next to the f32 MFMA chain's
        hipLaunchKernelGGL(gemm_f32_chain, dim3(32, N / 32), dim3(64), 0, 0, dx, K, (long long)K, dw, dy32, (long long)N);
        std::vector<float> gy((size_t)1024 * N), gy32((size_t)1024 * N);
        CK(hipMemcpy(gy.data(), dy, gy.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gy32.data(), dy32, gy32.size() * 4, hipMemcpyDeviceToHost));
        double es = 0, e32 = 0, mx = 0, rs = 0, r32 = 0;
        for (int r = 0; r < 1024; r += 7)
            for (int c = 0; c < N; c += 3) {
                double ref = 0, mag = 0;
                for (int k = 0; k < K; ++k) {
                    const double p = (double)hx[(size_t)r * K + k] * (double)hw[(size_t)c * K + k];
                    ref += p;
                    mag += fabs(p);
                }
                const double a = fabs(gy[(size_t)r * N + c] - ref), b = fabs(gy32[(size_t)r * N + c] - ref);
                es = fmax(es, a);
                e32 = fmax(e32, b);
                rs = fmax(rs, a / mag);
                r32 = fmax(r32, b / mag);
                mx = fmax(mx, fabs(ref));
            }
        printf("    max |err| vs float64: split %.3e  f32 MFMA chain %.3e  (max |y| %.3f);  max err / sum|a b|: split %.2e  f32 chain %.2e\n", es, e32, mx, rs, r32);
        CK(hipFree(dx));
        CK(hipFree(dw));
        CK(hipFree(dy));
        CK(hipFree(dy32));
        CK(hipFree(dp));
        CK(hipFree(dp2));
    }
    return 0;
}
