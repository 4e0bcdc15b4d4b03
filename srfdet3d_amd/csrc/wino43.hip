// wino43.hip -- 3x3 / stride 1 convolutions as Winograd F(4x4, 3x3) on the f32 MFMA, channels-last (NHWC).
//
// Reference call sites: the same layers as srf_wino3x3 (conv.hip): the 3x3 Conv2d + BatchNorm2d + ReLU layers of VoVNet's
// OSA blocks (mmdet3d_plugin/models/backbones/vovnet.py:116-133, :180-216), the image FPN's output convolutions, the head's
// `img_convs` (mmdet3d_plugin/models/sparse_heads/srfdet_head.py:404-416) and SECONDCustom's dense blocks
// (mmdet3d_plugin/models/backbones/second_custom.py:41-63).
//
// Why: F(2x2, 3x3) multiplies 16 frequencies per 4 outputs (direct / 2.25), F(4x4, 3x3) 36 per 16 (direct / 4): 1.78x fewer
// MFMA FLOPs for the layers that are 60 % of the LC frame.
//
// Structure (two launches per layer; the accumulators of 36 frequencies are what shapes it):
//   1. srf_wino43_xform_k  (HBM-bound): every 6 x 6 input patch (tiles of 4 x 4 outputs, numbered row-major over all images)
//      goes through V = B^T d B in registers and is written in the OPERAND ORDER of the multiply kernel:
//      V[tile block of 32][chunk of 8 channels][frequency 36][k quad 2][tile 32][4] -- a (frequency, chunk) piece is 1 KB,
//      exactly what one wave reads with one buffer_load_dwordx4 and feeds to four MFMA k-steps.
//   2. srf_wino43_mm_k: one workgroup = 32 tiles x 64 output channels x all 36 frequencies.  The register file of a CU holds
//      exactly that: 36 x 32 x 64 accumulators = 1152 of its 2048 registers per lane.  4 waves, one per SIMD, wave w owns the
//      frequencies 9 w .. 9 w + 8 of BOTH 32-channel halves (18 accumulator tiles of v_mfma_f32_32x32x2_f32 = 288 registers).
//      Nothing is shared between the waves in the reduction loop, so there is no LDS staging and no barrier in it: A (V pieces)
//      and B (U pieces, packed the same way once per layer) go L2 -> registers with buffer loads whose chunk offset is a
//      scalar, one chunk (27 KB per wave) ahead, in place: the registers of frequency j are reloaded right after the MFMAs of
//      frequency j have issued.  Per chunk and wave: 72 MFMAs, 27 loads, no vector arithmetic at all (an f32 MFMA does not
//      overlap with vector instructions of its own wave).
//      Epilogue: the 36 frequencies of a (tile, channel) sit in four different waves, so the accumulators change hands
//      through LDS (X[f][register pair][lane][2], 144 KB, one 32-channel half at a time); every thread then owns whole
//      (tile, channel) elements, applies Y = A^T M A, scale / shift / ReLU and stores 128-byte channel runs.
// Numerics: f32 throughout.  B^T and A^T have the entries {0, +-1, +-2, +-4, +-5, 8}: exact scalings, each 1-D transform is
// written as the fma sequence below.  U = G g G^T is computed in double and rounded once.  The result differs from a direct
// convolution by f32 roundings amplified by the transforms: ~5e-6 (Cin = 64) to ~2e-5 (Cin = 1024) of the map's maximum
// (tests/test_wino43_emulation.py tabulates it per layer shape; tests/test_gpu_conv.py holds the kernel to 3e-5).
// Deterministic: every accumulator is one k-ascending fma chain, the transforms have a fixed operation order.
#include "common.hpp"
#include <algorithm>
#include <stdlib.h>

typedef float w43_f32x16 __attribute__((ext_vector_type(16)));
typedef float w43_f32x4 __attribute__((ext_vector_type(4)));
typedef float w43_v2 __attribute__((ext_vector_type(2)));

#define W43_TB 32                       // tiles per tile block
#define W43_PIECE 1024                  // bytes of one (frequency, chunk) operand piece: 2 k quads x 32 rows x 16 B
#define W43_X1_BYTES (36 * 8 * 64 * 4)   // NB = 1: eight registers per phase (two workgroups per CU)
#define W43_X_BYTES (36 * 16 * 64 * 4)  // epilogue exchange image: X[f 36][accumulator register 16][lane 64] floats

struct W43Args {
    const float *x;
    float *y;
    float *V;          // workspace: ntb * nchunk * 36 pieces
    const float *U;    // ncb * nchunk * 36 * 2 pieces
    const float *scale, *shift;
    long long x_ld, y_ld;
    int N, H, W, Cout;
    int tilesX, tilesY;
    long long ntiles;
    int ntb, nchunk, ncb, relu;
    int tb0, n0;       // this launch covers the tile blocks tb0 .. ntb - 1 of the layer (a slab); V holds them from its start;
                       // x and y point at image n0, the image of the slab's first tile (32-bit offsets from there)
#ifdef SRF_DEV
    long long *stamps;   // developer build only: 6 s_memtime values per workgroup of srf_wino43_mm_k
#endif
};
#ifdef SRF_DEV
#define W43_STAMP(I) if (a.stamps && tid == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); a.stamps[(size_t)blockIdx.x * 6 + (I)] = __builtin_amdgcn_s_memtime(); }
#define W43_STAMP_NW(I) if (a.stamps && tid == 0) { a.stamps[(size_t)blockIdx.x * 6 + (I)] = __builtin_amdgcn_s_memtime(); }
#else
#define W43_STAMP(I)
#define W43_STAMP_NW(I)
#endif

// ---------------------------------------------------------------------------------------------------------------------
// weights: W (Cout, Cin, 3, 3) -> U = G g G^T, G = [[1/4, 0, 0], [-1/6, -1/6, -1/6], [-1/6, 1/6, -1/6], [1/24, 1/12, 1/6],
// [1/24, -1/12, 1/6], [0, 0, 1]], in double, rounded once; layout U[cout block 64][chunk][f 36][half 2][quad 2][co 32][4];
// channels >= Cout are zero.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void srf_wino43_pack_k(const float *__restrict__ Wt, int Cout, int Cin, int nchunk, float *__restrict__ P,
                                                        long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int kk = (int)(t & 3), co = (int)((t >> 2) & 31), q = (int)((t >> 7) & 1), h = (int)((t >> 8) & 1);
    long long rest = t >> 9;
    const int f = (int)(rest % 36);
    rest /= 36;
    const int chunk = (int)(rest % nchunk), cb = (int)(rest / nchunk);
    const int cog = cb * 64 + h * 32 + co, ci = chunk * 8 + q * 4 + kk;
    float r = 0.f;
    if (cog < Cout) {
        const double G[6][3] = {{0.25, 0.0, 0.0},
                                {-1.0 / 6, -1.0 / 6, -1.0 / 6},
                                {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                {1.0 / 24, 1.0 / 12, 1.0 / 6},
                                {1.0 / 24, -1.0 / 12, 1.0 / 6},
                                {0.0, 0.0, 1.0}};
        const float *g = Wt + ((size_t)cog * Cin + ci) * 9;
        const int fr = f / 6, fs = f - fr * 6;
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) s += G[fr][a] * (double)g[a * 3 + b] * G[fs][b];
        r = (float)s;
    }
    P[t] = r;
}

// ---------------------------------------------------------------------------------------------------------------------
// 1-D input transform B^T x (6 -> 6) on float4 (4 channels):
//   v0 = 4 x0 - 5 x2 + x4          v1 = (x4 - 4 x2) + (x3 - 4 x1)    v2 = (x4 - 4 x2) - (x3 - 4 x1)
//   v3 = (x4 - x2) + 2 (x3 - x1)   v4 = (x4 - x2) - 2 (x3 - x1)      v5 = 4 x1 - 5 x3 + x5
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 w43_fma4(float s, float4 a, float4 b)
{
    return make_float4(__fmaf_rn(s, a.x, b.x), __fmaf_rn(s, a.y, b.y), __fmaf_rn(s, a.z, b.z), __fmaf_rn(s, a.w, b.w));
}
__device__ __forceinline__ float4 w43_add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 w43_sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

#define W43_BT(X0, X1, X2, X3, X4, X5, V0, V1, V2, V3, V4, V5)   \
    do {                                                         \
        const float4 a_ = w43_fma4(-4.f, X2, X4);                \
        const float4 b_ = w43_fma4(-4.f, X1, X3);                \
        const float4 c_ = w43_sub4(X4, X2);                      \
        const float4 e_ = w43_sub4(X3, X1);                      \
        const float4 v0_ = w43_fma4(4.f, X0, w43_fma4(-5.f, X2, X4)); \
        const float4 v5_ = w43_fma4(4.f, X1, w43_fma4(-5.f, X3, X5)); \
        V0 = v0_;                                                \
        V1 = w43_add4(a_, b_);                                   \
        V2 = w43_sub4(a_, b_);                                   \
        V3 = w43_fma4(2.f, e_, c_);                              \
        V4 = w43_fma4(-2.f, e_, c_);                             \
        V5 = v5_;                                                \
    } while (0)

// One workgroup = one tile block x 4 chunks (32 channels).  Wave w owns the tiles 8 w .. 8 w + 7, lane = tile * 8 + channel quad:
// the 8 lanes of a tile read one whole 128-byte line per pixel (a lane per (tile, chunk) touched 32 lines per load
// instruction and ran at 4.4 TB/s whether V went to HBM or stayed in the Infinity Cache: bound by the line lookups).  Loads go
// through one buffer descriptor over the slab's images -- a pixel outside its image is an out-of-range offset and reads as
// zero (the convolution's padding).  Every thread transforms its 36 pixels x 4 channels in registers and writes 36 float4;
// the 8 tiles x 2 quads of a chunk form two 128-byte runs of that chunk's 1 KB piece.
// Measured (128 channels @ 6 x 232 x 400: 285 MB in, 641 MB out), every thread loading its whole patch: 194 us = 4.8 TB/s; stores
// alone 94 us (6.8 TB/s), loads alone 77 us.  Not changed by: a float2-per-lane form at four waves per SIMD instead of two (187 us), nontemporal stores, writing V
// into a 75 MB slab that stays in the Infinity Cache -- the read/write mix on HBM is what bounds it.
__global__ __launch_bounds__(256) void srf_wino43_xform_k(W43Args a)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cgroups = (a.nchunk + 3) >> 2;
    const int tbl = blockIdx.x / cgroups, cg = blockIdx.x - tbl * cgroups;   // tile block within the slab
    const int qd = lane & 7, t = wave * 8 + (lane >> 3);
    const int c = cg * 4 + (qd >> 1), q = qd & 1;
    const bool c_ok = c < a.nchunk;
    const int g = (a.tb0 + tbl) * W43_TB + t;  // ntiles < 2^31 (checked by the host)
    const int per_img = a.tilesX * a.tilesY;
    const bool live = g < (int)a.ntiles;
    const int n = live ? g / per_img : a.n0;
    const int rem = live ? g - n * per_img : 0;
    const int ty = rem / a.tilesX, tx = rem - ty * a.tilesX;
    const long long img_b = (long long)a.H * a.W * a.x_ld * 4;
    const long long left_b = (long long)(a.N - a.n0) * img_b;
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0,
                                                                 (int)(unsigned)(left_b < 0xFFFFFFF0ll ? left_b : 0xFFFFFFF0ll), 0x00020000);
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    const unsigned px_b = (unsigned)(a.x_ld * 4);
    const unsigned chan_b = (unsigned)((c * 8 + q * 4) * 4) + (unsigned)(n - a.n0) * (unsigned)img_b;
    // Horizontally adjacent tiles share two of their six patch columns: when the next tile of the wave (lane + 8: same channel
    // quad) is this tile's right neighbour in the same image row, columns 4 and 5 are ITS columns 0 and 1 -- taken from its
    // registers (48 ds_bpermute per thread) instead of from memory: 24 + ~1.5 loads per thread instead of 36 (the vertical halo
    // between tile rows lives in other workgroups and is still read twice).
    const bool share = (lane < 56) && live && (g + 1 < (int)a.ntiles) && (tx + 1 < a.tilesX);
    float4 d[6][6];
#pragma unroll
    for (int py = 0; py < 6; ++py) {
        const int y = y0 + py;
        const bool yok = live && c_ok && y >= 0 && y < a.H;
        const unsigned row_b = (unsigned)(y * a.W) * px_b + chan_b;
#pragma unroll
        for (int px = 0; px < 6; ++px) {
            const int x = x0 + px;
            const bool ok = yok && x >= 0 && x < a.W && !(px >= 4 && share);
            const unsigned off = ok ? row_b + (unsigned)x * px_b : 0xFFFFFFF8u;
            auto v_ = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)off, 0, 0);
            d[py][px] = *reinterpret_cast<float4 *>(&v_);
        }
    }
#pragma unroll
    for (int py = 0; py < 6; ++py)
#pragma unroll
        for (int px = 4; px < 6; ++px) {
            const float4 nb = d[py][px - 4];
            const float sx = __shfl_down(nb.x, 8), sy = __shfl_down(nb.y, 8), sz = __shfl_down(nb.z, 8), sw = __shfl_down(nb.w, 8);
            if (share) d[py][px] = make_float4(sx, sy, sz, sw);
        }
    if (!c_ok) return;
    // vertical stage (over py) per column, in place
#pragma unroll
    for (int px = 0; px < 6; ++px)
        W43_BT(d[0][px], d[1][px], d[2][px], d[3][px], d[4][px], d[5][px], d[0][px], d[1][px], d[2][px], d[3][px], d[4][px], d[5][px]);
    // horizontal stage per frequency row + stores
    float4 *vp = reinterpret_cast<float4 *>(a.V) + ((size_t)tbl * a.nchunk + c) * (36 * 64) + q * 32 + t;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        float4 v0, v1, v2, v3, v4, v5;
        W43_BT(d[r][0], d[r][1], d[r][2], d[r][3], d[r][4], d[r][5], v0, v1, v2, v3, v4, v5);
        vp[(r * 6 + 0) * 64] = v0;
        vp[(r * 6 + 1) * 64] = v1;
        vp[(r * 6 + 2) * 64] = v2;
        vp[(r * 6 + 3) * 64] = v3;
        vp[(r * 6 + 4) * 64] = v4;
        vp[(r * 6 + 5) * 64] = v5;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// packed f32 pairs for the output transform (two (tile, channel) elements per thread and step)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ w43_v2 w43_pk_add(w43_v2 a, w43_v2 b)
{
    w43_v2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ w43_v2 w43_pk_sub(w43_v2 a, w43_v2 b)
{
    w43_v2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ w43_v2 w43_pk_fma(w43_v2 s, w43_v2 a, w43_v2 b)   // s a + b, one rounding per component
{
    w43_v2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(s), "v"(a), "v"(b));
    return r;
}
// 1-D output transform A^T m (6 -> 4): z0 = (m0 + (m1 + m2)) + (m3 + m4), z1 = (m1 - m2) + 2 (m3 - m4),
// z2 = (m1 + m2) + 4 (m3 + m4), z3 = ((m1 - m2) + 8 (m3 - m4)) + m5
#define W43_AT(M0, M1, M2, M3, M4, M5, Z0, Z1, Z2, Z3)           \
    do {                                                         \
        const w43_v2 s1_ = w43_pk_add(M1, M2), d1_ = w43_pk_sub(M1, M2); \
        const w43_v2 s2_ = w43_pk_add(M3, M4), d2_ = w43_pk_sub(M3, M4); \
        Z0 = w43_pk_add(w43_pk_add(M0, s1_), s2_);               \
        Z1 = w43_pk_fma(k2_, d2_, d1_);                          \
        Z2 = w43_pk_fma(k4_, s2_, s1_);                          \
        Z3 = w43_pk_add(w43_pk_fma(k8_, d2_, d1_), M5);          \
    } while (0)

#define W43_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, ACC, 0, 0, 0)

// NB = 32-channel halves per workgroup (2: the full 64-channel block; 1: one half -- the last block of a layer whose channel
// count leaves it at most half full, and small maps that would otherwise fill few CUs).  `hb` = index of the 32-channel half.
template <int NB>
__device__ __forceinline__ void srf_wino43_mm_body(const W43Args &a, const int tb, const int hb)
{
    extern __shared__ __attribute__((aligned(16))) float s_x[];   // epilogue only
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int cb = hb >> 1, h0 = hb & 1;
    const int nchunk = a.nchunk;
    // operand streams: piece (chunk c, frequency f) of V at ((tb nchunk + c) 36 + f) KB; of U at (((cb nchunk + c) 36 + f) 2 + h) KB
    __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.V) + (size_t)(tb - a.tb0) * nchunk * (36 * 256), 0, (int)((long long)nchunk * 36 * W43_PIECE), 0x00020000);
    __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.U) + (size_t)cb * nchunk * (36 * 2 * 256), 0, (int)((long long)nchunk * 36 * 2 * W43_PIECE), 0x00020000);
    const int voff = lane * 16;
    const int fbase = wave * 9;

    w43_f32x16 acc[9][NB];
    w43_f32x4 fa[9], fb[9][NB];
#define W43_LOAD(J, C)                                                                                                   \
    do {                                                                                                                 \
        const int sa_ = ((C) * 36 + fbase + (J)) * W43_PIECE;                                                            \
        auto va_ = __builtin_amdgcn_raw_buffer_load_b128(vr, voff, sa_, 0);                                              \
        fa[J] = *reinterpret_cast<w43_f32x4 *>(&va_);                                                                    \
        _Pragma("unroll") for (int h_ = 0; h_ < NB; ++h_) {                                                              \
            const int sb_ = (((C) * 36 + fbase + (J)) * 2 + (NB == 2 ? h_ : h0)) * W43_PIECE;                            \
            auto vb_ = __builtin_amdgcn_raw_buffer_load_b128(ur, voff, sb_, 0);                                          \
            fb[J][h_] = *reinterpret_cast<w43_f32x4 *>(&vb_);                                                            \
        }                                                                                                                \
    } while (0)
// 18 accumulator tiles = 288 registers: 16 tiles fill the 256 AGPRs, the two tiles of the wave's last frequency live in
// VGPRs.  The compiler gives every MFMA of a kernel the same accumulator register class and, left alone, moved those two
// tiles to AGPRs and back around their MFMAs in every chunk (128 v_accvgpr copies + a drain of the MFMA pipe); their MFMAs
// are therefore written as inline assembly with VGPR accumulators ("+v").  A dependent MFMA on the same accumulator needs
// no wait states; the epilogue's first read of them sits behind explicit s_nops (the compiler does not see an MFMA here).
#define W43_MFMA_V(ACC, A, B) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define W43_MFMA_V0(ACC, A, B) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(ACC) : "v"(A), "v"(B))
#define W43_GROUP(J, FIRST)                                                                                              \
    do {                                                                                                                 \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                                                 \
            _Pragma("unroll") for (int h_ = 0; h_ < NB; ++h_) {                                                          \
                if (NB == 2 && (J) == 8) {                                                                               \
                    if ((FIRST) && s_ == 0) W43_MFMA_V0(acc[J][h_], fa[J][s_], fb[J][h_][s_]);                           \
                    else W43_MFMA_V(acc[J][h_], fa[J][s_], fb[J][h_][s_]);                                               \
                } else if ((FIRST) && s_ == 0) {                                                                         \
                    const w43_f32x16 z_ = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; \
                    acc[J][h_] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[J][s_], fb[J][h_][s_], z_, 0, 0, 0);            \
                } else                                                                                                   \
                    W43_MFMA(acc[J][h_], fa[J][s_], fb[J][h_][s_]);                                                      \
            }                                                                                                            \
    } while (0)

    W43_STAMP_NW(0);
#pragma unroll
    for (int j = 0; j < 9; ++j) W43_LOAD(j, 0);
    W43_STAMP(1);
    // chunk 0 .. nchunk - 2: multiply chunk c, reload every frequency's registers with chunk c + 1 right behind its MFMAs
    if (nchunk > 1) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            W43_GROUP(j, 1);
            W43_LOAD(j, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int c = 1; c < nchunk - 1; ++c) {
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                W43_GROUP(j, 0);
                W43_LOAD(j, c + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) W43_GROUP(j, 0);
    } else {
#pragma unroll
        for (int j = 0; j < 9; ++j) W43_GROUP(j, 1);
    }

    // ---- epilogue ----
    W43_STAMP_NW(2);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the inline-assembly MFMAs have written their accumulators
    const w43_v2 k2_ = {2.f, 2.f}, k4_ = {4.f, 4.f}, k8_ = {8.f, 8.f};
    float *X = s_x;
    const int per_img = a.tilesX * a.tilesY;
    const unsigned ypx_b = (unsigned)(a.y_ld * 4), yrow_b = (unsigned)a.W * ypx_b;
    const long long yleft_b = (long long)(a.N - a.n0) * a.H * a.W * a.y_ld * 4;
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(unsigned)(yleft_b < 0xFFFFFFF0ll ? yleft_b : 0xFFFFFFF0ll), 0x00020000);
    if constexpr (NB == 1) {
        // Two workgroups per CU (the other one multiplies while this one is here), so the exchange image may take half of the
        // LDS: X[f 36][8 accumulator registers][lane 64], the 16 registers in two phases.  Reader thread (er = tid / 32,
        // ep = tid % 32): register 8 ph + er, lanes 2 ep and 2 ep + 1 = one tile x two consecutive channels as one packed pair;
        // one ds_read_b64 per frequency, 8-byte stores (16 lanes = one 128-byte channel run of a pixel).
        const w43_v2 *X2 = reinterpret_cast<const w43_v2 *>(s_x);
        const int er = tid >> 5, ep = tid & 31;
        const int co = cb * 64 + h0 * 32 + 2 * (ep & 15);
        const bool co_ok = co < a.Cout;
        w43_v2 sc = {1.f, 1.f}, sh = {0.f, 0.f};
        if (co_ok && a.scale) sc = *reinterpret_cast<const w43_v2 *>(a.scale + co);
        if (co_ok && a.shift) sh = *reinterpret_cast<const w43_v2 *>(a.shift + co);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int r = ph * 8 + er;
            const int et = (r & 3) + 8 * (r >> 2) + 4 * (ep >> 4);
            const int eg = tb * W43_TB + et;
            const bool elive = eg < (int)a.ntiles;
            const int en = elive ? eg / per_img : a.n0;
            const int erem = elive ? eg - en * per_img : 0;
            const int ety = erem / a.tilesX, etx = erem - ety * a.tilesX;
            const int rows_ok = elive ? min(4, a.H - 4 * ety) : 0, cols_ok = min(4, a.W - 4 * etx);
            const unsigned ebase = (unsigned)(en - a.n0) * ((unsigned)a.H * yrow_b) + (unsigned)(4 * ety) * yrow_b + (unsigned)(4 * etx) * ypx_b;
            __syncthreads();   // the readers of the previous phase are done
#pragma unroll
            for (int j = 0; j < 9; ++j)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) X[((fbase + j) * 8 + rr) * 64 + lane] = acc[j][0][ph * 8 + rr];
            __syncthreads();
            if (ph == 0) W43_STAMP_NW(3);
            w43_v2 z[4][6];
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const w43_v2 m0 = X2[((0 * 6 + s) * 8 + er) * 32 + ep], m1 = X2[((1 * 6 + s) * 8 + er) * 32 + ep],
                             m2 = X2[((2 * 6 + s) * 8 + er) * 32 + ep], m3 = X2[((3 * 6 + s) * 8 + er) * 32 + ep],
                             m4 = X2[((4 * 6 + s) * 8 + er) * 32 + ep], m5 = X2[((5 * 6 + s) * 8 + er) * 32 + ep];
                W43_AT(m0, m1, m2, m3, m4, m5, z[0][s], z[1][s], z[2][s], z[3][s]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                w43_v2 o[4];
                W43_AT(z[i][0], z[i][1], z[i][2], z[i][3], z[i][4], z[i][5], o[0], o[1], o[2], o[3]);
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    w43_v2 v = w43_pk_fma(o[jx], sc, sh);
                    if (a.relu) {
                        v[0] = fmaxf(v[0], 0.f);
                        v[1] = fmaxf(v[1], 0.f);
                    }
                    const bool ok = co_ok && i < rows_ok && jx < cols_ok;
                    const unsigned off = ok ? ebase + (unsigned)i * yrow_b + (unsigned)jx * ypx_b + (unsigned)co * 4u : 0xFFFFFFF8u;
                    typedef unsigned w43_u2 __attribute__((ext_vector_type(2)));
                    const w43_u2 bits = {__float_as_uint(v[0]), __float_as_uint(v[1])};
                    __builtin_amdgcn_raw_buffer_store_b64(bits, yr, (int)off, 0, 0);
                }
            }
            if (ph == 0) W43_STAMP_NW(4);
        }
    } else {
    // Exchange image X[f 36][accumulator register 16][lane 64] floats of one 32-channel half.  Writers: every wave, its 9
    // frequencies, one ds_write_b32 per register (lanes consecutive).  Readers: thread (er = tid / 16, eq = tid % 16) takes the
    // four consecutive lanes 4 eq .. 4 eq + 3 of register er = one tile x four consecutive channels, one ds_read_b128 per
    // frequency (a wave reads 1 KB contiguous), transforms them as two packed pairs and stores 16 pixels x 16 bytes.
    const float4 *X4 = reinterpret_cast<const float4 *>(s_x);
    const int er = tid >> 4, eq = tid & 15;
    const int et = (er & 3) + 8 * (er >> 2) + 4 * (eq >> 3);   // tile behind accumulator register er in lane half eq / 8
    const int co4 = (eq & 7) * 4;
    const int eg = tb * W43_TB + et;
    const bool elive = eg < (int)a.ntiles;
    const int en = elive ? eg / per_img : a.n0;
    const int erem = elive ? eg - en * per_img : 0;
    const int ety = erem / a.tilesX, etx = erem - ety * a.tilesX;
    const int rows_ok = elive ? min(4, a.H - 4 * ety) : 0, cols_ok = min(4, a.W - 4 * etx);
    const unsigned ebase = (unsigned)(en - a.n0) * ((unsigned)a.H * yrow_b) + (unsigned)(4 * ety) * yrow_b + (unsigned)(4 * etx) * ypx_b;
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        __syncthreads();   // the readers of the previous half are done
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) X[((fbase + j) * 16 + r) * 64 + lane] = acc[j][p][r];
        __syncthreads();
        if (p == 0) W43_STAMP_NW(3);
        const int co = cb * 64 + (NB == 2 ? p : h0) * 32 + co4;   // Cout % 4 == 0: a quad is all inside or all outside
        const bool co_ok = co < a.Cout;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (co_ok && a.scale) sc = *reinterpret_cast<const float4 *>(a.scale + co);
        if (co_ok && a.shift) sh = *reinterpret_cast<const float4 *>(a.shift + co);
        const w43_v2 scl = {sc.x, sc.y}, sch = {sc.z, sc.w}, shl = {sh.x, sh.y}, shh = {sh.z, sh.w};
        w43_v2 zl[4][6], zh[4][6];
        // column stage: the six frequency rows of column s -> Z[i][s]
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            float4 m[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) m[r] = X4[((r * 6 + s) * 16 + er) * 16 + eq];
            const w43_v2 l0 = {m[0].x, m[0].y}, l1 = {m[1].x, m[1].y}, l2 = {m[2].x, m[2].y}, l3 = {m[3].x, m[3].y}, l4 = {m[4].x, m[4].y},
                         l5 = {m[5].x, m[5].y};
            const w43_v2 h0_ = {m[0].z, m[0].w}, h1 = {m[1].z, m[1].w}, h2 = {m[2].z, m[2].w}, h3 = {m[3].z, m[3].w}, h4 = {m[4].z, m[4].w},
                         h5 = {m[5].z, m[5].w};
            W43_AT(l0, l1, l2, l3, l4, l5, zl[0][s], zl[1][s], zl[2][s], zl[3][s]);
            W43_AT(h0_, h1, h2, h3, h4, h5, zh[0][s], zh[1][s], zh[2][s], zh[3][s]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w43_v2 ol[4], oh[4];
            W43_AT(zl[i][0], zl[i][1], zl[i][2], zl[i][3], zl[i][4], zl[i][5], ol[0], ol[1], ol[2], ol[3]);
            W43_AT(zh[i][0], zh[i][1], zh[i][2], zh[i][3], zh[i][4], zh[i][5], oh[0], oh[1], oh[2], oh[3]);
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                w43_v2 vl = w43_pk_fma(ol[jx], scl, shl), vh = w43_pk_fma(oh[jx], sch, shh);
                if (a.relu) {
                    vl[0] = fmaxf(vl[0], 0.f);
                    vl[1] = fmaxf(vl[1], 0.f);
                    vh[0] = fmaxf(vh[0], 0.f);
                    vh[1] = fmaxf(vh[1], 0.f);
                }
                const bool ok = co_ok && i < rows_ok && jx < cols_ok;
                const unsigned off = ok ? ebase + (unsigned)i * yrow_b + (unsigned)jx * ypx_b + (unsigned)co * 4u : 0xFFFFFFF8u;
                typedef unsigned w43_u4 __attribute__((ext_vector_type(4)));
                const w43_u4 bits = {__float_as_uint(vl[0]), __float_as_uint(vl[1]), __float_as_uint(vh[0]), __float_as_uint(vh[1])};
                __builtin_amdgcn_raw_buffer_store_b128(bits, yr, (int)off, 0, 0);
            }
        }
        if (p == 0) W43_STAMP_NW(4);
    }
    }
    W43_STAMP(5);
}

// work item -> (half-block index, tile block): items b and b + 8 share an XCD (round-robin dispatch), so the channel blocks of
// one tile block sit on one L2 and read its V pieces together
template <int NB>
__global__ __launch_bounds__(256, NB == 1 ? 2 : 1) void srf_wino43_mm_k(W43Args a, int nitems_cb)
{
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int cbi = jq % nitems_cb;
    const int tb = a.tb0 + (jq / nitems_cb) * 8 + xcd;
    if (tb >= a.ntb) return;
    srf_wino43_mm_body<NB>(a, tb, NB == 2 ? 2 * cbi : cbi);
}

extern "C" size_t srf_wino43_packed_weight_bytes(int Cout, int Cin)
{
    if (Cout <= 0 || Cin <= 0 || (Cin & 7)) return 0;
    return (size_t)srf_ceil_div(Cout, 64) * (Cin / 8) * 36 * 2 * W43_PIECE;
}

extern "C" int srf_wino43_pack_weights(const float *W, int Cout, int Cin, float *packed, srf_stream_t stream)
{
    if (Cout <= 0 || Cin <= 0 || !W || !packed) return SRF_EINVAL;
    if (Cin & 7) return SRF_EUNSUPPORTED;
    const long long total = (long long)(srf_wino43_packed_weight_bytes(Cout, Cin) / 4);
    hipLaunchKernelGGL(srf_wino43_pack_k, dim3((unsigned)srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, W, Cout, Cin, Cin / 8,
                       packed, total);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// Slabs.  V is 2.25x the layer's input (1.3 GB for 256 channels on six 232 x 400 maps) and is addressed through 32-bit buffer
// descriptors: a layer whose V would reach 2 GB runs in slabs of tile blocks -- transform slab i -> multiply slab i -- in ONE
// slab-sized workspace that every slab overwrites.  (Round 3 also tried slabs small enough for V to stay in the 256 MB Infinity
// Cache: no gain, the transform runs at the rate of its read / write mix either way, so the cut is only about the descriptor
// range.)  A slab is sized to whole rounds of workgroups (one per CU).
#define W43_SLAB_BYTES (2048ll << 20)

static long long w43_slab_tb(long long ntb, int nchunk, int ncb)
{
    static const long long forced = getenv("SRF_W43_SLAB_TB") ? atoll(getenv("SRF_W43_SLAB_TB")) : 0;   // developer A/B knob, read once
    if (forced > 0) return forced < ntb ? forced : ntb;
    const long long per_tb = (long long)nchunk * 36 * W43_PIECE;
    long long fit = W43_SLAB_BYTES / per_tb;
    if (fit < 8) fit = 8;
    if (fit >= ntb) return ntb;
    // whole rounds of 256 workgroups: tile blocks per round = 256 / ncb
    const long long round = (256 + ncb - 1) / ncb;
    long long k = fit / round;
    if (k < 1) k = 1;
    long long slab = k * round;
    slab = (slab + 7) / 8 * 8;
    return slab < ntb ? slab : ntb;
}

extern "C" size_t srf_wino43_workspace_bytes(int N, int H, int W, int Cin, int Cout)
{
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 7)) return 0;
    const long long ntiles = (long long)N * srf_ceil_div(H, 4) * srf_ceil_div(W, 4);
    const long long ntb = (ntiles + W43_TB - 1) / W43_TB;
    return (size_t)w43_slab_tb(ntb, Cin / 8, srf_ceil_div(Cout, 64)) * (Cin / 8) * 36 * W43_PIECE;
}

// argument block shared by srf_wino43 and the developer bench (tools/micro/wino43_bench.hip)
#define W43_NEED_X 1   /* the transform kernel's operands are checked */
#define W43_NEED_Y 2   /* the multiply kernel's operands are checked */
static int w43_make_args(W43Args &a, const float *x, int N, int H, int W, int Cin, long long x_ld, const float *U_packed, int Cout,
                         const float *scale, const float *shift, int relu, float *y, long long y_ld, void *workspace, size_t workspace_bytes,
                         int need = W43_NEED_X | W43_NEED_Y)
{
    if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SRF_EINVAL;
    if (((need & W43_NEED_X) && x_ld < Cin) || ((need & W43_NEED_Y) && y_ld < Cout)) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!workspace || ((need & W43_NEED_X) && !x) || ((need & W43_NEED_Y) && (!U_packed || !y))) return SRF_EINVAL;
    if ((Cin & 7) || (Cout & 3) || ((uintptr_t)workspace & 15)) return SRF_EUNSUPPORTED;
    if ((need & W43_NEED_X) && ((x_ld & 3) || ((uintptr_t)x & 15))) return SRF_EUNSUPPORTED;
    if ((need & W43_NEED_Y) && ((y_ld & 3) || ((uintptr_t)y & 15) || ((uintptr_t)U_packed & 15) || (scale && ((uintptr_t)scale & 15)) ||
                                (shift && ((uintptr_t)shift & 15))))
        return SRF_EUNSUPPORTED;
    if (workspace_bytes < srf_wino43_workspace_bytes(N, H, W, Cin, Cout)) return SRF_EWORKSPACE;
    a.x = x;
    a.y = y;
    a.V = (float *)workspace;
    a.U = U_packed;
    a.scale = scale;
    a.shift = shift;
    a.x_ld = x_ld;
    a.y_ld = y_ld;
    a.N = N;
    a.H = H;
    a.W = W;
    a.Cout = Cout;
    a.tilesX = srf_ceil_div(W, 4);
    a.tilesY = srf_ceil_div(H, 4);
    a.ntiles = (long long)N * a.tilesX * a.tilesY;
    const long long ntb = (a.ntiles + W43_TB - 1) / W43_TB;
    a.nchunk = Cin / 8;
    a.ncb = srf_ceil_div(Cout, 64);
    a.relu = relu;
    a.tb0 = 0;
    a.n0 = 0;
    if (a.ntiles + W43_TB >= (1ll << 31) || ntb * ((a.nchunk + 3) / 4) >= (1ll << 31) || ntb * a.ncb * 2 >= (1ll << 30)) return SRF_EUNSUPPORTED;
    if ((long long)a.nchunk * 36 * 2 * W43_PIECE >= (1ll << 31)) return SRF_EUNSUPPORTED;   // descriptor range of one U block
    a.ntb = (int)ntb;
#ifdef SRF_DEV
    a.stamps = nullptr;
#endif
    return SRF_OK;
}

// slab [tb0, tb0 + cnt) of the layer described by `full`: pointers moved to the slab's first image; SRF_EUNSUPPORTED when the
// images the slab touches span 4 GB or more (32-bit buffer offsets)
static int w43_slab_args(const W43Args &full, long long tb0, long long cnt, W43Args &s)
{
    s = full;
    const long long per_img = (long long)full.tilesX * full.tilesY;
    const long long g0 = tb0 * W43_TB, g1 = std::min((tb0 + cnt) * W43_TB, full.ntiles) - 1;
    const long long n0 = g0 / per_img, n1 = g1 / per_img;
    const long long ximg = (long long)full.H * full.W * full.x_ld * 4, yimg = (long long)full.H * full.W * full.y_ld * 4;
    if ((n1 - n0 + 1) * ximg >= 0xFFFFFFF0ll || (n1 - n0 + 1) * yimg >= 0xFFFFFFF0ll) return SRF_EUNSUPPORTED;
    s.tb0 = (int)tb0;
    s.ntb = (int)(tb0 + cnt);
    s.n0 = (int)n0;
    s.x = full.x + n0 * (ximg / 4);
    s.y = full.y + n0 * (yimg / 4);
    return SRF_OK;
}

static int w43_launch_xform(const W43Args &a, hipStream_t stream)
{
    hipLaunchKernelGGL(srf_wino43_xform_k, dim3((unsigned)((long long)(a.ntb - a.tb0) * ((a.nchunk + 3) / 4))), dim3(256), 0, stream, a);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

static int w43_launch_mm(const W43Args &a, hipStream_t stream)
{
    int dev = 0;
    SRF_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return SRF_EUNSUPPORTED;
    static bool attr_set[64] = {false};
    if (!attr_set[dev]) {
        SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_wino43_mm_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, W43_X_BYTES));
        SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_wino43_mm_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, W43_X1_BYTES));
        attr_set[dev] = true;
    }
    const long long tb8 = (((long long)(a.ntb - a.tb0) + 7) / 8) * 8;
    static const int force_nb = getenv("SRF_W43_NB") ? atoi(getenv("SRF_W43_NB")) : 0;   // developer A/B knob, read once
    // 64-channel blocks at one workgroup per CU read half the operand bytes per FLOP; 32-channel halves at two per CU hide each
    // other's prologue / epilogue, waste nothing on Cout = 160 / 224 and give small maps twice the workgroups.  Measured per
    // layer (tools/micro/wino43_bench.hip): halves win except on the large maps whose channel count is a multiple of 64.
    const int nb = force_nb ? (force_nb == 1 ? 1 : 2) : (((a.Cout & 63) == 0 && (long long)(a.ntb - a.tb0) * a.ncb >= 1024) ? 2 : 1);
    if (nb == 1) {
        const int nhb = srf_ceil_div(a.Cout, 32);
        hipLaunchKernelGGL((srf_wino43_mm_k<1>), dim3((unsigned)(tb8 * nhb)), dim3(256), W43_X1_BYTES, stream, a, nhb);
    } else
        hipLaunchKernelGGL((srf_wino43_mm_k<2>), dim3((unsigned)(tb8 * a.ncb)), dim3(256), W43_X_BYTES, stream, a, a.ncb);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_wino43(const float *x, int N, int H, int W, int Cin, long long x_ld, const float *U_packed, int Cout, const float *scale,
                          const float *shift, int relu, float *y, long long y_ld, void *workspace, size_t workspace_bytes, srf_stream_t stream)
{
    W43Args a;
    const int rc = w43_make_args(a, x, N, H, W, Cin, x_ld, U_packed, Cout, scale, shift, relu, y, y_ld, workspace, workspace_bytes);
    if (rc != SRF_OK || N == 0) return rc;
    const long long slab = w43_slab_tb(a.ntb, a.nchunk, a.ncb);
    // all slabs are checked before the first launch: a layer either runs whole or not at all
    for (long long tb0 = 0; tb0 < a.ntb; tb0 += slab) {
        W43Args s;
        const int r0 = w43_slab_args(a, tb0, std::min(slab, (long long)a.ntb - tb0), s);
        if (r0 != SRF_OK) return r0;
    }
    for (long long tb0 = 0; tb0 < a.ntb; tb0 += slab) {
        W43Args s;
        w43_slab_args(a, tb0, std::min(slab, (long long)a.ntb - tb0), s);
        const int r1 = w43_launch_xform(s, (hipStream_t)stream);
        if (r1 != SRF_OK) return r1;
        const int r2 = w43_launch_mm(s, (hipStream_t)stream);
        if (r2 != SRF_OK) return r2;
    }
    return SRF_OK;
}

// The two kernels of a layer as separate calls (measurement: bench.py times them apart; a caller that wants to overlap the
// HBM-bound transform with other work).  Whole layer in one slab only: SRF_EUNSUPPORTED when srf_wino43 would cut it.
extern "C" int srf_wino43_transform(const float *x, int N, int H, int W, int Cin, long long x_ld, int Cout, void *workspace,
                                    size_t workspace_bytes, srf_stream_t stream)
{
    W43Args a;
    const int rc = w43_make_args(a, x, N, H, W, Cin, x_ld, nullptr, Cout, nullptr, nullptr, 0, nullptr, 4, workspace, workspace_bytes, W43_NEED_X);
    if (rc != SRF_OK || N == 0) return rc;
    if (w43_slab_tb(a.ntb, a.nchunk, a.ncb) < a.ntb) return SRF_EUNSUPPORTED;
    W43Args s;
    const int r0 = w43_slab_args(a, 0, a.ntb, s);
    if (r0 != SRF_OK) return r0;
    return w43_launch_xform(s, (hipStream_t)stream);
}

extern "C" int srf_wino43_multiply(const void *workspace, size_t workspace_bytes, int N, int H, int W, int Cin, const float *U_packed,
                                   int Cout, const float *scale, const float *shift, int relu, float *y, long long y_ld, srf_stream_t stream)
{
    W43Args a;
    const int rc = w43_make_args(a, nullptr, N, H, W, Cin, 4, U_packed, Cout, scale, shift, relu, y, y_ld, const_cast<void *>(workspace),
                                 workspace_bytes, W43_NEED_Y);
    if (rc != SRF_OK || N == 0) return rc;
    if (w43_slab_tb(a.ntb, a.nchunk, a.ncb) < a.ntb) return SRF_EUNSUPPORTED;
    W43Args s;   // (x_ld = y_ld stand-ins of 4 keep the unused tensor out of the slab's 32-bit extent check)
    const int r0 = w43_slab_args(a, 0, a.ntb, s);
    if (r0 != SRF_OK) return r0;
    return w43_launch_mm(s, (hipStream_t)stream);
}
