// conv.hip -- dense convolutions of the camera branch and the BEV backbone on the f32 MFMA, channels-last (NHWC).
//
// Reference call sites: the 3x3 / stride 1 Conv2d + BatchNorm2d + ReLU layers of VoVNet's OSA blocks
// (mmdet3d_plugin/models/backbones/vovnet.py:116-133, :180-230), the 3x3 output convolutions of the image FPN
// (configs/nus/srfdet_voxel_nusc_LC.py:55-64 -> mmdet FPN), `img_convs` of the head (srfdet_head.py:404-416), the dense
// blocks of SECONDCustom (second_custom.py:41-63, :78-91) and the BEV FPN; the 1x1 `concat` convolution of every OSA block
// and the FPN laterals.  The reference runs them as cuDNN calls through torch; round 1 of this repo ran them on MIOpen,
// whose fp32 3x3 algorithm is a VALU Winograd kernel (57.8 % of the LC frame's kernel time).
//
// Layout: activations are (N, H, W, ld) f32 with `ld` >= channels floats per pixel: a layer reads a channel SLICE of its
// source buffer and writes a slice of its destination buffer, so the OSA concatenation (vovnet.py:222: torch.cat of
// the block input and the five branch outputs) is never built -- the six producers write side by side into one buffer
// and the 1x1 convolution reads it as a plain (pixels x K) matrix.
//
// srf_wino3x3: Winograd F(2x2, 3x3) with every stage inside one kernel:
//   * a workgroup owns 64 output tiles (8 x 8 tiles of 2 x 2 pixels; tile rows run over the N images stacked on top of
//     each other) x 64 output channels; 4 waves = 2 tile halves x 2 channel halves, each wave keeps all 16 Winograd
//     frequencies of its 32 tiles x 32 channels in 16 accumulators of v_mfma_f32_32x32x2_f32 (256 AGPRs, one wave per
//     SIMD), so the output transform needs no exchange: a lane holds the 16 frequencies of the same (tile, channel);
//   * the reduction runs over input channels in chunks of 8.  Per chunk every thread loads the 3 x 4 pixels x 4 channels
//     its half of an input tile needs (float4 per pixel, straight from global memory, predicated at the image border),
//     applies B^T d B in registers and writes the 8 frequencies it owns into the LDS operand image
//     V[f][k quad][tile][4]; the pre-transformed weights U = G g G^T arrive packed in the same image form
//     U[f][k quad][channel][4] and are copied linearly.  Both images are read with conflict-free ds_read_b128; the four
//     floats of a lane feed four consecutive MFMA k-steps (lane half h supplies channel 4 h + s at step s, for A and B
//     alike);
//   * software pipeline: global loads run two chunks ahead in registers, the transform + LDS writes of chunk c + 1 are
//     spread between the MFMA groups of chunk c, fragments are read one group (4 frequencies = 16 MFMAs) ahead; ONE
//     barrier per chunk, placed so that the first fragments of the next chunk are read behind it, under the last group.
//   * epilogue: A^T m A per (tile, channel) in registers, then y = acc * scale[co] + shift[co] (folded eval BatchNorm or
//     the conv bias), optional ReLU, 128-byte channel runs per pixel.
// Numerics: f32 throughout; the transforms use only +, - and the exact factor 0.5 (in the weight transform).  The result
// differs from a direct convolution by a few f32 roundings (tested against torch at 2e-4 of the map's max, the bar of the
// SECOND / FPN tests).
#include "common.hpp"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WN_RAW 924                               // float4 per patch buffer (the 32 x 2 tile block: 2 x 66 rows x 7)
#define WN_LDS_BYTES ((4 * 2048 + 2 * WN_RAW) * 16)  // V[2], U[2]: 2048 float4 each; RAW[2]

struct WinoArgs {
    const float *x;
    float *y;
    const float4 *U;
    const float *scale, *shift;
    long long x_ld, y_ld;
    int N, H, W, Cout;
    int rowBlocks, colBlocks, coutBlocks, nchunk, nspatial, relu;
    int cb0, ncb;       // channel blocks of this launch: cb0 .. cb0 + ncb - 1 (half-block form: in units of 32 channels)
    int nfull;          // srf_wino3x3_mixed_k: workgroups of the full form (the half-block ones follow)
    int ntail;          // srf_wino3x3_mixed_k: full work items nfull .. nfull + ntail - 1 run as two half-block workgroups each
    long long *stamps;  // developer timing hook (srf_dev_set_stamp_buffer): 4 s_memtime values per workgroup, else NULL
};

// ---------------------------------------------------------------------------------------------------------------------
// weight transform + packing: W (Cout, Cin, 3, 3) -> U[chunk][cout block][f][quad][co 64][4] (floats), U = G g G^T,
// G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]]; channels >= Cout are zero.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void srf_wino3x3_pack_k(const float *__restrict__ Wt, int Cout, int Cin, int coutBlocks,
                                                         float *__restrict__ P, long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int kk = (int)(t & 3), co = (int)((t >> 2) & 63), q = (int)((t >> 8) & 1), f = (int)((t >> 9) & 15);
    const long long rest = t >> 13;
    const int cb = (int)(rest % coutBlocks), chunk = (int)(rest / coutBlocks);
    const int cog = cb * 64 + co, ci = chunk * 8 + q * 4 + kk;
    float r = 0.f;
    if (cog < Cout) {
        const float *g = Wt + ((size_t)cog * Cin + ci) * 9;
        const int fr = f >> 2, fs = f & 3;
        // rows of G applied to the kernel's rows (fr) and columns (fs)
        float col[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float g0 = g[b], g1 = g[3 + b], g2 = g[6 + b];
            col[b] = fr == 0 ? g0 : fr == 1 ? 0.5f * (g0 + g1 + g2) : fr == 2 ? 0.5f * (g0 - g1 + g2) : g2;
        }
        r = fs == 0 ? col[0] : fs == 1 ? 0.5f * (col[0] + col[1] + col[2]) : fs == 2 ? 0.5f * (col[0] - col[1] + col[2]) : col[2];
    }
    P[t] = r;
}

typedef float wn_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ wn_f32x16 wn_zero16()
{
    return wn_f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
}

// float4 arithmetic as two packed instructions each: hipcc packs only about half of the component-wise form (28 of the 64
// transform operations of a chunk stayed scalar) and lowers a vector subtraction to four v_sub_f32
typedef float wn_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wn_v2 wn_pk_add(wn_v2 a, wn_v2 b)
{
    wn_v2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ wn_v2 wn_pk_sub(wn_v2 a, wn_v2 b)
{
    wn_v2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ wn_v2 wn_pk_fma(wn_v2 s, wn_v2 a, wn_v2 b)
{
    wn_v2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(s), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float4 wn_sub(float4 a, float4 b)
{
    const wn_v2 lo = wn_pk_sub(wn_v2{a.x, a.y}, wn_v2{b.x, b.y}), hi = wn_pk_sub(wn_v2{a.z, a.w}, wn_v2{b.z, b.w});
    return make_float4(lo[0], lo[1], hi[0], hi[1]);
}
__device__ __forceinline__ float4 wn_add(float4 a, float4 b)
{
    const wn_v2 lo = wn_pk_add(wn_v2{a.x, a.y}, wn_v2{b.x, b.y}), hi = wn_pk_add(wn_v2{a.z, a.w}, wn_v2{b.z, b.w});
    return make_float4(lo[0], lo[1], hi[0], hi[1]);
}
__device__ __forceinline__ float4 wn_fma(float s, float4 a, float4 b)   // s a + b, one rounding per component
{
    const wn_v2 sv = {s, s};
    const wn_v2 lo = wn_pk_fma(sv, wn_v2{a.x, a.y}, wn_v2{b.x, b.y}), hi = wn_pk_fma(sv, wn_v2{a.z, a.w}, wn_v2{b.z, b.w});
    return make_float4(lo[0], lo[1], hi[0], hi[1]);
}

#define WN_MFMA4(ACC, A, B)                                                        \
    do {                                                                           \
        ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((A).x, (B).x, ACC, 0, 0, 0);    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((A).y, (B).y, ACC, 0, 0, 0);    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((A).z, (B).z, ACC, 0, 0, 0);    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((A).w, (B).w, ACC, 0, 0, 0);    \
    } while (0)

// TWL = log2 of the tile block's width: the 64 tiles of a workgroup form an 8 x 8, 16 x 4 or 32 x 2 (rows x columns) block --
// the host picks the shape that covers the map with the fewest blocks (a 29 x 50 tile map: 28 / 26 / 25 blocks).
// HALFB: the launch covers ONE channel block that holds at most 32 real channels (Cout = 160, 224: the last block).  The four
// waves then split (tile half) x (FREQUENCY half) instead of (tile half) x (channel half): 8 frequencies = 8 accumulator
// tiles and 32 MFMAs per chunk and wave instead of 64 (the all-zero channel half is not multiplied); the waves of the
// upper frequency half hand their 8 accumulator tiles (the frequency rows 2 and 3 of M) to the lower half through LDS,
// which then runs the SAME output transform in the same order of operations as the full kernel: identical bits.
template <int DBG, int TWL, bool HALFB>
__device__ __forceinline__ void srf_wino3x3_body(const WinoArgs &a, const int cbi /* HALFB: index of a 32-channel half */, const int sp)
{
    constexpr int TW = 1 << TWL, TH = 64 >> TWL;       // tiles per block row / column
    constexpr int PR = 2 * TH + 2, PC = 2 * TW + 2;    // patch rows / columns (pixels)
    constexpr int HP = TW + 1 + (TWL == 3 ? 1 : 0);    // float4 per (row, column parity)
    constexpr int RP = TWL == 1 ? 7 : 2 * HP;          // row pitch: spreads the tiles of 16 lanes over the 16 bank slots
    constexpr int NPIX = PR * PC * 2;                  // float4 per chunk (2 k quads)
    constexpr int NL = (NPIX + 255) / 256;             // loads per thread
    static_assert(2 * PR * RP <= WN_RAW && 2 * HP <= RP + 1, "patch buffer");
    extern __shared__ __attribute__((aligned(16))) float4 s_w[];  // V[2][2048] | U[2][2048] | RAW[2][WN_RAW]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = HALFB ? cbi >> 1 : cbi;
    if (sp >= a.nspatial) return;
    const int per_img = a.rowBlocks * a.colBlocks;
    const int n = sp / per_img;
    const int rb = (sp - n * per_img) / a.colBlocks, cbk = sp - n * per_img - rb * a.colBlocks;
    long long st0 = 0, st1 = 0, st2 = 0;
    if (DBG & 8) st0 = __builtin_amdgcn_s_memtime();

    // ---- loader role: the (2 TH + 2) x (2 TW + 2) pixel patch of the block, 8 channels per chunk = NPIX float4, <= NL per
    // thread.  Buffer loads: the descriptor covers image n; a pixel outside the image gets an offset beyond the range and the
    // hardware returns zeros (the convolution's zero padding) -- no predicates, no selects.  LDS image of the patch:
    // RAW[k quad][row][column parity][column / 2] with a row pitch RP chosen so that the tiles of a wave (column stride 2)
    // read consecutive float4 and successive tile rows fall on different bank slots.
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x) + (long long)n * a.H * a.W * a.x_ld, 0, (int)((long long)a.H * a.W * a.x_ld * 4), 0x00020000);
    unsigned goff[4];
    int gdst[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = tid + 256 * j;
        const int qq = i & 1, p = i >> 1;
        const int py = p / PC, px = p - py * PC;
        const int y = 2 * TH * rb - 1 + py, x = 2 * TW * cbk - 1 + px;
        const bool ok = i < NPIX && y >= 0 && y < a.H && x >= 0 && x < a.W;
        goff[j] = ok ? (unsigned)((((long long)y * a.W + x) * a.x_ld + qq * 4) * 4) : 0x80000000u;
        // past the patch: a slot no pixel uses (behind the image if the buffer has room, else the spare slot of row 0)
        gdst[j] = i < NPIX ? 8192 + (qq * PR + py) * RP + (px & 1) * HP + (px >> 1) : 8192 + (2 * PR * RP < WN_RAW ? 2 * PR * RP : RP - 1);
    }
    const float4 z4_ = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gr_0 = z4_, gr_1 = z4_, gr_2 = z4_, gr_3 = z4_;
#define WN_GL(J)                                                                                    \
    if (!(DBG & 1) && (J) < NL) {                                                                   \
        auto v_ = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)goff[J], soff_, 0);             \
        gr_##J = *reinterpret_cast<float4 *>(&v_);                                                  \
    }
#define WN_LOAD_RAW(CH)                     \
    do {                                    \
        const int soff_ = (CH) * 32;        \
        WN_GL(0) WN_GL(1) WN_GL(2) WN_GL(3) \
    } while (0)
#define WN_STORE_RAW(RAWB)                             \
    do {                                               \
        s_w[gdst[0] + (RAWB)] = gr_0;                  \
        s_w[gdst[1] + (RAWB)] = gr_1;                  \
        s_w[gdst[2] + (RAWB)] = gr_2;                  \
        if (NL > 3) s_w[gdst[3] + (RAWB)] = gr_3;      \
    } while (0)

    // ---- transform role: tile t = lane, k quad q, half hh of the 4 frequency rows ----
    // Rows of the 4 x 4 input tile this half needs, in the order (RA, RB, RC) = (d[2 hh], d[1 + 2 hh], d[2 - hh]):
    // frequency rows  t0 = RA - RC,  t1 = RC + s RB  with s = +1 (hh = 0: d0 - d2, d1 + d2) or -1 (hh = 1: d2 - d1, d1 - d3).
    const int q = wave & 1, hh = wave >> 1;
    const float sgn = hh ? -1.f : 1.f;
    int rsrc_row[3];
    {
        const int tx = lane & (TW - 1), ty = lane >> TWL;
        const int rsel[3] = {2 * hh, 1 + 2 * hh, 2 - hh};
#pragma unroll
        for (int r = 0; r < 3; ++r) rsrc_row[r] = 8192 + (q * PR + 2 * ty + rsel[r]) * RP + tx;
    }
    const int u_chunk_bytes = a.coutBlocks * 2048 * 16;   // one chunk of all channel blocks (the host checks the total < 2^31)
    __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float4 *>(a.U) + (size_t)cb * 2048, 0, (int)((long long)a.nchunk * u_chunk_bytes - (long long)cb * 2048 * 16), 0x00020000);
    const int u_voff = tid * 16;

    // staging registers as named scalars (arrays indexed from macro loops were left in scratch by hipcc)
    float4 pr_0, pr_1, pr_2, pr_3, pr_4, pr_5, pr_6, pr_7, pr_8, pr_9, pr_10, pr_11;
    float4 ur_0 = z4_, ur_1 = z4_, ur_2 = z4_, ur_3 = z4_, ur_4 = z4_, ur_5 = z4_, ur_6 = z4_, ur_7 = z4_;
    // pixel column 2 tx + c of the patch: parity c & 1, index tx + (c >> 1)
#define WN_RD(R, C, I) pr_##I = s_w[rsrc_row[R] + (RAWB_) + ((C) & 1) * HP + ((C) >> 1)];
#define WN_READ_RAW(RAWB)                                               \
    do {                                                                \
        const int RAWB_ = (RAWB);                                       \
        WN_RD(0, 0, 0) WN_RD(0, 1, 1) WN_RD(0, 2, 2) WN_RD(0, 3, 3)     \
        WN_RD(1, 0, 4) WN_RD(1, 1, 5) WN_RD(1, 2, 6) WN_RD(1, 3, 7)     \
        WN_RD(2, 0, 8) WN_RD(2, 1, 9) WN_RD(2, 2, 10) WN_RD(2, 3, 11)   \
    } while (0)
    // weights through a buffer descriptor: the chunk / slab part of the address is a scalar offset (no 64-bit vector adds
    // in the loop: vector instructions take MFMA time away)
#define WN_UL(I, J)                                                                                             \
    {                                                                                                           \
        auto v_ = __builtin_amdgcn_raw_buffer_load_b128(ursrc, u_voff, u_soff_ + (J) * 4096, 0);                 \
        ur_##I = *reinterpret_cast<float4 *>(&v_);                                                              \
    }
#define WN_LOAD_U_LO(CH)                                              \
    do {                                                              \
        const int u_soff_ = (CH) * u_chunk_bytes;                     \
        if (!(DBG & 2)) { WN_UL(0, 0) WN_UL(1, 1) WN_UL(2, 2) WN_UL(3, 3) } \
    } while (0)
#define WN_LOAD_U_HI(CH)                                              \
    do {                                                              \
        const int u_soff_ = (CH) * u_chunk_bytes;                     \
        if (!(DBG & 2)) { WN_UL(4, 4) WN_UL(5, 5) WN_UL(6, 6) WN_UL(7, 7) } \
    } while (0)
#define WN_STORE_U_LO(WB)                                             \
    do {                                                              \
        float4 *ud_ = s_w + 4096 + (WB) + tid;                        \
        ud_[0] = ur_0; ud_[256] = ur_1; ud_[512] = ur_2; ud_[768] = ur_3; \
    } while (0)
#define WN_STORE_U_HI(WB)                                             \
    do {                                                              \
        float4 *ud_ = s_w + 4096 + (WB) + tid;                        \
        ud_[1024] = ur_4; ud_[1280] = ur_5; ud_[1536] = ur_6; ud_[1792] = ur_7; \
    } while (0)
    // vertical stage: the two frequency rows this half owns, for the 4 columns (register rows RA = 0-3, RB = 4-7, RC = 8-11)
#define WN_S1(C, A, B, D)                                                                                           \
    t0_##C = wn_sub(pr_##A, pr_##D);                                                                                \
    t1_##C = wn_fma(sgn, pr_##B, pr_##D);
#define WN_STAGE1()         \
    do {                    \
        WN_S1(0, 0, 4, 8)   \
        WN_S1(1, 1, 5, 9)   \
        WN_S1(2, 2, 6, 10)  \
        WN_S1(3, 3, 7, 11)  \
    } while (0)
    // horizontal stage of one frequency row + its 4 LDS writes
#define WN_STAGE2(T, FR, WB)                                                     \
    do {                                                                         \
        const int o_ = (WB) + (((FR) * 4) * 2 + q) * 64 + lane;                  \
        s_w[o_] = wn_sub(T##_0, T##_2);                                          \
        s_w[o_ + 128] = wn_add(T##_1, T##_2);                                    \
        s_w[o_ + 256] = wn_sub(T##_2, T##_1);                                    \
        s_w[o_ + 384] = wn_sub(T##_1, T##_3);                                    \
    } while (0)

    // ---- MFMA role: tile half th, channel half chh (HALFB: frequency half fh = wave >> 1, channels 0 .. 31 of the block) ----
    const int th = wave & 1, chh = HALFB ? (cbi & 1) : wave >> 1, fh = HALFB ? wave >> 1 : 0;
    const int li = lane & 31, lh = lane >> 5;
    const int a_off = lh * 64 + th * 32 + li + fh * 1024;          // + f * 128 (+ buffer)
    const int b_off = 4096 + lh * 64 + chh * 32 + li + fh * 1024;  // + f * 128 (+ buffer)
#define WN_READ_GROUP(SET, G, RB)                                                 \
    do {                                                                          \
        _Pragma("unroll") for (int e_ = 0; e_ < 2; ++e_) {                        \
            fa[SET][e_] = *reinterpret_cast<const f32x4 *>(&s_w[(RB) + a_off + ((G) * 2 + e_) * 128]); \
            fb[SET][e_] = *reinterpret_cast<const f32x4 *>(&s_w[(RB) + b_off + ((G) * 2 + e_) * 128]); \
        }                                                                         \
    } while (0)
// hipcc moves the register-only MFMAs across sched_barrier (they carry no chain): without the two empty asm pins below it
// merged pairs of groups and waited for the NEXT group's fragments right after issuing their reads.  The first pin holds
// the group's MFMAs behind this point (they consume the pinned fragments), the second keeps them in front of the
// region's end (it consumes their accumulators).
// FIRST (literal 0 / 1): the first chunk starts every accumulator chain from the inline constant 0 (srcC of the MFMA), so
// the 256 accumulator registers are never zero-filled (256 v_accvgpr_write per workgroup before)
#define WN_MFMA_GROUP(SET, G, FIRST)                                                            \
    do {                                                                                        \
        f32x4 pa0_ = fa[SET][0], pa1_ = fa[SET][1], pb0_ = fb[SET][0], pb1_ = fb[SET][1];       \
        asm volatile("" : "+v"(pa0_), "+v"(pa1_), "+v"(pb0_), "+v"(pb1_));                      \
        f32x16 c0_ = (FIRST) ? wn_zero16() : acc[(G) * 2], c1_ = (FIRST) ? wn_zero16() : acc[(G) * 2 + 1]; \
        if (!(DBG & 4)) {                                                                       \
            WN_MFMA4(c0_, pa0_, pb0_);                                                          \
            WN_MFMA4(c1_, pa1_, pb1_);                                                          \
        }                                                                                       \
        asm volatile("" : "+a"(c0_), "+a"(c1_));                                                \
        acc[(G) * 2] = c0_;                                                                     \
        acc[(G) * 2 + 1] = c1_;                                                                 \
    } while (0)
#define WN_MFMA_HALF(SET, G, E, FIRST)                                                          \
    do {                                                                                        \
        f32x4 pa_ = fa[SET][E], pb_ = fb[SET][E];                                               \
        asm volatile("" : "+v"(pa_), "+v"(pb_));                                                \
        f32x16 c_ = (FIRST) ? wn_zero16() : acc[(G) * 2 + (E)];                                 \
        if (!(DBG & 4)) WN_MFMA4(c_, pa_, pb_);                                                 \
        asm volatile("" : "+a"(c_));                                                            \
        acc[(G) * 2 + (E)] = c_;                                                                \
    } while (0)
#define WN_FENCE() __builtin_amdgcn_sched_barrier(0)

    f32x16 acc[16];   // written by the first chunk (FIRST = 1), never zero-filled
    f32x4 fa[2][2], fb[2][2];  // fragments of two groups (2 frequencies each): one in use, one in flight
    float4 t0_0, t0_1, t0_2, t0_3, t1_0, t1_1, t1_2, t1_3;
    const int nchunk = a.nchunk;
    const int last = nchunk - 1;

    // ---- prologue: chunk 0 transformed into V[0] / U[0], chunk 1's vertical stage in registers, chunk 2's patch and
    // chunk 1's weights in the staging registers ----
    WN_LOAD_RAW(0);
    WN_LOAD_U_LO(0);
    WN_LOAD_U_HI(0);
    WN_STORE_RAW(0);
    WN_LOAD_RAW(last < 1 ? last : 1);
    WN_STORE_RAW(WN_RAW);
    WN_STORE_U_LO(0);
    WN_STORE_U_HI(0);
    WN_LOAD_RAW(last < 2 ? last : 2);
    WN_LOAD_U_LO(last < 1 ? last : 1);
    WN_LOAD_U_HI(last < 1 ? last : 1);
    __syncthreads();
    WN_READ_RAW(0);
    WN_STAGE1();
    WN_STAGE2(t0, 2 * hh, 0);
    WN_STAGE2(t1, 2 * hh + 1, 0);
    WN_READ_RAW(WN_RAW);
    WN_STAGE1();
    __syncthreads();
    WN_READ_GROUP(0, 0, 0);
    if (DBG & 8) st1 = __builtin_amdgcn_s_memtime();

    // Iteration c: multiplies chunk c (8 groups of 2 frequencies = 8 MFMAs each; group g + 1 is read while group g
    // multiplies); writes the patch of chunk c + 2 (registers -> RAW) and loads the one of chunk c + 3; finishes the
    // transform of chunk c + 1 (horizontal stage -> V) and stages its weights (-> U), loads the weights of chunk c + 2;
    // behind the barrier, under the last group: chunk c + 2's patch RAW -> registers -> vertical stage.
    // one chunk of the main loop (FIRST: see WN_MFMA_GROUP)
#define WN_CHUNK_FULL(FIRST, c)                                                                                       \
    do {                                                                                                              \
        const int rbuf = (c & 1) * 2048, wbuf = 2048 - rbuf;                                                          \
        const int rawb = (c & 1) * WN_RAW;                                                                            \
        const int c2 = c + 2 < nchunk ? c + 2 : last, c3 = c + 3 < nchunk ? c + 3 : last;                             \
        WN_READ_GROUP(1, 1, rbuf);                                                                                    \
        WN_FENCE();                                                                                                   \
        WN_STORE_RAW(rawb);                                                                                           \
        WN_LOAD_RAW(c3);                                                                                              \
        WN_MFMA_GROUP(0, 0, FIRST);                                                                                   \
        WN_FENCE();                                                                                                   \
        WN_READ_GROUP(0, 2, rbuf);                                                                                    \
        WN_FENCE();                                                                                                   \
        WN_STAGE2(t0, 2 * hh, wbuf);                                                                                  \
        WN_MFMA_GROUP(1, 1, FIRST);                                                                                   \
        WN_FENCE();                                                                                                   \
        WN_READ_GROUP(1, 3, rbuf);                                                                                    \
        WN_FENCE();                                                                                                   \
        WN_STORE_U_LO(wbuf);                                                                                          \
        WN_LOAD_U_LO(c2);                                                                                             \
        WN_MFMA_GROUP(0, 2, FIRST);                                                                                   \
        WN_FENCE();                                                                                                   \
        WN_READ_GROUP(0, 4, rbuf);                                                                                    \
        WN_FENCE();                                                                                                   \
        WN_STAGE2(t1, 2 * hh + 1, wbuf);                                                                              \
        WN_MFMA_GROUP(1, 3, FIRST);                                                                                   \
        WN_FENCE();                                                                                                   \
        WN_READ_GROUP(1, 5, rbuf);                                                                                    \
        WN_FENCE();                                                                                                   \
        WN_STORE_U_HI(wbuf);                                                                                          \
        WN_LOAD_U_HI(c2);                                                                                             \
        WN_MFMA_GROUP(0, 4, FIRST);                                                                                   \
        WN_FENCE();                                                                                                   \
        WN_READ_GROUP(0, 6, rbuf);                                                                                    \
        WN_FENCE();                                                                                                   \
        WN_MFMA_GROUP(1, 5, FIRST);                                                                                   \
        WN_FENCE();                                                                                                   \
        WN_READ_GROUP(1, 7, rbuf);                                                                                    \
        WN_FENCE();                                                                                                   \
        WN_MFMA_GROUP(0, 6, FIRST);                                                                                   \
        WN_FENCE();                                                                                                   \
        __syncthreads();                                                                                              \
        WN_READ_GROUP(0, 0, wbuf);                                                                                    \
        WN_READ_RAW(rawb);                                                                                            \
        WN_FENCE();                                                                                                   \
        WN_MFMA_HALF(1, 7, 0, FIRST);                                                                                 \
        WN_FENCE();                                                                                                   \
        WN_STAGE1();                                                                                                  \
        WN_MFMA_HALF(1, 7, 1, FIRST);                                                                                 \
        WN_FENCE();                                                                                                   \
    } while (0)
#define WN_CHUNK_HALF(FIRST, c)                                                                                       \
    do {                                                                                                              \
            const int rbuf = (c & 1) * 2048, wbuf = 2048 - rbuf;                                                      \
            const int rawb = (c & 1) * WN_RAW;                                                                        \
            const int c2 = c + 2 < nchunk ? c + 2 : last, c3 = c + 3 < nchunk ? c + 3 : last;                         \
            WN_READ_GROUP(1, 1, rbuf);                                                                                \
            WN_FENCE();                                                                                               \
            WN_STORE_RAW(rawb);                                                                                       \
            WN_LOAD_RAW(c3);                                                                                          \
            WN_MFMA_GROUP(0, 0, FIRST);                                                                               \
            WN_FENCE();                                                                                               \
            WN_READ_GROUP(0, 2, rbuf);                                                                                \
            WN_FENCE();                                                                                               \
            WN_STAGE2(t0, 2 * hh, wbuf);                                                                              \
            WN_STORE_U_LO(wbuf);                                                                                      \
            WN_LOAD_U_LO(c2);                                                                                         \
            WN_MFMA_GROUP(1, 1, FIRST);                                                                               \
            WN_FENCE();                                                                                               \
            WN_READ_GROUP(1, 3, rbuf);                                                                                \
            WN_FENCE();                                                                                               \
            WN_STAGE2(t1, 2 * hh + 1, wbuf);                                                                          \
            WN_STORE_U_HI(wbuf);                                                                                      \
            WN_LOAD_U_HI(c2);                                                                                         \
            WN_MFMA_GROUP(0, 2, FIRST);                                                                               \
            WN_FENCE();                                                                                               \
            __syncthreads();                                                                                          \
            WN_READ_GROUP(0, 0, wbuf);                                                                                \
            WN_READ_RAW(rawb);                                                                                        \
            WN_FENCE();                                                                                               \
            WN_MFMA_HALF(1, 3, 0, FIRST);                                                                             \
            WN_FENCE();                                                                                               \
            WN_STAGE1();                                                                                              \
            WN_MFMA_HALF(1, 3, 1, FIRST);                                                                             \
            WN_FENCE();                                                                                               \
    } while (0)
    if (HALFB) {
        // four groups (8 frequencies) per chunk and wave: the schedule of the full loop with the groups 4 .. 7 taken out
        WN_CHUNK_HALF(1, 0);
        for (int c = 1; c < nchunk; ++c) WN_CHUNK_HALF(0, c);
    } else {
        // 4 MFMAs of group 7 cover the latency of the 12 patch reads the vertical stage waits for
        WN_CHUNK_FULL(1, 0);
        for (int c = 1; c < nchunk; ++c) WN_CHUNK_FULL(0, c);
    }

    // ---- epilogue: A^T m A, affine, ReLU, store ----
    if (DBG & 8) st2 = __builtin_amdgcn_s_memtime();
    // HALFB: frequency rows 2 and 3 of M (the accumulator tiles of the waves with fh = 1) change hands through LDS:
    // X[th][f 8][register 16][lane 64] floats = 64 KB over the V / U images, which nobody reads any more
    float *xch = reinterpret_cast<float *>(s_w) + th * (8 * 16 * 64) + lane;
    if (HALFB) {
        __syncthreads();  // every wave is past the (unused) fragment / patch reads of the loop's last trip
        if (fh == 1) {
#pragma unroll
            for (int f = 0; f < 8; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(f * 16 + r) * 64] = acc[f][r];
        }
        __syncthreads();
        if (fh == 1) return;
    }
    const int co = cb * 64 + chh * 32 + li;
    const bool co_ok = co < a.Cout;
    const float sc = (co_ok && a.scale) ? a.scale[co] : 1.f;
    const float sh = (co_ok && a.shift) ? a.shift[co] : 0.f;
    // stores through a buffer descriptor of image n: a pixel outside the map (or a channel past Cout) gets an offset beyond
    // the range and the hardware drops the store -- no exec-mask branches, no 64-bit address arithmetic (the predicated
    // global stores cost ~400 of the epilogue's ~1400 instructions)
    __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + (long long)n * a.H * a.W * a.y_ld, 0,
                                                                     (int)((long long)a.H * a.W * a.y_ld * 4), 0x00020000);
    const unsigned px_b = (unsigned)(a.y_ld * 4), row_b = (unsigned)(a.W * a.y_ld * 4);
    const unsigned OOB = 0x80000000u;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = rg * 4 + rr;
            const int t = th * 32 + rr + 8 * rg + 4 * lh;   // tile of accumulator register r in this lane half
            const int oy = 2 * (rb * TH + (t >> TWL)), ox = 2 * (cbk * TW + (t & (TW - 1)));
            const bool row_ok = oy < a.H && co_ok && ox < a.W;
            const unsigned o00_ = (unsigned)((oy * a.W + ox) * (int)a.y_ld + co) * 4u;
            float s[4], d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m0 = acc[j][r], m1 = acc[4 + j][r];
                const float m2 = HALFB ? xch[(j * 16 + r) * 64] : acc[8 + j][r], m3 = HALFB ? xch[((4 + j) * 16 + r) * 64] : acc[12 + j][r];
                s[j] = (m0 + m1) + m2;
                d[j] = (m1 - m2) - m3;
            }
            float o00 = (s[0] + s[1]) + s[2], o01 = (s[1] - s[2]) - s[3];
            float o10 = (d[0] + d[1]) + d[2], o11 = (d[1] - d[2]) - d[3];
            o00 = __fmaf_rn(o00, sc, sh);
            o01 = __fmaf_rn(o01, sc, sh);
            o10 = __fmaf_rn(o10, sc, sh);
            o11 = __fmaf_rn(o11, sc, sh);
            if (a.relu) {
                o00 = fmaxf(o00, 0.f);
                o01 = fmaxf(o01, 0.f);
                o10 = fmaxf(o10, 0.f);
                o11 = fmaxf(o11, 0.f);
            }
            const bool x1 = ox + 1 < a.W, y1 = oy + 1 < a.H;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o00), yrsrc, (int)(row_ok ? o00_ : OOB), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o01), yrsrc, (int)(row_ok && x1 ? o00_ + px_b : OOB), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o10), yrsrc, (int)(row_ok && y1 ? o00_ + row_b : OOB), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o11), yrsrc, (int)(row_ok && x1 && y1 ? o00_ + row_b + px_b : OOB), 0, 0);
        }
    }
    if ((DBG & 8) && a.stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long *o = a.stamps + (size_t)blockIdx.x * 4;
        o[0] = st0;
        o[1] = st1;
        o[2] = st2;
        o[3] = __builtin_amdgcn_s_memtime();
    }
}

// work item -> (channel block, spatial block): items b and b + 8 share an XCD (round-robin dispatch), so the channel blocks of
// one spatial block sit on one L2
__device__ __forceinline__ void srf_wino_decode(unsigned item, int cb0, int ncb, int &cbi, int &sp)
{
    const int xcd = item & 7, jq = item >> 3;
    cbi = cb0 + jq % ncb;
    sp = (jq / ncb) * 8 + xcd;
}

template <int DBG, int TWL, bool HALFB = false>
__global__ __launch_bounds__(256, 1) void srf_wino3x3_k(WinoArgs a)
{
    int cbi, sp;
    srf_wino_decode(blockIdx.x, a.cb0, a.ncb, cbi, sp);
    srf_wino3x3_body<DBG, TWL, HALFB>(a, cbi, sp);
}

// One launch, full and half-block workgroups: the first a.nfull workgroups are full ones (work items 0 .. nfull - 1 over the
// channel blocks 0 .. ncb - 1); then the a.ntail work items that would form a partly filled last round run as TWO half-block
// workgroups each (a round of half blocks is ~0.7 of a full one: 276 items on 256 CUs take 1.7 rounds instead of 2); then, for
// a layer whose last channel block holds <= 32 channels (cb0 = that block's first half, else -1), one half-block workgroup per
// spatial block.  The short workgroups fill the tail of the long ones instead of waiting for a launch of their own.
template <int TWL>
__global__ __launch_bounds__(256, 1) void srf_wino3x3_mixed_k(WinoArgs a)
{
    int cbi, sp;
    if (blockIdx.x < (unsigned)a.nfull) {
        srf_wino_decode(blockIdx.x, 0, a.ncb, cbi, sp);
        srf_wino3x3_body<0, TWL, false>(a, cbi, sp);
        return;
    }
    const unsigned h = blockIdx.x - (unsigned)a.nfull;
    if (h < 2u * (unsigned)a.ntail) {
        srf_wino_decode((unsigned)a.nfull + (h >> 1), 0, a.ncb, cbi, sp);
        cbi = 2 * cbi + (int)(h & 1);
    } else {
        srf_wino_decode(h - 2u * (unsigned)a.ntail, a.cb0, 1, cbi, sp);
    }
    srf_wino3x3_body<0, TWL, true>(a, cbi, sp);
}

// =====================================================================================================================
// srf_conv1x1_nhwc: Y[p][co] = sum_k X[p][k] W[co][k] on channels-last activations -- the `concat` 1x1 convolution of the
// OSA blocks over the whole concat buffer (vovnet.py:222-223) and the FPN laterals -- with scale / shift / ReLU as the
// epilogue.  A plain GEMM on v_mfma_f32_32x32x2_f32:
//   * workgroup = 4 waves = 2 x 2 wave tiles of (32 RM) pixels x 128 channels; RM = 4: 256 pixels x 256 channels per
//     workgroup, 16 accumulators per wave (one wave per SIMD), RM = 2 for small maps;
//   * the reduction runs in chunks of 32 channels: one 128-byte line per pixel row, so a wave-instruction of the loader
//     reads 8 whole lines (the first version read 16 bytes from 64 different lines per instruction and spent its time
//     in the texture path: 61 % MFMA-busy against rocBLAS's 91 % on the same GEMM).  LDS images X[pixel][8 quads],
//     W[channel][8 quads], the 16-byte slot of quad Q in row r is Q ^ ((r >> 1) & 7): loader writes (8 lanes = one row)
//     and fragment reads (16 lanes = 16 rows, one quad) are both conflict-free.  The weights are packed in that image
//     order once per layer and copied linearly;
//   * one ds_read_b128 per operand block feeds 4 MFMA k-steps (lane half h supplies quad 2 s + h in sub-step s): 8 reads
//     per 64 MFMAs; global loads run two chunks ahead in registers (buffer loads: pixel rows past the end return zeros),
//     the LDS stores of chunk c + 1 are spread over the four 64-MFMA sub-steps of chunk c; one barrier per chunk.
// f32 MFMAs do not overlap with vector instructions of their own wave (tools/micro/mfma_overlap.hip), so the loop is kept
// almost free of them.
// =====================================================================================================================
struct GemmArgs {
    const float *x;
    float *y;
    const f32x4 *Wp;
    const float *scale, *shift;
    long long x_ld, y_ld, M;
    int K, Cout, coutBlocks, nchunk, relu;
    long long mblocks;
    // convolution mode (CONV): rows are OUTPUT pixels (n, oy, ox), k runs over (tap, input channel): implicit im2col
    int H, W, Ho, Wo, kw, stride, pad, cin_chunks;  // cin_chunks = Cin / 32
    long long x_bytes;                               // extent of the input tensor (buffer descriptor range)
    // per-image row tiling + column sums of the stored outputs (eSE pooling fused into the OSA concat convolution):
    // bpi > 0: M = N HW rows, image n owns row blocks [n bpi, (n + 1) bpi), block (n, lb) starts at row n HW + lb TM and
    // ends with its image; colsum[(n bpi + lb) Cout + co] = sum over the block's rows of y[row][co]
    float *colsum;
    long long HW;
    int bpi;
    // row range of this launch (the mixed launch gives the head of the rows to 128 x 128 tiles and the tail to 64 x 64 ones):
    // row0 = first row (flat tiling) / first row inside every image (per-image tiling); the column sums of block lb of image n
    // go to slot n * slots + slot0 + lb
    long long row0;
    int slot0, slots;
    // FPN top-down step fused into the lateral convolution: y += top[n][floor(py sy)][floor(px sx)][co] (nearest upsampling by
    // size, as F.interpolate / srf_nhwc_upsample_add); rows are the pixels (n, py, px) of an (N, mapH, mapW) map
    const float *top;
    long long top_ld;
    int mapH, mapW, topH, topW;
    float sy, sx;
};

// W (Cout, K) row-major -> [chunk of 32][cout block of 256][channel 256][slot 8][4]; slot s of row co holds quad
// s ^ ((co >> 1) & 7); channels >= Cout are zero
__global__ __launch_bounds__(256) void srf_conv1x1_nhwc_pack_k(const float *__restrict__ Wt, int Cout, int K, int coutBlocks,
                                                              float *__restrict__ P, long long total)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int kk = (int)(t & 3), slot = (int)((t >> 2) & 7), co = (int)((t >> 5) & 255);
    const long long rest = t >> 13;
    const int cb = (int)(rest % coutBlocks), chunk = (int)(rest / coutBlocks);
    const int Q = slot ^ ((co >> 1) & 7);
    const int cog = cb * 256 + co, k = chunk * 32 + Q * 4 + kk;
    P[t] = cog < Cout ? Wt[(size_t)cog * K + k] : 0.f;
}

// Workgroup tile: (64 RM) pixels x (64 RN) channels, 4 waves = 2 x 2 wave tiles of (32 RM) x (32 RN); the weights are packed
// in blocks of 256 channels, a workgroup reads the (64 RN)-channel part `cs` of its block.
//   <2, 2>: 128 x 128, 64 accumulator registers per wave, 32 KB of LDS (one stage + register prefetch) -> several
//           workgroups per CU: the waves of different workgroups fill each other's barrier / staging gaps (an f32 MFMA
//           overlaps with another wave's vector and LDS instructions, not with its own wave's);
//   <4, 4>: 256 x 256, 16 accumulators per wave, one workgroup per CU (kept for A/B timing: SRF_GEMM_BIG=1).
template <int RM, int RN, bool CONV>
__device__ __forceinline__ void srf_gemm_body(const GemmArgs &a, const unsigned bid)
{
    constexpr int TM = 64 * RM, TN = 64 * RN;
    constexpr int ASZ = 8 * TM;          // float4 per A stage
    constexpr int BSZ = 8 * TN;          // float4 per B stage
    constexpr int NA = ASZ / 256;        // A float4 per thread and chunk (= 2 RM)
    constexpr int NB = BSZ / 256;
    constexpr int NCS = 256 / TN;        // channel sub-blocks per packed block
    extern __shared__ __attribute__((aligned(16))) f32x4 s_g[];  // A[ASZ] | B[BSZ]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = bid & 7, jq = bid >> 3;
    // column tiles that hold real channels only: a launch whose every second workgroup returned at once (Cout = 128 in a
    // 256-channel packed block) ran at half occupancy -- the slots of the no-op workgroups are not refilled fast enough
    const int nct = (a.Cout + TN - 1) / TN;
    const int ct = jq % nct;
    const int cb = ct / NCS, cs = ct - cb * NCS;
    const long long mb = (long long)(jq / nct) * 8 + xcd;
    if (mb >= a.mblocks) return;
    if ((cb * 256 + cs * TN) >= a.Cout) return;
    long long p0 = (CONV ? 0 : a.row0) + mb * TM, rows_blk = a.M - p0;
    long long slot = mb;
    if (!CONV && a.bpi > 0) {
        const long long n = mb / a.bpi, lb = mb - n * a.bpi;
        p0 = n * a.HW + a.row0 + lb * TM;
        rows_blk = a.HW - a.row0 - lb * TM;
        slot = n * a.slots + a.slot0 + lb;
    }

    __amdgpu_buffer_rsrc_t xrsrc;
    if (CONV) {
        xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.x_bytes, 0x00020000);
    } else {
        long long rows = rows_blk;
        if (rows > TM) rows = TM;
        xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x) + p0 * a.x_ld, 0, (int)(rows * a.x_ld * 4), 0x00020000);
    }
    const unsigned aoff0 = (unsigned)(((tid >> 3) * a.x_ld + (tid & 7) * 4) * 4);
    const unsigned aoff_step = (unsigned)(32 * a.x_ld * 4);  // 32 rows per j
    // CONV: the output pixel of each of this thread's rows, as the input coordinates of tap (0, 0)
    int cy[CONV ? 2 * RM : 1], cx[CONV ? 2 * RM : 1], cn[CONV ? 2 * RM : 1];
    if (CONV) {
#pragma unroll
        for (int j = 0; j < 2 * RM; ++j) {
            const long long p = p0 + (tid >> 3) + 32 * j;
            const long long hw = (long long)a.Ho * a.Wo;
            const int n = (int)(p / hw);
            const int rem = (int)(p - n * hw);
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            cn[j] = p < a.M ? n : -1;
            cy[j] = oy * a.stride - a.pad;
            cx[j] = ox * a.stride - a.pad;
        }
    }
    const int a_dst = (tid >> 3) * 8 + ((tid & 7) ^ ((tid >> 4) & 7));  // + 256 j: rows advance by 32, the swizzle repeats
    const f32x4 *Bg = a.Wp + (size_t)cb * 2048 + cs * BSZ + tid;
    const size_t b_chunk_stride = (size_t)a.coutBlocks * 2048;
    f32x4 ar[NA], br[NB];
#define GM_LOAD(CH)                                                                               \
    do {                                                                                          \
        if (CONV) {                                                                               \
            const int tap_ = (CH) / a.cin_chunks, cc_ = (CH) - tap_ * a.cin_chunks;               \
            const int ky_ = tap_ / a.kw, kx_ = tap_ - ky_ * a.kw;                                 \
            _Pragma("unroll") for (int j_ = 0; j_ < NA; ++j_) {                                   \
                const int iy_ = cy[j_] + ky_, ix_ = cx[j_] + kx_;                                 \
                const bool ok_ = cn[j_] >= 0 && iy_ >= 0 && iy_ < a.H && ix_ >= 0 && ix_ < a.W;   \
                const unsigned off_ = ok_ ? (unsigned)(((((long long)cn[j_] * a.H + iy_) * a.W + ix_) * a.x_ld + cc_ * 32 + (tid & 7) * 4) * 4) \
                                          : 0x80000000u;                                          \
                auto v_ = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)off_, 0, 0);          \
                ar[j_] = *reinterpret_cast<f32x4 *>(&v_);                                         \
            }                                                                                     \
        } else {                                                                                  \
            const int soff_ = (CH) * 128;                                                         \
            _Pragma("unroll") for (int j_ = 0; j_ < NA; ++j_) {                                   \
                auto v_ = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(aoff0 + j_ * aoff_step), soff_, 0); \
                ar[j_] = *reinterpret_cast<f32x4 *>(&v_);                                         \
            }                                                                                     \
        }                                                                                         \
        const f32x4 *bb_ = Bg + (size_t)(CH) * b_chunk_stride;                                    \
        _Pragma("unroll") for (int j_ = 0; j_ < NB; ++j_) br[j_] = bb_[j_ * 256];                 \
    } while (0)
#define GM_STORE()                                                                                \
    do {                                                                                          \
        _Pragma("unroll") for (int j_ = 0; j_ < NA; ++j_) s_g[a_dst + 256 * j_] = ar[j_];         \
        _Pragma("unroll") for (int j_ = 0; j_ < NB; ++j_) s_g[ASZ + tid + 256 * j_] = br[j_];     \
    } while (0)

    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int swz = (li >> 1) & 7;
    const int a_row = (wm * 32 * RM + li) * 8;        // + im * 256
    const int b_row = ASZ + (wn * 32 * RN + li) * 8;  // + jn * 256
    int qs[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) qs[s2] = (2 * s2 + lh) ^ swz;
    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 fa[2][RM], fb[2][RN];
#define GM_READ(SET, S2)                                                                                    \
    do {                                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < RM; ++i_) fa[SET][i_] = s_g[a_row + i_ * 256 + qs[S2]];     \
        _Pragma("unroll") for (int j_ = 0; j_ < RN; ++j_) fb[SET][j_] = s_g[b_row + j_ * 256 + qs[S2]];     \
    } while (0)
#define GM_MFMA(SET)                                                                                        \
    do {                                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < RM; ++i_) asm volatile("" : "+v"(fa[SET][i_]));             \
        _Pragma("unroll") for (int j_ = 0; j_ < RN; ++j_) asm volatile("" : "+v"(fb[SET][j_]));             \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)                                                 \
            _Pragma("unroll") for (int i_ = 0; i_ < RM; ++i_)                                               \
                _Pragma("unroll") for (int j_ = 0; j_ < RN; ++j_)                                           \
                    acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][i_][ks_], fb[SET][j_][ks_], acc[i_][j_], 0, 0, 0); \
    } while (0)

    // One LDS stage: chunk c is multiplied out of LDS while chunk c + 1 waits in registers; at the end of the chunk a
    // barrier retires the readers, the registers go to LDS, chunk c + 2 is requested, a second barrier publishes.
    const int nchunk = a.nchunk, last = nchunk - 1;
    GM_LOAD(0);
    GM_STORE();
    GM_LOAD(last < 1 ? last : 1);
    __syncthreads();
    GM_READ(0, 0);
    for (int c = 0; c < nchunk; ++c) {
        const int c2 = c + 2 < nchunk ? c + 2 : last;
        GM_READ(1, 1);
        WN_FENCE();
        GM_MFMA(0);
        WN_FENCE();
        GM_READ(0, 2);
        WN_FENCE();
        GM_MFMA(1);
        WN_FENCE();
        GM_READ(1, 3);
        WN_FENCE();
        GM_MFMA(0);
        WN_FENCE();
        GM_MFMA(1);
        WN_FENCE();
        __syncthreads();
        GM_STORE();
        GM_LOAD(c2);
        __syncthreads();
        GM_READ(0, 0);
        WN_FENCE();
    }

    // epilogue: lane = channel (li) within RN blocks of 32, accumulator register = pixel row
    float sc[RN], sh[RN];
    bool co_ok[RN];
    const int co0 = cb * 256 + cs * TN + wn * 32 * RN + li;
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int co = co0 + j * 32;
        co_ok[j] = co < a.Cout;
        sc[j] = (co_ok[j] && a.scale) ? a.scale[co] : 1.f;
        sh[j] = (co_ok[j] && a.shift) ? a.shift[co] : 0.f;
    }
    // stores through a buffer descriptor over this block's rows: a row past the block (or the image) and a channel past Cout
    // are offsets beyond the range, which the hardware drops -- no exec-mask branch per row, 32-bit offsets
    const long long rows_here = rows_blk < TM ? rows_blk : TM;
    __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + p0 * a.y_ld, 0, (int)(rows_here * a.y_ld * 4), 0x00020000);
    const int row_base = wm * 32 * RM + 4 * lh;     // + i * 32 + (r & 3) + 8 * (r >> 2)
    unsigned ybase[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) ybase[j] = co_ok[j] ? (unsigned)((row_base * a.y_ld + co0 + j * 32) * 4) : 0x80000000u;
    const unsigned yrow_b = (unsigned)(a.y_ld * 4);
    const long long rows_left = rows_blk - row_base;
    const bool want_sum = !CONV && a.colsum != nullptr;
    const bool want_top = !CONV && a.top != nullptr;
    int *s_top = reinterpret_cast<int *>(s_g);  // [TM]: offset (floats) of the top-level pixel each row of this block adds
    if (want_top) {
        __syncthreads();  // the last chunk's fragment reads are done
        if (tid < TM) {
            const long long row = p0 + tid;
            int off = 0;
            if (row < a.M) {
                const int hw = a.mapH * a.mapW;
                const int n = (int)(row / hw), rem = (int)(row - (long long)n * hw);
                const int yy = rem / a.mapW, xx = rem - yy * a.mapW;
                int ys = (int)floorf((float)yy * a.sy), xs = (int)floorf((float)xx * a.sx);
                if (ys > a.topH - 1) ys = a.topH - 1;
                if (xs > a.topW - 1) xs = a.topW - 1;
                off = (int)((((long long)n * a.topH + ys) * a.topW + xs) * a.top_ld);
            }
            s_top[tid] = off;
        }
        __syncthreads();
    }
    float csum[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) csum[j] = 0.f;
#pragma unroll
    for (int i = 0; i < RM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                float v = __fmaf_rn(acc[i][j][r], sc[j], sh[j]);
                if (a.relu) v = fmaxf(v, 0.f);
                if (want_top && co_ok[j]) v = __fadd_rn(v, a.top[s_top[row_base + dr] + co0 + j * 32]);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrsrc, (int)(ybase[j] + (unsigned)dr * yrow_b), 0, 0);
                if (want_sum) csum[j] += dr < rows_left ? v : 0.f;
            }
        }
        WN_FENCE();
    }
    if (!CONV && a.colsum) {
        // rows of a column: 2 lane halves x (4 / WN) waves; fixed order -> the sums are reproducible bit for bit
        float *red = reinterpret_cast<float *>(s_g);  // [wm][TN]
        __syncthreads();                               // the last chunk's fragment reads are done
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const float o = __shfl_xor(csum[j], 32);
            if (lh == 0) red[wm * TN + wn * 32 * RN + j * 32 + li] = csum[j] + o;
        }
        __syncthreads();
        constexpr int WM = 4 / (TN / (32 * RN));       // waves along the rows
        if (tid < TN) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) s += red[w * TN + tid];
            const int co = cb * 256 + cs * TN + tid;
            if (co < a.Cout) a.colsum[slot * a.Cout + co] = s;
        }
    }
}

template <int RM, int RN, int WPE, bool CONV>
__global__ __launch_bounds__(256, WPE) void srf_conv1x1_nhwc_k(GemmArgs a)
{
    srf_gemm_body<RM, RN, CONV>(a, blockIdx.x);
}

// One launch, two tile sizes: the first `nbig` workgroups run 128 x 128 tiles over the head of the rows (whole rounds of three
// workgroups per CU), the rest cover the tail of the rows with 64 x 64 tiles.  A last, partly filled round of 128 x 128 tiles
// costs the full latency of a tile (~100 us at K = 1728: 2.125 rounds took 812 us against 720 us for 2.0); the small tiles
// start as the big ones retire and are a quarter as long.  Every output is the same k-ordered fma chain in either tile.
__global__ __launch_bounds__(256, 3) void srf_conv1x1_nhwc_mixed_k(GemmArgs big, GemmArgs tail, unsigned nbig)
{
    if (blockIdx.x < nbig) srf_gemm_body<2, 2, false>(big, blockIdx.x);
    else srf_gemm_body<1, 1, false>(tail, blockIdx.x - nbig);
}

#ifdef SRF_DEV
// developer build only (python -m srfdet3d_amd.build --dev): device buffer of 4 * gridDim.x int64 for SRF_WINO_DBG=8; the
// production library has neither this symbol nor the ablation kernels
static long long *g_wino_stamps = nullptr;
extern "C" void srf_dev_set_stamp_buffer(long long *p) { g_wino_stamps = p; }
#endif

// Developer A/B knobs choose between forms that produce IDENTICAL bits (tile-block shape, half-block / tail mixes); they are
// read once per process -- the parity tests run each setting in its own interpreter.
static int srf_knob(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

static int srf_cu_count(int dev)
{
    static int cus[64] = {0};
    if (dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

extern "C" size_t srf_wino3x3_packed_weight_bytes(int Cout, int Cin)
{
    if (Cout <= 0 || Cin <= 0 || (Cin & 7)) return 0;
    return (size_t)(Cin / 8) * srf_ceil_div(Cout, 64) * 2048 * 16;
}

extern "C" int srf_wino3x3_pack_weights(const float *W, int Cout, int Cin, float *packed, srf_stream_t stream)
{
    if (Cout <= 0 || Cin <= 0 || !W || !packed) return SRF_EINVAL;
    if (Cin & 7) return SRF_EUNSUPPORTED;
    const int coutBlocks = srf_ceil_div(Cout, 64);
    const long long total = (long long)(Cin / 8) * coutBlocks * 8192;
    hipLaunchKernelGGL(srf_wino3x3_pack_k, dim3(srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, W, Cout, Cin, coutBlocks,
                       packed, total);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_wino3x3(const float *x, int N, int H, int W, int Cin, long long x_ld, const float *U_packed, int Cout,
                           const float *scale, const float *shift, int relu, float *y, long long y_ld, srf_stream_t stream)
{
    if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || x_ld < Cin || y_ld < Cout) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!x || !U_packed || !y) return SRF_EINVAL;
    if ((Cin & 7) || (x_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)U_packed & 15)) return SRF_EUNSUPPORTED;
    // per-lane byte offsets inside one image must stay below 2^30 (the buffer descriptor covers one image)
    if ((long long)H * W * x_ld * 4 >= (1ll << 30) || (long long)H * W * y_ld * 4 >= (1ll << 31)) return SRF_EUNSUPPORTED;
    WinoArgs a;
    a.x = x;
    a.y = y;
    a.U = reinterpret_cast<const float4 *>(U_packed);
    a.scale = scale;
    a.shift = shift;
    a.x_ld = x_ld;
    a.y_ld = y_ld;
    a.N = N;
    a.H = H;
    a.W = W;
    a.Cout = Cout;
    // tile-block shape (rows x columns of 2 x 2-pixel tiles): the one that covers the map with the fewest 64-tile blocks
    const int tilesY = (H + 1) / 2, tilesX = (W + 1) / 2;
    int twl = 3;
    long long best = (long long)srf_ceil_div(tilesY, 8) * srf_ceil_div(tilesX, 8);
    for (int l = 2; l >= 1; --l) {
        const long long nb = (long long)srf_ceil_div(tilesY, 64 >> l) * srf_ceil_div(tilesX, 1 << l);
        if (nb < best) {
            best = nb;
            twl = l;
        }
    }
    static const int force_twl = srf_knob("SRF_WINO_TWL", 0);
    if (force_twl >= 1 && force_twl <= 3) twl = force_twl;
    a.rowBlocks = srf_ceil_div(tilesY, 64 >> twl);
    a.colBlocks = srf_ceil_div(tilesX, 1 << twl);
    a.coutBlocks = srf_ceil_div(Cout, 64);
    a.nchunk = Cin / 8;
    const long long nspatial = (long long)N * a.rowBlocks * a.colBlocks;
    if (nspatial * a.coutBlocks >= (1ll << 30)) return SRF_EUNSUPPORTED;
    if (srf_wino3x3_packed_weight_bytes(Cout, Cin) >= ((size_t)1 << 31)) return SRF_EUNSUPPORTED;  // buffer-descriptor range of U
    a.nspatial = (int)nspatial;
    a.relu = relu;
#ifdef SRF_DEV
    a.stamps = g_wino_stamps;
#else
    a.stamps = nullptr;
#endif
    const long long blocks = ((nspatial + 7) / 8) * 8 * a.coutBlocks;
    int dev = 0;
    SRF_HIP_TRY(hipGetDevice(&dev));
    static bool attr_set[64] = {false};
    if (dev < 0 || dev >= 64) return SRF_EUNSUPPORTED;
#ifdef SRF_DEV
    static const int dbg = srf_knob("SRF_WINO_DBG", 0);  // timing ablations: skip loads / MFMAs, stamps (WRONG outputs by design)
#endif
    if (!attr_set[dev]) {
#define WN_ATTR(D, L, HB) SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_wino3x3_k<D, L, HB>, hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS_BYTES))
        WN_ATTR(0, 1, false);
        WN_ATTR(0, 2, false);
        WN_ATTR(0, 3, false);
        WN_ATTR(0, 1, true);
        WN_ATTR(0, 2, true);
        WN_ATTR(0, 3, true);
        SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_wino3x3_mixed_k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS_BYTES));
        SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_wino3x3_mixed_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS_BYTES));
        SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_wino3x3_mixed_k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS_BYTES));
#ifdef SRF_DEV
        WN_ATTR(1, 3, false);
        WN_ATTR(4, 3, false);
        WN_ATTR(8, 3, false);
#endif
#undef WN_ATTR
        attr_set[dev] = true;
    }
    const dim3 blk(256);
    a.cb0 = 0;
    a.ncb = a.coutBlocks;
#ifdef SRF_DEV
    if (dbg == 1 || dbg == 4 || dbg == 8) {   // ablation builds exist for the 8 x 8 shape only
        a.rowBlocks = srf_ceil_div(tilesY, 8);
        a.colBlocks = srf_ceil_div(tilesX, 8);
        a.nspatial = N * a.rowBlocks * a.colBlocks;
        const dim3 gd((unsigned)(((a.nspatial + 7) / 8) * 8 * a.coutBlocks));
        if (dbg == 1) hipLaunchKernelGGL((srf_wino3x3_k<1, 3>), gd, blk, WN_LDS_BYTES, (hipStream_t)stream, a);
        else if (dbg == 4) hipLaunchKernelGGL((srf_wino3x3_k<4, 3>), gd, blk, WN_LDS_BYTES, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((srf_wino3x3_k<8, 3>), gd, blk, WN_LDS_BYTES, (hipStream_t)stream, a);
        SRF_LAUNCH_CHECK();
        return SRF_OK;
    }
#endif
    // A last channel block with at most 32 real channels (Cout = 160, 224, ...) can run on the half-block kernel, whose
    // workgroups take ~0.62 of a full one, as a launch of its own.  Worth it when it saves rounds of 256 workgroups
    // (SRF_WINO_HALF=0 / 1 forces the choice: developer A/B knob).
    const long long sp8 = ((nspatial + 7) / 8) * 8;
    const int rem = Cout - (a.coutBlocks - 1) * 64;
    static const int force_half = srf_knob("SRF_WINO_HALF", -1);
    bool split = false, all_half = false;
    const int cus = srf_cu_count(dev);
    double rounds_best = (double)srf_ceil_div(sp8 * a.coutBlocks, cus);   // rounds of full workgroups
    if (rem <= 32) {
        // one mixed launch: the half workgroups (0.62 of a full one each) fill the tail of the full ones
        const double work = (double)(sp8 * (a.coutBlocks - 1)) + 0.62 * (double)sp8;
        const double then = a.coutBlocks > 1 ? std::max(work / cus + 0.3, (double)srf_ceil_div(sp8 * (a.coutBlocks - 1), cus))
                                             : 0.62 * (double)srf_ceil_div(sp8, cus);
        split = force_half < 0 ? then < rounds_best : force_half == 1;
        if (split) rounds_best = then;
    }
    // Every block as two half-block workgroups (32 channels each, the waves split the frequencies): measured, not faster
    // anywhere -- a launch of half blocks only runs at ~0.7 of the full form's rate (the input transform is done twice), so a
    // shorter last round does not pay it back (SECOND 128 -> 128 @ 184 x 184: 96 -> 104 us).  Kept behind SRF_WINO_HALF=2: the
    // parity test runs it against the full form.
    const int nhb = srf_ceil_div(Cout, 32);
    all_half = force_half == 2;
    if (all_half) split = false;
#define WN_LAUNCH(HB, GRID)                                                                                                    \
    do {                                                                                                                       \
        if (twl == 3) hipLaunchKernelGGL((srf_wino3x3_k<0, 3, HB>), dim3((unsigned)(GRID)), blk, WN_LDS_BYTES, (hipStream_t)stream, a);      \
        else if (twl == 2) hipLaunchKernelGGL((srf_wino3x3_k<0, 2, HB>), dim3((unsigned)(GRID)), blk, WN_LDS_BYTES, (hipStream_t)stream, a); \
        else hipLaunchKernelGGL((srf_wino3x3_k<0, 1, HB>), dim3((unsigned)(GRID)), blk, WN_LDS_BYTES, (hipStream_t)stream, a);               \
    } while (0)
    a.nfull = 0;
    a.ntail = 0;
#define WN_LAUNCH_MIXED(GRID)                                                                                                  \
    do {                                                                                                                       \
        const dim3 gm_((unsigned)(GRID));                                                                                      \
        if (twl == 3) hipLaunchKernelGGL((srf_wino3x3_mixed_k<3>), gm_, blk, WN_LDS_BYTES, (hipStream_t)stream, a);            \
        else if (twl == 2) hipLaunchKernelGGL((srf_wino3x3_mixed_k<2>), gm_, blk, WN_LDS_BYTES, (hipStream_t)stream, a);       \
        else hipLaunchKernelGGL((srf_wino3x3_mixed_k<1>), gm_, blk, WN_LDS_BYTES, (hipStream_t)stream, a);                     \
    } while (0)
    // a partly filled last round of full workgroups (SECOND 128 -> 128 @ 184 x 184: 288 work items on 256 CUs) as half-block
    // workgroups in the same launch, when they all fit beside each other (SRF_WINO_HALF=3 forces it on half of the items)
    const long long items = sp8 * a.coutBlocks, tail_items = items % cus;
    const bool tail = !split && !all_half && (force_half < 0 ? (items > cus && tail_items > 0 && 2 * tail_items <= cus) : force_half == 3);
    if (split && !all_half && a.coutBlocks > 1) {
        a.cb0 = 2 * (a.coutBlocks - 1);
        a.ncb = a.coutBlocks - 1;
        a.nfull = (int)(sp8 * a.ncb);
        WN_LAUNCH_MIXED(sp8 * a.coutBlocks);
    } else if (split) {
        a.cb0 = 2 * (a.coutBlocks - 1);
        a.ncb = 1;
        WN_LAUNCH(true, sp8);
    } else if (all_half) {
        a.cb0 = 0;
        a.ncb = nhb;
        WN_LAUNCH(true, sp8 * nhb);
    } else if (tail) {
        a.cb0 = -1;
        a.ncb = a.coutBlocks;
        a.ntail = (int)(force_half == 3 ? items / 2 : tail_items);
        a.nfull = (int)(items - a.ntail);
        WN_LAUNCH_MIXED((long long)a.nfull + 2ll * a.ntail);
    } else {
        WN_LAUNCH(false, blocks);
    }
#undef WN_LAUNCH_MIXED
#undef WN_LAUNCH
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" size_t srf_conv1x1_nhwc_packed_weight_bytes(int Cout, int K)
{
    if (Cout <= 0 || K <= 0 || (K & 31)) return 0;
    return (size_t)(K / 32) * srf_ceil_div(Cout, 256) * 2048 * 16;
}

extern "C" int srf_conv1x1_nhwc_pack_weights(const float *W, int Cout, int K, float *packed, srf_stream_t stream)
{
    if (Cout <= 0 || K <= 0 || !W || !packed) return SRF_EINVAL;
    if (K & 31) return SRF_EUNSUPPORTED;
    const int coutBlocks = srf_ceil_div(Cout, 256);
    const long long total = (long long)(K / 32) * coutBlocks * 8192;
    hipLaunchKernelGGL(srf_conv1x1_nhwc_pack_k, dim3(srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, W, Cout, K, coutBlocks,
                       packed, total);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// shared launcher: bpi == 0 -> flat row tiling; bpi > 0 -> per-image tiling with column sums into `colsum`
struct TopDown {
    const float *top;
    long long top_ld;
    int mapH, mapW, topH, topW;
};

static int conv1x1_launch(const float *x, long long M, int K, long long x_ld, const float *W_packed, int Cout, const float *scale,
                          const float *shift, int relu, float *y, long long y_ld, float *colsum, long long HW, hipStream_t stream,
                          int *bpi_out = nullptr, const TopDown *td = nullptr)
{
    GemmArgs a;
    a.x = x;
    a.y = y;
    a.Wp = reinterpret_cast<const f32x4 *>(W_packed);
    a.scale = scale;
    a.shift = shift;
    a.x_ld = x_ld;
    a.y_ld = y_ld;
    a.M = M;
    a.K = K;
    a.Cout = Cout;
    a.coutBlocks = srf_ceil_div(Cout, 256);
    a.nchunk = K / 32;
    a.relu = relu;
    a.H = a.W = a.Ho = a.Wo = a.kw = a.stride = a.pad = a.cin_chunks = 0;
    a.x_bytes = 0;
    a.colsum = colsum;
    a.HW = HW;
    a.top = td ? td->top : nullptr;
    a.top_ld = td ? td->top_ld : 0;
    a.mapH = td ? td->mapH : 0;
    a.mapW = td ? td->mapW : 0;
    a.topH = td ? td->topH : 0;
    a.topW = td ? td->topW : 0;
    a.sy = td ? (float)td->topH / (float)td->mapH : 0.f;
    a.sx = td ? (float)td->topW / (float)td->mapW : 0.f;
    int dev = 0;
    SRF_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return SRF_EUNSUPPORTED;
    static bool attr_set[64] = {false};
    constexpr int LDS_BIG = (8 * 256 + 8 * 256) * 16, LDS_STD = (8 * 128 + 8 * 128) * 16;
    if (!attr_set[dev]) {
        SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_conv1x1_nhwc_k<4, 4, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BIG));
        attr_set[dev] = true;
    }
    static const int big = srf_knob("SRF_GEMM_BIG", 0);
    int TM = (big && !colsum) ? 256 : 128, ncs = (big && !colsum) ? 1 : 2;
    // below ~0.8 rounds of 128 x 128 tiles at three per CU the 64 x 64 tiles win (stage 5 of VoVNet: 544 tiles, 417 -> 386 us);
    // SRF_GEMM_SMALL overrides the threshold (developer A/B knob)
    static const int small_thr = srf_knob("SRF_GEMM_SMALL", 640);
    if (srf_ceil_div(M, 128) * srf_ceil_div(Cout, 128) < small_thr) TM = 64, ncs = 4;  // small problem: 64 x 64 tiles
    a.row0 = 0;
    a.slot0 = 0;
    const long long nimg = colsum ? M / HW : 1;
    if (colsum) {
        a.bpi = (int)srf_ceil_div(HW, TM);
        a.slots = a.bpi;
        a.mblocks = nimg * a.bpi;
        if (bpi_out) *bpi_out = a.bpi;
    } else {
        a.bpi = 0;
        a.slots = 0;
        a.mblocks = srf_ceil_div(M, TM);
    }
    // tail of the last, partly filled round of 128 x 128 tiles as 64 x 64 tiles (SRF_GEMM_TAIL=0 turns it off; developer A/B knob)
    static const int tail_on = srf_knob("SRF_GEMM_TAIL", 1);
    static const double tail_frac = getenv("SRF_GEMM_TAIL_FRAC") ? atof(getenv("SRF_GEMM_TAIL_FRAC")) : 1.0;
    static const int tail_extra = srf_knob("SRF_GEMM_TAIL_EXTRA", 0);
    if (TM == 128 && tail_on) {
        const int cus = srf_cu_count(dev);
        const long long slots_cu = 3ll * cus, nct = srf_ceil_div(Cout, 128);
        const long long tiles = a.mblocks * nct;
        long long full = tiles / slots_cu;
        const double frac = (double)(tiles - full * slots_cu) / (double)slots_cu;
        if (full > tail_extra) full -= tail_extra;
        if (full >= 1 && frac > 0.0 && frac <= tail_frac) {
            // row blocks (per image) the big tiles keep: whole rounds, a multiple of 8 blocks in total where that is possible
            const long long keep = (full * slots_cu) / (nct * nimg);   // per image
            GemmArgs t = a;
            GemmArgs b = a;
            b.mblocks = keep * nimg;
            t.row0 = keep * 128;
            if (colsum) {
                b.bpi = (int)keep;
                t.bpi = (int)srf_ceil_div(HW - t.row0, 64);
                t.slot0 = (int)keep;
                b.slots = t.slots = b.bpi + t.bpi;
                t.mblocks = nimg * t.bpi;
                if (bpi_out) *bpi_out = b.slots;
            } else {
                t.mblocks = srf_ceil_div(M - t.row0, 64);
            }
            if (keep >= 1 && t.mblocks >= 1) {
                const long long gb = ((b.mblocks + 7) / 8) * 8 * nct, gt = ((t.mblocks + 7) / 8) * 8 * srf_ceil_div(Cout, 64);
                if (gb + gt >= (1ll << 31)) return SRF_EUNSUPPORTED;
                hipLaunchKernelGGL(srf_conv1x1_nhwc_mixed_k, dim3((unsigned)(gb + gt)), dim3(256), LDS_STD, stream, b, t, (unsigned)gb);
                SRF_LAUNCH_CHECK();
                return SRF_OK;
            }
            if (bpi_out && colsum) *bpi_out = a.bpi;
        }
    }
    const long long blocks = ((a.mblocks + 7) / 8) * 8 * srf_ceil_div(Cout, 256 / ncs);
    if (blocks >= (1ll << 31)) return SRF_EUNSUPPORTED;
    if (TM == 256)
        hipLaunchKernelGGL((srf_conv1x1_nhwc_k<4, 4, 1, false>), dim3((unsigned)blocks), dim3(256), LDS_BIG, stream, a);
    else if (TM == 64)
        hipLaunchKernelGGL((srf_conv1x1_nhwc_k<1, 1, 4, false>), dim3((unsigned)blocks), dim3(256), (8 * 64 + 8 * 64) * 16, stream, a);
    else
        hipLaunchKernelGGL((srf_conv1x1_nhwc_k<2, 2, 3, false>), dim3((unsigned)blocks), dim3(256), LDS_STD, stream, a);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

extern "C" int srf_conv1x1_nhwc(const float *x, long long M, int K, long long x_ld, const float *W_packed, int Cout, const float *scale,
                                const float *shift, int relu, float *y, long long y_ld, srf_stream_t stream)
{
    if (M < 0 || K <= 0 || Cout <= 0 || x_ld < K || y_ld < Cout) return SRF_EINVAL;
    if (M == 0) return SRF_OK;
    if (!x || !W_packed || !y) return SRF_EINVAL;
    if ((K & 31) || (x_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)W_packed & 15)) return SRF_EUNSUPPORTED;
    if (x_ld * 256 * 4 >= (1ll << 31)) return SRF_EUNSUPPORTED;
    return conv1x1_launch(x, M, K, x_ld, W_packed, Cout, scale, shift, relu, y, y_ld, nullptr, 0, (hipStream_t)stream);
}

// srf_conv1x1_nhwc_topdown: the lateral 1x1 convolution of an FPN level with the top-down step in its epilogue:
// y[n][py][px][co] = act(conv) + top[n][floor(py Ht / H)][floor(px Wt / W)][co] -- the `laterals[i - 1] += F.interpolate(
// laterals[i], size=..., mode="nearest")` of mmdet's FPN.forward without a pass of its own over the finer level (the same
// two floats are added as in srf_nhwc_upsample_add: identical bits).
extern "C" int srf_conv1x1_nhwc_topdown(const float *x, int N, int H, int W, int K, long long x_ld, const float *W_packed, int Cout,
                                        const float *scale, const float *shift, int relu, const float *top, int Ht, int Wt, long long top_ld,
                                        float *y, long long y_ld, srf_stream_t stream)
{
    if (N < 0 || H <= 0 || W <= 0 || Ht <= 0 || Wt <= 0 || K <= 0 || Cout <= 0 || x_ld < K || y_ld < Cout || top_ld < Cout) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!x || !W_packed || !y || !top) return SRF_EINVAL;
    if ((K & 31) || (x_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)W_packed & 15)) return SRF_EUNSUPPORTED;
    if (x_ld * 256 * 4 >= (1ll << 31) || (long long)N * Ht * Wt * top_ld >= (1ll << 31)) return SRF_EUNSUPPORTED;
    const TopDown td = {top, top_ld, H, W, Ht, Wt};
    return conv1x1_launch(x, (long long)N * H * W, K, x_ld, W_packed, Cout, scale, shift, relu, y, y_ld, nullptr, 0, (hipStream_t)stream,
                          nullptr, &td);
}

// srf_conv1x1_nhwc_pooled: the same convolution on N images of HW pixels each, plus mean[n][co] = the mean over the
// image's pixels of the stored outputs (after scale / shift / ReLU) -- VoVNet's eSE pooling (vovnet.py:165-177) without
// a second pass over the map.  Row blocks end with their image; a workgroup leaves the column sums of its block in the
// workspace, a second kernel adds the blocks of an image in a fixed order (deterministic, unlike an atomic reduction).
#define PM_GROUPS 16
// 256 threads = 16 channels x 16 groups; group g adds the blocks g, g + 16, ... of its image, the 16 group sums are then added
// in group order: a fixed order whatever the grid
__global__ __launch_bounds__(256) void srf_conv1x1_pool_finish_k(const float *__restrict__ partial, int bpi, int C, float inv,
                                                                 float *__restrict__ mean)
{
    __shared__ float s[PM_GROUPS][16];
    const int n = blockIdx.y, cl = threadIdx.x & 15, c = blockIdx.x * 16 + cl, g = threadIdx.x >> 4;
    float acc = 0.f;
    if (c < C)
        for (int b = g; b < bpi; b += PM_GROUPS) acc += partial[((long long)n * bpi + b) * C + c];
    s[g][cl] = acc;
    __syncthreads();
    if (g == 0 && c < C) {
        float t = s[0][cl];
#pragma unroll
        for (int k = 1; k < PM_GROUPS; ++k) t += s[k][cl];
        mean[(long long)n * C + c] = t * inv;
    }
}

extern "C" size_t srf_conv1x1_nhwc_pooled_workspace_bytes(int N, long long HW, int Cout)
{
    return (N <= 0 || HW <= 0 || Cout <= 0) ? 0 : (size_t)N * (size_t)srf_ceil_div(HW, 64) * Cout * 4;  // row blocks of >= 64
}

extern "C" int srf_conv1x1_nhwc_pooled(const float *x, int N, long long HW, int K, long long x_ld, const float *W_packed, int Cout,
                                       const float *scale, const float *shift, int relu, float *y, long long y_ld, float *mean, void *workspace,
                                       size_t workspace_bytes, srf_stream_t stream)
{
    if (N < 0 || HW <= 0 || K <= 0 || Cout <= 0 || x_ld < K || y_ld < Cout) return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!x || !W_packed || !y || !mean || !workspace) return SRF_EINVAL;
    if ((K & 31) || (x_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)W_packed & 15) || N > 65535) return SRF_EUNSUPPORTED;
    if (x_ld * 256 * 4 >= (1ll << 31)) return SRF_EUNSUPPORTED;
    if (workspace_bytes < srf_conv1x1_nhwc_pooled_workspace_bytes(N, HW, Cout)) return SRF_EWORKSPACE;
    int bpi = 0;
    const int rc = conv1x1_launch(x, (long long)N * HW, K, x_ld, W_packed, Cout, scale, shift, relu, y, y_ld, (float *)workspace, HW,
                                  (hipStream_t)stream, &bpi);
    if (rc != SRF_OK) return rc;
    hipLaunchKernelGGL(srf_conv1x1_pool_finish_k, dim3(srf_ceil_div(Cout, 16), N), dim3(256), 0, (hipStream_t)stream,
                       (const float *)workspace, bpi, Cout, 1.0f / (float)HW, mean);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// srf_conv_gemm_nhwc: Conv2d(Cin, Cout, (kh, kw), stride, padding) on channels-last activations as the same GEMM with an
// implicit im2col: rows = output pixels, k = (tap, input channel).  Used for the stride-2 3x3 layers (VoVNet stem_3, the
// first layer of SECONDCustom's second block, the extra levels of the BEV FPN): MIOpen's channels-last choice for them is
// a split-K kernel that accumulates with atomics (`igemm_fwd_gtcx35_nhwc_*_gkgs`), i.e. results that differ from run to
// run in the last bits -- here every output is one k-ordered fma chain.  W_packed = srf_conv1x1_nhwc_pack_weights of the
// weight reordered to (Cout, kh * kw * Cin) with the tap index slowest.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int srf_conv_gemm_nhwc(const float *x, int N, int H, int W, int Cin, long long x_ld, const float *W_packed, int Cout, int kh,
                                  int kw, int stride, int pad, const float *scale, const float *shift, int relu, float *y, long long y_ld,
                                  srf_stream_t stream)
{
    if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 || x_ld < Cin || y_ld < Cout)
        return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!x || !W_packed || !y) return SRF_EINVAL;
    if ((Cin & 31) || (x_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)W_packed & 15)) return SRF_EUNSUPPORTED;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return SRF_EINVAL;
    const long long x_bytes = (long long)N * H * W * x_ld * 4;
    if (x_bytes >= (1ll << 31)) return SRF_EUNSUPPORTED;
    GemmArgs a;
    a.x = x;
    a.y = y;
    a.Wp = reinterpret_cast<const f32x4 *>(W_packed);
    a.scale = scale;
    a.shift = shift;
    a.x_ld = x_ld;
    a.y_ld = y_ld;
    a.M = (long long)N * Ho * Wo;
    a.K = kh * kw * Cin;
    a.Cout = Cout;
    a.coutBlocks = srf_ceil_div(Cout, 256);
    a.nchunk = a.K / 32;
    a.relu = relu;
    a.H = H;
    a.W = W;
    a.Ho = Ho;
    a.Wo = Wo;
    a.kw = kw;
    a.stride = stride;
    a.pad = pad;
    a.cin_chunks = Cin / 32;
    a.x_bytes = x_bytes;
    a.colsum = nullptr;
    a.row0 = 0;
    a.slot0 = a.slots = 0;
    a.HW = 0;
    a.bpi = 0;
    a.top = nullptr;
    a.top_ld = 0;
    a.mapH = a.mapW = a.topH = a.topW = 0;
    a.sy = a.sx = 0.f;
    a.mblocks = srf_ceil_div(a.M, 128);
    long long blocks = ((a.mblocks + 7) / 8) * 8 * srf_ceil_div(Cout, 128);
    if (blocks >= (1ll << 31)) return SRF_EUNSUPPORTED;
    // count of 128-channel parts that hold real channels
    const long long live = a.mblocks * srf_ceil_div(Cout, 128);
    if (live < 384) {
        // a small map (the stride-2 layers of the BEV FPN: 46 x 46 outputs = 17 tiles of 128 pixels, each a chain of
        // 36 chunks on one CU with the rest of the chip idle): 64 x 64 tiles, four times the workgroups, a quarter of the chain
        a.mblocks = srf_ceil_div(a.M, 64);
        blocks = ((a.mblocks + 7) / 8) * 8 * srf_ceil_div(Cout, 64);
        hipLaunchKernelGGL((srf_conv1x1_nhwc_k<1, 1, 4, true>), dim3((unsigned)blocks), dim3(256), (8 * 64 + 8 * 64) * 16, (hipStream_t)stream, a);
    } else
        hipLaunchKernelGGL((srf_conv1x1_nhwc_k<2, 2, 3, true>), dim3((unsigned)blocks), dim3(256), (8 * 128 + 8 * 128) * 16, (hipStream_t)stream, a);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// srf_stem_conv_nchw: the first layer of the image backbone, Conv2d(Cin <= 4, 64, 3, stride 2, padding 1) + scale / shift
// + ReLU, reading the NCHW camera images and writing channels-last (VoVNet stem_1: 3 -> 64 on 6 x 928 x 1600; 7.7 GFLOP,
// HBM-bound: 107 MB in, 570 MB out).  A workgroup owns a tile of 4 x 64 output pixels: its 9 x 129 input patch per channel
// is read once, row by row (coalesced; the earlier versions gathered the 9 Cin taps of every pixel with stride-2 loads),
// into LDS with the even and the odd columns apart, so that the A operand of tap (ky, kx) is a unit-stride read:
// P[ci][row][column parity][65].  Wave w multiplies the 64 pixels of output row w by the 64 channels on
// v_mfma_f32_32x32x2_f32: k = ci * 9 + ky * 3 + kx ascending, one fma chain per output, zero taps add +0.
// ---------------------------------------------------------------------------------------------------------------------
#define ST_R 4
#define ST_C 64
#define ST_ROWP 130
template <int CIN>   // the input channel count at compile time: the k loop unrolls and its LDS reads batch (a run-time K kept the chain
                     // offset table -> operand -> MFMA serial per step)
__global__ __launch_bounds__(256) void srf_stem_conv_nchw_k(const float *__restrict__ x, int N, int Cin_rt, int H, int W, int Ho, int Wo,
                                                           const float *__restrict__ wt, const float *__restrict__ scale,
                                                           const float *__restrict__ shift, int relu, float *__restrict__ y, long long y_ld,
                                                           int tilesX, int tilesY)
{
    __shared__ float s_p[4 * 9 * ST_ROWP];  // [ci][patch row 0 .. 8][column parity][65]
    __shared__ float s_w[36][65];           // [k][channel] (rows padded: the transposing copy below writes a column per lane group)
    __shared__ int s_off[36];               // patch offset of tap k
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    constexpr int Cin = CIN, K = CIN * 9, KP = (K + 1) & ~1;
    const int bx = blockIdx.x % tilesX, rest = blockIdx.x / tilesX;
    const int by = rest % tilesY, n = rest / tilesY;
    const int ox0 = bx * ST_C, oy0 = by * ST_R;
    // weights (Cout, K) -> s_w[k][co]: consecutive lanes read consecutive k of one output channel (runs of K floats; with consecutive
    // co per lane every load instruction touched 64 lines K floats apart, ~1500 line lookups per workgroup for 7 KB)
    for (int e = tid; e < KP * 64; e += 256) {
        const int co = e / KP, k = e - co * KP;
        s_w[k][co] = k < K ? wt[(size_t)co * K + k] : 0.f;
    }
    if (tid < KP) {
        const int k = tid < K ? tid : 0;
        const int ci = k / 9, r = k - ci * 9, ky = r / 3, kx = r - ky * 3;
        s_off[tid] = (ci * 9 + ky) * ST_ROWP + (kx & 1) * 65 + (kx >> 1);
    }
    {
        // the patch: all of a thread's loads first (a constant trip count, fully unrolled), then its LDS stores.  Written as
        // `for (e = tid; e < total; e += 256) s_p[...] = xn[...]` the loop kept its run-time trip count and every iteration waited
        // for its own 4-byte load: 14 memory latencies in a row per workgroup, 290 us for the layer at 2.3 TB/s.
        const int cols = 2 * ST_C + 1, total = Cin * 9 * cols;
        const float *xn = x + (size_t)n * Cin * H * W;
        constexpr int NL = (CIN * 9 * (2 * ST_C + 1) + 255) / 256;
        float v[NL];
        int dst[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * 256;
            const int r = e / cols, c = e - r * cols;
            const int ci = r / 9, iyl = r - ci * 9;
            const int iy = 2 * oy0 - 1 + iyl, ix = 2 * ox0 - 1 + c;
            const bool ok = e < total && iy >= 0 && iy < H && ix >= 0 && ix < W;
            v[i] = ok ? xn[((size_t)ci * H + iy) * W + ix] : 0.f;
            dst[i] = e < total ? r * ST_ROWP + (c & 1) * 65 + (c >> 1) : -1;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if (dst[i] >= 0) s_p[dst[i]] = v[i];
    }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int ksteps = KP >> 1;
    const int a_base = 2 * wave * ST_ROWP + li;   // output row `wave` of the tile: patch rows 2 wave + ky
#pragma unroll
    for (int s2 = 0; s2 < ksteps; ++s2) {
        const int k = 2 * s2 + lh;
        const int off = s_off[k] + a_base;
        const bool live = k < K;
        const float a0 = live ? s_p[off] : 0.f, a1 = live ? s_p[off + 32] : 0.f;
        const float b0 = s_w[k][li], b1 = s_w[k][32 + li];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    float sc[2], sh[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        sc[j] = scale ? scale[j * 32 + li] : 1.f;
        sh[j] = shift ? shift[j * 32 + li] : 0.f;
    }
    // stores through a buffer descriptor of image n: a pixel outside the map gets an offset beyond the range and is dropped
    __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(y + (long long)n * Ho * Wo * y_ld, 0,
                                                                     (int)((long long)Ho * Wo * y_ld * 4), 0x00020000);
    const int oy = oy0 + wave;
    const unsigned px_b = (unsigned)(y_ld * 4);
    const unsigned row_off = (unsigned)(((long long)oy * Wo + ox0) * y_ld * 4) + (unsigned)(li * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int px = i * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
            const bool ok = oy < Ho && ox0 + px < Wo;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float o = acc[i][j][r];
                if (scale) o = __fmaf_rn(o, sc[j], sh[j]);
                else if (shift) o = __fadd_rn(o, sh[j]);
                if (relu) o = fmaxf(o, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), yrsrc, ok ? (int)(row_off + (unsigned)px * px_b + j * 128) : (int)0x80000000, 0, 0);
            }
        }
}

extern "C" int srf_stem_conv_nchw(const float *x, int N, int Cin, int H, int W, const float *Wt, int Cout, const float *scale,
                                  const float *shift, int relu, float *y, long long y_ld, srf_stream_t stream)
{
    if (N < 0 || Cin <= 0 || H <= 0 || W <= 0 || y_ld < Cout) return SRF_EINVAL;
    if (Cin > 4 || Cout != 64 || (y_ld & 3) || ((uintptr_t)y & 15)) return SRF_EUNSUPPORTED;
    if (N == 0) return SRF_OK;
    if (!x || !Wt || !y) return SRF_EINVAL;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if ((long long)Ho * Wo * y_ld * 4 >= (1ll << 31)) return SRF_EUNSUPPORTED;   // buffer-descriptor range of one image's output
    const int tilesX = srf_ceil_div(Wo, ST_C), tilesY = srf_ceil_div(Ho, ST_R);
    const long long blocks = (long long)N * tilesX * tilesY;
    if (blocks >= (1ll << 31)) return SRF_EUNSUPPORTED;
#define SRF_STEM(C)                                                                                                             \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_stem_conv_nchw_k<C>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, N, Cin, H, W, Ho, Wo, \
                       Wt, scale, shift, relu, y, y_ld, tilesX, tilesY)
    switch (Cin) {
    case 1: SRF_STEM(1); break;
    case 2: SRF_STEM(2); break;
    case 3: SRF_STEM(3); break;
    default: SRF_STEM(4); break;
    }
#undef SRF_STEM
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
