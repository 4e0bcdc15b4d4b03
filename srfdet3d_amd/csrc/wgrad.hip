// wgrad.hip -- srf_conv_wgrad_nhwc: the weight gradient of a stride-1 convolution (1x1, or 3x3 with padding 1) on channels-last
// tensors, for the trainable layers of config 4 (tools/train.py:220-234 trains VoVNet stages 4-5, the image FPN, the head's
// img_convs; mmdet3d_plugin/models/backbones/vovnet.py:354-374; torch autograd runs these on MIOpen's `igemm_wrw_*_gkgs` kernels:
// float-atomic split-K, results differing from run to run, 57 ms of a 268 ms step, and the 1x1 layers on a rocBLAS TN GEMM, 26 ms).
//
//     dW[co][ci][ky][kx] = sum over pixels p = (n, y, x) of  g[p][co] * X[n][y + ky - pad][x + kx - pad][ci]
//
// = a GEMM  C (Cout x taps Cin) = G^T (Cout x P)  .  im2col(X) (P x taps Cin)  whose REDUCTION runs over the pixels -- the slow index of
// both operands (channels-last rows).  Computed here as an f32 GEMM on the bf16 MFMA through the exact three-way split of
// gemm_split.hip (x = xh + xm + xl in bf16, six of the nine exact partial products accumulated in f32: f32 accuracy at 6/16 of the
// f32 MFMA's cycles), with both operands split on the fly (both are activations) and the transposition done by the LDS:
//   * a K chunk = 32 consecutive output pixels; a workgroup tile = 128 output channels x 128 input channels of ONE tap; 4 waves =
//     2 x 2 wave tiles of 64 x 64 (four 32 x 32 accumulator tiles each), 48 MFMAs of 32x32x16 per chunk and wave;
//   * loads: thread (pixel t / 8, quad t % 8 + 8 j) reads 4 x 16 bytes of g and 4 x 16 bytes of X (the tap's shifted pixel; a pixel
//     outside the image or a channel past the layer's is an out-of-range buffer offset and reads as zero), splits them into the three
//     planes and stores 8 bytes per plane into ROW-major images [pixel 32][channel 128] bf16 (256-byte rows, 16-byte chunks
//     XOR-swizzled by ((row & 3) << 2) | ((row >> 2) & 3): layout (b) of cdna_hip_programming.md T10);
//   * fragments: the MFMA wants, per lane, 8 consecutive k = 8 consecutive PIXELS of one channel -- a column of the image.
//     ds_read_b64_tr_b16 reads a 4-row x 16-column block per 16 lanes and hands every lane one column: two of them per operand
//     fragment, conflict-free on that layout;
//   * the pixels are cut into `nsplit` ranges (so that a layer fills the chip: 36 tiles x 21 ranges for 256 -> 256); every range
//     writes its partial C to a workspace and srf_wgrad_reduce_k adds the ranges IN ORDER into dW in the (Cout, Cin, kh, kw) layout
//     of nn.Conv2d.weight: deterministic, bitwise repeatable.
// Not the bits of an f32 chain (16 products per MFMA are summed inside the instruction; the ranges are added last), within the
// same error bound: tests/test_gpu_wgrad.py holds it to float64 autograd.
#include "common.hpp"

typedef __bf16 wg_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf2 __attribute__((ext_vector_type(2)));
typedef float wg_f2 __attribute__((ext_vector_type(2)));
typedef float wg_f4 __attribute__((ext_vector_type(4)));
typedef float wg_f16 __attribute__((ext_vector_type(16)));
typedef unsigned wg_u2 __attribute__((ext_vector_type(2)));
typedef unsigned wg_u4 __attribute__((ext_vector_type(4)));

#define WG_PLANE 8192              /* bytes of one plane image: 32 pixels x 256 B */
#define WG_OPER (3 * WG_PLANE)     /* the three planes of one operand */

struct WgArgs {
    const float *g, *x;
    float *partial;
    long long g_ld, x_ld, P;
    int N, H, W, Cin, Cout, kw, pad, taps;
    int mtiles, ctiles, nsplit, chunks_per_split, nchunks;
    int g_bytes, x_bytes;
    int flat;   // column tiles cut the flattened (tap, input channel) axis in 128s (Cin % 32 == 0) instead of every tap's channels apart
};

__device__ __forceinline__ unsigned wg_pk_bf16(float a, float b)
{
    const wg_f2 v = {a, b};
    const wg_bf2 h = __builtin_convertvector(v, wg_bf2);   // v_cvt_pk_bf16_f32: round to nearest even
    return *reinterpret_cast<const unsigned *>(&h);
}

// (x0, x1) -> the packed pairs (h0, h1), (m0, m1), (l0, l1) with x = h + m + l exactly (gemm_split.hip, "Domain")
__device__ __forceinline__ void wg_split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l)
{
    h = wg_pk_bf16(x0, x1);
    const float r0 = __fsub_rn(x0, __uint_as_float(h << 16)), r1 = __fsub_rn(x1, __uint_as_float(h & 0xffff0000u));
    m = wg_pk_bf16(r0, r1);
    const float s0 = __fsub_rn(r0, __uint_as_float(m << 16)), s1 = __fsub_rn(r1, __uint_as_float(m & 0xffff0000u));
    l = wg_pk_bf16(s0, s1);
}

// ds_read_b64_tr_b16 with a literal offset (the macro arguments are literals at every use)
#define WG_TR(DST, ADDR, IMM) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(IMM) : "memory")

__global__ __launch_bounds__(256, 2) void srf_wgrad_split_k(WgArgs a)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * WG_OPER];   // planes of the g tile | planes of the X tile
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // work item -> (pixel range, column tile = (tap, input-channel tile), row tile)
    int b = blockIdx.x;
    const int mt = b % a.mtiles;
    b /= a.mtiles;
    // column tiles: `flat` (Cin % 32 == 0): 128 consecutive columns of the (tap, input channel) axis -- a tile may straddle taps, every
    // 32-channel group of it lies in one (192 -> 192, VoVNet stage 4: 14 column tiles instead of 9 x 2 half-empty ones); else the tiles
    // of every tap apart
    const int ntiles = a.flat ? a.ctiles : a.taps * a.ctiles;
    const int nt = b % ntiles, sp = b / ntiles;
    const int co0 = mt * 128;
    int tapj[4], cij0[4], kyj[4], kxj[4];   // per 32-channel group j of the tile: its tap and its first input channel (uniform: SGPRs)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int tp_, c0_;
        if (a.flat) {
            const int f0 = nt * 128 + 32 * j;
            tp_ = f0 / a.Cin;
            c0_ = f0 - tp_ * a.Cin;
            if (tp_ >= a.taps) {   // past the last column: nothing to load (channel a.Cin does not exist)
                tp_ = a.taps - 1;
                c0_ = a.Cin;
            }
        } else {
            tp_ = nt / a.ctiles;
            c0_ = (nt - tp_ * a.ctiles) * 128 + 32 * j;
        }
        tapj[j] = tp_;
        cij0[j] = c0_;
        kyj[j] = tp_ / a.kw;
        kxj[j] = tp_ - kyj[j] * a.kw;
    }
    const int c_begin = sp * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    c_end = c_end < a.nchunks ? c_end : a.nchunks;

    __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.g), 0, a.g_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, a.x_bytes, 0x00020000);

    // ---- loader: this thread's pixel of the chunk and its four channel quads ----
    const int pxl = tid >> 3, qg = tid & 7;
    long long p = (long long)c_begin * 32 + pxl;           // output pixel of the NEXT chunk to be loaded
    int ox, oy, n;
    {
        const long long hw = (long long)a.H * a.W;
        n = (int)(p / hw);
        const int rem = (int)(p - (long long)n * hw);
        oy = rem / a.W;
        ox = rem - oy * a.W;
    }
    unsigned gcol[4], xcol[4];   // byte offset of the quad inside a pixel row, out of range when the channel does not exist
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * qg + 32 * j;
        gcol[j] = co0 + c < a.Cout ? (unsigned)((co0 + c) * 4) : 0x80000000u;
        xcol[j] = cij0[j] + 4 * qg < a.Cin ? (unsigned)((cij0[j] + 4 * qg) * 4) : 0x80000000u;
    }
    wg_f4 graw[4], xraw[4];
#define WG_LOAD()                                                                                                          \
    do {                                                                                                                   \
        const bool pok_ = p < a.P;                                                                                         \
        const unsigned grow_ = pok_ ? (unsigned)(p * a.g_ld * 4) : 0x80000000u;                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                 \
            const int iy_ = oy + kyj[j_] - a.pad, ix_ = ox + kxj[j_] - a.pad;                                              \
            const bool xok_ = pok_ && iy_ >= 0 && iy_ < a.H && ix_ >= 0 && ix_ < a.W;                                      \
            const unsigned xrow_ = xok_ ? (unsigned)((((long long)n * a.H + iy_) * a.W + ix_) * a.x_ld * 4) : 0x80000000u; \
            const unsigned go_ = ((grow_ | gcol[j_]) & 0x80000000u) ? 0x80000000u : grow_ + gcol[j_];                      \
            const unsigned xo_ = ((xrow_ | xcol[j_]) & 0x80000000u) ? 0x80000000u : xrow_ + xcol[j_];                      \
            auto vg_ = __builtin_amdgcn_raw_buffer_load_b128(gr, (int)go_, 0, 0);                                          \
            graw[j_] = *reinterpret_cast<wg_f4 *>(&vg_);                                                                   \
            auto vx_ = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)xo_, 0, 0);                                          \
            xraw[j_] = *reinterpret_cast<wg_f4 *>(&vx_);                                                                   \
        }                                                                                                                  \
        p += 32;                                                                                                           \
        ox += 32;                                                                                                          \
        if (ox >= a.W) {                                                                                                   \
            ox -= a.W;                                                                                                     \
            if (++oy >= a.H) {                                                                                             \
                oy = 0;                                                                                                    \
                ++n;                                                                                                       \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)

    // split registers: [operand][plane][quad j] = 4 bf16 (two packed pairs)
    unsigned spl[2][3][4][2];
#define WG_SPLIT()                                                                                                         \
    do {                                                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                 \
            wg_split2(graw[j_][0], graw[j_][1], spl[0][0][j_][0], spl[0][1][j_][0], spl[0][2][j_][0]);                     \
            wg_split2(graw[j_][2], graw[j_][3], spl[0][0][j_][1], spl[0][1][j_][1], spl[0][2][j_][1]);                     \
            wg_split2(xraw[j_][0], xraw[j_][1], spl[1][0][j_][0], spl[1][1][j_][0], spl[1][2][j_][0]);                     \
            wg_split2(xraw[j_][2], xraw[j_][3], spl[1][0][j_][1], spl[1][1][j_][1], spl[1][2][j_][1]);                     \
        }                                                                                                                  \
    } while (0)
    // store addresses: row pxl, chunk (qg >> 1) + 4 j, half qg & 1
    const int swr = ((pxl & 3) << 2) | ((pxl >> 2) & 3);
    unsigned sto[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sto[j] = (unsigned)(256 * pxl + 16 * ((((qg >> 1) + 4 * j)) ^ swr) + 8 * (qg & 1));
#define WG_STORE()                                                                                                         \
    do {                                                                                                                   \
        _Pragma("unroll") for (int o_ = 0; o_ < 2; ++o_)                                                                   \
            _Pragma("unroll") for (int pl_ = 0; pl_ < 3; ++pl_)                                                            \
                _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                           \
                    *reinterpret_cast<wg_u2 *>(lds + o_ * WG_OPER + pl_ * WG_PLANE + sto[j_]) = wg_u2{spl[o_][pl_][j_][0], spl[o_][pl_][j_][1]}; \
    } while (0)

    // ---- fragments: lane (group g = lane / 16, i = lane % 16 = 4 q + p) of a transposed read supplies the address of block row q,
    // columns 4 p .. 4 p + 3; the block of (k step, 32-row tile T, half h) starts at pixel 16 ks + 8 (g >> 1) + 4 h, channel
    // 32 T + 16 (g & 1) ----
    const int wm = wave & 1, wn = wave >> 1;
    const int grp = lane >> 4, li16 = lane & 15, tq = li16 >> 2, tp = li16 & 3;
    unsigned fad[2][2][2];   // [operand][tile of the wave's two][half], LDS byte address for k step 0, plane 0
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int T = (o == 0 ? wm : wn) * 2 + t;
                const int row = 8 * (grp >> 1) + 4 * h + tq;
                const int ch = T * 4 + 2 * (grp & 1) + (tp >> 1);
                const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
                fad[o][t][h] = lds0 + (unsigned)(o * WG_OPER + 256 * row + 16 * (ch ^ sw) + 8 * (tp & 1));
            }
    wg_f16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    wg_bf8 fa[3][2], fb[3][2];
    // k step KS: rows + 16 = + 4096 bytes (the swizzle term (row >> 2) & 3 is unchanged by + 16 rows); plane PL: + WG_PLANE -- both as
    // the instruction's immediate offset: 8 address registers serve the 48 transposed reads of a chunk
#define WG_FRAG(DST, O, T, PL, KS)                                                                                         \
    do {                                                                                                                   \
        wg_u2 r0_, r1_;                                                                                                    \
        WG_TR(r0_, fad[O][T][0], (KS) * 4096 + (PL) * WG_PLANE);                                                           \
        WG_TR(r1_, fad[O][T][1], (KS) * 4096 + (PL) * WG_PLANE);                                                           \
        DST##_raw = wg_u4{r0_[0], r0_[1], r1_[0], r1_[1]};                                                                 \
    } while (0)
    wg_u4 fa00_raw, fa01_raw, fa10_raw, fa11_raw, fa20_raw, fa21_raw, fb00_raw, fb01_raw, fb10_raw, fb11_raw, fb20_raw, fb21_raw;
#define WG_READ(KS)                                                                                                        \
    do {                                                                                                                   \
        WG_FRAG(fa00, 0, 0, 0, KS); WG_FRAG(fa01, 0, 1, 0, KS); WG_FRAG(fb00, 1, 0, 0, KS); WG_FRAG(fb01, 1, 1, 0, KS);    \
        WG_FRAG(fa10, 0, 0, 1, KS); WG_FRAG(fa11, 0, 1, 1, KS); WG_FRAG(fb10, 1, 0, 1, KS); WG_FRAG(fb11, 1, 1, 1, KS);    \
        WG_FRAG(fa20, 0, 0, 2, KS); WG_FRAG(fa21, 0, 1, 2, KS); WG_FRAG(fb20, 1, 0, 2, KS); WG_FRAG(fb21, 1, 1, 2, KS);    \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa00_raw), "+v"(fa01_raw), "+v"(fa10_raw), "+v"(fa11_raw), "+v"(fa20_raw), "+v"(fa21_raw)); \
        asm volatile("" : "+v"(fb00_raw), "+v"(fb01_raw), "+v"(fb10_raw), "+v"(fb11_raw), "+v"(fb20_raw), "+v"(fb21_raw)); \
        fa[0][0] = *reinterpret_cast<const wg_bf8 *>(&fa00_raw); fa[0][1] = *reinterpret_cast<const wg_bf8 *>(&fa01_raw);   \
        fa[1][0] = *reinterpret_cast<const wg_bf8 *>(&fa10_raw); fa[1][1] = *reinterpret_cast<const wg_bf8 *>(&fa11_raw);   \
        fa[2][0] = *reinterpret_cast<const wg_bf8 *>(&fa20_raw); fa[2][1] = *reinterpret_cast<const wg_bf8 *>(&fa21_raw);   \
        fb[0][0] = *reinterpret_cast<const wg_bf8 *>(&fb00_raw); fb[0][1] = *reinterpret_cast<const wg_bf8 *>(&fb01_raw);   \
        fb[1][0] = *reinterpret_cast<const wg_bf8 *>(&fb10_raw); fb[1][1] = *reinterpret_cast<const wg_bf8 *>(&fb11_raw);   \
        fb[2][0] = *reinterpret_cast<const wg_bf8 *>(&fb20_raw); fb[2][1] = *reinterpret_cast<const wg_bf8 *>(&fb21_raw);   \
    } while (0)
#define WG_MM(PA, PB)                                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                   \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][i_], fb[PB][j_], acc[i_][j_], 0, 0, 0)
    // the six products of a k step, smallest terms first: l h, h l, m m (2^-16), m h, h m (2^-8), h h
#define WG_MFMA6() do { WG_MM(2, 0); WG_MM(0, 2); WG_MM(1, 1); WG_MM(1, 0); WG_MM(0, 1); WG_MM(0, 0); } while (0)

    if (c_begin < c_end) {
        WG_LOAD();
        WG_SPLIT();
        WG_STORE();
        WG_LOAD();   // (a chunk past the range reads pixels that belong to the next range or past the end: loaded, never multiplied)
        __syncthreads();
        for (int c = c_begin; c < c_end; ++c) {
            WG_READ(0);
            WG_MFMA6();
            WG_SPLIT();      // the next chunk's operands: vector work beside this chunk's MFMAs
            WG_READ(1);
            WG_MFMA6();
            __syncthreads();
            WG_STORE();
            WG_LOAD();
            __syncthreads();
        }
    }
#undef WG_LOAD
#undef WG_SPLIT
#undef WG_STORE
#undef WG_READ
#undef WG_FRAG
#undef WG_MM
#undef WG_MFMA6

    // ---- epilogue: accumulator register r of tile (i, j) = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the 32 x 32 tile, column lane & 31 ----
    const int li = lane & 31, lh = lane >> 5;
    const long long NC = (long long)a.taps * a.Cin;
    float *dst = a.partial + (size_t)sp * a.Cout * NC;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int jj = wn * 2 + j;                     // the 32-channel group of the tile this accumulator tile holds
            const int ci = cij0[jj] + li;
            if (ci >= a.Cin) continue;
            const size_t col = (size_t)tapj[jj] * a.Cin + ci;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < a.Cout) dst[(size_t)co * NC + col] = acc[i][j][r];
            }
        }
}

// dW[co][ci][tap] = sum over the pixel ranges s = 0, 1, ... (in this order) of partial[s][co][tap Cin + ci]
__global__ __launch_bounds__(256) void srf_wgrad_reduce_k(const float *__restrict__ partial, int nsplit, int Cout, int Cin, int taps,
                                                        float *__restrict__ dW)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long NC = (long long)taps * Cin, total = (long long)Cout * NC;
    if (e >= total) return;
    const int co = (int)(e / NC);
    const int rem = (int)(e - (long long)co * NC);
    const int tap = rem / Cin, ci = rem - tap * Cin;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s = __fadd_rn(s, partial[(size_t)k * total + e]);
    dW[((size_t)co * Cin + ci) * taps + tap] = s;
}

static bool wg_flat(int Cin, int taps) { return taps > 1 && (Cin & 31) == 0 && (Cin & 127) != 0; }

static int wg_plan(long long P, int Cin, int Cout, int taps, int *mtiles, int *ctiles, int *nsplit, int *cps, int *nchunks)
{
    *mtiles = srf_ceil_div(Cout, 128);
    *ctiles = wg_flat(Cin, taps) ? srf_ceil_div(taps * Cin, 128) : srf_ceil_div(Cin, 128);
    const long long nch = (P + 31) / 32;
    if (nch >= (1ll << 30)) return SRF_EUNSUPPORTED;
    *nchunks = (int)nch;
    const int tiles = *mtiles * (wg_flat(Cin, taps) ? *ctiles : *ctiles * taps);
    int ns = srf_ceil_div(768, tiles);          // ~1.5 rounds of 512 co-resident workgroups
    const int most = (int)((nch + 15) / 16);    // a range is at least 16 chunks deep
    ns = ns > most ? most : ns;
    ns = ns < 1 ? 1 : ns;
    *cps = (int)((nch + ns - 1) / ns);
    *nsplit = (int)((nch + *cps - 1) / *cps);
    return SRF_OK;
}

extern "C" size_t srf_conv_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int ksize)
{
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3)) return 0;
    int mt, ct, ns, cps, nch;
    if (wg_plan((long long)N * H * W, Cin, Cout, ksize * ksize, &mt, &ct, &ns, &cps, &nch) != SRF_OK) return 0;
    return (size_t)ns * Cout * (size_t)(ksize * ksize) * Cin * sizeof(float);
}

// g: (N, H, W, g_ld) channels-last gradient of the layer's output (Cout channels); x: (N, H, W, x_ld) its input (Cin channels);
// dW: (Cout, Cin, ksize, ksize) contiguous.  ksize 1 (padding 0) or 3 (padding 1), stride 1.  Cin, Cout multiples of 4, W >= 32,
// every tensor below 2 GB, g_ld / x_ld multiples of 4 with 16-byte aligned bases; else SRF_EUNSUPPORTED.
extern "C" int srf_conv_wgrad_nhwc(const float *g, long long g_ld, const float *x, long long x_ld, int N, int H, int W, int Cin, int Cout,
                                   int ksize, void *workspace, size_t workspace_bytes, float *dW, srf_stream_t stream)
{
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !g || !x || !dW || !workspace) return SRF_EINVAL;
    if ((ksize != 1 && ksize != 3) || (Cin & 3) || (Cout & 3) || W < 32 || (g_ld & 3) || (x_ld & 3) || g_ld < Cout || x_ld < Cin)
        return SRF_EUNSUPPORTED;
    if (((uintptr_t)g & 15) || ((uintptr_t)x & 15)) return SRF_EUNSUPPORTED;
    const long long P = (long long)N * H * W;
    if (P * g_ld * 4 >= (1ll << 31) || P * x_ld * 4 >= (1ll << 31)) return SRF_EUNSUPPORTED;
    WgArgs a;
    a.g = g;
    a.x = x;
    a.partial = (float *)workspace;
    a.g_ld = g_ld;
    a.x_ld = x_ld;
    a.P = P;
    a.N = N;
    a.H = H;
    a.W = W;
    a.Cin = Cin;
    a.Cout = Cout;
    a.kw = ksize;
    a.pad = ksize / 2;
    a.taps = ksize * ksize;
    const int rc = wg_plan(P, Cin, Cout, a.taps, &a.mtiles, &a.ctiles, &a.nsplit, &a.chunks_per_split, &a.nchunks);
    if (rc != SRF_OK) return rc;
    if (workspace_bytes < (size_t)a.nsplit * Cout * (size_t)a.taps * Cin * sizeof(float)) return SRF_EINVAL;
    a.g_bytes = (int)(P * g_ld * 4);
    a.x_bytes = (int)(P * x_ld * 4);
    a.flat = wg_flat(Cin, a.taps) ? 1 : 0;
    const long long blocks = (long long)a.nsplit * (a.flat ? a.ctiles : a.taps * a.ctiles) * a.mtiles;
    if (blocks >= (1ll << 31)) return SRF_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(srf_wgrad_split_k, dim3((unsigned)blocks), dim3(256), 0, st, a);
    const long long total = (long long)Cout * a.taps * Cin;
    hipLaunchKernelGGL(srf_wgrad_reduce_k, dim3((unsigned)srf_ceil_div(total, 256)), dim3(256), 0, st, a.partial, a.nsplit, Cout, Cin, a.taps, dW);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
