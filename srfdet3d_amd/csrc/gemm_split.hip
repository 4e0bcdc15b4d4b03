// gemm_split.hip -- srf_conv1x1_nhwc_split: the 1x1 convolutions of the camera branch (VoVNet's OSA `concat` layers,
// mmdet3d_plugin/models/backbones/vovnet.py:222-223, with the eSE average pool of :165-177 from the same pass; the image
// FPN's lateral convolutions with the top-down step) as an f32 GEMM computed on the bf16 MFMA.
//
// Why.  v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate (64 FLOP / clk / SIMD); srf_gemm_direct_k already delivers that
// rate at the clock the chip holds (129-131 TFLOP/s = 0.82 of the 157.3 nominal), so the 1.80 TFLOP of these layers cost
// 13.9 ms of a 38.7 ms LC frame and cannot get cheaper on that instruction.  v_mfma_f32_32x32x16_bf16 does 16x the FLOPs per
// clock.  An f32 value is the EXACT sum of three bf16 values,
//     x = xh + xm + xl,   xh = bf16(x), xm = bf16(x - xh), xl = bf16((x - xh) - xm)      (round to nearest even; 8 + 8 + 8 bits),
// so a product a b is exactly the sum of nine bf16 x bf16 products, each of which is exact in f32.  The six of relative
// magnitude >= 2^-16 (hh, hm, mh, mm, hl, lh) are accumulated here, in f32, on the bf16 MFMA: 6/16 of the f32 MFMA's
// cycles.  The three dropped terms (ml, lm, ll) are together below 2^-23 |a b| -- the size of ONE f32 rounding of the product,
// which the f32 fma chain commits for every term anyway.  Measured against float64 (tools/micro/gemm_split_bench.hip, the
// five GEMM shapes of an LC frame, post-ReLU-like data): max error / sum |a b| = 2.0-2.8e-7 for this kernel, 1.7-3.0e-7 for the
// f32 MFMA chain.  It is NOT bit-identical to the f32 chain (another summation order: 16 products per MFMA are summed inside
// the instruction), deterministic, and within the same bound; tests/test_gpu_gemm_split.py holds it to the f32 chain's own
// error against float64.  SRF_GEMM_SPLIT=0 (ops.conv1x1_nhwc) keeps these layers on the f32-MFMA kernels.
//
// Domain (what "exact split" covers, and what happens outside; tests/test_gpu_gemm_split.py holds every line of this):
//   * x = xh + xm + xl EXACTLY for every finite x with |x| <= 0x1.FEp127 (the largest bf16, 3.3895e38) whose lowest set bit is
//     worth >= 2^-133 (the smallest bf16 subnormal): zero, and every |x| >= 2^-110.  Rows that mix magnitudes (2^-60 ... 2^+60 in
//     one dot product) and sums that cancel to 1e-6 of sum |a b| are inside the domain: the bound is relative to sum |a b|, as the
//     f32 chain's own.
//   * tiny operands, |x| < 2^-110 (f32 subnormals included): the low planes fall under the bf16 range and are rounded to the bf16
//     subnormal grid or flushed, an ABSOLUTE error of at most 2^-126 per operand (times its partner): the result carries an extra
//     <= 4 x 2^-126 (sum |a| + sum |b|) over the terms in question -- a flush-to-zero-sized effect, below anything a network's
//     activations can resolve.
//   * |x| in (3.3895e38, FLT_MAX]: bf16(x) rounds to infinity, the remainders become inf - inf: every output of that row (activation)
//     or column (weight) is NaN where the f32 chain stays finite.  OUTSIDE the domain.  Weights are checked when they are packed
//     (ops.gemm_split_weight_in_domain: such a layer stays on the f32-MFMA kernels); activations are not checked -- a value of
//     3.4e38 behind a BatchNorm means the network has already diverged.
//   * +-inf / NaN inputs: every output that the f32 chain makes non-finite (inf or NaN) is non-finite here too and vice versa, but
//     an infinity arrives as NaN (inf - bf16(inf) = NaN in the low planes; inf x 0 in a zero plane) -- same finiteness pattern,
//     NaN in place of +-inf.
//
// Structure (the LDS-staged form of srf_conv1x1_nhwc_k; the bf16 MFMA, unlike the f32 one, leaves the vector ALU to its wave:
// "an MFMA holds the SIMD's vector issue for 8 of its 32 cycles", MI355X_MICROARCH.md):
//   * workgroup tile 128 pixels x 128 channels, 4 waves = 2 x 2 wave tiles of 64 x 64 (2 x 2 accumulator tiles), K in chunks of
//     32 channels (one 128-byte line per pixel row), three workgroups per CU (48 KB of LDS each);
//   * A: thread (row = t / 8 + 32 j, quad = t % 8) loads 4 floats, splits them (11 vector instructions per pair: 3 v_cvt_pk_bf16_f32,
//     4 shifts / masks, 4 subtractions) and stores 8 bytes into each of the three plane images A[plane][row][slot 4][8 bf16],
//     slot = oct ^ ((row >> 2) & 3): conflict-free for the loader's ds_write_b64 and for the fragment's ds_read_b128;
//   * B: pre-split and packed once per layer in exactly that image order ([chunk][column tile][plane][col][slot][8]): 24 KB per
//     (chunk, tile), copied linearly;
//   * per chunk a wave issues 2 k-steps x 4 tiles x 6 products = 48 MFMAs (1536 cycles) from 24 ds_read_b128; loads run two
//     chunks ahead in registers; one LDS stage, two barriers per chunk (the other two workgroups of the CU fill them).
// First version measured (same shapes): 189-199 TFLOP/s f32-equivalent (1.13-1.20 PFLOP/s of bf16 issued) on the three large
// layers against 122-135 for the f32 kernels.
#include "common.hpp"

typedef __bf16 gs_bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 gs_bf8 __attribute__((ext_vector_type(8)));
typedef float gs_f2 __attribute__((ext_vector_type(2)));
typedef float gs_f4 __attribute__((ext_vector_type(4)));
typedef float gs_f16 __attribute__((ext_vector_type(16)));
typedef unsigned gs_u2 __attribute__((ext_vector_type(2)));
typedef unsigned gs_u4 __attribute__((ext_vector_type(4)));

#define GS_IMG 24576   // bytes of one operand image: 3 planes x 128 rows x 64 B

struct GsArgs {
    const float *x;
    float *y;
    const unsigned char *Wp;
    const float *scale, *shift;
    long long x_ld, y_ld, M;
    int K, Cout, nchunk, nct, relu;
    long long mblocks;
    // per-image row tiling + column sums of the stored outputs (eSE pooling): bpi > 0: image n owns row blocks [n bpi, (n + 1) bpi)
    float *colsum;
    long long HW;
    int bpi;
    // FPN top-down step in the epilogue (as srf_conv1x1_nhwc_topdown): y += top[n][floor(py sy)][floor(px sx)][co]
    const float *top;
    long long top_ld;
    int mapH, mapW, topH, topW;
    float sy, sx;
    // GS_CONV: rows are OUTPUT pixels (n, oy, ox) of a Conv2d(Cin, Cout, (kh, kw), stride, pad), k runs over (tap, input channel): implicit
    // im2col as in srf_conv_gemm_nhwc (conv.hip); a chunk of 32 channels lies inside one tap (Cin % 32 == 0)
    int H, W, Ho, Wo, kw, stride, pad, cin_chunks;
    long long x_bytes;
};
#define GS_PLAIN 0
#define GS_POOL 1
#define GS_TOPDOWN 2
#define GS_CONV 3

__device__ __forceinline__ unsigned gs_pk_bf16(float a, float b)
{
    const gs_f2 v = {a, b};
    const gs_bf2 h = __builtin_convertvector(v, gs_bf2);   // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return *reinterpret_cast<const unsigned *>(&h);
}

// (x0, x1) -> the packed pairs (h0, h1), (m0, m1), (l0, l1) with x = h + m + l exactly
__device__ __forceinline__ void gs_split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l)
{
    h = gs_pk_bf16(x0, x1);
    const float r0 = __fsub_rn(x0, __uint_as_float(h << 16)), r1 = __fsub_rn(x1, __uint_as_float(h & 0xffff0000u));
    m = gs_pk_bf16(r0, r1);
    const float s0 = __fsub_rn(r0, __uint_as_float(m << 16)), s1 = __fsub_rn(r1, __uint_as_float(m & 0xffff0000u));
    l = gs_pk_bf16(s0, s1);
}

// W (Cout, K) row-major -> [chunk of 32][column tile of 128][plane 3][col 128][slot 4][8 bf16]; slot s of column n holds the
// channels 8 (s ^ ((n >> 2) & 3)) .. + 7 of the chunk; columns >= Cout are zero
__global__ __launch_bounds__(256) void srf_gemm_split_pack_k(const float *__restrict__ Wt, int Cout, int K, int nct, unsigned short *__restrict__ P,
                                                            long long total)
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
    const float x = co < Cout ? Wt[(size_t)co * K + k] : 0.f;
    unsigned h, m, l;
    gs_split2(x, 0.f, h, m, l);
    P[t] = (unsigned short)((p == 0 ? h : p == 1 ? m : l) & 0xffffu);
}

template <int MODE>
__global__ __launch_bounds__(256, MODE == GS_CONV ? 2 : 3) void srf_gemm_split_k(GsArgs a)
{
    constexpr bool POOL = MODE == GS_POOL, TOPDOWN = MODE == GS_TOPDOWN, CONV = MODE == GS_CONV;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * GS_IMG];   // A planes | B planes
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // work item -> (column tile, row block): items b and b + 8 share an XCD, the column tiles of a row block sit on one L2
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int ct = jq % a.nct;
    const long long mb = (long long)(jq / a.nct) * 8 + xcd;
    if (mb >= a.mblocks) return;
    long long p0 = mb * 128, rows_blk = a.M - p0;
    long long slot = mb;
    if (POOL) {
        const long long n = mb / a.bpi, lb = mb - n * a.bpi;
        p0 = n * a.HW + lb * 128;
        rows_blk = a.HW - lb * 128;
        slot = n * a.bpi + lb;
    }
    const long long rows_here = rows_blk < 128 ? rows_blk : 128;
    __amdgpu_buffer_rsrc_t xr = CONV ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.x_bytes, 0x00020000)
                                     : __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x) + p0 * a.x_ld, 0, (int)(rows_here * a.x_ld * 4), 0x00020000);
    const int nchunk = a.nchunk;
    const size_t chunk_stride = (size_t)a.nct * GS_IMG;
    const unsigned char *bsrc = a.Wp + (size_t)ct * GS_IMG + (size_t)tid * 16;

    const int q = tid & 7, r0 = tid >> 3;
    const unsigned aoff0 = (unsigned)((r0 * a.x_ld + q * 4) * 4), aoff_step = (unsigned)(32 * a.x_ld * 4);   // rows past the block read as zero
    gs_f4 araw[4];
    gs_u4 braw[6];
    unsigned sp[3][4][2];
    // CONV: the output pixel of each of this thread's rows, as the input coordinates of tap (0, 0)
    int cy[CONV ? 4 : 1], cx[CONV ? 4 : 1], cn[CONV ? 4 : 1];
    if (CONV) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long long p = p0 + r0 + 32 * j;
            const long long hw = (long long)a.Ho * a.Wo;
            const int n = (int)(p / hw);
            const int rem = (int)(p - n * hw);
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            cn[j] = p < a.M ? n : -1;
            cy[j] = oy * a.stride - a.pad;
            cx[j] = ox * a.stride - a.pad;
        }
    }
#define GS_LOAD(C)                                                                                                         \
    do {                                                                                                                   \
        if (CONV) {                                                                                                        \
            const int tap_ = (C) / a.cin_chunks, cc_ = (C) - tap_ * a.cin_chunks;                                          \
            const int ky_ = tap_ / a.kw, kx_ = tap_ - ky_ * a.kw;                                                          \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                             \
                const int iy_ = cy[j_] + ky_, ix_ = cx[j_] + kx_;                                                          \
                const bool ok_ = cn[j_] >= 0 && iy_ >= 0 && iy_ < a.H && ix_ >= 0 && ix_ < a.W;                            \
                const unsigned off_ = ok_ ? (unsigned)(((((long long)cn[j_] * a.H + iy_) * a.W + ix_) * a.x_ld + cc_ * 32 + q * 4) * 4) \
                                          : 0x80000000u;                                                                   \
                auto v_ = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)off_, 0, 0);                                      \
                araw[j_] = *reinterpret_cast<gs_f4 *>(&v_);                                                                \
            }                                                                                                              \
        } else {                                                                                                           \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                             \
                auto v_ = __builtin_amdgcn_raw_buffer_load_b128(xr, (int)(aoff0 + j_ * aoff_step), (C) * 128, 0);          \
                araw[j_] = *reinterpret_cast<gs_f4 *>(&v_);                                                                \
            }                                                                                                              \
        }                                                                                                                  \
        const gs_u4 *bb_ = reinterpret_cast<const gs_u4 *>(bsrc + (size_t)(C) * chunk_stride);                             \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) braw[i_] = bb_[i_ * 256];                                         \
    } while (0)
#define GS_SPLIT()                                                                                                         \
    do {                                                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                 \
            gs_split2(araw[j_][0], araw[j_][1], sp[0][j_][0], sp[1][j_][0], sp[2][j_][0]);                                 \
            gs_split2(araw[j_][2], araw[j_][3], sp[0][j_][1], sp[1][j_][1], sp[2][j_][1]);                                 \
        }                                                                                                                  \
    } while (0)
#define GS_STORE()                                                                                                         \
    do {                                                                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                 \
            const int row_ = r0 + 32 * j_;                                                                                 \
            const int off_ = row_ * 64 + (((q >> 1) ^ ((row_ >> 2) & 3)) << 4) + (q & 1) * 8;                              \
            _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) {                                                             \
                const gs_u2 w_ = {sp[p_][j_][0], sp[p_][j_][1]};                                                           \
                *reinterpret_cast<gs_u2 *>(lds + p_ * 8192 + off_) = w_;                                                   \
            }                                                                                                              \
        }                                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) *reinterpret_cast<gs_u4 *>(lds + GS_IMG + (tid + i_ * 256) * 16) = braw[i_]; \
    } while (0)

    const int wm = wave & 1, wn = wave >> 1;
    const int li = lane & 31, lh = lane >> 5;
    int a_off[2], b_off[2], swz_a[2], swz_b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wm * 64 + i * 32 + li, col = wn * 64 + i * 32 + li;
        a_off[i] = row * 64;
        b_off[i] = GS_IMG + col * 64;
        swz_a[i] = (row >> 2) & 3;
        swz_b[i] = (col >> 2) & 3;
    }
    gs_f16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    gs_bf8 fa[3][2], fb[3][2];
#define GS_READ(S)                                                                                                         \
    do {                                                                                                                   \
        _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_)                                                                   \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                             \
                fa[p_][i_] = *reinterpret_cast<const gs_bf8 *>(lds + p_ * 8192 + a_off[i_] + (((2 * (S) + lh) ^ swz_a[i_]) << 4)); \
                fb[p_][i_] = *reinterpret_cast<const gs_bf8 *>(lds + p_ * 8192 + b_off[i_] + (((2 * (S) + lh) ^ swz_b[i_]) << 4)); \
            }                                                                                                              \
    } while (0)
#define GS_MM(PA, PB)                                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                                       \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                   \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][i_], fb[PB][j_], acc[i_][j_], 0, 0, 0)
    // the six products of a k-step, smallest terms first: l h, h l, m m (2^-16), m h, h m (2^-8), h h
#define GS_MFMA6() do { GS_MM(2, 0); GS_MM(0, 2); GS_MM(1, 1); GS_MM(1, 0); GS_MM(0, 1); GS_MM(0, 0); } while (0)

    const int last = nchunk - 1;
    GS_LOAD(0);
    GS_SPLIT();
    GS_STORE();
    GS_LOAD(last < 1 ? last : 1);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int c2 = c + 2 < nchunk ? c + 2 : last;
        GS_READ(0);
        GS_MFMA6();
        GS_SPLIT();      // the next chunk's A: vector work beside this chunk's MFMAs
        GS_READ(1);
        GS_MFMA6();
        __syncthreads();
        GS_STORE();
        GS_LOAD(c2);
        __syncthreads();
    }
#undef GS_LOAD
#undef GS_SPLIT
#undef GS_STORE
#undef GS_READ
#undef GS_MM
#undef GS_MFMA6

    // epilogue (that of srf_gemm_direct_k): lane = channel li of block j, accumulator register = pixel row (r & 3) + 8 (r >> 2) + 4 lh of block i
    float sc[2], sh[2];
    bool co_ok[2];
    const int co0 = ct * 128 + wn * 64 + li;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = co0 + j * 32;
        co_ok[j] = co < a.Cout;
        sc[j] = (co_ok[j] && a.scale) ? a.scale[co] : 1.f;
        sh[j] = (co_ok[j] && a.shift) ? a.shift[co] : 0.f;
    }
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(a.y + p0 * a.y_ld, 0, (int)(rows_here * a.y_ld * 4), 0x00020000);
    const int row_base = wm * 64 + 4 * lh;
    unsigned ybase[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) ybase[j] = co_ok[j] ? (unsigned)((row_base * a.y_ld + co0 + j * 32) * 4) : 0x80000000u;
    const unsigned yrow_b = (unsigned)(a.y_ld * 4);
    const long long rows_left = rows_blk - row_base;
    int *s_top = reinterpret_cast<int *>(lds);   // [128]: offset (floats) of the top-level pixel each row of this block adds
    if (TOPDOWN) {
        // (the loop's last barrier is behind every fragment read of this workgroup)
        if (tid < 128) {
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
    float csum[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v = __fmaf_rn(acc[i][j][r], sc[j], sh[j]);
                if (a.relu) v = fmaxf(v, 0.f);
                if (TOPDOWN && co_ok[j]) v = __fadd_rn(v, a.top[s_top[row_base + dr] + co0 + j * 32]);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, (int)(ybase[j] + (unsigned)dr * yrow_b), 0, 0);
                if (POOL) csum[j] += dr < rows_left ? v : 0.f;
            }
        }
    if (POOL) {
        // column sums of the block: the two lane halves of a wave (shuffle), then the two waves that share the columns (LDS), in a
        // fixed order: reproducible bit for bit
        float *red = reinterpret_cast<float *>(lds);   // [2][128]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float o = __shfl_xor(csum[j], 32);
            if (lh == 0) red[wm * 128 + wn * 64 + j * 32 + li] = csum[j] + o;
        }
        __syncthreads();
        if (tid < 128) {
            const int co = ct * 128 + tid;
            if (co < a.Cout) a.colsum[slot * a.Cout + co] = red[tid] + red[128 + tid];
        }
    }
}

extern "C" size_t srf_conv1x1_nhwc_split_packed_weight_bytes(int Cout, int K)
{
    if (Cout <= 0 || K <= 0 || (K & 31)) return 0;
    return (size_t)(K / 32) * srf_ceil_div(Cout, 128) * GS_IMG;
}

extern "C" int srf_conv1x1_nhwc_split_pack_weights(const float *W, int Cout, int K, void *packed, srf_stream_t stream)
{
    if (Cout <= 0 || K <= 0 || !W || !packed) return SRF_EINVAL;
    if (K & 31) return SRF_EUNSUPPORTED;
    const int nct = srf_ceil_div(Cout, 128);
    const long long total = (long long)(srf_conv1x1_nhwc_split_packed_weight_bytes(Cout, K) / 2);
    hipLaunchKernelGGL(srf_gemm_split_pack_k, dim3((unsigned)srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, W, Cout, K, nct,
                       (unsigned short *)packed, total);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

struct GsTop {
    const float *top;
    long long top_ld;
    int mapH, mapW, topH, topW;
};

static int gs_launch(const float *x, long long M, int K, long long x_ld, const void *W_packed, int Cout, const float *scale, const float *shift,
                     int relu, float *y, long long y_ld, float *colsum, long long HW, hipStream_t stream, int *bpi_out, const GsTop *td = nullptr)
{
    GsArgs a;
    a.x = x;
    a.y = y;
    a.Wp = (const unsigned char *)W_packed;
    a.scale = scale;
    a.shift = shift;
    a.x_ld = x_ld;
    a.y_ld = y_ld;
    a.M = M;
    a.K = K;
    a.Cout = Cout;
    a.nchunk = K / 32;
    a.nct = srf_ceil_div(Cout, 128);
    a.relu = relu;
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
    a.H = a.W = a.Ho = a.Wo = a.kw = a.stride = a.pad = a.cin_chunks = 0;
    a.x_bytes = 0;
    if (colsum) {
        a.bpi = (int)srf_ceil_div(HW, 128);
        a.mblocks = (M / HW) * a.bpi;
        if (bpi_out) *bpi_out = a.bpi;
    } else {
        a.bpi = 0;
        a.mblocks = srf_ceil_div(M, 128);
    }
    const long long blocks = ((a.mblocks + 7) / 8) * 8 * a.nct;
    if (blocks >= (1ll << 31)) return SRF_EUNSUPPORTED;
    if (colsum)
        hipLaunchKernelGGL((srf_gemm_split_k<GS_POOL>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else if (td)
        hipLaunchKernelGGL((srf_gemm_split_k<GS_TOPDOWN>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((srf_gemm_split_k<GS_PLAIN>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// srf_conv_gemm_nhwc_split: Conv2d(Cin, Cout, (kh, kw), stride, padding) on channels-last activations as the split GEMM with an
// implicit im2col (the stride-2 3x3 layers: VoVNet stem_3, SECONDCustom's second block, the BEV FPN extras -- the layers
// srf_conv_gemm_nhwc runs on the f32 MFMA).  W_packed = srf_conv1x1_nhwc_split_pack_weights of the weight reordered to
// (Cout, kh * kw * Cin), tap index slowest.
extern "C" int srf_conv_gemm_nhwc_split(const float *x, int N, int H, int W, int Cin, long long x_ld, const void *W_packed, int Cout, int kh,
                                        int kw, int stride, int pad, const float *scale, const float *shift, int relu, float *y,
                                        long long y_ld, srf_stream_t stream)
{
    if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 || x_ld < Cin || y_ld < Cout)
        return SRF_EINVAL;
    if (N == 0) return SRF_OK;
    if (!x || !W_packed || !y) return SRF_EINVAL;
    if ((Cin & 31) || (x_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)W_packed & 15)) return SRF_EUNSUPPORTED;
    const int Ho = (H + 2 * pad - kh) / stride + 1, Wo = (W + 2 * pad - kw) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return SRF_EINVAL;
    const long long x_bytes = (long long)N * H * W * x_ld * 4;
    if (x_bytes >= (1ll << 31) || y_ld * 128 * 4 >= (1ll << 31)) return SRF_EUNSUPPORTED;
    GsArgs a;
    a.x = x;
    a.y = y;
    a.Wp = (const unsigned char *)W_packed;
    a.scale = scale;
    a.shift = shift;
    a.x_ld = x_ld;
    a.y_ld = y_ld;
    a.M = (long long)N * Ho * Wo;
    a.K = kh * kw * Cin;
    a.Cout = Cout;
    a.nchunk = a.K / 32;
    a.nct = srf_ceil_div(Cout, 128);
    a.relu = relu;
    a.colsum = nullptr;
    a.HW = 0;
    a.bpi = 0;
    a.top = nullptr;
    a.top_ld = 0;
    a.mapH = a.mapW = a.topH = a.topW = 0;
    a.sy = a.sx = 0.f;
    a.H = H;
    a.W = W;
    a.Ho = Ho;
    a.Wo = Wo;
    a.kw = kw;
    a.stride = stride;
    a.pad = pad;
    a.cin_chunks = Cin / 32;
    a.x_bytes = x_bytes;
    a.mblocks = srf_ceil_div(a.M, 128);
    const long long blocks = ((a.mblocks + 7) / 8) * 8 * a.nct;
    if (blocks >= (1ll << 31)) return SRF_EUNSUPPORTED;
    hipLaunchKernelGGL((srf_gemm_split_k<GS_CONV>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

static int gs_check(const float *x, int K, long long x_ld, const void *W_packed, int Cout, long long y_ld)
{
    if (K <= 0 || Cout <= 0 || x_ld < K || y_ld < Cout) return SRF_EINVAL;
    if ((K & 31) || (x_ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)W_packed & 15)) return SRF_EUNSUPPORTED;
    if (x_ld * 128 * 4 >= (1ll << 31) || y_ld * 128 * 4 >= (1ll << 31)) return SRF_EUNSUPPORTED;
    return SRF_OK;
}

extern "C" int srf_conv1x1_nhwc_split(const float *x, long long M, int K, long long x_ld, const void *W_packed, int Cout, const float *scale,
                                      const float *shift, int relu, float *y, long long y_ld, srf_stream_t stream)
{
    if (M < 0) return SRF_EINVAL;
    const int rc = gs_check(x, K, x_ld, W_packed, Cout, y_ld);
    if (rc != SRF_OK) return rc;
    if (M == 0) return SRF_OK;
    if (!x || !W_packed || !y) return SRF_EINVAL;
    return gs_launch(x, M, K, x_ld, W_packed, Cout, scale, shift, relu, y, y_ld, nullptr, 0, (hipStream_t)stream, nullptr);
}

extern "C" int srf_conv1x1_nhwc_split_topdown(const float *x, int N, int H, int W, int K, long long x_ld, const void *W_packed, int Cout,
                                              const float *scale, const float *shift, int relu, const float *top, int Ht, int Wt,
                                              long long top_ld, float *y, long long y_ld, srf_stream_t stream)
{
    if (N < 0 || H <= 0 || W <= 0 || Ht <= 0 || Wt <= 0 || top_ld < Cout) return SRF_EINVAL;
    const int rc = gs_check(x, K, x_ld, W_packed, Cout, y_ld);
    if (rc != SRF_OK) return rc;
    if (N == 0) return SRF_OK;
    if (!x || !W_packed || !y || !top) return SRF_EINVAL;
    if ((long long)N * Ht * Wt * top_ld >= (1ll << 31)) return SRF_EUNSUPPORTED;
    const GsTop td = {top, top_ld, H, W, Ht, Wt};
    return gs_launch(x, (long long)N * H * W, K, x_ld, W_packed, Cout, scale, shift, relu, y, y_ld, nullptr, 0, (hipStream_t)stream, nullptr, &td);
}

// the pooled form: as srf_conv1x1_nhwc_pooled (conv.hip); workspace = srf_conv1x1_nhwc_pooled_workspace_bytes(N, HW, Cout)
__global__ __launch_bounds__(256) void srf_gemm_split_pool_finish_k(const float *__restrict__ partial, int bpi, int C, float inv, float *__restrict__ mean)
{
    __shared__ float s[16][16];
    const int n = blockIdx.y, cl = threadIdx.x & 15, c = blockIdx.x * 16 + cl, g = threadIdx.x >> 4;
    float acc = 0.f;
    if (c < C)
        for (int b = g; b < bpi; b += 16) acc += partial[((long long)n * bpi + b) * C + c];
    s[g][cl] = acc;
    __syncthreads();
    if (g == 0 && c < C) {
        float t = s[0][cl];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += s[k][cl];
        mean[(long long)n * C + c] = t * inv;
    }
}

extern "C" int srf_conv1x1_nhwc_split_pooled(const float *x, int N, long long HW, int K, long long x_ld, const void *W_packed, int Cout,
                                             const float *scale, const float *shift, int relu, float *y, long long y_ld, float *mean,
                                             void *workspace, size_t workspace_bytes, srf_stream_t stream)
{
    if (N < 0 || HW <= 0) return SRF_EINVAL;
    const int rc = gs_check(x, K, x_ld, W_packed, Cout, y_ld);
    if (rc != SRF_OK) return rc;
    if (N == 0) return SRF_OK;
    if (!x || !W_packed || !y || !mean || !workspace) return SRF_EINVAL;
    if (N > 65535) return SRF_EUNSUPPORTED;
    if (workspace_bytes < (size_t)N * (size_t)srf_ceil_div(HW, 128) * Cout * 4) return SRF_EWORKSPACE;
    int bpi = 0;
    const int r2 = gs_launch(x, (long long)N * HW, K, x_ld, W_packed, Cout, scale, shift, relu, y, y_ld, (float *)workspace, HW,
                             (hipStream_t)stream, &bpi);
    if (r2 != SRF_OK) return r2;
    hipLaunchKernelGGL(srf_gemm_split_pool_finish_k, dim3(srf_ceil_div(Cout, 16), N), dim3(256), 0, (hipStream_t)stream,
                       (const float *)workspace, bpi, Cout, 1.0f / (float)HW, mean);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
