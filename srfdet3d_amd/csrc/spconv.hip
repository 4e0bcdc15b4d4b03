// spconv.hip -- K5 sparse convolution forward on the f32 MFMA pipe, K6 densify (gfx950).
//
// Reference call sites: every SubMConv3d / SparseConv3d (+ BN1d + ReLU, + residual in SparseBasicBlock) of
// SparseEncoderCustom.forward, mmdet3d_plugin/models/middle_encoders/sparse_encoder_custom.py:125-134, and
// SparseConvTensor.dense() at :135-138.
//
// Output-stationary implicit GEMM.  A workgroup owns TM consecutive output rows and all COUT columns; for each
// kernel offset k (skipped when no row of the tile has a neighbour at k) it gathers the TM input rows named by
// nbr[k][.] into LDS in 32-channel chunks next to the matching 32 x COUT slab of W[k], and accumulates with
// v_mfma_f32_32x32x2_f32 (v_mfma_f32_16x16x4_f32 for COUT = 16).  Missing neighbours contribute exact zeros.
// The f32 MFMA is a k-ordered fma chain, so every output element is the chain over (k ascending, c ascending)
// that oracle/srf_oracle.c:orc_spconv_fwd forms -- results compare exactly.  Each output row is written once,
// with eval-BatchNorm (y = fma(x, alpha, beta)), residual add and ReLU applied in registers.
//
// LDS per workgroup (COUT = 128): nbr tile 27*64*4 = 6.9 KB, A chunk 64*33*4 = 8.4 KB, W chunk 32*128*4 = 16 KB
// -> 5 workgroups per CU; latency is hidden by occupancy rather than by an explicit pipeline.
#include "common.hpp"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SRF_KC 32
#define SRF_KMAX 27

template <int TM, int COUT>
__device__ __forceinline__ void srf_stage_tile(const float *__restrict__ in, int Cin, const float *__restrict__ Wk, int c0,
                                               const int *s_nbr_k, float (*s_a)[SRF_KC + 1], float (*s_w)[COUT])
{
    const int tid = threadIdx.x;
    const bool vec = (Cin & 3) == 0;
    // gathered input rows: 8 threads per row, one float4 each
    for (int e = tid; e < TM * (SRF_KC / 4); e += 256) {
        const int r = e / (SRF_KC / 4), q = e % (SRF_KC / 4);
        const int i = s_nbr_k[r];
        const int c = c0 + q * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i >= 0 && c < Cin) {
            const float *src = in + (size_t)i * Cin + c;
            if (vec) {
                v = *reinterpret_cast<const float4 *>(src);
            } else {
                v.x = src[0];
                if (c + 1 < Cin) v.y = src[1];
                if (c + 2 < Cin) v.z = src[2];
                if (c + 3 < Cin) v.w = src[3];
            }
        }
        float *dst = &s_a[r][q * 4];
        dst[0] = v.x;
        dst[1] = v.y;
        dst[2] = v.z;
        dst[3] = v.w;
    }
    // weight slab: rows c0..c0+31 of W[k] (Cin x COUT), zero beyond Cin
    for (int e = tid; e < SRF_KC * (COUT / 4); e += 256) {
        const int c = e / (COUT / 4), j = e % (COUT / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + c < Cin) v = *reinterpret_cast<const float4 *>(Wk + (size_t)(c0 + c) * COUT + j * 4);
        *reinterpret_cast<float4 *>(&s_w[c][j * 4]) = v;
    }
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Give every XCD one contiguous
// range of output tiles, so that the input rows its tiles gather (rows that are close in index are close in space)
// stay in that XCD's 4 MB L2 instead of being re-fetched through the fabric.  Bijective for any grid size.
__device__ __forceinline__ int srf_xcd_tile(int b, int n)
{
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

template <int TM>
__device__ __forceinline__ void srf_load_nbr_tile(const int *__restrict__ nbr, int nbr_stride, int K, int row0, int A_out,
                                                  int *s_nbr, int *s_any)
{
    if (threadIdx.x < SRF_KMAX) s_any[threadIdx.x] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < K * TM; t += 256) {
        const int k = t / TM, r = t % TM;
        const int row = row0 + r;
        const int v = row < A_out ? nbr[(size_t)k * nbr_stride + row] : -1;
        s_nbr[t] = v;
        if (v >= 0) s_any[k] = 1;  // benign race: every writer stores 1
    }
    __syncthreads();
}

// ---- pipelined wide kernel (COUT in {32, 64, 128}) ------------------------------------------------------------------
// Branch-free loads (Cin % 4 == 0 is required by the launcher): a missing neighbour or a channel beyond Cin reads a
// valid clamped address and is zeroed by a select, so all loads of a step stay in flight together.  The staging
// registers are native vectors (f32x4), not HIP's float4 struct: arrays of the union-based struct are not split into
// registers by the compiler and end up round-tripping through LDS.
template <int COUT, int TM, int NA, int NW>
__device__ __forceinline__ void srf_step_load(const float *__restrict__ in, int Cin, const float *__restrict__ W,
                                              const int *nbr_k /* this offset's TM entries: LDS tile or nbr + k*stride + row0 */, int rows_left,
                                              int k, int c0, f32x4 (&ra)[NA], f32x4 (&rw)[NW], unsigned &okmask)
{
    const int tid = threadIdx.x;
    const float *Wk = W + (size_t)k * Cin * COUT;
    // the zeroing select is applied at store time (srf_step_store): touching the loaded value here would make the
    // compiler wait for the load before the MFMA block it is meant to overlap
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int e = tid + j * 256;
        const int r = e / (SRF_KC / 4), q = e % (SRF_KC / 4);
        const int i = r < rows_left ? nbr_k[r] : -1;  // L2-resident; 8 lanes share one entry
        const int c = c0 + q * 4;
        const bool ok = (i >= 0) & (c < Cin);
        const int ii = i >= 0 ? i : 0;
        const int cc = c < Cin ? c : Cin - 4;
        ra[j] = *reinterpret_cast<const f32x4 *>(in + (size_t)ii * Cin + cc);
        m |= (ok ? 1u : 0u) << j;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int e = tid + j * 256;
        const int c = e / (COUT / 4), q = e % (COUT / 4);
        const bool ok = c0 + c < Cin;
        const int cc = ok ? c0 + c : Cin - 1;
        rw[j] = *reinterpret_cast<const f32x4 *>(Wk + (size_t)cc * COUT + q * 4);
        m |= (ok ? 1u : 0u) << (NA + j);
    }
    okmask = m;
}

template <int COUT, int TM, int NA, int NW>
__device__ __forceinline__ void srf_step_store(float (*s_a)[SRF_KC + 1], float (*s_w)[COUT], const f32x4 (&ra)[NA],
                                               const f32x4 (&rw)[NW], unsigned okmask)
{
    const int tid = threadIdx.x;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int e = tid + j * 256;
        float *dst = &s_a[e / (SRF_KC / 4)][(e % (SRF_KC / 4)) * 4];
        const f32x4 v = ((okmask >> j) & 1u) ? ra[j] : zero;
        dst[0] = v[0];
        dst[1] = v[1];
        dst[2] = v[2];
        dst[3] = v[3];
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int e = tid + j * 256;
        *reinterpret_cast<f32x4 *>(&s_w[e / (COUT / 4)][(e % (COUT / 4)) * 4]) = ((okmask >> (NA + j)) & 1u) ? rw[j] : zero;
    }
}

// Waves arranged WR x WC, each owning 32 x (CT*32) outputs.  Software pipeline: the global loads of step t+1 (gathered
// rows + W slab, held in registers) are in flight while the MFMAs of step t run from LDS buffer t&1; they are written
// to buffer (t+1)&1 afterwards; one barrier per step.  Steps enumerate (active kernel offset, 32-channel chunk)
// pairs; offsets that no row of the tile uses are not visited.  A chunk is always 32 deep (zero padded), so the MFMA
// loop is fully unrolled and its LDS reads are hoisted ahead of the MFMAs by the compiler.
template <int COUT, int TM, int WR, int WC>
__global__ __launch_bounds__(256) void srf_spconv_mfma32_k(const float *__restrict__ in, int Cin,
                                                         const float *__restrict__ W, int K,
                                                         const int *__restrict__ nbr, int nbr_stride, int A_out,
                                                         const float *__restrict__ alpha, const float *__restrict__ beta,
                                                         const float *__restrict__ residual, int relu,
                                                         float *__restrict__ out, const int *__restrict__ rows_dev)
{
    static_assert(WR * WC == 4 && TM == WR * 32, "one 32-row tile per wave row");
    constexpr int CT = COUT / WC / 32;
    constexpr int NA = TM * (SRF_KC / 4) / 256;    // float4 gathers per thread and step
    constexpr int NW = SRF_KC * (COUT / 4) / 256;  // float4 weight loads per thread and step
    // COUT = 128 leaves the 27 x TM neighbour tile in L2 (it would cost the third resident workgroup per CU);
    // the narrower kernels have LDS to spare and keep it on chip
    constexpr bool NBR_LDS = COUT < 128;
    __shared__ int s_nbr[NBR_LDS ? SRF_KMAX * TM : 1];
    __shared__ int s_any[SRF_KMAX];
    __shared__ int s_klist[SRF_KMAX + 1];
    __shared__ float s_a[2][TM][SRF_KC + 1];
    __shared__ __attribute__((aligned(16))) float s_w[2][SRF_KC][COUT];

    if (rows_dev) {  // static-shape levels: rows >= *rows_dev are padding; their tiles do nothing
        const int live = *rows_dev;
        A_out = A_out < live ? A_out : live;
    }
    // tiles of LIVE rows only, dealt XCD-contiguously over the first n_tiles workgroups (a capacity-sized launch must not
    // leave whole XCDs with nothing but padding)
    const int n_tiles = (A_out + TM - 1) / TM;
    if ((int)blockIdx.x >= n_tiles) return;
    const int row0 = srf_xcd_tile(blockIdx.x, n_tiles) * TM;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int rows_left = A_out - row0;
    if (tid < SRF_KMAX) s_any[tid] = 0;
    __syncthreads();
    for (int t = tid; t < K * TM; t += 256) {
        const int k = t / TM, r = t % TM;
        const int v = r < rows_left ? nbr[(size_t)k * nbr_stride + row0 + r] : -1;
        if (NBR_LDS) s_nbr[t] = v;
        if (v >= 0) s_any[k] = 1;  // benign race: every writer stores 1
    }
    __syncthreads();
    if (tid < 64) {  // compact the used offsets with one ballot (K <= 27 < 64) instead of a serial loop on one lane
        const bool used = tid < K && s_any[tid];
        const unsigned long long m = __ballot(used);
        if (used) s_klist[__popcll(m & ((1ull << tid) - 1ull))] = tid;
        if (tid == 0) s_klist[SRF_KMAX] = __popcll(m);
    }
    __syncthreads();
    const int nchunk = (Cin + SRF_KC - 1) / SRF_KC;
    const int T = s_klist[SRF_KMAX] * nchunk;

    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[ct][j] = 0.0f;

    f32x4 ra[NA], rw[NW];
    unsigned okmask = 0;
    const int ar = wr * 32 + (lane & 31);
    const int kh = lane >> 5;
    if (T > 0) {
        const int k0 = s_klist[0];
        srf_step_load<COUT, TM, NA, NW>(in, Cin, W, NBR_LDS ? s_nbr + k0 * TM : nbr + (size_t)k0 * nbr_stride + row0,
                                        NBR_LDS ? TM : rows_left, k0, 0, ra, rw, okmask);
        srf_step_store<COUT, TM, NA, NW>(s_a[0], s_w[0], ra, rw, okmask);
    }
    __syncthreads();
    int tk = 0, tc = 0;  // (offset index, chunk index) of step t+1
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        if (++tc == nchunk) {
            tc = 0;
            ++tk;
        }
        const bool more = t + 1 < T;
        if (more) {
            const int kn = s_klist[tk];
            srf_step_load<COUT, TM, NA, NW>(in, Cin, W, NBR_LDS ? s_nbr + kn * TM : nbr + (size_t)kn * nbr_stride + row0,
                                            NBR_LDS ? TM : rows_left, kn, tc * SRF_KC, ra, rw, okmask);
        }
#pragma unroll
        for (int kk = 0; kk < SRF_KC; kk += 2) {
            const float a = s_a[buf][ar][kk + kh];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const float b = s_w[buf][kk + kh][(wc * CT + ct) * 32 + (lane & 31)];
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ct], 0, 0, 0);
            }
        }
        if (more) srf_step_store<COUT, TM, NA, NW>(s_a[buf ^ 1], s_w[buf ^ 1], ra, rw, okmask);
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32: col = lane & 31, row = (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = (wc * CT + ct) * 32 + (lane & 31);
        const float al = alpha ? alpha[col] : 1.0f;
        const float be = alpha ? beta[col] : 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = row0 + wr * 32 + (j & 3) + 8 * (j >> 2) + 4 * kh;
            if (row < A_out) {
                float v = acc[ct][j];
                if (alpha) v = __fmaf_rn(v, al, be);
                if (residual) v = __fadd_rn(v, residual[(size_t)row * COUT + col]);
                if (relu) v = v > 0.0f ? v : 0.0f;
                out[(size_t)row * COUT + col] = v;
            }
        }
    }
}

// =====================================================================================================================
// Packed-weight kernel (the fast path; weights are constants at inference, so they are re-laid-out once per layer).
//
// Operand images in LDS are built so that a lane fetches its 16 k-values of a 32-channel chunk with four ds_read_b128:
//   row of the image = one gathered input row (A) or one output column (B), 32 floats = 8 slots of 16 B;
//   logical slot s = 4*h + g holds channels c0 + 2*(4g + i) + h, i = 0..3  (h = parity of the channel: the f32 32x32x2
//   MFMA takes channel 2j from lanes 0-31 and 2j+1 from lanes 32-63);
//   physical slot = s ^ ((row >> 1) & 7): with 128-B rows this XOR swizzle puts the 16 lanes of every ds_read_b128
//   lane group on 16 different (half, slot) pairs, i.e. conflict-free without padding (48 KB per workgroup -> three
//   workgroups per CU).
// srf_spconv_pack_weights writes W in exactly that image, per (offset k, chunk), so the slab copy global -> LDS is
// linear.  The accumulation order per output element is unchanged (k ascending, channel ascending): results stay
// bit-identical to srf_spconv_fwd and to the oracle.
// =====================================================================================================================
// direct (LDS-free B operand) layout of the COUT = 128 kernel, see srf_spconv_direct_k below
static bool srf_direct64_enabled()
{
    static const bool on = [] {
        const char *e = getenv("SRF_SPCONV_PACKED64");  // developer switch: the LDS-staged kernel for 64 -> 64, for A/B timing
        return !(e && e[0] == '1');
    }();
    return on;
}
static bool srf_direct_layout(int Cin, int Cout)
{
    return (Cout == 128 && (Cin == 64 || Cin == 128)) || (Cout == 64 && Cin == 64 && srf_direct64_enabled());
}
// compacted-offset layout of the same shapes (srf_spconv_gs_k below), the default
static bool srf_gs_layout(int Cin, int Cout);
__global__ void srf_pack_weights_gs_k(const float *__restrict__ W, int K, int Cin, int Cout, int nchunk, float *__restrict__ P);

__global__ __launch_bounds__(256) void srf_pack_weights_direct_k(const float *__restrict__ W, int K, int Cin, int Cout,
                                                               int nchunk, float *__restrict__ P)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)K * nchunk * Cout * 32;
    if (t >= total) return;
    const int nwc = Cout / 32;  // 32-column slices: 4 (COUT = 128) or 2 (COUT = 64)
    const int i = (int)(t & 3), lane = (int)((t >> 2) & 63), g = (int)((t >> 8) & 3);
    long long rest = t >> 10;
    const int wc = (int)(rest % nwc);
    rest /= nwc;
    const int chunk = (int)(rest % nchunk), k = (int)(rest / nchunk);
    const int col = wc * 32 + (lane & 31);
    const int c = chunk * 32 + 2 * (4 * g + i) + (lane >> 5);
    P[t] = c < Cin ? W[((size_t)k * Cin + c) * Cout + col] : 0.0f;
}

__global__ __launch_bounds__(256) void srf_pack_weights_k(const float *__restrict__ W, int K, int Cin, int Cout, int nchunk,
                                                        float *__restrict__ P)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)K * nchunk * Cout * 32;
    if (t >= total) return;
    const int e = (int)(t & 31);
    long long rest = t >> 5;
    const int col = (int)(rest % Cout);
    rest /= Cout;
    const int chunk = (int)(rest % nchunk), k = (int)(rest / nchunk);
    const int phys = e >> 2, i = e & 3;
    const int sl = phys ^ ((col >> 1) & 7);
    const int c = chunk * 32 + 2 * ((sl & 3) * 4 + i) + (sl >> 2);
    P[t] = c < Cin ? W[((size_t)k * Cin + c) * Cout + col] : 0.0f;
}

// wave-private layout of the 32-output-channel layers (srf_spconv_w32_k below): [k][group of 4 steps][channel parity][col 32][4]
static bool srf_w32_layout(int K, int Cin, int Cout)
{
    static const bool off = [] {
        const char *e = getenv("SRF_SPCONV_W32");  // developer switch: 0 = the LDS-staged tile kernel, for A/B timing
        return e && e[0] == '0';
    }();
    return !off && K == SRF_KMAX && Cout == 32 && (Cin == 16 || Cin == 32);
}

__global__ __launch_bounds__(256) void srf_pack_weights_w32_k(const float *__restrict__ W, int K, int Cin, float *__restrict__ P)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int total = K * Cin * 32;
    if (t >= total) return;
    const int nb = Cin / 8;
    const int i = t & 3, col = (t >> 2) & 31, kh = (t >> 7) & 1;
    const int rest = t >> 8;
    const int sg = rest % nb, k = rest / nb;
    const int c = 2 * (4 * sg + i) + kh;   // MFMA step 4 sg + i multiplies the channels 2 s (lanes 0-31) and 2 s + 1 (lanes 32-63)
    P[t] = W[((size_t)k * Cin + c) * 32 + col];
}

extern "C" size_t srf_spconv_packed_weight_bytes(int K, int Cin, int Cout)
{
    if (K <= 0 || Cin <= 0 || Cout <= 0) return 0;
    return (size_t)K * ((Cin + 31) / 32) * Cout * 32 * sizeof(float);
}

extern "C" int srf_spconv_pack_weights(const float *W, int K, int Cin, int Cout, float *packed, srf_stream_t stream)
{
    if (!W || !packed || K <= 0 || K > SRF_KMAX || Cin <= 0 || Cout <= 0) return SRF_EINVAL;
    const int nchunk = (Cin + 31) / 32;
    const long long total = (long long)K * nchunk * Cout * 32;
    if (srf_w32_layout(K, Cin, Cout))
        hipLaunchKernelGGL(srf_pack_weights_w32_k, dim3(srf_ceil_div((long long)K * Cin * 32, 256)), dim3(256), 0, (hipStream_t)stream, W, K,
                           Cin, packed);
    else if (srf_gs_layout(Cin, Cout))
        hipLaunchKernelGGL(srf_pack_weights_gs_k, dim3(srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, W, K, Cin,
                           Cout, nchunk, packed);
    else if (srf_direct_layout(Cin, Cout))
        hipLaunchKernelGGL(srf_pack_weights_direct_k, dim3(srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, W, K,
                           Cin, Cout, nchunk, packed);
    else
        hipLaunchKernelGGL(srf_pack_weights_k, dim3(srf_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, W, K, Cin, Cout,
                           nchunk, packed);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

template <int COUT, int TM, int NA, int NW>
__device__ __forceinline__ void srf_pk_load(const float *__restrict__ in, int Cin, const float *__restrict__ slab,
                                            const int *nbr_k, int rows_left, int c0, f32x4 (&ra)[NA], f32x4 (&rw)[NW],
                                            unsigned &okmask)
{
    const int tid = threadIdx.x;
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int e = tid + j * 256;
        const int r = e >> 3, q = e & 7;
        const int i = r < rows_left ? nbr_k[r] : -1;
        const int c = c0 + q * 4;
        const bool ok = (i >= 0) & (c < Cin);
        const int ii = i >= 0 ? i : 0;
        const int cc = c < Cin ? c : Cin - 4;
        ra[j] = *reinterpret_cast<const f32x4 *>(in + (size_t)ii * Cin + cc);
        m |= (ok ? 1u : 0u) << j;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) rw[j] = *reinterpret_cast<const f32x4 *>(slab + (size_t)(tid + j * 256) * 4);
    okmask = m;
}

template <int COUT, int TM, int NA, int NW>
__device__ __forceinline__ void srf_pk_store(float *s_a, float *s_w, const f32x4 (&ra)[NA], const f32x4 (&rw)[NW],
                                             unsigned okmask)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int e = tid + j * 256;
        const int r = e >> 3, q = e & 7;
        const f32x4 v = ((okmask >> j) & 1u) ? ra[j] : zero;
        const int swz = (r >> 1) & 7;
        const int off = 2 * (q & 1);
        const f32x2 ev = {v[0], v[2]}, od = {v[1], v[3]};
        *reinterpret_cast<f32x2 *>(s_a + r * 32 + (((q >> 1)) ^ swz) * 4 + off) = ev;
        *reinterpret_cast<f32x2 *>(s_a + r * 32 + ((4 + (q >> 1)) ^ swz) * 4 + off) = od;
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) *reinterpret_cast<f32x4 *>(s_w + (size_t)(tid + j * 256) * 4) = rw[j];
}

template <int COUT, int TM, int WR, int WC>
__global__ __launch_bounds__(256) void srf_spconv_packed_k(const float *__restrict__ in, int Cin,
                                                         const float *__restrict__ Wp, int K,
                                                         const int *__restrict__ nbr, int nbr_stride, int A_out,
                                                         const float *__restrict__ alpha, const float *__restrict__ beta,
                                                         const float *__restrict__ residual, int relu,
                                                         float *__restrict__ out, const int *__restrict__ rows_dev)
{
    static_assert(WR * WC == 4 && TM == WR * 32, "one 32-row tile per wave row");
    constexpr int CT = COUT / WC / 32;
    constexpr int NA = TM * 8 / 256;
    constexpr int NW = COUT * 8 / 256;
    constexpr bool NBR_LDS = COUT < 128;
    __shared__ int s_nbr[NBR_LDS ? SRF_KMAX * TM : 1];
    __shared__ int s_any[SRF_KMAX];
    __shared__ int s_klist[SRF_KMAX + 1];
    __shared__ __attribute__((aligned(16))) float s_a[2][TM * 32];
    __shared__ __attribute__((aligned(16))) float s_w[2][COUT * 32];

    if (rows_dev) {  // static-shape levels: rows >= *rows_dev are padding; their tiles do nothing
        const int live = *rows_dev;
        A_out = A_out < live ? A_out : live;
    }
    // tiles of LIVE rows only, dealt XCD-contiguously over the first n_tiles workgroups (a capacity-sized launch must not
    // leave whole XCDs with nothing but padding)
    const int n_tiles = (A_out + TM - 1) / TM;
    if ((int)blockIdx.x >= n_tiles) return;
    const int row0 = srf_xcd_tile(blockIdx.x, n_tiles) * TM;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int rows_left = A_out - row0;
    if (tid < SRF_KMAX) s_any[tid] = 0;
    __syncthreads();
    for (int t = tid; t < K * TM; t += 256) {
        const int k = t / TM, r = t % TM;
        const int v = r < rows_left ? nbr[(size_t)k * nbr_stride + row0 + r] : -1;
        if (NBR_LDS) s_nbr[t] = v;
        if (v >= 0) s_any[k] = 1;  // benign race: every writer stores 1
    }
    __syncthreads();
    if (tid < 64) {  // compact the used offsets with one ballot (K <= 27 < 64) instead of a serial loop on one lane
        const bool used = tid < K && s_any[tid];
        const unsigned long long m = __ballot(used);
        if (used) s_klist[__popcll(m & ((1ull << tid) - 1ull))] = tid;
        if (tid == 0) s_klist[SRF_KMAX] = __popcll(m);
    }
    __syncthreads();
    const int nchunk = (Cin + SRF_KC - 1) / SRF_KC;
    const int T = s_klist[SRF_KMAX] * nchunk;
    const size_t slab_floats = (size_t)COUT * 32;

    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[ct][j] = 0.0f;

    f32x4 ra[NA], rw[NW];
    unsigned okmask = 0;
    const int kh = lane >> 5;
    const int arow = wr * 32 + (lane & 31);
    const int a_swz = (arow >> 1) & 7;
    if (T > 0) {
        const int k0 = s_klist[0];
        srf_pk_load<COUT, TM, NA, NW>(in, Cin, Wp + (size_t)k0 * nchunk * slab_floats,
                                      NBR_LDS ? s_nbr + k0 * TM : nbr + (size_t)k0 * nbr_stride + row0,
                                      NBR_LDS ? TM : rows_left, 0, ra, rw, okmask);
        srf_pk_store<COUT, TM, NA, NW>(s_a[0], s_w[0], ra, rw, okmask);
    }
    __syncthreads();
    int tk = 0, tc = 0;
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        if (++tc == nchunk) {
            tc = 0;
            ++tk;
        }
        const bool more = t + 1 < T;
        if (more) {
            const int kn = s_klist[tk];
            srf_pk_load<COUT, TM, NA, NW>(in, Cin, Wp + ((size_t)kn * nchunk + tc) * slab_floats,
                                          NBR_LDS ? s_nbr + kn * TM : nbr + (size_t)kn * nbr_stride + row0,
                                          NBR_LDS ? TM : rows_left, tc * SRF_KC, ra, rw, okmask);
        }
        // operand fragments of this step: 4 + 4*CT ds_read_b128, then the MFMAs run without further LDS waits
        f32x4 af[4], bf[CT][4];
        const float *pa = s_a[buf] + arow * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) af[g] = *reinterpret_cast<const f32x4 *>(pa + (((kh << 2) + g) ^ a_swz) * 4);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = (wc * CT + ct) * 32 + (lane & 31);
            const float *pb = s_w[buf] + col * 32;
            const int b_swz = (col >> 1) & 7;
#pragma unroll
            for (int g = 0; g < 4; ++g) bf[ct][g] = *reinterpret_cast<const f32x4 *>(pb + (((kh << 2) + g) ^ b_swz) * 4);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 16; ++j)
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j >> 2][j & 3], bf[ct][j >> 2][j & 3], acc[ct], 0, 0, 0);
        if (more) srf_pk_store<COUT, TM, NA, NW>(s_a[buf ^ 1], s_w[buf ^ 1], ra, rw, okmask);
        __syncthreads();
    }

#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = (wc * CT + ct) * 32 + (lane & 31);
        const float al = alpha ? alpha[col] : 1.0f;
        const float be = alpha ? beta[col] : 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = row0 + wr * 32 + (j & 3) + 8 * (j >> 2) + 4 * kh;
            if (row < A_out) {
                float v = acc[ct][j];
                if (alpha) v = __fmaf_rn(v, al, be);
                if (residual) v = __fadd_rn(v, residual[(size_t)row * COUT + col]);
                if (relu) v = v > 0.0f ? v : 0.0f;
                out[(size_t)row * COUT + col] = v;
            }
        }
    }
}

// =====================================================================================================================
// COUT = 128, Cin in {64, 128}: the B operand never touches LDS.  Weights are packed so that every lane finds the 16
// values it feeds to the 16 MFMAs of a 32-channel chunk as four consecutive float4 (one coalesced 1 KB load per wave
// and group), and are fetched global(L2) -> registers one chunk ahead by the wave that uses them: no slab copy, no LDS
// traffic for B and one barrier per kernel offset (64 MFMAs per wave and row tile) instead of one per chunk.  LDS holds
// only the gathered input rows of one offset (double buffered) and the neighbour tile.  Same accumulation order as every
// other kernel here (offset ascending, channel ascending): bit-identical results.
//   direct layout: Wd[k][chunk][wc = col/32][g][lane][i] = W[k][chunk*32 + 2*(4g+i) + (lane>>5)][wc*32 + (lane&31)]
// =====================================================================================================================
template <int NCH, int NWC = 4>
__device__ __forceinline__ void srf_dir_load_b(f32x4 (&b)[4], const float *__restrict__ Wd, int k, int chunk, int wc, int lane)
{
#pragma unroll
    for (int g = 0; g < 4; ++g)
        b[g] = *reinterpret_cast<const f32x4 *>(Wd + (((((size_t)k * NCH + chunk) * NWC + wc) * 4 + g) * 64 + lane) * 4);
}

template <int TM, int NCH, int NA>
__device__ __forceinline__ void srf_dir_gather(const float *__restrict__ in, const int *s_nbr_k, f32x4 (&ra)[NA], unsigned &okmask)
{
    const int tid = threadIdx.x;
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int e = tid + j * 256;
        const int r = e / (8 * NCH), q = e % (8 * NCH);
        const int i = s_nbr_k[r];
        ra[j] = *reinterpret_cast<const f32x4 *>(in + (size_t)(i >= 0 ? i : 0) * (32 * NCH) + q * 4);
        m |= (i >= 0 ? 1u : 0u) << j;
    }
    okmask = m;
}

template <int TM, int NCH, int NA>
__device__ __forceinline__ void srf_dir_store(float *s_a, const f32x4 (&ra)[NA], unsigned okmask)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int e = tid + j * 256;
        const int r = e / (8 * NCH), qq = e % (8 * NCH);
        const int ch = qq >> 3, q = qq & 7;
        const f32x4 v = ((okmask >> j) & 1u) ? ra[j] : zero;
        const int swz = (r >> 1) & 7;
        const int off = 2 * (q & 1);
        const f32x2 ev = {v[0], v[2]}, od = {v[1], v[3]};
        float *img = s_a + ch * (TM * 32) + r * 32;
        *reinterpret_cast<f32x2 *>(img + ((q >> 1) ^ swz) * 4 + off) = ev;
        *reinterpret_cast<f32x2 *>(img + ((4 + (q >> 1)) ^ swz) * 4 + off) = od;
    }
}

template <int TM, int NCH, int NBUF>
__global__ __launch_bounds__(256) void srf_spconv_direct_k(const float *__restrict__ in, const float *__restrict__ Wd, int K,
                                                         const int *__restrict__ nbr, int nbr_stride, int A_out,
                                                         const float *__restrict__ alpha, const float *__restrict__ beta,
                                                         const float *__restrict__ residual, int relu,
                                                         float *__restrict__ out, const int *__restrict__ rows_dev)
{
    constexpr int COUT = 128, RT = TM / 32, NA = TM * 8 * NCH / 256;
    __shared__ int s_nbr[SRF_KMAX * TM];
    __shared__ int s_any[SRF_KMAX];
    __shared__ int s_klist[SRF_KMAX + 1];
    __shared__ __attribute__((aligned(16))) float s_a[NBUF][NCH * TM * 32];

    if (rows_dev) {  // static-shape levels: rows >= *rows_dev are padding; their tiles do nothing
        const int live = *rows_dev;
        A_out = A_out < live ? A_out : live;
    }
    // tiles of LIVE rows only, dealt XCD-contiguously over the first n_tiles workgroups (a capacity-sized launch must not
    // leave whole XCDs with nothing but padding)
    const int n_tiles = (A_out + TM - 1) / TM;
    if ((int)blockIdx.x >= n_tiles) return;
    const int row0 = srf_xcd_tile(blockIdx.x, n_tiles) * TM;
    const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6;
    srf_load_nbr_tile<TM>(nbr, nbr_stride, K, row0, A_out, s_nbr, s_any);
    if (tid < 64) {  // compact the used offsets with one ballot (K <= 27 < 64) instead of a serial loop on one lane
        const bool used = tid < K && s_any[tid];
        const unsigned long long m = __ballot(used);
        if (used) s_klist[__popcll(m & ((1ull << tid) - 1ull))] = tid;
        if (tid == 0) s_klist[SRF_KMAX] = __popcll(m);
    }
    __syncthreads();
    const int ntap = s_klist[SRF_KMAX];

    f32x16 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[rt][j] = 0.0f;

    const int kh = lane >> 5;
    // B fragments are fetched PD chunks ahead into one register slot per chunk position (PD = 2 for the 4-chunk layers:
    // a weight line that was evicted from L2 by the gathers takes longer than one chunk of MFMAs to arrive)
    constexpr int PD = NCH >= 4 ? 2 : 1, NS = NCH >= 4 ? NCH : 2;
    f32x4 ra[NA], bq[NS][4];
    unsigned okmask = 0;
    if (ntap > 0) {
        const int k0 = s_klist[0];
        srf_dir_load_b<NCH>(bq[0], Wd, k0, 0, wc, lane);
        if (PD == 2) srf_dir_load_b<NCH>(bq[1], Wd, k0, 1, wc, lane);
        srf_dir_gather<TM, NCH, NA>(in, s_nbr + k0 * TM, ra, okmask);
        srf_dir_store<TM, NCH, NA>(s_a[0], ra, okmask);
    }
    __syncthreads();
    for (int tk = 0; tk < ntap; ++tk) {
        const int buf = NBUF == 2 ? (tk & 1) : 0;
        const int kc = s_klist[tk];
        const bool more = tk + 1 < ntap;
        const int kn = more ? s_klist[tk + 1] : kc;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // B fragments of the next chunk (of this offset or the first of the next one); NCH is even, so the
            // register pair alternates consistently across offsets
            if (PD == 2) {
                if (c + 2 < NCH) srf_dir_load_b<NCH>(bq[c + 2], Wd, kc, c + 2, wc, lane);
                else srf_dir_load_b<NCH>(bq[c + 2 - NCH], Wd, kn, c + 2 - NCH, wc, lane);  // harmless re-read on the last offset
            } else {
                if (c + 1 < NCH) srf_dir_load_b<NCH>(bq[(c + 1) & 1], Wd, kc, c + 1, wc, lane);
                else srf_dir_load_b<NCH>(bq[(c + 1) & 1], Wd, kn, 0, wc, lane);
            }
            // the gather of the next offset goes out AFTER the last B load this offset still has to wait for: vmcnt
            // retires in order, so an earlier gather (an L2 miss more often than not) would be waited for at every
            // chunk; here it has two chunks of MFMAs to land and only the LDS store below waits for it
            if (c == NCH - 2 && more) srf_dir_gather<TM, NCH, NA>(in, s_nbr + kn * TM, ra, okmask);
            f32x4 af[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int arow = rt * 32 + (lane & 31);
                const float *pa = s_a[buf] + c * (TM * 32) + arow * 32;
                const int a_swz = (arow >> 1) & 7;
#pragma unroll
                for (int g = 0; g < 4; ++g) af[rt][g] = *reinterpret_cast<const f32x4 *>(pa + (((kh << 2) + g) ^ a_swz) * 4);
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rt][j >> 2][j & 3], bq[PD == 2 ? c : (c & 1)][j >> 2][j & 3], acc[rt], 0, 0, 0);
        }
        if (NBUF == 1) __syncthreads();  // single buffer: every wave is done reading before the rows are replaced
        if (more) srf_dir_store<TM, NCH, NA>(s_a[NBUF == 2 ? (buf ^ 1) : 0], ra, okmask);
        __syncthreads();
    }

    const int col = wc * 32 + (lane & 31);
    const float al = alpha ? alpha[col] : 1.0f;
    const float be = alpha ? beta[col] : 0.0f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = row0 + rt * 32 + (j & 3) + 8 * (j >> 2) + 4 * kh;
            if (row < A_out) {
                float v = acc[rt][j];
                if (alpha) v = __fmaf_rn(v, al, be);
                if (residual) v = __fadd_rn(v, residual[(size_t)row * COUT + col]);
                if (relu) v = v > 0.0f ? v : 0.0f;
                out[(size_t)row * COUT + col] = v;
            }
        }
}

// COUT = 64, Cin = 64 with the same LDS-free B operand: 64-row tiles, waves = 2 row halves x 2 column halves, each a
// 32 x 32 accumulator.  Against srf_spconv_packed_k<64, 64, 2, 2> (weight slab through LDS, a barrier per 32-channel chunk =
// per 16 MFMAs) there is one barrier per kernel offset (32 MFMAs per wave) and half the LDS traffic.
template <int NCH>
__global__ __launch_bounds__(256) void srf_spconv_direct64_k(const float *__restrict__ in, const float *__restrict__ Wd, int K,
                                                           const int *__restrict__ nbr, int nbr_stride, int A_out,
                                                           const float *__restrict__ alpha, const float *__restrict__ beta,
                                                           const float *__restrict__ residual, int relu,
                                                           float *__restrict__ out, const int *__restrict__ rows_dev)
{
    constexpr int COUT = 64, TM = 64, NA = TM * 8 * NCH / 256;
    static_assert(NCH == 2, "register slots below alternate over two chunks");
    __shared__ int s_nbr[SRF_KMAX * TM];
    __shared__ int s_any[SRF_KMAX];
    __shared__ int s_klist[SRF_KMAX + 1];
    __shared__ __attribute__((aligned(16))) float s_a[2][NCH * TM * 32];

    if (rows_dev) {  // static-shape levels: rows >= *rows_dev are padding; their tiles do nothing
        const int live = *rows_dev;
        A_out = A_out < live ? A_out : live;
    }
    const int n_tiles = (A_out + TM - 1) / TM;
    if ((int)blockIdx.x >= n_tiles) return;
    const int row0 = srf_xcd_tile(blockIdx.x, n_tiles) * TM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave & 1, wr = wave >> 1;
    srf_load_nbr_tile<TM>(nbr, nbr_stride, K, row0, A_out, s_nbr, s_any);
    if (tid < 64) {
        const bool used = tid < K && s_any[tid];
        const unsigned long long m = __ballot(used);
        if (used) s_klist[__popcll(m & ((1ull << tid) - 1ull))] = tid;
        if (tid == 0) s_klist[SRF_KMAX] = __popcll(m);
    }
    __syncthreads();
    const int ntap = s_klist[SRF_KMAX];

    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
    const int kh = lane >> 5;
    f32x4 ra[NA], bq[2][4];
    unsigned okmask = 0;
    if (ntap > 0) {
        const int k0 = s_klist[0];
        srf_dir_load_b<NCH, 2>(bq[0], Wd, k0, 0, wc, lane);
        srf_dir_gather<TM, NCH, NA>(in, s_nbr + k0 * TM, ra, okmask);
        srf_dir_store<TM, NCH, NA>(s_a[0], ra, okmask);
    }
    __syncthreads();
    const int arow = wr * 32 + (lane & 31);
    const int a_swz = (arow >> 1) & 7;
    for (int tk = 0; tk < ntap; ++tk) {
        const int buf = tk & 1;
        const int kc = s_klist[tk];
        const bool more = tk + 1 < ntap;
        const int kn = more ? s_klist[tk + 1] : kc;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // B fragments of the next chunk (of this offset, or the first of the next one) into the other register slot
            if (c + 1 < NCH) srf_dir_load_b<NCH, 2>(bq[(c + 1) & 1], Wd, kc, c + 1, wc, lane);
            else srf_dir_load_b<NCH, 2>(bq[(c + 1) & 1], Wd, kn, 0, wc, lane);  // harmless re-read on the last offset
            // the gather of the next offset goes out after the B load this chunk still waits for (vmcnt retires in order)
            if (c == NCH - 2 && more) srf_dir_gather<TM, NCH, NA>(in, s_nbr + kn * TM, ra, okmask);
            f32x4 af[4];
            const float *pa = s_a[buf] + c * (TM * 32) + arow * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) af[g] = *reinterpret_cast<const f32x4 *>(pa + (((kh << 2) + g) ^ a_swz) * 4);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j >> 2][j & 3], bq[c & 1][j >> 2][j & 3], acc, 0, 0, 0);
        }
        if (more) srf_dir_store<TM, NCH, NA>(s_a[buf ^ 1], ra, okmask);
        __syncthreads();
    }

    const int col = wc * 32 + (lane & 31);
    const float al = alpha ? alpha[col] : 1.0f;
    const float be = alpha ? beta[col] : 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int row = row0 + wr * 32 + (j & 3) + 8 * (j >> 2) + 4 * kh;
        if (row < A_out) {
            float v = acc[j];
            if (alpha) v = __fmaf_rn(v, al, be);
            if (residual) v = __fadd_rn(v, residual[(size_t)row * COUT + col]);
            if (relu) v = v > 0.0f ? v : 0.0f;
            out[(size_t)row * COUT + col] = v;
        }
    }
}

// =====================================================================================================================
// COUT = 128, Cin in {64, 128} and COUT = Cin = 64, compacted offsets ("gs": gather rows, scatter into an LDS-resident output
// tile).  (COUT = 64: a wave owns one 16-column MFMA tile instead of two, the output tile is half as wide and up to 120 rows
// tall; on the 64-channel level of a nuScenes sweep only 40 % of the (row, offset) pairs exist, and the output-stationary
// srf_spconv_direct64_k spent 60 % of its MFMAs on zeros: 159 -> 107 us per layer.)
// In the output-stationary kernels above a 32-row tile spends one full MFMA pass on every kernel offset any of its rows
// uses, although only 16-20 of the 27 neighbours of a row exist: 40 % of the issued MFMAs multiply zeros (59 % useful at
// the 128-channel level of a nuScenes sweep, 73 % on a Waymo sweep).  Here a workgroup owns up to 88 output rows whose
// accumulators live in LDS.  For every offset k the rows that HAVE a neighbour at k are compacted (one ballot per 64 rows)
// and processed in groups of 16 on v_mfma_f32_16x16x4_f32: the group's accumulators are read from the output tile (row =
// the group's output slot), run through the channel-ascending MFMA chain and written back, so every output element still
// sees exactly the chain (offset ascending, channel ascending) of the oracle -- minus the terms that were exact zeros:
// results stay bit-identical.  86 % of the issued MFMAs are useful on the nuScenes sweep (90 % Waymo); the kernel issues
// 70 % of the MFMA cycles of srf_spconv_direct_k (profiles/r01_pmc_spconv128_gs_traffic.json).
// B operands go global(L2) -> registers in MFMA order as in the direct kernel, once per offset and tile (not once per
// group), one offset ahead in a second register set; the gathered rows of the next group are fetched during the MFMAs of
// the current one (double-buffered LDS).  A wave owns 32 output channels of the tile, so no other wave touches its
// accumulators: one barrier per group (the A hand-over) suffices.
//   gs layout: Wg[k][chunk][wc = col/32][g][lane][i], idx = 4g + i, = W[k][chunk*32 + 4*(idx&7) + (lane>>4)]
//                                                                      [wc*32 + 16*(idx>>3) + (lane&15)]
// LDS: output tile (88 + 1 spare row for the padding slots of last groups) x 132 x 4 = 47 KB, A 2 x (NCH x 2 KB), row lists 27 x 96 x 5 B = 13 KB -> 76 KB, two workgroups
// per CU (224 VGPRs).  Which rows a workgroup owns: srf_spconv_tiles_build below (ranges of equal cost), or equal-height
// tiles when the caller passes no ranges.
// Measured (MI355X, nuScenes level 4, 35k rows, 556k pairs): 257 us vs 342 us for srf_spconv_direct_k; 514 vs 582 us on a
// 66k-row level with 19.5 pairs per row.  A workgroup spends ~45 % of a group's time issuing MFMAs and the two
// workgroups of a CU do not interleave perfectly (MFMA pipe busy 64 %): the remaining distance to the MFMA roofline.
// =====================================================================================================================
#define SRF_GS_TMAX 88   /* most output rows of a tile: 2 workgroups per CU still fit in LDS */
#define SRF_GS_LS 96     /* stride of the per-offset row lists (TMAX rounded up to whole groups) */
#define SRF_GS_OS 132    /* output-tile row stride in floats: rows 4 apart land 16 banks apart */
#define SRF_GS_CHS (16 * 32 + 8) /* chunk stride of the A image: the four chunks of a row start 8 banks apart */
#define SRF_GS_SLOTS 512 /* co-resident workgroups: 256 CUs x 2 */

// Tile height.  A tile is ~80 groups of MFMAs (~200 us): with a fixed height the last partial round of tiles would leave
// most of the chip idle for that long (547 tiles of 64 rows on 512 slots: 35 tiles run alone in a second round).  The
// height is therefore chosen from the live row count so that the tiles fill a whole number of rounds of 512: 1 round up
// to 45k rows, 2 rounds up to 90k, ...  Evaluated identically on the host (grid size from the capacity) and in the
// kernel (from the device row count of a static-shape level).
__host__ __device__ static inline int srf_gs_rounds(int A) { return A <= 0 ? 1 : (A + SRF_GS_SLOTS * SRF_GS_TMAX - 1) / (SRF_GS_SLOTS * SRF_GS_TMAX); }
__host__ __device__ static inline int srf_gs_tile_rows(int A)
{
    const int per = SRF_GS_SLOTS * srf_gs_rounds(A);
    int tm = ((A + per - 1) / per + 7) & ~7;
    return tm < 16 ? 16 : tm;
}

static bool srf_gs_enabled()
{
    static const bool on = [] {
        const char *e = getenv("SRF_SPCONV_DIRECT");  // developer switch: keep the previous (direct) kernel for A/B timing
        return !(e && e[0] == '1');
    }();
    return on;
}
static bool srf_gsp_enabled();
static bool srf_gs_layout(int Cin, int Cout)
{
    // (32 -> 64, the strided convolution into the 64-channel level: only the pipelined kernel srf_spconv_gsp_k has a one-chunk form)
    return srf_gs_enabled() && ((Cout == 128 && (Cin == 64 || Cin == 128)) || (Cout == 64 && (Cin == 64 || (Cin == 32 && srf_gsp_enabled()))));
}

__global__ __launch_bounds__(256) void srf_pack_weights_gs_k(const float *__restrict__ W, int K, int Cin, int Cout, int nchunk,
                                                           float *__restrict__ P)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)K * nchunk * Cout * 32;
    if (t >= total) return;
    const int i = (int)(t & 3), lane = (int)((t >> 2) & 63), g = (int)((t >> 8) & 3);
    const int per = Cout * 32;  // elements per (offset, chunk): Cout / 32 column blocks x 4 groups x 64 lanes x 4
    const int wc = (int)((t % per) >> 10);
    const long long rest = t / per;
    const int chunk = (int)(rest % nchunk), k = (int)(rest / nchunk);
    const int idx = 4 * g + i;
    const int col = wc * 32 + 16 * (idx >> 3) + (lane & 15);
    const int c = chunk * 32 + 4 * (idx & 7) + (lane >> 4);
    P[t] = c < Cin ? W[((size_t)k * Cin + c) * Cout + col] : 0.0f;
}

// rows of one group, global -> registers.  Padding entries of a last group read row 0 and accumulate into a spare row of
// the output tile that is never stored: any finite-or-not value will do, no zero fill and no predicates are needed
template <int NCH, int NA>
__device__ __forceinline__ void srf_gs_gather(const float *__restrict__ in, const int *s_rows, f32x4 (&ra)[NA])
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int e = tid + j * 256;
        const int r = e / (8 * NCH), q = e % (8 * NCH);
        const int i = s_rows[r];  // padding entries of a last group name row 0
        ra[j] = *reinterpret_cast<const f32x4 *>(in + (size_t)i * (32 * NCH) + q * 4);
    }
}

// A image of one group: [chunk][row 0..15][32], channel 4q + j of a chunk at position j*8 + q (lane (row, j) of the
// 16x16x4 MFMA reads its eight steps as two b128), 16-byte units XOR-swizzled by (row >> 1) & 7
template <int NCH, int NA, int CHS = SRF_GS_CHS>
__device__ __forceinline__ void srf_gs_store(float *s_a, const f32x4 (&ra)[NA])
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int e = tid + j * 256;
        const int r = e / (8 * NCH), qq = e % (8 * NCH);
        const int ch = qq >> 3, q = qq & 7;
        const int swz = (r >> 1) & 7;
        float *img = s_a + ch * CHS + r * 32 + (q & 3);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) img[((jj * 2 + (q >> 2)) ^ swz) << 2] = ra[j][jj];
    }
}

// B operands of one (offset, chunk) for this wave.  COUT = 128: the wave owns 32 columns (two 16-column MFMA tiles, four
// f32x4); COUT = 64: 16 columns (tile `wave & 1` of column block `wave >> 1`, two f32x4 of the same packed image)
template <int NCH, int COUT>
__device__ __forceinline__ void srf_gs_load_b(f32x4 (&b)[COUT / 32], const float *__restrict__ Wg, int k, int chunk, int wave, int lane)
{
    constexpr int NWC = COUT / 32;
    const int wc = COUT == 128 ? wave : (wave >> 1), g0 = COUT == 128 ? 0 : 2 * (wave & 1);
#pragma unroll
    for (int g = 0; g < COUT / 32; ++g)
        b[g] = *reinterpret_cast<const f32x4 *>(Wg + (((((size_t)k * NCH + chunk) * NWC + wc) * 4 + g0 + g) * 64 + lane) * 4);
}

// GP = groups of 16 rows per step (one gather / barrier per step).  COUT = 64 runs GP = 2: a wave's share of a group is
// only 16 MFMAs (512 cycles), less than the L2 latency of the next gather, so a step carries two groups (two independent
// accumulator chains) and the tile needs a third fewer steps.
template <int NCH, int NA, int COUT, int LS, int OS, int TMAX, int GP>
__device__ __forceinline__ void srf_gs_offset(const float *__restrict__ in, const float *__restrict__ Wg, int kc, int kn, bool more_k,
                                              const int *s_in, const unsigned char *s_slot, const int *s_cnt, float *s_out,
                                              float *s_a, int &buf, f32x4 (&bc)[NCH][COUT / 32], f32x4 (&bn)[NCH][COUT / 32],
                                              f32x4 (&ra)[NA])
{
    constexpr int NT = COUT / 64;  // 16-column MFMA tiles per wave
    constexpr int RS = 16 * GP, CHS = RS * 32 + 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int colb = COUT == 128 ? wave * 32 : wave * 16;
    const int n = s_cnt[kc];
    const int ng = (n + RS - 1) / RS;
    const int ar = lane & 15, aj = lane >> 4;
    const int a_swz = (ar >> 1) & 7;
    for (int g = 0; g < ng; ++g) {
        const bool last = g + 1 == ng;
        const bool has_next = !last || more_k;
        if (last && more_k) {  // B of the next offset, one whole step of MFMAs ahead, into the other register set
#pragma unroll
            for (int c = 0; c < NCH; ++c) srf_gs_load_b<NCH, COUT>(bn[c], Wg, kn, c, wave, lane);
        }
        if (has_next) srf_gs_gather<NCH, NA>(in, last ? s_in + kn * LS : s_in + kc * LS + (g + 1) * RS, ra);
        // all A fragments of the step first (ds_read_b128 in flight together, one exposed LDS latency per step instead of
        // one per chunk), then the accumulators of its rows out of the output tile
        f32x4 af[GP][NCH][2];
        f32x4 acc[GP][NT];
        int oaddr[GP][4];
#pragma unroll
        for (int gp = 0; gp < GP; ++gp) {
            const float *abase = s_a + buf * (NCH * CHS) + (gp * 16 + ar) * 32;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                af[gp][c][0] = *reinterpret_cast<const f32x4 *>(abase + c * CHS + (((aj << 1) ^ a_swz) << 2));
                af[gp][c][1] = *reinterpret_cast<const f32x4 *>(abase + c * CHS + ((((aj << 1) + 1) ^ a_swz) << 2));
            }
        }
#pragma unroll
        for (int gp = 0; gp < GP; ++gp) {
            // output slots of the group's 16 rows; padding rows of a last step name the spare row TMAX
            const unsigned sl4 = *reinterpret_cast<const unsigned *>(s_slot + kc * LS + (g * GP + gp) * 16 + aj * 4);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                oaddr[gp][jj] = (int)((sl4 >> (8 * jj)) & 255u) * OS + colb + ar;
#pragma unroll
                for (int cb = 0; cb < NT; ++cb) acc[gp][cb][jj] = s_out[oaddr[gp][jj] + cb * 16];
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the reads above the MFMA block
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int gp = 0; gp < GP; ++gp) {
                    const float a = af[gp][c][s >> 2][s & 3];
                    acc[gp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bc[c][s >> 2][s & 3], acc[gp][0], 0, 0, 0);
                    if (NT == 2)
                        acc[gp][NT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bc[c][COUT / 32 - 2 + (s >> 2)][s & 3], acc[gp][NT - 1], 0, 0, 0);
                }
            }
        }
        // a padding slot (spare row TMAX) may appear in both groups of a step: both write garbage there, never read back
#pragma unroll
        for (int gp = 0; gp < GP; ++gp)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                for (int cb = 0; cb < NT; ++cb) s_out[oaddr[gp][jj] + cb * 16] = acc[gp][cb][jj];
            }
        // every load issued in this step has landed before the next one starts.  Stated explicitly (s_waitcnt vmcnt(0)):
        // the compiler cannot tie "B was prefetched" to "the gather was waited for" across the two branches and would
        // otherwise guard the next step's MFMAs with vmcnt waits that also catch that step's own fresh loads
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (has_next) srf_gs_store<NCH, NA, CHS>(s_a + (buf ^ 1) * (NCH * CHS), ra);
        __syncthreads();
        buf ^= 1;
    }
}

// ---- work-balanced row ranges ------------------------------------------------------------------------------------------
// One round of co-resident workgroups finishes when its heaviest tile does, and the pairs per 72 rows of a sweep vary by
// 1.5x around their mean (dense clusters vs. isolated returns).  srf_spconv_tiles_build therefore cuts the rows of a
// rulebook into T ranges of (almost) equal PAIR count -- range t = the rows whose exclusive pair prefix lies in
// [t*P/T, (t+1)*P/T) -- once per rulebook (the four SubM layers of a level share it).  A workgroup walks its range in
// sub-tiles of at most SRF_GS_TMAX rows.
#define SRF_TB_ROWS 256
#define SRF_GS_ROW_COST 8 /* cost of a row = its pairs + 8: the constant carries the row's share of the per-sub-tile work (list
                             build, epilogue: ~20 k cycles per sub-tile against ~270 cycles per pair in the step loop of
                             srf_spconv_gsp_k) and keeps sparse ranges from growing taller than a tile, which costs a second
                             sub-tile (+35 k cycles).  Round 1's kernel, whose steps were twice as long, was fitted with 12.
                             Measured (round 5, in-kernel stamps + bench_spconv): 128 -> 128 (35k rows, 68 rows per range)
                             226 / 219 us with 12 / 4 on one box, 228 / 228 on another; 64 -> 64 (60k rows, 116 rows per range of
                             at most 128) 88 / 91.5 us: the taller the ranges of a level against its tile, the larger the constant
                             it wants */
__host__ __device__ static inline int srf_gs_ranges(int A_cap)
{
    const int t = A_cap / 32;
    return t < 1 ? 1 : (t > SRF_GS_SLOTS ? SRF_GS_SLOTS : t);
}

__device__ __forceinline__ int srf_block_scan_excl(int v, int *s_scan /* 256 + 4 */, int &total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) s_scan[wave] = x;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wave) base += s_scan[w];
    }
    total = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
    __syncthreads();
    return base + x - v;
}

// cost per row, exclusive prefix inside blocks of SRF_TB_ROWS rows (row order), block totals
__global__ __launch_bounds__(256) void srf_gs_rowpairs_k(const int *__restrict__ nbr, int nbr_stride, int K, int A,
                                                       const int *__restrict__ rows_dev, int *__restrict__ local,
                                                       int *__restrict__ blocksum, int row_cost)
{
    __shared__ int s_scan[4];
    if (rows_dev) {
        const int live = *rows_dev;
        A = A < live ? A : live;
    }
    const int r0 = blockIdx.x * SRF_TB_ROWS;
    int carry = 0;
    for (int j = 0; j < SRF_TB_ROWS / 256; ++j) {  // block-uniform trip count: the barriers inside the scan are safe
        const int r = r0 + j * 256 + threadIdx.x;
        int c = 0;
        {
            // all K entries of the row in flight at once (as a loop of `c += nbr[..] >= 0` every entry was its own round trip: 27 in a
            // row, the whole of this 9-10 us launch); rows past the end read row A - 1 and count nothing
            const int rc = r < A ? r : (A > 0 ? A - 1 : 0);   // (A == 0, a level without live rows: row 0 of the capacity-sized table, counted as nothing)
            int v[SRF_KMAX];
#pragma unroll
            for (int k = 0; k < SRF_KMAX; ++k) v[k] = nbr[(size_t)(k < K ? k : K - 1) * nbr_stride + rc];
#pragma unroll
            for (int k = 0; k < SRF_KMAX; ++k) c += (k < K && v[k] >= 0) ? 1 : 0;
            c = r < A ? c + row_cost : 0;
        }
        int total;
        const int ex = srf_block_scan_excl(c, s_scan, total);
        if (r < A) local[r] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) blocksum[blockIdx.x] = carry;
}

// exclusive scan of the block totals (one workgroup; nb <= a few hundred), total pair count at [nb]
__global__ __launch_bounds__(256) void srf_gs_blockscan_k(const int *__restrict__ blocksum, int nb, int *__restrict__ blockoff)
{
    __shared__ int s_scan[4];
    int carry = 0;
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const int v = b < nb ? blocksum[b] : 0;
        int total;
        const int ex = srf_block_scan_excl(v, s_scan, total);
        if (b < nb) blockoff[b] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) blockoff[nb] = carry;
}

// tiles[t] = first row whose exclusive pair prefix reaches t*P/T (binary search over blocks, then inside the block)
__global__ __launch_bounds__(256) void srf_gs_cut_k(const int *__restrict__ local, const int *__restrict__ blockoff, int nb, int A,
                                                  const int *__restrict__ rows_dev, int T, int *__restrict__ tiles)
{
    if (rows_dev) {
        const int live = *rows_dev;
        A = A < live ? A : live;
    }
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t > T) return;
    const int nbl = (A + SRF_TB_ROWS - 1) / SRF_TB_ROWS;  // live blocks (<= nb)
    if (t == T || A == 0) {
        tiles[t] = A;
        return;
    }
    const long long P = blockoff[nbl];
    const long long target = (P * t + T - 1) / T;
    // last block whose offset is <= target
    int lo = 0, hi = nbl - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (blockoff[mid] <= target) lo = mid;
        else hi = mid - 1;
    }
    const long long boff = blockoff[lo];
    const int r0 = lo * SRF_TB_ROWS;
    const int r1 = r0 + SRF_TB_ROWS < A ? r0 + SRF_TB_ROWS : A;
    // first row in [r0, r1) with boff + local[r] >= target, else r1
    int a = r0, b = r1;
    while (a < b) {
        const int mid = (a + b) >> 1;
        if (boff + local[mid] >= target) b = mid;
        else a = mid + 1;
    }
    tiles[t] = a;
}

extern "C" int srf_spconv_tiles_count(int A_out) { return A_out <= 0 ? 1 : srf_gs_ranges(A_out); }

static int srf_gs_row_cost_value()
{
    static const int row_cost = [] {
        const char *e = getenv("SRF_GS_ROW_COST");   // developer switch (A/B timing of the cost model)
        return e ? atoi(e) : SRF_GS_ROW_COST;
    }();
    return row_cost;
}

extern "C" int srf_spconv_tiles_row_cost(void) { return srf_gs_row_cost_value(); }

extern "C" size_t srf_spconv_tiles_workspace_bytes(int A_out)
{
    if (A_out < 0) return 0;
    const size_t nb = (size_t)(A_out + SRF_TB_ROWS - 1) / SRF_TB_ROWS;
    return ((size_t)A_out + 2 * nb + 8) * sizeof(int);
}

extern "C" int srf_spconv_tiles_build(const int *nbr, int nbr_stride, int K, int A_out, const int *rows_dev, void *workspace,
                                      int *tiles, srf_stream_t stream)
{
    if (A_out < 0 || K <= 0 || K > SRF_KMAX || nbr_stride < A_out || !tiles) return SRF_EINVAL;
    if (A_out > 0 && (!nbr || !workspace)) return SRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int T = srf_spconv_tiles_count(A_out);
    const int nb = (A_out + SRF_TB_ROWS - 1) / SRF_TB_ROWS;
    int *local = (int *)workspace, *blocksum = local + A_out, *blockoff = blocksum + nb + 1;
    if (nb > 0) {
        const int row_cost = srf_gs_row_cost_value();
        hipLaunchKernelGGL(srf_gs_rowpairs_k, dim3(nb), dim3(256), 0, st, nbr, nbr_stride, K, A_out, rows_dev, local, blocksum, row_cost);
        hipLaunchKernelGGL(srf_gs_blockscan_k, dim3(1), dim3(256), 0, st, blocksum, nb, blockoff);
    }
    hipLaunchKernelGGL(srf_gs_cut_k, dim3(srf_ceil_div(T + 1, 256)), dim3(256), 0, st, local, blockoff, nb, A_out, rows_dev, T, tiles);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

template <int NCH, int COUT>
__global__ __launch_bounds__(256, 2) void srf_spconv_gs_k(const float *__restrict__ in, const float *__restrict__ Wg, int K,
                                                        const int *__restrict__ nbr, int nbr_stride, int A_out,
                                                        const float *__restrict__ alpha, const float *__restrict__ beta,
                                                        const float *__restrict__ residual, int relu,
                                                        float *__restrict__ out, const int *__restrict__ rows_dev,
                                                        const int *__restrict__ tiles)
{
    // COUT = 64 (Cin = 64): a wave owns one 16-column MFMA tile, the output tile is half as wide and may be taller
    // groups of 16 rows per step (srf_gs_offset).  COUT = 64, nuScenes level 3 (59.6k rows, 10.9 pairs per row): 107 us with 1,
    // 123 us with 2; 64-row tiles walked by two workgroups per range (four per CU) 118 us; B fetched a whole offset ahead 113 us.
    // Ablations of the 107 us: without MFMAs 70, without the gathers 99, without barriers 100, prologue + epilogue alone 14.
    // Three workgroups per CU (88-row tiles, 768 ranges, 46 KB of LDS each): 105 us -- occupancy is not what holds it.
    // Round 3: starting the workgroup in the odd wave slots of a CU (HW_ID) 1-4 k cycles late, so that the two co-resident
    // workgroups begin out of phase: no change (108 / 262 us with and without, both shapes) -- they do not run in lock-step.
    constexpr int GP = 1;
    constexpr int NA = 16 * GP * 8 * NCH / 256, NKW = (SRF_KMAX + 3) / 4, CHS = 16 * GP * 32 + 8;
    constexpr int TMAX = COUT == 128 ? SRF_GS_TMAX : 120, LS = COUT == 128 ? SRF_GS_LS : 128, OS = COUT + 4;
    constexpr int SPLIT = 1;  // workgroups per range of the row cut (2 with 64-row tiles was slower for COUT = 64: 118 vs 107 us)
    static_assert(COUT == 128 || COUT == 64, "column tiling of the waves");
    static_assert(TMAX <= 128 && TMAX <= LS && LS % (16 * GP) == 0, "two ballot segments of 64 rows; whole groups per list");
    static_assert(NA >= 1, "a group is at least one f32x4 per thread");
    __shared__ int s_in[SRF_KMAX * LS];                 // per offset: input rows of the outputs that have this neighbour
    __shared__ __attribute__((aligned(4))) unsigned char s_slot[SRF_KMAX * LS];  // ... and their slot in the output tile
    __shared__ int s_cnt[SRF_KMAX];
    __shared__ int s_klist[SRF_KMAX + 1];
    __shared__ __attribute__((aligned(16))) float s_out[(TMAX + 1) * OS];  // + the spare row of padding slots
    __shared__ __attribute__((aligned(16))) float s_a[2 * NCH * CHS];

    const int A_cap = A_out;
    if (rows_dev) {  // static-shape levels: rows >= *rows_dev are padding; their tiles do nothing
        const int live = *rows_dev;
        A_out = A_out < live ? A_out : live;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this workgroup's rows: a work-balanced range of the rulebook (walked in equal sub-tiles), or one equal-height tile
    int range0, range1;
    if (tiles) {
        const int T = srf_gs_ranges(A_cap);
        if ((int)blockIdx.x >= T * SPLIT) return;
        const int ts = srf_xcd_tile(blockIdx.x, T * SPLIT);
        const int t = ts / SPLIT, part = ts - t * SPLIT;
        range0 = tiles[t];
        range1 = tiles[t + 1];
        range1 = range1 < A_out ? range1 : A_out;
        if (SPLIT > 1 && range1 > range0) {  // this workgroup's share of the range: whole multiples of 8 rows
            const int per = (((range1 - range0 + SPLIT - 1) / SPLIT) + 7) & ~7;
            range0 += part * per;
            range1 = range0 + per < range1 ? range0 + per : range1;
        }
    } else {
        const int tm = srf_gs_tile_rows(A_out);
        const int n_tiles = (A_out + tm - 1) / tm;
        if ((int)blockIdx.x >= n_tiles) return;
        range0 = srf_xcd_tile(blockIdx.x, n_tiles) * tm;
        range1 = range0 + tm < A_out ? range0 + tm : A_out;
    }
    if (range1 <= range0) return;
    const int nsub = (range1 - range0 + TMAX - 1) / TMAX;
    const int TM = (((range1 - range0 + nsub - 1) / nsub) + 7) & ~7;  // <= TMAX (a multiple of 8)
    for (int row0 = range0; row0 < range1; row0 += TM) {
    const int row_end = row0 + TM < range1 ? row0 + TM : range1;  // rows of this sub-tile: [row0, row_end)
    // An opaque zero: the address arithmetic of the prologue / epilogue below is invariant across sub-tiles, and hoisted
    // above this loop it stays live through the main loop, where every register is taken -- the compiler then spills it
    // (11 MB of scratch write-back per launch).  Tied to this value it is recomputed per sub-tile instead.
    int zero = 0;
    asm volatile("" : "+s"(zero));
    for (int e = tid; e < TM * OS / 4; e += 256) reinterpret_cast<f32x4 *>(s_out)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    // compaction: the wave's offsets (wave, wave + 4, ...), rows in two segments of 64; all loads in flight together
    int nv[NKW][2];
#pragma unroll
    for (int i = 0; i < NKW; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = wave + 4 * i, r = h * 64 + lane;
            nv[i][h] = (k < K && row0 + r < row_end) ? nbr[(size_t)(k + zero) * nbr_stride + row0 + r] : -1;
        }
#pragma unroll
    for (int i = 0; i < NKW; ++i) {
        const int k = wave + 4 * i;
        if (k >= SRF_KMAX) break;
        int *lin = s_in + (k + zero) * LS;
        unsigned char *lsl = s_slot + (k + zero) * LS;
        lin[lane] = 0;  // padding of the last group: input row 0 into the spare output row (same wave: ordered before the
        lsl[lane] = (unsigned char)TMAX;  // compacted stores below)
        if (lane < LS - 64) {
            lin[64 + lane] = 0;
            lsl[64 + lane] = (unsigned char)TMAX;
        }
        int base = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int v = nv[i][h];
            const unsigned long long m = __ballot(v >= 0);
            if (v >= 0) {
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                lin[pos] = v;
                lsl[pos] = (unsigned char)(h * 64 + lane);
            }
            base += __popcll(m);
        }
        if (lane == 0) s_cnt[k] = base;
    }
    __syncthreads();
    if (tid < 64) {
        const bool used = tid < K && s_cnt[tid] > 0;
        const unsigned long long m = __ballot(used);
        if (used) s_klist[__popcll(m & ((1ull << tid) - 1ull))] = tid;
        if (tid == 0) s_klist[SRF_KMAX] = __popcll(m);
    }
    __syncthreads();
    const int ntap = s_klist[SRF_KMAX];

    f32x4 b0[NCH][COUT / 32], b1[NCH][COUT / 32], ra[NA];
    int buf = 0;
    if (ntap > 0) {
        const int k0 = s_klist[0];
#pragma unroll
        for (int c = 0; c < NCH; ++c) srf_gs_load_b<NCH, COUT>(b0[c], Wg, k0, c, wave, lane);
        srf_gs_gather<NCH, NA>(in, s_in + k0 * LS, ra);
        srf_gs_store<NCH, NA, CHS>(s_a, ra);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing is in flight when the first group starts (see srf_gs_offset)
    __syncthreads();
    for (int tk = 0; tk < ntap; tk += 2) {  // two offsets per trip: the B register sets swap roles without moves
        {
            const bool more = tk + 1 < ntap;
            const int kc = s_klist[tk], kn = more ? s_klist[tk + 1] : kc;
            srf_gs_offset<NCH, NA, COUT, LS, OS, TMAX, GP>(in, Wg, kc, kn, more, s_in, s_slot, s_cnt, s_out, s_a, buf, b0, b1, ra);
        }
        if (tk + 1 < ntap) {
            const bool more = tk + 2 < ntap;
            const int kc = s_klist[tk + 1], kn = more ? s_klist[tk + 2] : kc;
            srf_gs_offset<NCH, NA, COUT, LS, OS, TMAX, GP>(in, Wg, kc, kn, more, s_in, s_slot, s_cnt, s_out, s_a, buf, b1, b0, ra);
        }
    }

    // epilogue: every output row once, BN / residual / ReLU in registers, 512 B per row and store
    constexpr int CQ = COUT / 4;  // float4 per output row
    const int c4 = ((tid & (CQ - 1)) + zero) * 4;
    f32x4 al = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
    if (alpha) {
        al = *reinterpret_cast<const f32x4 *>(alpha + c4);
        be = *reinterpret_cast<const f32x4 *>(beta + c4);
    }
    for (int r = tid / CQ; r < TM; r += 256 / CQ) {
        const int row = row0 + r;
        if (row >= row_end) break;
        f32x4 v = *reinterpret_cast<const f32x4 *>(s_out + r * OS + c4);
        f32x4 rs = {0.f, 0.f, 0.f, 0.f};
        if (residual) rs = *reinterpret_cast<const f32x4 *>(residual + (size_t)row * COUT + c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = v[j];
            if (alpha) x = __fmaf_rn(x, al[j], be[j]);
            if (residual) x = __fadd_rn(x, rs[j]);
            if (relu) x = x > 0.0f ? x : 0.0f;
            v[j] = x;
        }
        *reinterpret_cast<f32x4 *>(out + (size_t)row * COUT + c4) = v;
    }
    __syncthreads();  // the output tile and the row lists are rebuilt by the next sub-tile
    }
}

// =====================================================================================================================
// srf_spconv_gsp_k: the compacted-offset kernel above with its step SOFTWARE-PIPELINED (round 5).
// In srf_spconv_gs_k a step (one group of 16 compacted rows) is a serial chain in every wave: read the A fragments and the
// group's accumulators from LDS (0.71 us of exposed LDS latency and address arithmetic in the in-kernel stamps), 64 MFMAs
// (0.92 us), accumulators back (0.18), wait for the gather + store it (0.22), barrier: the MFMA pipe sees 45 % of a wave's time
// and the two workgroups of a CU interleave only partly (62 % busy).  Same arithmetic here -- the same (offset, channel)-
// ascending chain per output, bit-identical results -- but every wave's MFMA block is the only thing it waits for:
//   * the rows of ALL offsets of a sub-tile form ONE flat list of steps (per offset padded to whole groups of 16; s_pin holds
//     the byte offset of the input row, s_pslot the output slot): step i's rows are entries 16 i .. 16 i + 15 whatever its
//     offset, so looking two steps ahead is plain address arithmetic (the per-offset lists of the kernel above needed the
//     offset after next for that);
//   * step i issues, in this order: the chunk-0 fragments of A[i]; the LDS stores of A[i + 1] (gathered during step i - 1:
//     landed long ago, no wait); the gather of step i + 2; then per 32-channel chunk the fragment reads of the NEXT chunk
//     followed by the chunk's MFMAs (fragments double-buffered per chunk: 16 registers instead of 32), the output slots of
//     step i + 1 inside the first chunk; accumulators of step i back to the tile and those of step i + 1 out of it (the
//     same wave's LDS operations execute in order, so rows shared by the two steps are read after they were written);
//     barrier.  LDS instructions issued between the MFMAs of the own wave cost ~1 cycle per MFMA; what stays exposed per
//     step is one LDS latency behind the barrier and the accumulator round trip;
//   * address arithmetic per step: rows are gathered through a buffer descriptor (32-bit offset row + quad: one add per
//     load, padding entries are out-of-range offsets that read zeros without touching memory), the store addresses of the A
//     image are precomputed per thread: ~20 vector instructions per step (they take MFMA issue time on this chip);
//   * the last step of an offset is a separate instantiation (LAST) that also fetches the next offset's B operands into the
//     other register set right behind the gather: no branch inside a step, the compiler's vmcnt bookkeeping stays exact.
// The flat lists carry two dummy steps behind the real ones (zeros into the spare row), so nothing in a step is conditional.
// =====================================================================================================================

static bool srf_gsp_enabled()
{
    static const bool on = [] {
        const char *e = getenv("SRF_SPCONV_GSP");  // developer switch: 0 keeps srf_spconv_gs_k (A/B timing, parity of the two forms)
        return !(e && e[0] == '0');
    }();
    return on;
}

typedef __attribute__((address_space(3))) float srf_lds_float;
typedef __attribute__((address_space(3))) f32x4 srf_lds_f32x4;
__device__ __forceinline__ unsigned srf_lds_addr(const void *p) { return (unsigned)(uintptr_t)p; }   // LDS byte address of a __shared__ object

// per-thread constants of a step (LDS byte addresses for buffer 0 of the A image; the other buffer is `abuf_bytes` further)
struct SrfGspLane {
    unsigned fo0, fo1;     // the lane's two fragment quads inside chunk 0
    unsigned sto[4];       // where the four floats of the lane's first gathered quad go (its second quad: two chunks further)
    unsigned goff;         // byte offset of that quad inside the input row
    unsigned colbase;      // LDS byte address of (slot 0, the lane's first column) of the output tile
    unsigned boff;         // byte offset of the lane's B fragment inside a (offset, chunk) image
};

// ABL: timing ablations of the developer build (-DSRF_DEV; wrong outputs by design): 1 = no MFMAs, 2 = no gathers / A stores,
// 3 = no accumulator round trip through the tile, 4 = no barrier
#ifdef SRF_DEV
__device__ long long srf_gsp_stamps[512 * 16];   // developer build: per workgroup the cycles of a step spent before / in / behind the MFMA block and at the barrier, the step count, prologue / loop / epilogue
extern "C" int srf_dev_gsp_stamps(long long *host, int n)
{
    SRF_HIP_TRY(hipDeviceSynchronize());
    SRF_HIP_TRY(hipMemcpyFromSymbol(host, HIP_SYMBOL(srf_gsp_stamps), sizeof(long long) * (n < 512 * 16 ? n : 512 * 16)));
    return SRF_OK;
}
#endif
// what a wave carries from step to step: every address the LDS-only parts of a step use is prepared INSIDE the MFMA block before
template <int GP>
struct SrfGspRegs {
    unsigned f0, f1;          // fragment quads of the A image the step reads
    unsigned st[4];           // where it stores the rows of the step after it
    unsigned gaddr[GP];       // buffer offsets of the rows it requests (step + 2)
    unsigned oaddr[GP][4];    // tile addresses of its accumulators
    unsigned pin_addr;        // LDS address of the row-list entries it reads (step + 3: the request after next)
    unsigned slot_addr;       // LDS address of the slots it reads (step + 1)
    unsigned info_addr;       // LDS address of the offset flag it reads (step + 1)
};

// GP = groups of 16 rows per step.  COUT = 64 runs GP = 2: a wave owns ONE 16-column MFMA tile there, i.e. one dependent chain of 16
// MFMAs per group (40 cycles each instead of 32) behind a fixed chain of LDS latencies (fragments -> MFMAs -> accumulators back and
// out -> barrier); two groups per step are two independent chains and half the barriers per pair.  (The rows of an offset are padded
// to whole steps: 27 x 16 instead of 27 x 8 padding rows per sub-tile on average.)
template <int NCH, int COUT, int GP, bool LAST, int ABL = 0>
__device__ __forceinline__ void srf_gsp_step(__amdgpu_buffer_rsrc_t rs, __amdgpu_buffer_rsrc_t wrs, int kn, int &buf, const SrfGspLane &L,
                                             SrfGspRegs<GP> &G, f32x4 (&bc)[NCH][COUT / 32], f32x4 (&bn)[NCH][COUT / 32],
                                             f32x4 (&ra)[GP][NCH >= 2 ? NCH / 2 : 1], f32x4 (&acc)[GP][COUT / 64], long long (&stamp)[8], int &info_next)
{
    constexpr int NT = COUT / 64, NB = COUT / 32, NA = NCH >= 2 ? NCH / 2 : 1, RS = 16 * GP, CHS = RS * 32 + 8, OS = COUT + 4;
    constexpr int TPR = NCH >= 2 ? 16 : 8;     // threads per gathered row (a quad each; rows of 128 channels: two quads each)
    constexpr unsigned ABUF = NCH * CHS * 4;   // bytes of one A buffer
    typedef const __attribute__((address_space(3))) unsigned srf_lds_u32;
    const bool gathers = TPR == 16 || threadIdx.x < 16 * TPR;   // 32-channel rows: 128 threads (two waves) carry a group of 16 rows
    long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (ABL == 5) t0 = __builtin_amdgcn_s_memtime();
    // ---- LDS / memory instructions only.  In-kernel stamps (the ping-pong experiment, tools/micro/spconv_gsq_experiment.patch): a wave
    // issues NO vector instruction while the partner wave of its SIMD -- here: the CU's other workgroup -- streams f32 MFMAs, at any
    // s_setprio; a step whose first instruction is an address add therefore stands still until the partner's MFMA block is over, and
    // the two workgroups take turns instead of overlapping.  Every address below comes out of registers prepared inside this
    // wave's own previous MFMA block ----
    f32x4 af[2][GP][2];
#pragma unroll
    for (int gp = 0; gp < GP; ++gp) {
        af[0][gp][0] = *reinterpret_cast<const srf_lds_f32x4 *>(G.f0 + gp * 2048);
        af[0][gp][1] = *reinterpret_cast<const srf_lds_f32x4 *>(G.f1 + gp * 2048);
    }
    unsigned ro[GP], sl4[GP];
#pragma unroll
    for (int gp = 0; gp < GP; ++gp) {
        ro[gp] = *reinterpret_cast<srf_lds_u32 *>(G.pin_addr + gp * 64);     // list entries of step i + 3
        sl4[gp] = *reinterpret_cast<srf_lds_u32 *>(G.slot_addr + gp * 16);   // slots of step i + 1
    }
    const unsigned inf = *reinterpret_cast<const __attribute__((address_space(3))) unsigned char *>(G.info_addr);  // ... and its offset flag
    if (ABL != 2 && gathers) {
        // A[i + 1]: rows gathered during step i - 1, into the buffer the previous step read
#pragma unroll
        for (int gp = 0; gp < GP; ++gp)
#pragma unroll
            for (int j = 0; j < NA; ++j)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<srf_lds_float *>(G.st[jj] + gp * 2048 + j * (2 * CHS * 4)) = ra[gp][j][jj];
        // rows of step i + 2 (dummy steps behind the last one: out-of-range offsets, zeros)
#pragma unroll
        for (int gp = 0; gp < GP; ++gp)
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)G.gaddr[gp] + j * 256, 0, 0);
                ra[gp][j] = *reinterpret_cast<f32x4 *>(&v);
            }
    }
    if (LAST) {   // B of the next offset into the other register set, a whole step ahead (through a descriptor: not an invariant
                  // load the compiler may sink to its first use behind the barrier)
        const int so = kn * (NCH * NB * 4096);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int g = 0; g < NB; ++g) {
                auto v = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)(L.boff + g * 1024), so + c * (NB * 4096), 0);
                bn[c][g] = *reinterpret_cast<f32x4 *>(&v);
            }
    }
    unsigned oaddr_n[GP][4];
    if (ABL == 5) t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    // ---- the MFMA block: fragments of the next chunk requested a chunk ahead; the vector instructions of the step in its last chunk ----
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) {
#pragma unroll
            for (int gp = 0; gp < GP; ++gp) {
                af[(c + 1) & 1][gp][0] = *reinterpret_cast<const srf_lds_f32x4 *>(G.f0 + gp * 2048 + (c + 1) * (CHS * 4));
                af[(c + 1) & 1][gp][1] = *reinterpret_cast<const srf_lds_f32x4 *>(G.f1 + gp * 2048 + (c + 1) * (CHS * 4));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int gp = 0; gp < GP; ++gp) {
                const float a = af[c & 1][gp][s >> 2][s & 3];
                if (ABL == 1) {   // keep the operands alive, issue nothing
                    asm volatile("" ::"v"(a), "v"(bc[c][s >> 2][s & 3]), "v"(bc[c][NB - 2 + (s >> 2)][s & 3]));
                    continue;
                }
                acc[gp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bc[c][s >> 2][s & 3], acc[gp][0], 0, 0, 0);
                if (NT == 2) acc[gp][NT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bc[c][NB - 2 + (s >> 2)][s & 3], acc[gp][NT - 1], 0, 0, 0);
            }
            if (c == NCH - 1 && s == 3) {
                // (all fragment reads of the step are issued: the address registers are free to move on)
                __builtin_amdgcn_sched_barrier(0);
                buf ^= 1;
                const int d = buf ? (int)ABUF : -(int)ABUF;   // the A buffers swap roles
                unsigned inf_here = inf;   // (opaque copies: the arithmetic on the loaded values must not be hoisted to the loads)
                asm volatile("" : "+v"(inf_here));
                info_next = __builtin_amdgcn_readfirstlane((int)inf_here);
                G.f0 += d;
                G.f1 += d;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) G.st[jj] -= d;
#pragma unroll
                for (int gp = 0; gp < GP; ++gp) {
                    unsigned ro_here = ro[gp], sl_here = sl4[gp];
                    asm volatile("" : "+v"(ro_here), "+v"(sl_here));
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) oaddr_n[gp][jj] = ((sl_here >> (8 * jj)) & 255u) * (unsigned)(OS * 4) + L.colbase;
                    G.gaddr[gp] = ro_here + L.goff;
                    asm volatile("" : "+v"(G.gaddr[gp]), "+v"(oaddr_n[gp][0]), "+v"(oaddr_n[gp][1]), "+v"(oaddr_n[gp][2]), "+v"(oaddr_n[gp][3]));
                }
                G.pin_addr += RS * 4;
                G.slot_addr += RS;
                G.info_addr += 1;
                asm volatile("" : "+v"(G.f0), "+v"(G.f1), "+v"(G.st[0]), "+v"(G.st[1]), "+v"(G.st[2]), "+v"(G.st[3]), "+v"(G.pin_addr),
                             "+v"(G.slot_addr), "+v"(G.info_addr));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (ABL == 5) t2 = __builtin_amdgcn_s_memtime();
    if (ABL == 3) {
#pragma unroll
        for (int gp = 0; gp < GP; ++gp)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) G.oaddr[gp][jj] = oaddr_n[gp][jj];
        __syncthreads();
        return;
    }
    // ---- LDS only again: accumulators of step i back into the tile, those of step i + 1 out of it (in-order LDS: shared rows are
    // read after they were written); plain ds_read_b32 into the accumulator registers themselves (the compiler pairs them as
    // ds_read2 and then shuffles registers behind a wait), awaited before the barrier together with the stores ----
#pragma unroll
    for (int gp = 0; gp < GP; ++gp)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int cb = 0; cb < NT; ++cb) *reinterpret_cast<srf_lds_float *>(G.oaddr[gp][jj] + cb * 64) = acc[gp][cb][jj];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int gp = 0; gp < GP; ++gp)
#pragma unroll
        for (int cb = 0; cb < NT; ++cb)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float t;
                if (cb == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(oaddr_n[gp][jj]) : "memory");
                else asm volatile("ds_read_b32 %0, %1 offset:64" : "=v"(t) : "v"(oaddr_n[gp][jj]) : "memory");
                acc[gp][cb][jj] = t;
            }
#pragma unroll
    for (int gp = 0; gp < GP; ++gp)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) G.oaddr[gp][jj] = oaddr_n[gp][jj];   // (a renaming: the copies, if any, land in the next MFMA block)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ABL == 5) t3 = __builtin_amdgcn_s_memtime();
    if (ABL != 4) __syncthreads();
#ifdef SRF_DEV
    if (ABL == 5) {
        const long long t4 = __builtin_amdgcn_s_memtime();
        stamp[0] += t1 - t0;
        stamp[1] += t2 - t1;
        stamp[2] += t3 - t2;
        stamp[3] += t4 - t3;
        stamp[4] += 1;
    }
#endif
}

template <int NCH, int COUT, int ABL = 0, int GP = 1>
__global__ __launch_bounds__(256, 2) void srf_spconv_gsp_k(const float *__restrict__ in, int A_in, const float *__restrict__ Wg, int K,
                                                         const int *__restrict__ nbr, int nbr_stride, int A_out,
                                                         const float *__restrict__ alpha, const float *__restrict__ beta,
                                                         const float *__restrict__ residual, int relu,
                                                         float *__restrict__ out, const int *__restrict__ rows_dev,
                                                         const int *__restrict__ tiles)
{
    constexpr int NT = COUT / 64, NB = COUT / 32;
    constexpr int RS = 16 * GP;   // rows per step (srf_gsp_step)
    constexpr int NA = NCH >= 2 ? NCH / 2 : 1, TPR = NCH >= 2 ? 16 : 8, NKW = (SRF_KMAX + 3) / 4, CHS = RS * 32 + 8;
    constexpr int TMAX = COUT == 128 ? SRF_GS_TMAX : 128, OS = COUT + 4;   // (64 channels: 128 rows = the two ballot segments; 63 KB of LDS)
    constexpr int FL = ((TMAX * SRF_KMAX + SRF_KMAX * (RS - 1) + RS - 1) / RS) * RS + 2 * RS;   // every offset padded to whole steps, two dummy steps
    static_assert(COUT == 128 || COUT == 64, "column tiling of the waves");
    static_assert(TMAX <= 128 && TMAX < 255, "two ballot segments of 64 rows; slots are bytes");
    static_assert(NCH == 1 || (NCH & 1) == 0, "a gathered row is one quad (32 channels) or NCH / 2 quads per thread");
    __shared__ unsigned s_pin[FL];                          // flat step list: byte offset of the input row (padding: out of range)
    __shared__ __attribute__((aligned(4))) unsigned char s_pslot[FL];  // ... and the row's slot in the output tile
    __shared__ int s_cnt[SRF_KMAX];
    __shared__ int s_gstart[SRF_KMAX + 1];                  // first step of an offset; [KMAX] = steps of the sub-tile
    __shared__ int s_klist[SRF_KMAX + 2];                   // used offsets ascending; [KMAX] their count, [KMAX + 1] their bit mask
    __shared__ unsigned char s_sinfo[FL / RS + 4];          // per step: bit 0 = last step of its offset, bits 1.. = the next used offset
    __shared__ __attribute__((aligned(16))) float s_out[(TMAX + 1) * OS];
    __shared__ __attribute__((aligned(16))) float s_a[2 * NCH * CHS];

    const int A_cap = A_out;
    if (rows_dev) {
        const int live = *rows_dev;
        A_out = A_out < live ? A_out : live;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int range0, range1;
    if (tiles) {
        const int T = srf_gs_ranges(A_cap);
        if ((int)blockIdx.x >= T) return;
        const int t = srf_xcd_tile(blockIdx.x, T);
        range0 = tiles[t];
        range1 = tiles[t + 1];
        range1 = range1 < A_out ? range1 : A_out;
    } else {
        const int tm = srf_gs_tile_rows(A_out);
        const int n_tiles = (A_out + tm - 1) / tm;
        if ((int)blockIdx.x >= n_tiles) return;
        range0 = srf_xcd_tile(blockIdx.x, n_tiles) * tm;
        range1 = range0 + tm < A_out ? range0 + tm : A_out;
    }
    if (range1 <= range0) return;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)((long long)A_in * (32 * NCH) * 4), 0x00020000);
    long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tk0 = 0;
    if (ABL == 5) tk0 = __builtin_amdgcn_s_memtime();
    const int nsub = (range1 - range0 + TMAX - 1) / TMAX;
    const int TM = (((range1 - range0 + nsub - 1) / nsub) + 7) & ~7;
    for (int row0 = range0; row0 < range1; row0 += TM) {
    const int row_end = row0 + TM < range1 ? row0 + TM : range1;
    int zero = 0;   // opaque: keeps the prologue's address arithmetic from being hoisted above the sub-tile loop and spilled (see srf_spconv_gs_k)
    asm volatile("" : "+s"(zero));
    long long tp0 = 0, tl0 = 0, te0 = 0;
    if (ABL == 5) tp0 = __builtin_amdgcn_s_memtime();
    for (int e = tid; e < TM * OS / 4; e += 256) reinterpret_cast<f32x4 *>(s_out)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    int nv[NKW][2];
#pragma unroll
    for (int i = 0; i < NKW; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = wave + 4 * i, r = h * 64 + lane;
            nv[i][h] = (k < K && row0 + r < row_end) ? nbr[(size_t)(k + zero) * nbr_stride + row0 + r] : -1;
        }
    // pass 1: pairs per offset
#pragma unroll
    for (int i = 0; i < NKW; ++i) {
        const int k = wave + 4 * i;
        if (k >= SRF_KMAX) break;
        const int c = __popcll(__ballot(nv[i][0] >= 0)) + __popcll(__ballot(nv[i][1] >= 0));
        if (lane == 0) s_cnt[k] = c;
    }
    __syncthreads();
    if (tid < 64) {   // first step of every offset (exclusive prefix of the group counts), the used offsets in ascending order
        const int c = tid < SRF_KMAX ? s_cnt[tid] : 0;
        const int ng = (c + RS - 1) / RS;
        int x = ng;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const int y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (tid < SRF_KMAX) s_gstart[tid] = x - ng;
        if (tid == SRF_KMAX - 1) s_gstart[SRF_KMAX] = x;
        const unsigned long long m = __ballot(c > 0);
        if (c > 0) s_klist[__popcll(m & ((1ull << tid) - 1ull))] = tid;
        if (tid == 0) {
            s_klist[SRF_KMAX] = __popcll(m);
            s_klist[SRF_KMAX + 1] = (int)(unsigned)m;
        }
    }
    __syncthreads();
    const int S = __builtin_amdgcn_readfirstlane(s_gstart[SRF_KMAX]);
    const unsigned used = (unsigned)s_klist[SRF_KMAX + 1];
    // pass 2: compaction into the flat list
#pragma unroll
    for (int i = 0; i < NKW; ++i) {
        const int k = wave + 4 * i;
        if (k >= SRF_KMAX) break;
        const int g0 = s_gstart[k + zero] * RS;
        int base = 0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int v = nv[i][h];
            const unsigned long long m = __ballot(v >= 0);
            if (v >= 0) {
                const int pos = g0 + base + __popcll(m & ((1ull << lane) - 1ull));
                s_pin[pos] = (unsigned)v * (unsigned)(32 * NCH * 4);
                s_pslot[pos] = (unsigned char)(h * 64 + lane);
            }
            base += __popcll(m);
        }
        const int ng = (base + RS - 1) / RS;
        const int pad = ng * RS - base;   // padding of the offset's last step (< RS <= 32 entries): zeros into the spare row
        if (lane < pad) {
            s_pin[g0 + base + lane] = 0x80000000u;
            s_pslot[g0 + base + lane] = (unsigned char)TMAX;
        }
        if (lane < ng) {
            const unsigned later = k + 1 < 32 ? (used >> (k + 1)) : 0u;
            const int kn = later ? k + 1 + __builtin_ctz(later) : k;   // the next used offset (the last one names itself)
            s_sinfo[g0 / RS + lane] = (unsigned char)(lane == ng - 1 ? ((kn << 1) | 1) : 0);
        }
    }
    if (tid < 2 * RS) {   // two dummy steps behind the last one
        s_pin[S * RS + tid] = 0x80000000u;
        s_pslot[S * RS + tid] = (unsigned char)TMAX;
    }
    __syncthreads();
    const int ntap = __builtin_amdgcn_readfirstlane(s_klist[SRF_KMAX]);

    // per-thread constants of a step
    SrfGspLane L;
    {
        const int ar = lane & 15, aj = lane >> 4, a_swz = (ar >> 1) & 7;
        const unsigned a0 = srf_lds_addr(s_a) + (unsigned)zero;
        L.fo0 = a0 + (unsigned)(ar * 32 + (((aj << 1) ^ a_swz) << 2)) * 4u;
        L.fo1 = a0 + (unsigned)(ar * 32 + ((((aj << 1) + 1) ^ a_swz) << 2)) * 4u;
        L.colbase = srf_lds_addr(s_out) + (unsigned)((COUT == 128 ? wave * 32 : wave * 16) + ar + zero) * 4u;
        // gather: thread = (row tid / 16, quad tid % 16 of the row, and for Cin = 128 the quad 16 further = two chunks further)
        const int r = (tid / TPR) & 15, qq = tid % TPR, ch = qq >> 3, q = qq & 7, swz = (r >> 1) & 7;
        L.goff = (unsigned)(qq * 16 + zero);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) L.sto[jj] = a0 + (unsigned)(ch * CHS + r * 32 + (q & 3) + (((jj * 2 + (q >> 2)) ^ swz) << 2)) * 4u;
        const int wc = COUT == 128 ? wave : (wave >> 1), g0 = COUT == 128 ? 0 : 2 * (wave & 1);
        L.boff = (unsigned)(lane * 16 + wc * 4096 + g0 * 1024 + zero);
    }
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Wg), 0, K * NCH * NB * 4096, 0x00020000);
    f32x4 b0[NCH][NB], b1[NCH][NB], ra[GP][NA], acc[GP][NT];
    SrfGspRegs<GP> G;
    int buf = 0;
    {
        constexpr unsigned ABUF = NCH * CHS * 4;
        const int aj = lane >> 4;
        G.f0 = L.fo0;
        G.f1 = L.fo1;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) G.st[jj] = L.sto[jj] + ABUF;
#pragma unroll
        for (int gp = 0; gp < GP; ++gp) {
            const unsigned sl0 = *reinterpret_cast<const unsigned *>(s_pslot + gp * 16 + aj * 4);   // slots of step 0
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                G.oaddr[gp][jj] = ((sl0 >> (8 * jj)) & 255u) * (unsigned)(OS * 4) + L.colbase;
#pragma unroll
                for (int cb = 0; cb < NT; ++cb) acc[gp][cb][jj] = 0.0f;   // the tile was just zeroed
            }
            G.gaddr[gp] = s_pin[2 * RS + gp * 16 + ((tid / TPR) & 15)] + L.goff;        // step 0 requests the rows of step 2
        }
        G.pin_addr = srf_lds_addr(s_pin) + (unsigned)(3 * RS + ((tid / TPR) & 15)) * 4u;   // ... and reads the list entries of step 3
        G.slot_addr = srf_lds_addr(s_pslot) + (unsigned)(RS + aj * 4);                  // ... and the slots of step 1
        G.info_addr = srf_lds_addr(s_sinfo) + 1u + (unsigned)zero;                      // ... and its offset flag
    }
    if (ntap > 0) {
        const int k0 = s_klist[0];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int g = 0; g < NB; ++g) {
                auto v = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)(L.boff + g * 1024), (k0 * NCH + c) * (NB * 4096), 0);
                b0[c][g] = *reinterpret_cast<f32x4 *>(&v);
            }
#pragma unroll
        for (int gp = 0; gp < GP; ++gp) {
            const unsigned ro = s_pin[gp * 16 + ((tid / TPR) & 15)];
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ro + L.goff + j * 256), 0, 0);
                ra[gp][j] = *reinterpret_cast<f32x4 *>(&v);
            }
        }
        if (TPR == 16 || tid < 16 * TPR) {
#pragma unroll
            for (int gp = 0; gp < GP; ++gp)
#pragma unroll
                for (int j = 0; j < NA; ++j)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<srf_lds_float *>(L.sto[jj] + gp * 2048 + j * (2 * CHS * 4)) = ra[gp][j][jj];
        }
#pragma unroll
        for (int gp = 0; gp < GP; ++gp) {
            const unsigned ro = s_pin[RS + gp * 16 + ((tid / TPR) & 15)];
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ro + L.goff + j * 256), 0, 0);
                ra[gp][j] = *reinterpret_cast<f32x4 *>(&v);
            }
        }
    }
    __syncthreads();
    if (ABL == 5) {
        tl0 = __builtin_amdgcn_s_memtime();
        stamp[5] += tl0 - tp0;
    }
    // ONE flat loop over the steps: whether a step ends its offset (and which offset follows) is a byte of the step list, read in the
    // step's LDS section and turned into a scalar inside its MFMA block -- no vector instruction outside the MFMA blocks (a lookup
    // of the offset tables between two offsets used to cost three dependent LDS reads and their readfirstlanes, stalled like every
    // vector instruction while the partner workgroup multiplies)
    int info = S > 0 ? __builtin_amdgcn_readfirstlane((int)s_sinfo[0]) : 0;
    for (int i = 0; i < S;) {   // two offsets per trip: the B register sets swap roles without moves
        int info_next = 0;
        for (; !(info & 1); ++i) {
            srf_gsp_step<NCH, COUT, GP, false, ABL>(rs, wrs, 0, buf, L, G, b0, b1, ra, acc, stamp, info_next);
            info = info_next;
        }
        srf_gsp_step<NCH, COUT, GP, true, ABL>(rs, wrs, info >> 1, buf, L, G, b0, b1, ra, acc, stamp, info_next);
        info = info_next;
        if (++i >= S) break;
        for (; !(info & 1); ++i) {
            srf_gsp_step<NCH, COUT, GP, false, ABL>(rs, wrs, 0, buf, L, G, b1, b0, ra, acc, stamp, info_next);
            info = info_next;
        }
        srf_gsp_step<NCH, COUT, GP, true, ABL>(rs, wrs, info >> 1, buf, L, G, b1, b0, ra, acc, stamp, info_next);
        info = info_next;
        ++i;
    }
    // the last step's write-back went through raw LDS addresses: make the whole tile visible to the epilogue's plain reads
    __syncthreads();
    if (ABL == 5) {
        te0 = __builtin_amdgcn_s_memtime();
        stamp[6] += te0 - tl0;
    }

    // epilogue: every output row once, BN / residual / ReLU in registers, 512 B per row and store
    constexpr int CQ = COUT / 4;
    const int c4 = ((tid & (CQ - 1)) + zero) * 4;
    f32x4 al = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
    if (alpha) {
        al = *reinterpret_cast<const f32x4 *>(alpha + c4);
        be = *reinterpret_cast<const f32x4 *>(beta + c4);
    }
    for (int r = tid / CQ; r < TM; r += 256 / CQ) {
        const int row = row0 + r;
        if (row >= row_end) break;
        f32x4 v = *reinterpret_cast<const f32x4 *>(s_out + r * OS + c4);
        f32x4 rsd = {0.f, 0.f, 0.f, 0.f};
        if (residual) rsd = *reinterpret_cast<const f32x4 *>(residual + (size_t)row * COUT + c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = v[j];
            if (alpha) x = __fmaf_rn(x, al[j], be[j]);
            if (residual) x = __fadd_rn(x, rsd[j]);
            if (relu) x = x > 0.0f ? x : 0.0f;
            v[j] = x;
        }
        *reinterpret_cast<f32x4 *>(out + (size_t)row * COUT + c4) = v;
    }
    __syncthreads();
    if (ABL == 5) {
        stamp[7] += __builtin_amdgcn_s_memtime() - te0;
        stamp[4] += 1000000;   // sub-tiles in the millions digit
    }
    }
#ifdef SRF_DEV
    if (ABL == 5 && tid == 0 && blockIdx.x < 512) {
        long long *dst = srf_gsp_stamps + (size_t)blockIdx.x * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j] = stamp[j];
        dst[8] = __builtin_amdgcn_s_memtime() - tk0;
    }
#endif
}

// =====================================================================================================================
// srf_spconv_w32_k: the 32-output-channel layers (SubM 32 -> 32 x 4 and the strided 16 -> 32 of the nuScenes encoder,
// sparse_encoder_custom.py:109-140 through spconv's SubMConv3d / SparseConv3d) -- 2 GFLOP per frame that took 0.33 ms: the
// tile kernel above walks its 27 offsets as gather -> LDS -> barrier -> 16 MFMAs with one step of lookahead, i.e. at the
// latency of a gather per step (2.4 us) whatever the arithmetic.  Here nothing is shared but the weights:
//   * a workgroup = 8 waves x 32 output rows; all 27 x Cin x 32 weights sit in LDS (110 KB at Cin = 32), copied once;
//   * every wave gathers ITS 32 rows straight into the MFMA's A layout -- lane (row, half) loads half of the neighbour's
//     channels with Cin / 8 buffer_load_dwordx4 (a missing neighbour is an out-of-range offset and reads as zero), then
//     v_permlane32_swap hands the odd channels to lanes 32-63 and the even ones to lanes 0-31 (the f32 32x32x2 MFMA takes
//     channel 2 s from lanes 0-31 and 2 s + 1 from lanes 32-63) -- three offsets in flight, no barrier, no LDS round trip;
//   * offsets none of the wave's rows has are skipped (one ballot per offset).
// Every output is the same chain as before (offset ascending, channel ascending, zero terms added as +0): bit-identical to
// the tile kernel and the oracle.
// =====================================================================================================================
// ---------------------------------------------------------------------------------------------------------------------
// Row order for the 32-channel layers (srf_spconv_w32_k): a wave multiplies EVERY offset that any of its 32 rows has -- 0.77 of the
// 27 offsets on the level-2 rulebook of a nuScenes sweep, 0.54 on the strided 16 -> 32 one, where 0.28 / 0.05 exist.  Rows with the same
// set of offsets put side by side share their zeros: sorted by an 11-bit key of the offset mask (has any z - 1 neighbour, has any z + 1
// neighbour, the nine offsets of the row's own z plane) inside windows of 1024 rows a 32-row group executes 0.47 / 0.28 of the
// offsets.  One launch per rulebook (shared by the SubM layers of a level): a workgroup counting-sorts its window in LDS and writes
//   plan[0 .. A_pad)                      order[pos] = row at sorted position pos (-1: padding)
//   plan[A_pad + k A_pad + pos]           the rulebook entry nbr[k][order[pos]] (-1: none)
// Which rows share a wave changes nothing in any output (every row is its own fma chain): bit-identical.
// ---------------------------------------------------------------------------------------------------------------------
#define SRF_ORD_WIN 1024
#define SRF_ORD_BINS 2048
__host__ __device__ static inline int srf_ord_pad(int A) { return ((A + SRF_ORD_WIN - 1) / SRF_ORD_WIN) * SRF_ORD_WIN; }

__global__ __launch_bounds__(256) void srf_spconv_order_k(const int *__restrict__ nbr, int nbr_stride, int K, int A, int A_pad,
                                                         const int *__restrict__ rows_dev, int *__restrict__ plan)
{
    __shared__ int s_hist[SRF_ORD_BINS];
    __shared__ int s_part[256];
    __shared__ int s_row[SRF_ORD_WIN];
    if (rows_dev) {
        const int live = *rows_dev;
        A = A < live ? A : live;
    }
    const int tid = threadIdx.x, w0 = blockIdx.x * SRF_ORD_WIN;
    for (int b = tid; b < SRF_ORD_BINS; b += 256) s_hist[b] = 0;
    __syncthreads();
    int key[4], rank[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = w0 + i * 256 + tid;
        const int rc = row < A ? row : (A > 0 ? A - 1 : 0);
        unsigned m = 0;
        int v[SRF_KMAX];
#pragma unroll
        for (int k = 0; k < SRF_KMAX; ++k) v[k] = nbr[(size_t)(k < K ? k : K - 1) * nbr_stride + rc];
#pragma unroll
        for (int k = 0; k < SRF_KMAX; ++k) m |= (k < K && v[k] >= 0 ? 1u : 0u) << k;
        const unsigned p0 = m & 0x1ffu, p1 = (m >> 9) & 0x1ffu, p2 = (m >> 18) & 0x1ffu;
        unsigned kk = K == SRF_KMAX ? ((((p0 != 0 ? 1u : 0u) | (p2 != 0 ? 2u : 0u)) << 9) | p1) : (m & (SRF_ORD_BINS - 1));
        key[i] = row < A ? (int)kk : SRF_ORD_BINS - 1;   // rows past the end sort last (their order entry becomes -1)
        rank[i] = atomicAdd(&s_hist[key[i]], 1);
    }
    __syncthreads();
    // exclusive prefix of the 2048 bins: 8 consecutive bins per thread, then the 256 partial sums
    int loc[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        loc[j] = sum;
        sum += s_hist[tid * 8 + j];
    }
    s_part[tid] = sum;
    __syncthreads();
    int base = 0;
    for (int t = 0; t < tid; ++t) base += s_part[t];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) s_hist[tid * 8 + j] = base + loc[j];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) s_row[s_hist[key[i]] + rank[i]] = i * 256 + tid;   // (ranks inside a bin: arrival order, any order is right)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = i * 256 + tid, pos = w0 + p;
        if (pos >= A_pad) continue;
        const int row = w0 + s_row[p];
        const bool live = row < A;
        plan[pos] = live ? row : -1;
        const int rc = live ? row : (A > 0 ? A - 1 : 0);
        int v[SRF_KMAX];
#pragma unroll
        for (int k = 0; k < SRF_KMAX; ++k) v[k] = nbr[(size_t)(k < K ? k : K - 1) * nbr_stride + rc];
#pragma unroll
        for (int k = 0; k < SRF_KMAX; ++k)
            if (k < K) plan[(size_t)A_pad + (size_t)k * A_pad + pos] = live ? v[k] : -1;
    }
}

extern "C" size_t srf_spconv_order_ints(int A_out, int K)
{
    if (A_out <= 0 || K <= 0 || K > SRF_KMAX) return 0;
    return (size_t)srf_ord_pad(A_out) * (size_t)(1 + K);
}

extern "C" int srf_spconv_order_build(const int *nbr, int nbr_stride, int K, int A_out, const int *rows_dev, int *plan, srf_stream_t stream)
{
    if (A_out < 0 || K <= 0 || K > SRF_KMAX || nbr_stride < A_out) return SRF_EINVAL;
    if (A_out == 0) return SRF_OK;
    if (!nbr || !plan) return SRF_EINVAL;
    const int A_pad = srf_ord_pad(A_out);
    hipLaunchKernelGGL(srf_spconv_order_k, dim3(A_pad / SRF_ORD_WIN), dim3(256), 0, (hipStream_t)stream, nbr, nbr_stride, K, A_out, A_pad,
                       rows_dev, plan);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

#ifndef SRF_W32_DEPTH
#define SRF_W32_DEPTH 6
#endif
template <int CIN>
__global__ __launch_bounds__(512) void srf_spconv_w32_k(const float *__restrict__ in, int A_in, const float *__restrict__ Wp,
                                                       const int *__restrict__ nbr, int nbr_stride, int A_out,
                                                       const float *__restrict__ alpha, const float *__restrict__ beta,
                                                       const float *__restrict__ residual, int relu, float *__restrict__ out,
                                                       const int *__restrict__ rows_dev, const int *__restrict__ plan)
{
    constexpr int K = SRF_KMAX, H = CIN / 2, NV = H / 4, NB = CIN / 8, TM = 256;
    extern __shared__ __attribute__((aligned(16))) f32x4 s_w32[];   // [K][NB][2][32]
    const int A_cap = A_out;
    if (rows_dev) {
        const int live = *rows_dev;
        A_out = A_out < live ? A_out : live;
    }
    const int n_tiles = plan ? ((A_out + SRF_ORD_WIN - 1) / SRF_ORD_WIN) * 4 : (A_out + TM - 1) / TM;
    if ((int)blockIdx.x >= n_tiles) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kh = lane >> 5;
    // with a plan (srf_spconv_order_build) the wave's 32 rows are a group of the sorted order: `row` is a POSITION, the rulebook
    // entries come from the plan's sorted copy, the output row is order[position].  The 32 groups of a 1024-row window go to its
    // four workgroups in turn (group 4 w + b to wave w of workgroup b): every workgroup, and every pair of waves that shares a SIMD,
    // gets light and heavy groups alike.
    int row0 = 0, row;
    if (plan) {
        const int A_pad = srf_ord_pad(A_cap);
        const int win = blockIdx.x >> 2, b = blockIdx.x & 3;
        row = win * SRF_ORD_WIN + (4 * wave + b) * 32 + r;
        nbr = plan + A_pad;
        nbr_stride = A_pad;
    } else {
        row0 = srf_xcd_tile(blockIdx.x, n_tiles) * TM;
        row = row0 + wave * 32 + r;
    }
    const f32x4 *Wp4 = reinterpret_cast<const f32x4 *>(Wp);
    for (int i = tid; i < K * NB * 64; i += 512) s_w32[i] = Wp4[i];
    // all K rulebook entries of the row in flight at once (a load under `row < A_out` is a load under a branch: hipcc then waits for
    // each one before the next -- 27 round trips in a row): rows past the end read the last row's entries and drop them
    int idx[K];
    unsigned anym = 0;
    const int row_ld = (plan || row < A_out) ? row : A_out - 1;     // (every position of a plan exists: padding holds -1)
    const bool row_ok = plan || row < A_out;
    const int ord = plan ? plan[row] : row;                          // the output row behind this lane's row / position (-1: none)
#pragma unroll
    for (int k = 0; k < K; ++k) idx[k] = nbr[(size_t)k * nbr_stride + row_ld];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        idx[k] = row_ok ? idx[k] : -1;
        anym |= (__ballot(idx[k] >= 0) != 0ull ? 1u : 0u) << k;
    }
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)((long long)A_in * CIN * 4), 0x00020000);
    // D offsets in flight.  The loads are NOT under the `anym` test: a load inside a scalar branch makes hipcc's wait-count pass
    // give up counting (it emitted s_waitcnt vmcnt(0) in front of every offset's MFMAs, 109 of them: the "three offsets in flight"
    // of round 3 were drained at every step and each offset paid a full gather latency, 1.4 us against 0.43 us of MFMAs).  An
    // offset no row of the wave has is 64 out-of-range lanes: the descriptor returns zeros without touching memory.
    constexpr int D = SRF_W32_DEPTH;
    f32x4 a[D][NV];
#define W32_LOAD(SET, KK)                                                                                    \
    {                                                                                                        \
        const unsigned vo_ = idx[KK] >= 0 ? (unsigned)(idx[KK] * (CIN * 4) + kh * (H * 4)) : 0x80000000u;    \
        _Pragma("unroll") for (int j_ = 0; j_ < NV; ++j_) {                                                  \
            auto v_ = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(vo_ + j_ * 16), 0, 0);                      \
            a[SET][j_] = *reinterpret_cast<f32x4 *>(&v_);                                                    \
        }                                                                                                    \
    }
#pragma unroll
    for (int k = 0; k < D - 1; ++k) W32_LOAD(k, k)
    __syncthreads();   // the weights are in LDS
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k + D - 1 < K) W32_LOAD((k + D - 1) % D, k + D - 1)
        if (anym & (1u << k)) {
            f32x4 b[NB];
#pragma unroll
            for (int sg = 0; sg < NB; ++sg) b[sg] = s_w32[((k * NB + sg) * 2 + kh) * 32 + r];
            // lanes 0-31 hold the first half of the row's channels, lanes 32-63 the second: swap so that lanes 0-31 end up with
            // the even channels (first half in x / z, second half in y / w), lanes 32-63 with the odd ones
            float e0[NV], e1[NV], o0[NV], o1[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const f32x4 v = a[k % D][j];
                auto p0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[1]), false, false);
                auto p1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2]), __float_as_uint(v[3]), false, false);
                e0[j] = __uint_as_float(p0[0]);   // channel 4 j + kh
                o0[j] = __uint_as_float(p0[1]);   // channel H + 4 j + kh
                e1[j] = __uint_as_float(p1[0]);   // channel 4 j + 2 + kh
                o1[j] = __uint_as_float(p1[1]);   // channel H + 4 j + 2 + kh
            }
            // steps in channel order: s = 2 j (channels 4 j, 4 j + 1), 2 j + 1; then the second half from step H / 2 on
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e0[j], b[(2 * j) >> 2][(2 * j) & 3], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e1[j], b[(2 * j + 1) >> 2][(2 * j + 1) & 3], acc, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(o0[j], b[(H / 2 + 2 * j) >> 2][(H / 2 + 2 * j) & 3], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(o1[j], b[(H / 2 + 2 * j + 1) >> 2][(H / 2 + 2 * j + 1) & 3], acc, 0, 0, 0);
            }
        }
    }
#undef W32_LOAD
    const float al = alpha ? alpha[r] : 1.0f;
    const float be = alpha ? beta[r] : 0.0f;
    // the 16 residual values in flight together (one load per row under its own `orow < A_out` branch was 16 round trips in a row)
    int orow[16];   // accumulator register j holds row (j & 3) + 8 (j >> 2) + 4 kh of the wave's 32: its output row is lane that-index's `ord`
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int o = __shfl(ord, (j & 3) + 8 * (j >> 2) + 4 * kh);
        orow[j] = (o >= 0 && o < A_out) ? o : -1;
    }
    float res[16];
    if (residual) {
#pragma unroll
        for (int j = 0; j < 16; ++j) res[j] = residual[(size_t)(orow[j] >= 0 ? orow[j] : 0) * 32 + r];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float v = acc[j];
        if (alpha) v = __fmaf_rn(v, al, be);
        if (residual) v = __fadd_rn(v, res[j]);
        if (relu) v = v > 0.0f ? v : 0.0f;
        if (orow[j] >= 0) out[(size_t)orow[j] * 32 + r] = v;
    }
}

extern "C" int srf_spconv_fwd_packed(const float *in, int A_in, int Cin, const float *W_packed, int K, const int *nbr,
                                     int nbr_stride, int A_out, int Cout, const float *alpha, const float *beta,
                                     const float *residual, int relu, float *out, const int *rows_dev, const int *tiles,
                                     srf_stream_t stream)
{
    if (A_in < 0 || A_out < 0 || Cin <= 0 || Cin > 512 || K <= 0 || K > SRF_KMAX || nbr_stride < A_out) return SRF_EINVAL;
    if ((alpha == nullptr) != (beta == nullptr)) return SRF_EINVAL;
    if (A_out == 0) return SRF_OK;
    if (!in || !W_packed || !nbr || !out) return SRF_EINVAL;
    if ((Cin & 3) || A_in == 0) return SRF_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
#define SRF_ARGS in, Cin, W_packed, K, nbr, nbr_stride, A_out, alpha, beta, residual, relu, out, rows_dev
    switch (Cout) {
    case 32:
        // the weights of these shapes are packed for srf_spconv_w32_k alone (srf_spconv_pack_weights picks the layout from (K, Cin, Cout)):
        // an input beyond its 32-bit descriptor range (>= 16 M rows) must not fall through to a kernel that reads the tile layout
        if (srf_w32_layout(K, Cin, Cout) && (long long)A_in * Cin * 4 >= (1ll << 31)) return SRF_EUNSUPPORTED;
        if (srf_w32_layout(K, Cin, Cout)) {
            int dev = 0;
            SRF_HIP_TRY(hipGetDevice(&dev));
            static bool attr_set[64] = {false};
            if (dev < 0 || dev >= 64) return SRF_EUNSUPPORTED;
            if (!attr_set[dev]) {
                SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_spconv_w32_k<32>, hipFuncAttributeMaxDynamicSharedMemorySize, SRF_KMAX * 32 * 32 * 4));
                SRF_HIP_TRY(hipFuncSetAttribute((const void *)srf_spconv_w32_k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, SRF_KMAX * 16 * 32 * 4));
                attr_set[dev] = true;
            }
            // `tiles` of a 32-channel layer = the plan of srf_spconv_order_build (NULL: rows in their own order)
            const unsigned grid32 = tiles ? (unsigned)(srf_ord_pad(A_out) / 256) : (unsigned)srf_ceil_div(A_out, 256);
            if (Cin == 32)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_w32_k<32>), dim3(grid32), dim3(512), SRF_KMAX * 32 * 32 * 4, st, in,
                                   A_in, W_packed, nbr, nbr_stride, A_out, alpha, beta, residual, relu, out, rows_dev, tiles);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_w32_k<16>), dim3(grid32), dim3(512), SRF_KMAX * 16 * 32 * 4, st, in,
                                   A_in, W_packed, nbr, nbr_stride, A_out, alpha, beta, residual, relu, out, rows_dev, tiles);
            break;
        }
        hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_packed_k<32, 128, 4, 1>), dim3(srf_ceil_div(A_out, 128)), dim3(256),
                           0, st, SRF_ARGS);
        break;
    case 64:
        if (srf_gs_layout(Cin, Cout)) {
            const dim3 grid(tiles ? srf_gs_ranges(A_out) : SRF_GS_SLOTS * srf_gs_rounds(A_out));
#ifdef SRF_DEV
            if (Cin == 64 && srf_gsp_enabled() && (long long)A_in * Cin * 4 < (1ll << 31) && (getenv("SRF_GSP_ABL") || getenv("SRF_GSP_GP2"))) {
                const bool stamp = getenv("SRF_GSP_ABL") && atoi(getenv("SRF_GSP_ABL")) == 5, gp2 = getenv("SRF_GSP_GP2") != nullptr;
#define SRF_GSP_DEV64(A, G) hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_gsp_k<2, 64, A, G>), grid, dim3(256), 0, st, in, A_in, W_packed, K, nbr, nbr_stride, \
                                               A_out, alpha, beta, residual, relu, out, rows_dev, tiles)
                if (stamp && gp2) SRF_GSP_DEV64(5, 2);
                else if (stamp) SRF_GSP_DEV64(5, 1);
                else if (gp2) SRF_GSP_DEV64(0, 2);
                else SRF_GSP_DEV64(0, 1);
#undef SRF_GSP_DEV64
                break;
            }
#endif
            if (Cin == 32) {   // only the pipelined kernel has a one-chunk form (the weights are packed for it: srf_gs_layout)
                if ((long long)A_in * Cin * 4 >= (1ll << 31)) return SRF_EUNSUPPORTED;
                hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_gsp_k<1, 64>), grid, dim3(256), 0, st, in, A_in, W_packed, K, nbr, nbr_stride, A_out,
                                   alpha, beta, residual, relu, out, rows_dev, tiles);
            } else if (srf_gsp_enabled() && (long long)A_in * Cin * 4 < (1ll << 31))
                hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_gsp_k<2, 64>), grid, dim3(256), 0, st, in, A_in, W_packed, K, nbr, nbr_stride, A_out,
                                   alpha, beta, residual, relu, out, rows_dev, tiles);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_gs_k<2, 64>), grid, dim3(256), 0, st, in, W_packed, K, nbr, nbr_stride, A_out,
                                   alpha, beta, residual, relu, out, rows_dev, tiles);
            break;
        }
        if (srf_direct_layout(Cin, Cout)) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_direct64_k<2>), dim3(srf_ceil_div(A_out, 64)), dim3(256), 0, st, in, W_packed,
                               K, nbr, nbr_stride, A_out, alpha, beta, residual, relu, out, rows_dev);
            break;
        }
        hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_packed_k<64, 64, 2, 2>), dim3(srf_ceil_div(A_out, 64)), dim3(256), 0,
                           st, SRF_ARGS);
        break;
    case 128: {
        if (srf_gs_layout(Cin, Cout)) {
            // >= the tiles of any live row count <= A_out / the ranges srf_spconv_tiles_build cut for this capacity
            const dim3 grid(tiles ? srf_gs_ranges(A_out) : SRF_GS_SLOTS * srf_gs_rounds(A_out));
            if (srf_gsp_enabled() && (long long)A_in * Cin * 4 < (1ll << 31)) {
#ifdef SRF_DEV
                if (Cin == 128) {   // SRF_GSP_ABL = ablation (wrong outputs), SRF_GSP_PADLDS = extra LDS bytes (one workgroup per CU from 4000 on)
                    const char *e = getenv("SRF_GSP_ABL"), *pl = getenv("SRF_GSP_PADLDS");
                    const int abl = e ? atoi(e) : 0, pad = pl ? atoi(pl) : 0;
#define SRF_GSP_DEV(A) hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_gsp_k<4, 128, A>), grid, dim3(256), pad, st, in, A_in, W_packed, K, nbr, nbr_stride, \
                                          A_out, alpha, beta, residual, relu, out, rows_dev, tiles)
                    if (abl == 1) SRF_GSP_DEV(1);
                    else if (abl == 2) SRF_GSP_DEV(2);
                    else if (abl == 3) SRF_GSP_DEV(3);
                    else if (abl == 4) SRF_GSP_DEV(4);
                    else if (abl == 5) SRF_GSP_DEV(5);
                    else SRF_GSP_DEV(0);
#undef SRF_GSP_DEV
                    break;
                }
#endif
#ifdef SRF_DEV
                if (Cin == 64 && getenv("SRF_GSP_ABL") && atoi(getenv("SRF_GSP_ABL")) == 5) {
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_gsp_k<2, 128, 5>), grid, dim3(256), 0, st, in, A_in, W_packed, K, nbr, nbr_stride,
                                       A_out, alpha, beta, residual, relu, out, rows_dev, tiles);
                    break;
                }
#endif
                if (Cin == 128)
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_gsp_k<4, 128>), grid, dim3(256), 0, st, in, A_in, W_packed, K, nbr, nbr_stride,
                                       A_out, alpha, beta, residual, relu, out, rows_dev, tiles);
                else
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_gsp_k<2, 128>), grid, dim3(256), 0, st, in, A_in, W_packed, K, nbr, nbr_stride,
                                       A_out, alpha, beta, residual, relu, out, rows_dev, tiles);
                break;
            }
            if (Cin == 128)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_gs_k<4, 128>), grid, dim3(256), 0, st, in, W_packed, K, nbr, nbr_stride,
                                   A_out, alpha, beta, residual, relu, out, rows_dev, tiles);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_gs_k<2, 128>), grid, dim3(256), 0, st, in, W_packed, K, nbr, nbr_stride,
                                   A_out, alpha, beta, residual, relu, out, rows_dev, tiles);
            break;
        }
        if (srf_direct_layout(Cin, Cout)) {
#define SRF_DARGS in, W_packed, K, nbr, nbr_stride, A_out, alpha, beta, residual, relu, out, rows_dev
            if (Cin == 128)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_direct_k<32, 4, 2>), dim3(srf_ceil_div(A_out, 32)), dim3(256), 0, st,
                                   SRF_DARGS);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_direct_k<32, 2, 2>), dim3(srf_ceil_div(A_out, 32)), dim3(256), 0, st,
                                   SRF_DARGS);
#undef SRF_DARGS
            break;
        }
        // 64-row tiles halve the W-slab traffic, 32-row tiles balance better when there are only a few tiles per CU:
        // pick the one whose busiest CU (tiles dealt evenly over 256 CUs) carries fewer 32-row units
        const int units64 = 2 * srf_ceil_div(srf_ceil_div(A_out, 64), 256), units32 = srf_ceil_div(srf_ceil_div(A_out, 32), 256);
        if (units32 < units64)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_packed_k<128, 32, 1, 4>), dim3(srf_ceil_div(A_out, 32)), dim3(256), 0,
                               st, SRF_ARGS);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_packed_k<128, 64, 2, 2>), dim3(srf_ceil_div(A_out, 64)), dim3(256), 0,
                               st, SRF_ARGS);
        break;
    }
    default:
        return SRF_EUNSUPPORTED;
    }
#undef SRF_ARGS
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// COUT = 16: four waves, each 16 rows x 16 cols on v_mfma_f32_16x16x4_f32
template <int TM>
__global__ __launch_bounds__(256) void srf_spconv_mfma16_k(const float *__restrict__ in, int Cin,
                                                         const float *__restrict__ W, int K,
                                                         const int *__restrict__ nbr, int nbr_stride, int A_out,
                                                         const float *__restrict__ alpha, const float *__restrict__ beta,
                                                         const float *__restrict__ residual, int relu,
                                                         float *__restrict__ out, const int *__restrict__ rows_dev)
{
    constexpr int COUT = 16;
    static_assert(TM == 64, "four 16-row wave tiles");
    __shared__ int s_nbr[SRF_KMAX * TM];
    __shared__ int s_any[SRF_KMAX];
    __shared__ float s_a[TM][SRF_KC + 1];
    __shared__ __attribute__((aligned(16))) float s_w[SRF_KC][COUT];

    if (rows_dev) {  // static-shape levels: rows >= *rows_dev are padding; their tiles do nothing
        const int live = *rows_dev;
        A_out = A_out < live ? A_out : live;
    }
    const int row0 = blockIdx.x * TM;
    if (row0 >= A_out) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    srf_load_nbr_tile<TM>(nbr, nbr_stride, K, row0, A_out, s_nbr, s_any);

    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    const int ar = wave * 16 + (lane & 15);
    const int kq = lane >> 4;
    for (int k = 0; k < K; ++k) {
        if (!s_any[k]) continue;
        const float *Wk = W + (size_t)k * Cin * COUT;
        for (int c0 = 0; c0 < Cin; c0 += SRF_KC) {
            __syncthreads();
            srf_stage_tile<TM, COUT>(in, Cin, Wk, c0, s_nbr + k * TM, s_a, s_w);
            __syncthreads();
            const int kc = (Cin - c0) < SRF_KC ? (Cin - c0) : SRF_KC;
            for (int kk = 0; kk < kc; kk += 4) {
                const float a = s_a[ar][kk + kq];
                const float b = s_w[kk + kq][lane & 15];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
        }
    }
    // C/D layout of 16x16: col = lane & 15, row = (lane >> 4) * 4 + j
    const int col = lane & 15;
    const float al = alpha ? alpha[col] : 1.0f;
    const float be = alpha ? beta[col] : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = row0 + wave * 16 + kq * 4 + j;
        if (row < A_out) {
            float v = acc[j];
            if (alpha) v = __fmaf_rn(v, al, be);
            if (residual) v = __fadd_rn(v, residual[(size_t)row * COUT + col]);
            if (relu) v = v > 0.0f ? v : 0.0f;
            out[(size_t)row * COUT + col] = v;
        }
    }
}

// COUT = 16, Cin <= 16 (the 41 x 1472 x 1472 level: conv_input and the first two basic blocks).  These layers are bound by
// the gathers, not by arithmetic (27 x 64 B per output row out of L2), so the kernel has no LDS and no barriers at all:
// a wave owns 16 output rows; its B operands -- all K x Cin x 16 weights, STEPS*K registers per lane -- are loaded once
// and stay in registers; the neighbour indices of all offsets are fetched up front, and the gathered A operands go
// global -> registers in MFMA order (v_mfma_f32_16x16x4_f32: lane (row, q) supplies channel 4j + q of step j), nine
// offsets in flight at a time.  Accumulation order as everywhere: offset ascending, channel ascending.
template <int STEPS>  // ceil(Cin / 4)
__global__ __launch_bounds__(256) void srf_spconv_c16_k(const float *__restrict__ in, int Cin, const float *__restrict__ W, int K,
                                                      const int *__restrict__ nbr, int nbr_stride, int A_out,
                                                      const float *__restrict__ alpha, const float *__restrict__ beta,
                                                      const float *__restrict__ residual, int relu, float *__restrict__ out,
                                                      const int *__restrict__ rows_dev)
{
    constexpr int COUT = 16, KM = SRF_KMAX, G = 9;
    if (rows_dev) {  // static-shape levels: rows >= *rows_dev are padding; their tiles do nothing
        const int live = *rows_dev;
        A_out = A_out < live ? A_out : live;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * 16;
    if (row0 >= A_out) return;  // no barriers in this kernel: waves are independent
    const int r = lane & 15, q = lane >> 4;
    const int row = row0 + r;
    // weights: lane (col = r, q) holds W[k][4j + q][col]
    float wreg[KM][STEPS];
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
            const int c = 4 * j + q;
            wreg[k][j] = (k < K && c < Cin) ? W[((size_t)k * Cin + c) * COUT + r] : 0.f;
        }
    int idx[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) idx[k] = (k < K && row < A_out) ? nbr[(size_t)k * nbr_stride + row] : -1;

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float a[2][G][STEPS];
#pragma unroll
    for (int t = 0; t < G; ++t)
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
            const int c = 4 * j + q;
            a[0][t][j] = (idx[t] >= 0 && c < Cin) ? in[(size_t)idx[t] * Cin + c] : 0.f;  // predicated: 9 of 10 neighbours are absent here
        }
#pragma unroll
    for (int g0 = 0; g0 < KM; g0 += G) {
        const int cur = (g0 / G) & 1;
        if (g0 + G < KM) {
#pragma unroll
            for (int t = 0; t < G; ++t)
#pragma unroll
                for (int j = 0; j < STEPS; ++j) {
                    const int k = g0 + G + t, c = 4 * j + q;
                    a[cur ^ 1][t][j] = (idx[k] >= 0 && c < Cin) ? in[(size_t)idx[k] * Cin + c] : 0.f;
                }
        }
#pragma unroll
        for (int t = 0; t < G; ++t) {
            const int k = g0 + t;
#pragma unroll
            for (int j = 0; j < STEPS; ++j) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][t][j], wreg[k][j], acc, 0, 0, 0);
            }
        }
    }
    // C/D layout of 16x16: col = lane & 15, row = (lane >> 4) * 4 + i
    const float al = alpha ? alpha[r] : 1.f, be = alpha ? beta[r] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int orow = row0 + q * 4 + i;
        if (orow < A_out) {
            float v = acc[i];
            if (alpha) v = __fmaf_rn(v, al, be);
            if (residual) v = __fadd_rn(v, residual[(size_t)orow * COUT + r]);
            if (relu) v = v > 0.f ? v : 0.f;
            out[(size_t)orow * COUT + r] = v;
        }
    }
}

// srf_spconv_c16l_k: the same 16-row waves with the weights in LDS instead of 108 registers per lane, and the gathers in nine batches
// of three offsets, three batches in flight (36 registers): ~100 registers per lane = five waves per SIMD instead of two.  The kernel
// is bound by the latency of its gathers (three dependent batches per wave, two waves per SIMD to hide them): at nuScenes size every wave
// of the level is resident at once either way (26k rows = 6 waves per CU), at Waymo size (138k rows = 34 waves per CU) the register form
// runs 4.3 rounds of 8 waves per CU.  Same chain per output (offset ascending, channel ascending): identical bits.
template <int STEPS>  // ceil(Cin / 4)
__global__ __launch_bounds__(256) void srf_spconv_c16l_k(const float *__restrict__ in, int Cin, const float *__restrict__ W, int K,
                                                       const int *__restrict__ nbr, int nbr_stride, int A_out,
                                                       const float *__restrict__ alpha, const float *__restrict__ beta,
                                                       const float *__restrict__ residual, int relu, float *__restrict__ out,
                                                       const int *__restrict__ rows_dev)
{
    constexpr int COUT = 16, KM = SRF_KMAX, G = 3, NBATCH = KM / G, D = 3;
    __shared__ float s_w[KM * STEPS * 64];   // [k][j][lane]: lane (col r, q) holds W[k][4 j + q][r]
    if (rows_dev) {
        const int live = *rows_dev;
        A_out = A_out < live ? A_out : live;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    for (int e = tid; e < KM * STEPS * 64; e += 256) {
        const int l = e & 63, kj = e >> 6, k = kj / STEPS, j = kj - k * STEPS;
        const int c = 4 * j + (l >> 4);
        s_w[e] = (k < K && c < Cin) ? W[((size_t)k * Cin + c) * COUT + (l & 15)] : 0.f;
    }
    __syncthreads();   // the only barrier: the waves are independent from here on
    const int row0 = (blockIdx.x * 4 + wave) * 16;
    if (row0 >= A_out) return;
    const int row = row0 + r;
    const int row_ld = row < A_out ? row : A_out - 1;
    int idx[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) idx[k] = nbr[(size_t)(k < K ? k : K - 1) * nbr_stride + row_ld];
#pragma unroll
    for (int k = 0; k < KM; ++k) idx[k] = (k < K && row < A_out) ? idx[k] : -1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float a[D][G][STEPS];
#define C16L_LOAD(SET, B)                                                                                              \
    _Pragma("unroll") for (int t_ = 0; t_ < G; ++t_)                                                                   \
        _Pragma("unroll") for (int j_ = 0; j_ < STEPS; ++j_) {                                                         \
            const int k_ = (B) * G + t_, c_ = 4 * j_ + q;                                                              \
            a[SET][t_][j_] = (idx[k_] >= 0 && c_ < Cin) ? in[(size_t)idx[k_] * Cin + c_] : 0.f;                        \
        }
    C16L_LOAD(0, 0)
    C16L_LOAD(1, 1)
#pragma unroll
    for (int b = 0; b < NBATCH; ++b) {
        if (b + 2 < NBATCH) { C16L_LOAD((b + 2) % D, b + 2) }
        float wv[G][STEPS];
#pragma unroll
        for (int t = 0; t < G; ++t)
#pragma unroll
            for (int j = 0; j < STEPS; ++j) wv[t][j] = s_w[((b * G + t) * STEPS + j) * 64 + lane];
#pragma unroll
        for (int t = 0; t < G; ++t)
#pragma unroll
            for (int j = 0; j < STEPS; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b % D][t][j], wv[t][j], acc, 0, 0, 0);
    }
#undef C16L_LOAD
    const float al = alpha ? alpha[r] : 1.f, be = alpha ? beta[r] : 0.f;
    float res[4];
    if (residual) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int orow = row0 + q * 4 + i;
            res[i] = residual[(size_t)(orow < A_out ? orow : A_out - 1) * COUT + r];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int orow = row0 + q * 4 + i;
        float v = acc[i];
        if (alpha) v = __fmaf_rn(v, al, be);
        if (residual) v = __fadd_rn(v, res[i]);
        if (relu) v = v > 0.f ? v : 0.f;
        if (orow < A_out) out[(size_t)orow * COUT + r] = v;
    }
}

#ifndef SRF_C16L_MIN_ROWS
#define SRF_C16L_MIN_ROWS 60000   /* measured (MI355X): 26k rows 14.1 -> 14.7 us (every wave resident at once either way), 99k rows 47.9 -> 35.1 us */
#endif
static bool srf_c16l_enabled()
{
    static const bool on = [] {
        const char *e = getenv("SRF_SPCONV_C16L");   // developer switch: 0 = the register-weight form (A/B timing; identical bits)
        return !(e && e[0] == '0');
    }();
    return on;
}

extern "C" int srf_spconv_fwd(const float *in, int A_in, int Cin, const float *W, int K, const int *nbr, int nbr_stride,
                              int A_out, int Cout, const float *alpha, const float *beta, const float *residual,
                              int relu, float *out, const int *rows_dev, srf_stream_t stream)
{
    if (A_in < 0 || A_out < 0 || Cin <= 0 || Cin > 512 || K <= 0 || K > SRF_KMAX || nbr_stride < A_out) return SRF_EINVAL;
    if ((alpha == nullptr) != (beta == nullptr)) return SRF_EINVAL;
    if (A_out == 0) return SRF_OK;
    if (!in || !W || !nbr || !out) return SRF_EINVAL;
    if (Cout != 16 && ((Cin & 3) || A_in == 0)) return SRF_EUNSUPPORTED;  // the wide kernels gather whole float4s
    hipStream_t st = (hipStream_t)stream;
#define SRF_ARGS in, Cin, W, K, nbr, nbr_stride, A_out, alpha, beta, residual, relu, out, rows_dev
    switch (Cout) {
    case 16:
        if (Cin <= 16 && K == SRF_KMAX && A_in > 0) {  // register-resident weights, LDS-free gather
            // many rows (Waymo's first level): the LDS-weight form, five waves per SIMD; few rows: every wave is resident at once anyway
            const bool lds_w = srf_c16l_enabled() && A_out >= SRF_C16L_MIN_ROWS;
            if (Cin <= 8) {
                if (lds_w) hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_c16l_k<2>), dim3(srf_ceil_div(A_out, 64)), dim3(256), 0, st, SRF_ARGS);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_c16_k<2>), dim3(srf_ceil_div(A_out, 64)), dim3(256), 0, st, SRF_ARGS);
            } else {
                if (lds_w) hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_c16l_k<4>), dim3(srf_ceil_div(A_out, 64)), dim3(256), 0, st, SRF_ARGS);
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_c16_k<4>), dim3(srf_ceil_div(A_out, 64)), dim3(256), 0, st, SRF_ARGS);
            }
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_mfma16_k<64>), dim3(srf_ceil_div(A_out, 64)), dim3(256), 0, st,
                               SRF_ARGS);
        }
        break;
    case 32:
        hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_mfma32_k<32, 128, 4, 1>), dim3(srf_ceil_div(A_out, 128)), dim3(256),
                           0, st, SRF_ARGS);
        break;
    case 64:
        hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_mfma32_k<64, 64, 2, 2>), dim3(srf_ceil_div(A_out, 64)), dim3(256), 0,
                           st, SRF_ARGS);
        break;
    case 128:
        hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_spconv_mfma32_k<128, 64, 2, 2>), dim3(srf_ceil_div(A_out, 64)), dim3(256), 0,
                           st, SRF_ARGS);
        break;
    default:
        return SRF_EUNSUPPORTED;
    }
#undef SRF_ARGS
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// K6 densify: (A, C) rows at (b, z, y, x) -> (B, C, D, H, W).  One thread per (row, channel): the read is coalesced,
// the write scatters one dword per channel plane.
// ---------------------------------------------------------------------------------------------------------------------
// 64 rows x 64 channels per workgroup through LDS: the features are read along the channels (coalesced), the dense
// (B, C, D, H, W) tensor is written with the ROW index fastest -- consecutive rows of the sorted active set are neighbours
// in x (and z), so a channel's 64 values land in a few cache lines instead of one line per value (3.1 M scattered 4-byte
// stores took 31 us for 12 MB).
__global__ __launch_bounds__(256) void srf_densify_k(const float *__restrict__ feats, const int4 *__restrict__ indices,
                                                   int A, int C, int D, int H, int W, float *__restrict__ out)
{
    __shared__ float s_f[64][65];
    __shared__ long long s_off[64];
    const int row0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tid = threadIdx.x;
    const size_t plane = (size_t)D * H * W;
    if (tid < 64) {
        const int a = row0 + tid;
        long long off = -1;
        if (a < A) {
            const int4 p = indices[a];
            if (p.x >= 0) off = (long long)((size_t)p.x * C * plane + ((size_t)p.y * H + p.z) * W + p.w);  // padding rows: b < 0
        }
        s_off[tid] = off;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + i * 256, r = e >> 6, c = e & 63;
        const int a = row0 + r;
        s_f[r][c] = (a < A && c0 + c < C) ? feats[(size_t)a * C + c0 + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + i * 256, c = e >> 6, r = e & 63;
        const long long off = s_off[r];
        if (off >= 0 && c0 + c < C) out[(size_t)off + (size_t)(c0 + c) * plane] = s_f[r][c];
    }
}

extern "C" int srf_densify(const float *feats, const int *indices, int A, int C, int B, int D, int H, int W, float *out,
                           int zero_fill, srf_stream_t stream)
{
    if (A < 0 || C <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0 || !out) return SRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (zero_fill) SRF_HIP_TRY(srf_fill_bytes(out, 0, sizeof(float) * (size_t)B * C * D * H * W, st));
    if (A == 0) return SRF_OK;
    if (!feats || !indices) return SRF_EINVAL;
    hipLaunchKernelGGL(srf_densify_k, dim3(srf_ceil_div(A, 64), srf_ceil_div(C, 64)), dim3(256), 0, st, feats,
                       (const int4 *)indices, A, C, D, H, W, out);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
