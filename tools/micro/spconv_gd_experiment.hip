// NOT BUILT -- record of a round-4 experiment.  This section sat in csrc/spconv.hip between
// srf_spconv_gs_k and srf_spconv_w32_k, with `srf_gd_layout` consulted before `srf_gs_layout` in srf_spconv_pack_weights and in
// the Cout = 64 / 128 cases of srf_spconv_fwd_packed.  Bit-exact against the oracle (tests/test_gpu_spconv.py passed), but slower than
// srf_spconv_gs_k on the nuScenes levels (tools/bench_spconv.py, MI355X): 128 -> 128 on 35 k rows / 556 k pairs 328 us against 258,
// 64 -> 64 on 59.6 k rows 129 us against 106.  Compile-time ablations of the 128 -> 128 form (the ABL template bits below): without
// the A gathers 264 us, without the lane transposes 307, without the reloads of B 252, without the accumulator round trip
// through LDS 319 -- i.e. with every wave fetching its own operands (4x the A bytes of the cooperative gather, and B re-read per
// wave and offset: ~3.9 MB per workgroup against 2.25 MB, 6.7 TB/s from L2 for the launch) the loads are NOT hidden one group
// ahead by the two waves a SIMD holds, and what the missing barrier gains is lost again.  Stages on the way: the loads as C++
// (the compiler sinks them below the last MFMA of a group: 315 us), two straight-line copies of the group with / without the
// reload of B (loads into fresh registers + copies behind vmcnt(0): same), then the inline-assembly form below.
// =====================================================================================================================
// srf_spconv_gd_k (round 4): the compacted-offset kernel above without its step latency.  srf_spconv_gs_k walks a tile as
// gather (global -> registers) -> LDS A image -> barrier -> fragment reads -> 64 MFMAs -> accumulators back to LDS, one
// dependent chain per group of 16 pairs: ~1.9 us per step of which 0.85 us are MFMA issue, and removing any ONE link (the
// ablations in the comment of that kernel) gains little because the others remain.  Here the chain is gone:
//   * every wave gathers ITS OWN A operand straight into the MFMA's register layout -- lane (row = lane & 15, kq = lane >> 4)
//     loads the channel quads 16 j + 4 kq .. + 3 of its gathered row (Cin / 16 dwordx4: the four lanes of a row read 64
//     consecutive bytes per j), and a 4 x 4 transpose across the four lane rows (v_permlane32_swap + v_permlane16_swap: four
//     instructions per four registers) hands lane kq the channels 4 s + kq that step s of the 16x16x4 MFMA takes from it
//     -- so the A image, its 32 ds_write_b32 per thread and group, the fragment reads and the BARRIER of every step do
//     not exist;
//   * a wave still owns COUT / 4 output columns, so the four waves read the same rows (4x the L2 -> register bytes of the
//     cooperative gather: ~2 KB per pair, under 10 TB/s for the whole chip) but never wait for each other: the two
//     workgroups of a CU are eight independent instruction streams whose loads, LDS traffic and MFMAs interleave freely;
//   * the operands of the NEXT group are loaded IN PLACE: the registers of a quarter of the channel steps are reloaded right
//     behind the MFMAs that consumed them (A for the next group, B too when the offset changes), i.e. a full group of MFMAs
//     (2048 cycles) ahead of their use and without a second register set;
//   * the output tile stays in LDS (accumulators read / written per group: a wave only touches its own columns).
// Without the A image a tile holds 112 rows in 75 KB (two workgroups per CU).  Accumulation order per output unchanged: offsets
// ascending, channels ascending (step s adds the channels 4 s .. 4 s + 3 in order): the same f32 fma chain as every other
// kernel here and the oracle, bit for bit.  SRF_SPCONV_GD=0 keeps srf_spconv_gs_k (A/B switch).
// =====================================================================================================================
#define SRF_GD_TMAX 112
#define SRF_GD_LS 112

static int srf_gd_abl()
{
    static const int v = [] { const char *e = getenv("SRF_GD_ABL"); return e ? atoi(e) : 0; }();
    return v;
}

static bool srf_gd_layout(int Cin, int Cout)
{
    static const bool on = [] {
        const char *e = getenv("SRF_SPCONV_GD");
        return !(e && e[0] == '0');
    }();
    return on && srf_gs_layout(Cin, Cout);
}

// P[k][wave][t][lane][s] = W[k][4 s + kq][wave CW + 16 t + (lane & 15)],  S = Cin / 4 steps, kq = lane >> 4, CW = Cout / 4
__global__ __launch_bounds__(256) void srf_pack_weights_gd_k(const float *__restrict__ W, int K, int Cin, int Cout, float *__restrict__ P)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)K * Cin * Cout;
    if (t >= total) return;
    const int S = Cin / 4, CW = Cout / 4, NT = CW / 16;
    long long r = t;
    const int s = (int)(r % S);
    r /= S;
    const int lane = (int)(r & 63);
    r >>= 6;
    const int tt = (int)(r % NT);
    r /= NT;
    const int wave = (int)(r & 3), k = (int)(r >> 2);
    const int cin = 4 * s + (lane >> 4), col = wave * CW + 16 * tt + (lane & 15);
    P[t] = W[((size_t)k * Cin + cin) * Cout + col];
}

// One group of 16 pairs: NV quarters of (wait, lane transpose, 4 NT MFMAs, reload of the quarter's registers for the next group).
// The reloads are inline assembly ON the loop-carried registers ("+v"): written as C++ loads the compiler either sinks them all
// below the group's last MFMA or, with the reload of B in a second copy of the body, loads into fresh registers and copies them
// back behind an s_waitcnt vmcnt(0) at the end of every group -- both expose the whole load latency per group (measured: 315 us
// against 255 us for srf_spconv_gs_k on the nuScenes level).  The compiler does not know these loads, so the waits are explicit:
// loads return in order, and between the load of quarter v and its use one group later at least NV - 1 loads were issued (one
// per other quarter; more when B was reloaded too), so vmcnt(NV - 1) before quarter v is exact in the common case and safe
// in the others.
#define SRF_GD_LOAD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(dst) : "v"(ptr) : "memory")

template <int NV, int NT, int ABL>
__device__ __forceinline__ void srf_gd_group(f32x4 (&a)[NV], f32x4 (&b)[NT][NV], f32x4 (&acc)[NT], const float *nrow, const float *nw,
                                             int S, bool new_tap)
{
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (NV == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // 4 x 4 transpose across the lane rows: (lane row kq, register e) = channel 16 v + 4 kq + e  ->  channel 16 v + 4 e + kq
        auto p02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[v][0]), __float_as_uint(a[v][2]), false, false);
        auto p13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[v][1]), __float_as_uint(a[v][3]), false, false);
        auto q01 = __builtin_amdgcn_permlane16_swap(p02[0], p13[0], false, false);
        auto q23 = __builtin_amdgcn_permlane16_swap(p02[1], p13[1], false, false);
        const float av[4] = {(ABL & 2) ? a[v][0] : __uint_as_float(q01[0]), (ABL & 2) ? a[v][1] : __uint_as_float(q01[1]),
                             (ABL & 2) ? a[v][2] : __uint_as_float(q23[0]), (ABL & 2) ? a[v][3] : __uint_as_float(q23[1])};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (!(ABL & 16)) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b[t][v][i], acc[t], 0, 0, 0);
                else acc[t][i] += av[i] * b[t][v][i];
            }
        // this quarter's registers are free: the next group's operands come in behind the MFMAs that used them
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 1)) SRF_GD_LOAD(a[v], nrow + v * 16);
        if (new_tap && !(ABL & 4)) {
#pragma unroll
            for (int t = 0; t < NT; ++t) SRF_GD_LOAD(b[t][v], nw + (size_t)t * 64 * S + v * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NCH, int COUT, int ABL = 0>
__global__ __launch_bounds__(256, 2) void srf_spconv_gd_k(const float *__restrict__ in, const float *__restrict__ Wg, int K,
                                                        const int *__restrict__ nbr, int nbr_stride, int A_out,
                                                        const float *__restrict__ alpha, const float *__restrict__ beta,
                                                        const float *__restrict__ residual, int relu,
                                                        float *__restrict__ out, const int *__restrict__ rows_dev,
                                                        const int *__restrict__ tiles)
{
    constexpr int CIN = 32 * NCH, S = CIN / 4, NV = S / 4, NT = COUT / 64, CW = COUT / 4;
    constexpr int NKW = (SRF_KMAX + 3) / 4;
    constexpr int TMAX = SRF_GD_TMAX, LS = SRF_GD_LS, OS = COUT + 4;
    static_assert(COUT == 128 || COUT == 64, "column tiling of the waves");
    static_assert(TMAX <= 128 && TMAX <= LS && LS % 16 == 0, "two ballot segments of 64 rows; whole groups per list");
    __shared__ int s_in[SRF_KMAX * LS];                 // per offset: input rows of the outputs that have this neighbour
    __shared__ __attribute__((aligned(4))) unsigned char s_slot[SRF_KMAX * LS];  // ... and their slot in the output tile
    __shared__ int s_cnt[SRF_KMAX];
    __shared__ int s_klist[SRF_KMAX + 1];
    __shared__ __attribute__((aligned(16))) float s_out[(TMAX + 1) * OS];  // + the spare row of padding slots

    const int A_cap = A_out;
    if (rows_dev) {  // static-shape levels: rows >= *rows_dev are padding; their tiles do nothing
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
    const int nsub = (range1 - range0 + TMAX - 1) / TMAX;
    const int TM = (((range1 - range0 + nsub - 1) / nsub) + 7) & ~7;  // <= TMAX (a multiple of 8)
    const int ar = lane & 15, kq = lane >> 4;
    const int colb = wave * CW;
    const float *in_lane = in + kq * 4;   // quad j of the lane: channels 16 j + 4 kq .. + 3
    const float *w_lane = Wg + ((size_t)wave * NT * 64 + lane) * S;   // + (k * 4 * NT + t) * 64 * S
    for (int row0 = range0; row0 < range1; row0 += TM) {
    const int row_end = row0 + TM < range1 ? row0 + TM : range1;  // rows of this sub-tile: [row0, row_end)
    int zero = 0;  // see srf_spconv_gs_k: keeps the prologue / epilogue address arithmetic out of the main loop's registers
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

    // ---- the wave's own stream of (offset, group) items: no barrier until the tile is done ----
    f32x4 a[NV], b[NT][NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        a[v] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t][v] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int tk = 0, g = 0;
    if (ntap > 0) {   // the first group's operands (the same untracked loads as in the loop: the compiler must not guard them)
        const int k0 = s_klist[0];
        const float *arow = in_lane + (size_t)s_in[k0 * LS + ar] * CIN;
        const float *w0 = w_lane + (size_t)k0 * (4 * NT * 64 * S);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            SRF_GD_LOAD(a[v], arow + v * 16);
#pragma unroll
            for (int t = 0; t < NT; ++t) SRF_GD_LOAD(b[t][v], w0 + (size_t)t * 64 * S + v * 4);
        }
    }
    while (tk < ntap) {
        const int k = s_klist[tk];
        const int ng = (s_cnt[k] + 15) >> 4;
        const bool last = g + 1 >= ng;
        const bool new_tap = last && tk + 1 < ntap;
        const int kn = new_tap ? s_klist[tk + 1] : k;
        const int gn = last ? 0 : g + 1;
        // the next item's row (after the tile's last item: an in-range list entry whose loads are never used)
        const float *nrow = in_lane + (size_t)s_in[kn * LS + gn * 16 + ar] * CIN;
        const float *nw = w_lane + (size_t)kn * (4 * NT * 64 * S);
        // output slots of the group's 16 pairs and their accumulators out of the tile; padding entries name the spare row TMAX
        const unsigned sl4 = *reinterpret_cast<const unsigned *>(s_slot + k * LS + g * 16 + kq * 4);
        f32x4 acc[NT];
        int oaddr[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            oaddr[jj] = (int)((sl4 >> (8 * jj)) & 255u) * OS + colb + ar;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t][jj] = (ABL & 8) ? 0.f : s_out[oaddr[jj] + t * 16];
        }
        srf_gd_group<NV, NT, ABL>(a, b, acc, nrow, nw, S, new_tap);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (!(ABL & 8) || acc[t][jj] == 12345.f) s_out[oaddr[jj] + t * 16] = acc[t][jj];
        if (last) {
            ++tk;
            g = 0;
        } else {
            ++g;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the (unused) operands requested behind the last group have landed: their registers are free
    __syncthreads();  // every wave's columns of the tile are complete

    // epilogue: every output row once, BN / residual / ReLU in registers, one row of COUT floats per pass and store
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

