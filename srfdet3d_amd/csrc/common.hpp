// common.hpp -- shared device helpers for libsrfdet3d_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/srfdet3d.h"

#define SRF_WAVE 64

#define SRF_HIP_TRY(expr)                                    \
    do {                                                     \
        hipError_t _e = (expr);                              \
        if (_e != hipSuccess) return SRF_EHIP_BASE - (int)_e; \
    } while (0)

#define SRF_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t _e = hipGetLastError();                   \
        if (_e != hipSuccess) return SRF_EHIP_BASE - (int)_e; \
    } while (0)

static inline int srf_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t srf_align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---------------------------------------------------------------------------------------------
// open-addressing coordinate table: keys[cap] (uint32 linearised coordinate, EMPTY = 0xFFFFFFFF)
// followed by vals[cap] (int32).  cap is a power of two >= 2 * entries, so probes are short and
// every probe loop is bounded by cap.
// ---------------------------------------------------------------------------------------------
#define SRF_EMPTY_KEY 0xFFFFFFFFu

__device__ __forceinline__ uint32_t srf_hash32(uint32_t k)
{
    k ^= k >> 16;
    k *= 0x7feb352dU;
    k ^= k >> 15;
    k *= 0x846ca68bU;
    k ^= k >> 16;
    return k;
}

// returns the slot holding `key` (inserting it if absent), or -1 if the table is full
__device__ __forceinline__ int srf_table_insert(uint32_t *keys, uint32_t mask, uint32_t key)
{
    uint32_t h = srf_hash32(key) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        uint32_t old = atomicCAS(&keys[h], SRF_EMPTY_KEY, key);
        if (old == SRF_EMPTY_KEY || old == key) return (int)h;
        h = (h + 1) & mask;
    }
    return -1;
}

// lookup in a table completed by an earlier launch
__device__ __forceinline__ int srf_table_find(const uint32_t *__restrict__ keys, uint32_t mask, uint32_t key)
{
    uint32_t h = srf_hash32(key) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        uint32_t k = keys[h];
        if (k == key) return (int)h;
        if (k == SRF_EMPTY_KEY) return -1;
        h = (h + 1) & mask;
    }
    return -1;
}

// ---------------------------------------------------------------------------------------------
// device fill as a kernel: the library enqueues nothing but kernel nodes into a captured hipGraph (a kernel node costs the same
// as a memset node).  Round 1 introduced it as a workaround for "memset nodes not taking effect on replay"; that diagnosis was
// withdrawn (tools/micro/graph_memset.hip: 0 of 120 replays wrong) -- the fill kernel stays because it is
// what every graph of this library has been validated with.  nbytes must be a multiple of 4 and ptr 4-byte aligned (every
// use here is).
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void srf_fill_words_k(uint32_t *__restrict__ p, uint32_t v, size_t nwords)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride) p[i] = v;
}

static inline hipError_t srf_fill_bytes(void *ptr, int byte, size_t nbytes, hipStream_t st)
{
    if (nbytes == 0) return hipSuccess;
    if ((nbytes & 3) || ((uintptr_t)ptr & 3)) return hipMemsetAsync(ptr, byte, nbytes, st);
    const uint32_t b = (uint32_t)(byte & 0xFF);
    const size_t nwords = nbytes >> 2;
    size_t blocks = (nwords + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(srf_fill_words_k, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t *)ptr, b * 0x01010101u, nwords);
    return hipGetLastError();
}

// Up to four word regions, each with its own value, in ONE launch (a frame replayed from a hipGraph pays ~5 us per launch
// whatever its size: the bitmap clear of a sparse level and the -1 fill of its row list are one node this way).
struct SrfFillRegions {
    uint32_t *ptr[4];
    uint32_t value[4];
    size_t nwords[4];
};

static __global__ __launch_bounds__(256) void srf_fill_regions_k(SrfFillRegions r)
{
    const size_t stride = (size_t)gridDim.x * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t *p = r.ptr[j];
        const uint32_t v = r.value[j];
        const size_t n = r.nwords[j];
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = v;
    }
}

// regions with a null pointer or zero length are skipped; pointers must be 4-byte aligned
static inline hipError_t srf_fill_regions(const SrfFillRegions &regions, hipStream_t st)
{
    SrfFillRegions r = regions;
    size_t most = 0;
    for (int j = 0; j < 4; ++j) {
        if (!r.ptr[j]) r.nwords[j] = 0;
        if ((uintptr_t)r.ptr[j] & 3) return hipErrorInvalidValue;
        most = r.nwords[j] > most ? r.nwords[j] : most;
    }
    if (most == 0) return hipSuccess;
    size_t blocks = (most + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(srf_fill_regions_k, dim3((unsigned)blocks), dim3(256), 0, st, r);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// wave / block scans (256-thread blocks)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int srf_wave_inclusive_scan(int v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ int srf_wave_sum(int v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// exclusive scan of one int per thread across a 256-thread block; *total receives the block sum
__device__ __forceinline__ int srf_block_exclusive_scan_256(int v, int *total, int *lds4 /* >= 4 ints */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = srf_wave_inclusive_scan(v);
    if (lane == 63) lds4[wave] = inc;
    __syncthreads();
    int w0 = lds4[0], w1 = lds4[1], w2 = lds4[2], w3 = lds4[3];
    int base = (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
    *total = w0 + w1 + w2 + w3;
    __syncthreads();
    return base + inc - v;
}

// ---------------------------------------------------------------------------------------------
// device-wide exclusive scan in three launches, with the value producer and the consumer fused in:
//   val(i)            -> int contribution of element i
//   out(i, v, prefix) -> called once per element with its value and exclusive prefix
// partial must hold srf_scan_blocks(n) + 1 ints; partial[srf_scan_blocks(n)] ends up holding the total
// (also copied to *total_out when non-null).
// ---------------------------------------------------------------------------------------------
#define SRF_SCAN_THREADS 256
#define SRF_SCAN_ITEMS 8
#define SRF_SCAN_TILE (SRF_SCAN_THREADS * SRF_SCAN_ITEMS)

static inline int srf_scan_blocks(long long n) { return n > 0 ? srf_ceil_div(n, SRF_SCAN_TILE) : 1; }

template <class ValF>
__global__ __launch_bounds__(SRF_SCAN_THREADS) void srf_scan_reduce_k(int n, ValF val, int *__restrict__ partial)
{
    __shared__ int lds[4];
    const int base = blockIdx.x * SRF_SCAN_TILE;
    int s = 0;
#pragma unroll
    for (int j = 0; j < SRF_SCAN_ITEMS; ++j) {
        int i = base + j * SRF_SCAN_THREADS + threadIdx.x;
        if (i < n) s += val(i);
    }
    s = srf_wave_sum(s);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// single block: exclusive scan of partial[0..nb) in place, total to partial[nb] and *total_out
static __global__ __launch_bounds__(256) void srf_scan_partials_k(int nb, int *__restrict__ partial, int *__restrict__ total_out,
                                                         int clamp_max)
{
    __shared__ int lds[4];
    int carry = 0;
    for (int base = 0; base < nb; base += 256) {
        int i = base + threadIdx.x;
        int v = i < nb ? partial[i] : 0;
        int tot;
        int ex = srf_block_exclusive_scan_256(v, &tot, lds);
        if (i < nb) partial[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        partial[nb] = carry;
        if (total_out) *total_out = (clamp_max >= 0 && carry > clamp_max) ? clamp_max : carry;
    }
}

template <class ValF, class OutF>
__global__ __launch_bounds__(SRF_SCAN_THREADS) void srf_scan_apply_k(int n, ValF val, const int *__restrict__ partial,
                                                                   OutF out)
{
    __shared__ int lds[4];
    const int base = blockIdx.x * SRF_SCAN_TILE + threadIdx.x * SRF_SCAN_ITEMS;
    int v[SRF_SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int j = 0; j < SRF_SCAN_ITEMS; ++j) {
        int i = base + j;
        v[j] = i < n ? val(i) : 0;
        s += v[j];
    }
    int tot;
    int ex = srf_block_exclusive_scan_256(s, &tot, lds) + partial[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SRF_SCAN_ITEMS; ++j) {
        int i = base + j;
        if (i < n) out(i, v[j], ex);
        ex += v[j];
    }
}

// The same scan as ONE launch of one 1024-thread workgroup, for inputs of at most SRF_SCAN_SINGLE_MAX = 8192 elements (the bitmaps
// of the two coarsest sparse levels: 2-5 k words): the three-launch form costs ~15 us there whatever the size (three dependent
// graph nodes), this one 7 us.  One round only: a lone workgroup pays a full memory latency per round (measured: 43 us for the
// 44.5 k words of level 3, 124 us for the 35 k points of the voxelization flags -- against 15 and 22 us in three launches).
// Element i belongs to thread i / ITEMS, as in srf_scan_apply_k: identical prefixes, identical out() calls.
#define SRF_SCAN_SINGLE_THREADS 1024
#define SRF_SCAN_SINGLE_MAX (SRF_SCAN_SINGLE_THREADS * SRF_SCAN_ITEMS)

template <class ValF, class OutF>
__global__ __launch_bounds__(SRF_SCAN_SINGLE_THREADS) void srf_scan_single_k(int n, ValF val, OutF out, int *__restrict__ partial_total,
                                                                           int *__restrict__ total_out, int clamp_max)
{
    __shared__ int s_w[SRF_SCAN_SINGLE_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int base0 = 0; base0 < n; base0 += SRF_SCAN_SINGLE_THREADS * SRF_SCAN_ITEMS) {  // uniform trip count
        const int base = base0 + threadIdx.x * SRF_SCAN_ITEMS;
        int v[SRF_SCAN_ITEMS];
        int s = 0;
#pragma unroll
        for (int j = 0; j < SRF_SCAN_ITEMS; ++j) {
            const int i = base + j;
            v[j] = i < n ? val(i) : 0;
            s += v[j];
        }
        const int inc = srf_wave_inclusive_scan(s);
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int before = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < SRF_SCAN_SINGLE_THREADS / 64; ++w) {
            const int c = s_w[w];
            before += w < wave ? c : 0;
            tot += c;
        }
        __syncthreads();
        int ex = carry + before + inc - s;
#pragma unroll
        for (int j = 0; j < SRF_SCAN_ITEMS; ++j) {
            const int i = base + j;
            if (i < n) out(i, v[j], ex);
            ex += v[j];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) {
        *partial_total = carry;
        if (total_out) *total_out = (clamp_max >= 0 && carry > clamp_max) ? clamp_max : carry;
    }
}

template <class ValF, class OutF>
static inline int srf_device_scan(int n, ValF val, OutF out, int *partial, int *total_out, int clamp_max,
                                  hipStream_t st)
{
    const int nb = srf_scan_blocks(n);
    if (n <= SRF_SCAN_SINGLE_MAX) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_scan_single_k<ValF, OutF>), dim3(1), dim3(SRF_SCAN_SINGLE_THREADS), 0, st, n, val, out,
                           partial + nb, total_out, clamp_max);
        SRF_LAUNCH_CHECK();
        return SRF_OK;
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_scan_reduce_k<ValF>), dim3(nb), dim3(SRF_SCAN_THREADS), 0, st, n, val, partial);
    hipLaunchKernelGGL(srf_scan_partials_k, dim3(1), dim3(256), 0, st, nb, partial, total_out, clamp_max);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(srf_scan_apply_k<ValF, OutF>), dim3(nb), dim3(SRF_SCAN_THREADS), 0, st, n, val,
                       partial, out);
    SRF_LAUNCH_CHECK();
    return SRF_OK;
}
