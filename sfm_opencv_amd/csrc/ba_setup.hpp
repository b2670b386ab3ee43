// ba_setup.hpp -- construction of a bundle-adjustment problem ON THE DEVICE (included by ba.hip only).
//
// bundle_adjustment() of the reference is ONE call that builds the ceres::Problem and solves it
// (NViewReconstuct.cpp:1169-1224), and the "Time (s)" it prints covers both (NView:1239).  What has to be built here
// before the first LM iteration are orderings of the observation list:
//   * observations grouped by point, a point's observations by ascending camera (ties: caller's order);
//   * points stored sorted by the list of cameras that see them (lexicographic, ties: caller's order), so that the
//     per-camera and per-camera-pair walks of the kernels gather from runs of neighbouring records;
//   * a camera-ordered copy of (point slot, pixel);
//   * for every pair of free cameras that share points, the list of observation pairs, cut into chunks.
// Rounds 1-2 did this with std::sort / std::stable_sort on one host thread: 0.2 s at C4 (200 cameras / 1.2M observations)
// and 1.3 s at C5 (1000 / 8M) -- ten to a hundred times the LM loop it prepares.  All of it is now a handful of
// stable LSD radix sorts, prefix sums and gather kernels on the GPU; the host only sees per-camera-pair counts.
// The resulting tables are identical, entry by entry, to the ones the host code produced (tests/test_ba_setup_gpu.py
// restates the orderings in numpy), so the iteration results stay bit-identical.
//
// Everything here is HBM-bound integer work on a few hundred MB at most; the kernels are plain coalesced passes, the
// only structured one is the radix scatter (per-wave match-any ranking, so that the sort is stable).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

typedef unsigned long long su64;
typedef unsigned int su32;

// ------------------------------------------------------------------------------------------------
// exclusive prefix sum of 32-bit counters, in place: tile scan -> scan of the tile totals -> add
// ------------------------------------------------------------------------------------------------
#define SCAN_ITEMS 16
#define SCAN_TILE (256 * SCAN_ITEMS)

__device__ __forceinline__ su32 setup_wave_incl_scan(su32 v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const su32 t = __shfl_up(v, off); if (lane >= off) v += t; }
    return v;
}

// 256 threads: exclusive prefix of v over the workgroup, the workgroup's total in *total
__device__ __forceinline__ su32 setup_block_excl_scan(su32 v, su32* total, su32* sm)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const su32 incl = setup_wave_incl_scan(v, lane);
    if (lane == 63) sm[wave] = incl;
    __syncthreads();
    su32 wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const su32 s = sm[w]; if (w < wave) wbase += s; tot += s; }
    __syncthreads();
    *total = tot;
    return wbase + incl - v;
}

__global__ __launch_bounds__(256) void setup_scan_tile_kernel(su32* __restrict__ data, size_t n, su32* __restrict__ bsum)
{
    __shared__ su32 sm[4];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    su32 x[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { x[j] = base + j < n ? data[base + j] : 0u; sum += x[j]; }
    su32 tot;
    su32 run = setup_block_excl_scan(sum, &tot, sm);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { if (base + j < n) data[base + j] = run; run += x[j]; }
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// one workgroup: exclusive scan of the tile totals (64-bit carry: *total64 tells the caller whether 32 bits sufficed)
__global__ __launch_bounds__(256) void setup_scan_top_kernel(su32* __restrict__ bsum, int nb, su64* __restrict__ total64)
{
    __shared__ su32 sm[4];
    su64 carry = 0;
    for (int base = 0; base < nb; base += 256) {
        const int i = base + threadIdx.x;
        const su32 v = i < nb ? bsum[i] : 0u;
        su32 tot;
        const su32 ex = setup_block_excl_scan(v, &tot, sm);
        if (i < nb) bsum[i] = (su32)(carry + ex);
        carry += tot;
    }
    if (threadIdx.x == 0 && total64) *total64 = carry;
}

__global__ __launch_bounds__(256) void setup_scan_add_kernel(su32* __restrict__ data, size_t n, const su32* __restrict__ bsum)
{
    const su32 add = bsum[blockIdx.x];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { const size_t i = base + (size_t)j * 256 + threadIdx.x; if (i < n) data[i] += add; }
}

static inline size_t setup_scan_tiles(size_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

// data[0..n) -> exclusive prefix sums; bsum: scratch of setup_scan_tiles(n) counters; total64 (device, may be null) = the sum
static inline void setup_enqueue_scan(hipStream_t st, su32* data, size_t n, su32* bsum, su64* total64)
{
    if (n == 0) { if (total64) (void)hipMemsetAsync(total64, 0, sizeof(su64), st); return; }
    const int nb = (int)setup_scan_tiles(n);
    hipLaunchKernelGGL(setup_scan_tile_kernel, dim3(nb), dim3(256), 0, st, data, n, bsum);
    hipLaunchKernelGGL(setup_scan_top_kernel, dim3(1), dim3(256), 0, st, bsum, nb, total64);
    hipLaunchKernelGGL(setup_scan_add_kernel, dim3(nb), dim3(256), 0, st, data, n, (const su32*)bsum);
}

// ------------------------------------------------------------------------------------------------
// stable LSD radix sort of (64-bit key, 32-bit value) pairs, 8 bits per pass
//   tile = 4 waves x RS_ROUNDS x 64 consecutive elements; a wave owns a contiguous piece of the tile and walks it 64
//   elements at a time, ranking equal digits with a match-any over eight ballots, so equal keys keep their order.
// ------------------------------------------------------------------------------------------------
#define RS_ROUNDS 16
#define RS_TILE (256 * RS_ROUNDS)

__global__ __launch_bounds__(256) void setup_rs_hist_kernel(const su64* __restrict__ keys, size_t n, int shift, su32* __restrict__ hist, int nblocks)
{
    __shared__ su32 cnt[256];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const size_t i = base + (size_t)r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(su32)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = cnt[threadIdx.x];      // digit-major: one scan over the whole table gives the offsets
}

__global__ __launch_bounds__(256) void setup_rs_scatter_kernel(const su64* __restrict__ kin, const su32* __restrict__ vin, su64* __restrict__ kout,
                                                               su32* __restrict__ vout, size_t n, int shift, const su32* __restrict__ off, int nblocks)
{
    __shared__ su32 wcnt[4][256];
    __shared__ su32 wbase[4][256];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0;
    __syncthreads();
    const size_t w0 = (size_t)blockIdx.x * RS_TILE + (size_t)wave * (64 * RS_ROUNDS);
    su64 k[RS_ROUNDS]; su32 v[RS_ROUNDS], rk[RS_ROUNDS];
    const su64 lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const size_t i = w0 + (size_t)r * 64 + lane;
        const bool valid = i < n;
        k[r] = valid ? kin[i] : 0ull;
        v[r] = valid ? (vin ? vin[i] : (su32)i) : 0u;
        const su32 d = (su32)(k[r] >> shift) & 255u;
        su64 m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) { const bool bit = (d >> b) & 1u; const su64 bal = __ballot(bit); m &= bit ? bal : ~bal; }
        rk[r] = 0;
        if (valid) {
            // every lane of the group reads the wave's running count of this digit, then the group's last lane advances it
            // (one wave, program order: the read of all lanes precedes the write)
            const su32 prev = wcnt[wave][d];
            rk[r] = prev + (su32)__popcll(m & lt);
            if ((m >> lane) == 1ull) wcnt[wave][d] = prev + (su32)__popcll(m);
        }
    }
    __syncthreads();
    {
        su32 run = off[(size_t)tid * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < 4; ++w) { wbase[w][tid] = run; run += wcnt[w][tid]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const size_t i = w0 + (size_t)r * 64 + lane;
        if (i < n) {
            const su32 d = (su32)(k[r] >> shift) & 255u;
            const size_t pos = (size_t)wbase[wave][d] + rk[r];
            kout[pos] = k[r]; vout[pos] = v[r];
        }
    }
}

struct SetupSortBufs {
    su64* k[2]; su32* v[2];     // ping-pong; the input keys are in k[0]
    su32* hist; su32* bsum;     // 256 * tiles counters; setup_scan_tiles(256 * tiles) counters
};
static inline size_t setup_rs_tiles(size_t n) { return (n + RS_TILE - 1) / RS_TILE; }

// Sorts by the low `bits` bits of the keys.  Values: v[0], or the element index where identity_vals.  Returns the index
// (0 / 1) of the buffers that hold the result.
static inline int setup_radix_sort(hipStream_t st, const SetupSortBufs& B, size_t n, int bits, bool identity_vals)
{
    if (n == 0) return 0;
    const int passes = bits <= 0 ? 1 : (bits + 7) / 8;
    const int nblocks = (int)setup_rs_tiles(n);
    int cur = 0;
    for (int p = 0; p < passes; ++p) {
        hipLaunchKernelGGL(setup_rs_hist_kernel, dim3(nblocks), dim3(256), 0, st, (const su64*)B.k[cur], n, 8 * p, B.hist, nblocks);
        setup_enqueue_scan(st, B.hist, (size_t)256 * nblocks, B.bsum, nullptr);
        hipLaunchKernelGGL(setup_rs_scatter_kernel, dim3(nblocks), dim3(256), 0, st, (const su64*)B.k[cur],
                           (p == 0 && identity_vals) ? (const su32*)nullptr : (const su32*)B.v[cur], B.k[cur ^ 1], B.v[cur ^ 1], n, 8 * p,
                           (const su32*)B.hist, nblocks);
        cur ^= 1;
    }
    return cur;
}

// ------------------------------------------------------------------------------------------------
// the set-up passes
// ------------------------------------------------------------------------------------------------
static inline int setup_bit_width(su64 x) { int b = 0; while (x) { ++b; x >>= 1; } return b; }

// observation k -> key (point << cb | camera); flags[0] |= 1 on an index out of range (the key then stays inside the tables)
__global__ __launch_bounds__(256) void setup_obs_key_kernel(const int* __restrict__ rc, const int* __restrict__ rp, int nobs, int nc, int np, int cb,
                                                            su64* __restrict__ keys, int* __restrict__ flags)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nobs) return;
    const int c = rc[k], p = rp[k];
    if (c < 0 || c >= nc || p < 0 || p >= np) { atomicOr(flags, 1); keys[k] = 0; return; }
    keys[k] = ((su64)p << cb) | (su64)c;
}

// keys sorted; v(i) = keys[i] >> shift < nvals.  starts[v] = first i with v(i) >= v, for v = 0 .. nvals (starts[nvals] = n): the
// CSR offsets of the sorted list without counters or atomics (an atomic per observation on its camera's counter -- 8,000 per
// address at C5 -- made this pass 10 ms; as boundaries of the sorted keys it is one coalesced read)
__global__ __launch_bounds__(256) void setup_starts_kernel(const su64* __restrict__ keys, size_t n, int shift, su64 nvals, su32* __restrict__ starts)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i > n) return;
    const su64 cur = i < n ? keys[i] >> shift : nvals;
    su64 v = i > 0 ? (keys[i - 1] >> shift) + 1 : 0;
    for (; v <= cur && v <= nvals; ++v) starts[v] = (su32)i;
}

// max over p of st[p + 1] - st[p] (the longest track): one atomic per workgroup
__global__ __launch_bounds__(256) void setup_max_kernel(const su32* __restrict__ st, int n, su32* __restrict__ out)
{
    __shared__ su32 sm[4];
    su32 x = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) x = max(x, st[i + 1] - st[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = max(x, (su32)__shfl_xor(x, off));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) { x = max(max(sm[0], sm[1]), max(sm[2], sm[3])); if (x) atomicMax(out, x); }
}

// Sort key of a point for one group of positions [j0, j0 + npos) of its ascending camera list: fields of b bits,
// earlier positions in the higher bits, camera + 1 (0 = the list has ended: a prefix sorts first).
__global__ __launch_bounds__(256) void setup_ptkey_kernel(const su32* __restrict__ order, int np, const su32* __restrict__ st,
                                                          const su64* __restrict__ obs_keys, su64 cam_mask, int j0, int npos, int b, su64* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= np) return;
    const su32 p = order ? order[i] : (su32)i;
    const su32 base = st[p], m = st[p + 1] - base;
    su64 key = 0;
    for (int j = j0; j < j0 + npos; ++j) key = (key << b) | ((su32)j < m ? (obs_keys[base + j] & cam_mask) + 1ull : 0ull);
    out[i] = key;
}

// slot of every point, and the per-slot observation counts (scanned into pt_start by the caller)
__global__ __launch_bounds__(256) void setup_slot_kernel(const su32* __restrict__ order, int np, const su32* __restrict__ st, int* __restrict__ slot,
                                                         su32* __restrict__ cnt_slot)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= np) return;
    const su32 p = order[s];
    slot[p] = s; cnt_slot[s] = st[p + 1] - st[p];
}

// observations into their final order: by slot, inside a point by (camera, caller's index)
__global__ __launch_bounds__(256) void setup_fill_obs_kernel(const su64* __restrict__ obs_keys, const su32* __restrict__ obs_k, int nobs, int cb,
                                                             const su32* __restrict__ st, const int* __restrict__ slot, const int* __restrict__ pt_start,
                                                             const double2* __restrict__ ruv, int* __restrict__ ocam, int* __restrict__ opt,
                                                             double2* __restrict__ ouv, su64* __restrict__ cam_keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nobs) return;
    const su64 key = obs_keys[i];
    const su32 p = (su32)(key >> cb);
    const int cam = (int)(key & ((1ull << cb) - 1ull));
    const int s = slot[p];
    const int q = pt_start[s] + (int)((su32)i - st[p]);
    ocam[q] = cam; opt[q] = s; ouv[q] = ruv[obs_k[i]];
    cam_keys[q] = (su64)cam;
}

// dst[slot[p]] = src[p] (to_slot) or dst[p] = src[slot[p]] (back to the caller's order), 3 doubles per point
__global__ __launch_bounds__(256) void setup_permute_pts_kernel(const double* __restrict__ src, const int* __restrict__ slot, int np, double* __restrict__ dst, int to_slot)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= np) return;
    const size_t a = 3 * (size_t)p, b = 3 * (size_t)slot[p];
    const size_t from = to_slot ? a : b, to = to_slot ? b : a;
    dst[to] = src[from]; dst[to + 1] = src[from + 1]; dst[to + 2] = src[from + 2];
}

// camera-ordered copy of (point slot, pixel): q_sorted = observation indices sorted by camera (stable)
__global__ __launch_bounds__(256) void setup_cam_copy_kernel(const su32* __restrict__ q_sorted, int nobs, const int* __restrict__ opt, const double2* __restrict__ ouv,
                                                             int* __restrict__ cam_pt, double2* __restrict__ cam_uv)
{
    const int at = blockIdx.x * 256 + threadIdx.x;
    if (at >= nobs) return;
    const su32 q = q_sorted[at];
    cam_pt[at] = opt[q]; cam_uv[at] = ouv[q];
}

// observation pairs of a point between free cameras: a point's observations are in ascending camera order, so those of the
// constant camera 0 (if any) lead; with m others there are m (m - 1) / 2 pairs
__device__ __forceinline__ int setup_lead_fixed(const int* __restrict__ ocam, int lo, int hi, int fix0)
{
    int z = 0;
    if (fix0) while (lo + z < hi && ocam[lo + z] == 0) ++z;
    return z;
}
// total64 (zeroed by the caller): the number of pairs in 64 bits.  The per-point counters and the tile sums of the scan behind them are
// 32 bits wide and WRAP for absurd inputs (4096 points of 1449 observations each in one tile); the caller rejects the problem on this
// sum before it looks at anything the scan produced.
__global__ __launch_bounds__(256) void setup_pair_count_kernel(const int* __restrict__ pt_start, const int* __restrict__ ocam, int np, int fix0, su32* __restrict__ npair,
                                                               su64* __restrict__ total64)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    su64 c = 0;
    if (s < np) {
        const int lo = pt_start[s], hi = pt_start[s + 1];
        const su64 m = (su64)(hi - lo - setup_lead_fixed(ocam, lo, hi, fix0));
        c = m * (m - (m ? 1 : 0)) / 2;
        npair[s] = c > 0xffffffffull ? 0xffffffffu : (su32)c;
    }
    su64 v = c;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, off), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), off);
        v += ((su64)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0 && v) atomicAdd((unsigned long long*)total64, (unsigned long long)v);
}

// thread per observation i: its pairs (i, j > i) in the order "for i, for j" of the point -- key = ca * nc + cb with ca >= cb,
// raw = (observation of ca, observation of cb)
__global__ __launch_bounds__(256) void setup_pair_gen_kernel(const int* __restrict__ pt_start, const int* __restrict__ ocam, const int* __restrict__ opt, int nobs,
                                                             int fix0, int nc, const su32* __restrict__ pair_off, su64* __restrict__ keys, int2* __restrict__ raw)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nobs) return;
    const int s = opt[q], lo = pt_start[s], hi = pt_start[s + 1];
    const int z = setup_lead_fixed(ocam, lo, hi, fix0);
    const long long a = (long long)q - (lo + z), m = (long long)hi - lo - z;
    if (a < 0) return;
    size_t id = (size_t)pair_off[s] + (size_t)(a * (2 * m - a - 1) / 2);
    const int ci = ocam[q];
    for (int j = q + 1; j < hi; ++j, ++id) {
        const int cj = ocam[j];                   // cj >= ci
        if (ci < cj) { keys[id] = (su64)cj * (su64)nc + (su64)ci; raw[id] = make_int2(j, q); }
        else         { keys[id] = (su64)ci * (su64)nc + (su64)cj; raw[id] = make_int2(q, j); }
    }
}

// run starts of the sorted pair keys: flag -> (scan) -> compaction
__global__ __launch_bounds__(256) void setup_flag_kernel(const su64* __restrict__ keys, size_t n, su32* __restrict__ flag)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t > n) return;
    flag[t] = (t < n && (t == 0 || keys[t] != keys[t - 1])) ? 1u : 0u;       // n + 1 entries: the scan leaves the count in flag[n]
}
__global__ __launch_bounds__(256) void setup_compact_kernel(const su64* __restrict__ keys, const su32* __restrict__ rank, size_t n, su64* __restrict__ blk_key,
                                                            su32* __restrict__ blk_first)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    if (t == 0 || keys[t] != keys[t - 1]) { blk_key[rank[t]] = keys[t]; blk_first[rank[t]] = (su32)t; }
}

// items of the pair kernels, in sorted order: [observation i, observation j, point slot, 0]
__global__ __launch_bounds__(256) void setup_items_kernel(const su32* __restrict__ id_sorted, const int2* __restrict__ raw, const int* __restrict__ opt, size_t n,
                                                          int4* __restrict__ items)
{
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int2 r = raw[id_sorted[t]];
    items[t] = make_int4(r.x, r.y, opt[r.x], 0);
}

// lowest and highest camera among the observations of every block of 256 points (the back-substitution stages them in LDS)
__global__ __launch_bounds__(256) void setup_crange_kernel(const int* __restrict__ pt_start, const int* __restrict__ ocam, int np, int* __restrict__ crange)
{
    __shared__ int slo[4], shi[4];
    const int b = blockIdx.x;
    const int q0 = pt_start[min(np, b * 256)], q1 = pt_start[min(np, (b + 1) * 256)];
    int lo = INT_MAX, hi = -1;
    for (int q = q0 + threadIdx.x; q < q1; q += 256) { const int c = ocam[q]; lo = min(lo, c); hi = max(hi, c); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); }
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        crange[2 * (size_t)b] = min(min(slo[0], slo[1]), min(slo[2], slo[3]));
        crange[2 * (size_t)b + 1] = max(max(shi[0], shi[1]), max(shi[2], shi[3]));
    }
}
