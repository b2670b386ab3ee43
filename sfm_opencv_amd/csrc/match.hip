// match.hip -- brute-force kNN-2 descriptor matching for gfx950 (replaces cv::BFMatcher::knnMatch(k=2)
// + the ratio tail of match_features, NViewReconstuct.cpp:873-913; L2 twin TwoViewReconstruct.cpp:156-194).
//
// Paths
//   * int8 MFMA path (v_mfma_i32_32x32x32_i8): OpenCV SIFT descriptors are integer-valued floats in
//     [0,255]; biased to int8 (v-128), |a-b|^2 = |a'|^2 + |b'|^2 - 2 a'.b' is exact in int32, so the
//     kNN ordering is exact in any summation order.  The reference orders by sqrtf(d^2) (float32): distinct
//     integers d^2 >= 2^22 can round to the same float, so rows whose 2nd-best d^2 >= 2^22 are re-scored by
//   * the exact fp32 path: direct-difference sum in the accumulation order of OpenCV's SSE2 normL2Sqr_
//     (see oracle/orc_match.c), correctly rounded sqrtf, ordering key (sqrt bits, train index).
//   * Hamming2 (AKAZE, the live reference configuration): VALU popcount of non-zero 2-bit cells.
// Tie-breaking everywhere: smaller distance, then smaller train index (cv::batchDistance's stable insertion).
#include "common.hpp"
// float32 results must be bit-exact with the CPU restatement: no FMA contraction, correctly rounded sqrt
// (HIP's __fsqrt_rn/__fmul_rn are NOT the rounded forms on this toolchain: native sqrt / contractible mul).
#pragma clang fp contract(off)
#include <cfloat>
#include <climits>
#include <algorithm>
#include <emmintrin.h>
#include <type_traits>

#define KNN_BLOCK_ROWS 256     // descriptor sets are padded to a multiple of this (chunks of the kNN kernels are whole multiples)
#ifndef KNN_STAGE_ROWS
#define KNN_STAGE_ROWS 128     // train rows per staged LDS block of knn2_i8_kernel
#endif
#ifndef KNN_WGS_PER_CU
#define KNN_WGS_PER_CU 4       // occupancy target of knn2_i8_kernel: 4 workgroups = 4 waves per SIMD (<= 128 VGPRs)
#endif
#define PAD_NORM 8388607      // 2^23-1: larger than any real partial key, never selected
#define KEY_INVALID 0x7fffffffffffffffLL
#define RESCORE_D2 4194304    // 2^22: below this, distinct integers have distinct float32 square roots

typedef int   v4i  __attribute__((ext_vector_type(4)));
typedef int   v16i __attribute__((ext_vector_type(16)));
typedef float v4f  __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// descriptor preparation
// ------------------------------------------------------------------------------------------------
// Float rows -> biased int8 rows (b = v - 128, zero padded), exactness check (integers in [0,255]) and two per-row
// terms: norm[row] = sum b^2 and norm[rows_pad + row] = sum b^2 + 2 sum b (the train-side term of the kNN keys, see
// knn2_i8_kernel).  Each lane converts 4 consecutive values (one 16-byte load, one 4-byte store); a wave covers
// 64 / (dim_pad / 4) rows.
__device__ __forceinline__ void prep_l2_rows(const float* __restrict__ src, size_t ld, int rows, int dim, int dim_pad, int rows_pad,
                                             int8_t* __restrict__ dst, int32_t* __restrict__ norm, int* __restrict__ flag, int wave_index)
{
    const int lane = threadIdx.x & 63;
    const int lpr = dim_pad >> 2;                       // lanes per row: 8, 16 or 32
    const int row = wave_index * (64 / lpr) + lane / lpr, k = (lane % lpr) * 4;
    if (row >= rows_pad) return;
    int acc = 0, sum = 0, bad = 0;
    int packed = 0;
    if (row < rows) {
        float v[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        const float* rp = src + (size_t)row * ld + k;
        if (k + 3 < dim && ((((size_t)rp) & 15) == 0)) { const float4 t = *(const float4*)rp; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (k + i < dim) v[i] = rp[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (k + i < dim) {
                const float r = rintf(v[i]);
                if (!(v[i] >= 0.0f && v[i] <= 255.0f) || r != v[i]) bad = 1;
                const int q = (int)fminf(fmaxf(r, 0.0f), 255.0f) - 128;
                packed |= (q & 255) << (8 * i);
                acc += q * q; sum += q;
            }
        }
    }
    *(int*)(dst + (size_t)row * dim_pad + k) = packed;
    for (int off = lpr >> 1; off > 0; off >>= 1) { acc += __shfl_xor(acc, off); sum += __shfl_xor(sum, off); }
    if (lane % lpr == 0) {
        norm[row] = row < rows ? acc : PAD_NORM;
        norm[rows_pad + row] = row < rows ? acc + 2 * sum : PAD_NORM;
    }
    if (bad) atomicOr(flag, 1);
}

// The same for rows whose length is a multiple of 16 floats and whose int8 copy has no padding columns (SIFT: 128), 16 values per
// lane: four 16-byte loads, v_cvt_pk_u8_f32 (convert + clamp) and a convert-back compare per value for the exactness verdict, the
// bias as one XOR per dword, the two sums as v_dot4_i32_i8, one 16-byte store -- a third of the VALU work per byte of prep_l2_rows,
// which sat at the issue rate rather than at the memory rate.  (For values that are not integers in [0, 255] the int8 row differs
// from prep_l2_rows' -- and is never read: the set is flagged inexact.)
__device__ __forceinline__ void prep_l2_rows16(const float* __restrict__ src, size_t ld, int rows, int dim, int rows_pad,
                                               int8_t* __restrict__ dst, int32_t* __restrict__ norm, int* __restrict__ flag, int wave_index)
{
    const int lane = threadIdx.x & 63;
    const int lpr = dim >> 4;                           // lanes per row: 2, 4 or 8
    const int row = wave_index * (64 / lpr) + lane / lpr, k = (lane % lpr) * 16;
    if (row >= rows_pad) return;
    int acc = 0, sum = 0, bad = 0;
    v4i packed = { 0, 0, 0, 0 };
    if (row < rows) {
        const float4* rp = (const float4*)(src + (size_t)row * ld + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 t = rp[j];
            unsigned u = __builtin_amdgcn_cvt_pk_u8_f32(t.x, 0, 0);
            u = __builtin_amdgcn_cvt_pk_u8_f32(t.y, 1, u);
            u = __builtin_amdgcn_cvt_pk_u8_f32(t.z, 2, u);
            u = __builtin_amdgcn_cvt_pk_u8_f32(t.w, 3, u);
            bad |= ((float)(u & 255u) != t.x) | ((float)((u >> 8) & 255u) != t.y) | ((float)((u >> 16) & 255u) != t.z) | ((float)(u >> 24) != t.w);
            const int q = (int)(u ^ 0x80808080u);       // value - 128 in every byte
            packed[j] = q;
            acc = __builtin_amdgcn_sdot4(q, q, acc, false);
            sum = __builtin_amdgcn_sdot4(q, 0x01010101, sum, false);
        }
    }
    *(v4i*)(dst + (size_t)row * dim + k) = packed;
    for (int off = lpr >> 1; off > 0; off >>= 1) { acc += __shfl_xor(acc, off); sum += __shfl_xor(sum, off); }
    if (lane % lpr == 0) {
        norm[row] = row < rows ? acc : PAD_NORM;
        norm[rows_pad + row] = row < rows ? acc + 2 * sum : PAD_NORM;
    }
    if (bad) atomicOr(flag, 1);
}
// host side of the choice (per launch: every image of a batch must qualify)
static inline bool prep_l2_fast(const float* src, size_t ld, int dim, int dim_pad)
{
    return dim == dim_pad && (dim % 16) == 0 && dim >= 32 && (ld % 4) == 0 && (((uintptr_t)src) % 16) == 0;
}

__global__ void prep_l2_kernel(const float* __restrict__ src, size_t ld, int rows, int dim, int dim_pad,
                               int8_t* __restrict__ dst, int32_t* __restrict__ norm, int* __restrict__ flag, int rows_pad, int fast)
{
    const int wave_index = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (fast) prep_l2_rows16(src, ld, rows, dim, rows_pad, dst, norm, flag, wave_index);
    else prep_l2_rows(src, ld, rows, dim, dim_pad, rows_pad, dst, norm, flag, wave_index);
}

// batched form: one launch prepares many images (blockIdx.y = image)
struct PrepDesc { const float* src; size_t ld; int rows, dim, dim_pad, rows_pad; int8_t* dst; int32_t* norm; int* flag; };
__global__ void prep_l2_batched_kernel(const PrepDesc* __restrict__ tbl, int fast)
{
    const PrepDesc d = tbl[blockIdx.y];
    // table pointers are generic to the compiler; round-trip through the global address space so the loads/stores are global_*
    const float* src = (const float*)(const float __attribute__((address_space(1)))*)(uintptr_t)d.src;
    int8_t* dst = (int8_t*)(int8_t __attribute__((address_space(1)))*)(uintptr_t)d.dst;
    int32_t* norm = (int32_t*)(int32_t __attribute__((address_space(1)))*)(uintptr_t)d.norm;
    const int wave_index = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (fast) prep_l2_rows16(src, d.ld, d.rows, d.dim, d.rows_pad, dst, norm, d.flag, wave_index);
    else prep_l2_rows(src, d.ld, d.rows, d.dim, d.dim_pad, d.rows_pad, dst, norm, d.flag, wave_index);
}

// Rows that crossed PCIe as BYTES (sfmhip_descsets_create_l2_host: the staging threads convert integer-valued float rows on their way
// into the pinned ring, 128 B per SIFT row instead of 512): the biased int8 copy, the two norm terms and the float rows the exact
// kernels read (re-scoring of square-root collisions), one launch for all images.  16 values per lane, dim in {32, 64, 128}.
struct PrepU8Desc { const uint8_t* src; int rows, dim, rows_pad; int8_t* dst; int32_t* norm; float* f32; };
__global__ void prep_l2_u8_batched_kernel(const PrepU8Desc* __restrict__ tbl)
{
    const PrepU8Desc d = tbl[blockIdx.y];
    const uint8_t* src = (const uint8_t*)(const uint8_t __attribute__((address_space(1)))*)(uintptr_t)d.src;
    int8_t* dst = (int8_t*)(int8_t __attribute__((address_space(1)))*)(uintptr_t)d.dst;
    int32_t* norm = (int32_t*)(int32_t __attribute__((address_space(1)))*)(uintptr_t)d.norm;
    float* f32 = (float*)(float __attribute__((address_space(1)))*)(uintptr_t)d.f32;
    const int lane = threadIdx.x & 63, wave_index = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lpr = d.dim >> 4;                         // lanes per row: 2, 4 or 8
    const int row = wave_index * (64 / lpr) + lane / lpr, k = (lane % lpr) * 16;
    if (row >= d.rows_pad) return;
    int acc = 0, sum = 0;
    v4i packed = { 0, 0, 0, 0 };
    if (row < d.rows) {
        const v4i u = *(const v4i*)(src + (size_t)row * d.dim + k);
        float4* fo = (float4*)(f32 + (size_t)row * d.dim + k);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned w = (unsigned)u[j];
            fo[j] = make_float4((float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24));
            const int q = (int)(w ^ 0x80808080u);
            packed[j] = q;
            acc = __builtin_amdgcn_sdot4(q, q, acc, false);
            sum = __builtin_amdgcn_sdot4(q, 0x01010101, sum, false);
        }
    }
    *(v4i*)(dst + (size_t)row * d.dim + k) = packed;
    for (int off = lpr >> 1; off > 0; off >>= 1) { acc += __shfl_xor(acc, off); sum += __shfl_xor(sum, off); }
    if (lane % lpr == 0) {
        norm[row] = row < d.rows ? acc : PAD_NORM;
        norm[d.rows_pad + row] = row < d.rows ? acc + 2 * sum : PAD_NORM;
    }
}

// Hamming2: rows into 64-byte zero-padded rows, re-encoded so that one dword carries 32 two-bit cells' LOW bits and
// another their HIGH bits: with a[0..15] the 16 dwords of a row and M = 0x55555555,
//     L[i] = (a[2i] & M) | ((a[2i+1] & M) << 1),   H[i] = ((a[2i] >> 1) & M) | (a[2i+1] & ~M),   i = 0..7
// (cells of a[2i] on the even bit positions, cells of a[2i+1] on the odd ones).  A cell differs iff its low bits or its
// high bits differ, so NORM_HAMMING2(a, b) = sum_i popcount((La[i]^Lb[i]) | (Ha[i]^Hb[i])): 3 VALU ops per 32 cells
// (v_xor, v_bitop3, v_bcnt) instead of 5 per 16.  Stored as [L0..L7 | H0..H7].
__device__ __forceinline__ void prep_hamming_body(const uint8_t* __restrict__ src, size_t ld, int rows, int nbytes,
                                                  uint32_t* __restrict__ dst, int rows_pad, size_t i)
{
    if (i >= (size_t)rows_pad * 8) return;
    const int row = (int)(i >> 3), k = (int)(i & 7);
    uint32_t a0 = 0, a1 = 0;
    if (row < rows) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k0 = 8 * k + b, k1 = 8 * k + 4 + b;
            if (k0 < nbytes) a0 |= (uint32_t)src[(size_t)row * ld + k0] << (8 * b);
            if (k1 < nbytes) a1 |= (uint32_t)src[(size_t)row * ld + k1] << (8 * b);
        }
    }
    const uint32_t M = 0x55555555u;
    dst[(size_t)row * 16 + k] = (a0 & M) | ((a1 & M) << 1);
    dst[(size_t)row * 16 + 8 + k] = ((a0 >> 1) & M) | (a1 & ~M);
}
__global__ void prep_hamming_kernel(const uint8_t* __restrict__ src, size_t ld, int rows, int nbytes,
                                    uint32_t* __restrict__ dst, int rows_pad)
{
    prep_hamming_body(src, ld, rows, nbytes, dst, rows_pad, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}
// one launch for many images (blockIdx.y = image): sfmhip_descsets_create_hamming2_host
struct PrepHamDesc { const uint8_t* src; size_t ld; int rows, nbytes, rows_pad; uint32_t* u32; uint32_t* f4; };
__global__ void prep_hamming_batched_kernel(const PrepHamDesc* __restrict__ tbl)
{
    const PrepHamDesc d = tbl[blockIdx.y];
    prep_hamming_body((const uint8_t*)(const uint8_t __attribute__((address_space(1)))*)(uintptr_t)d.src, d.ld, d.rows, d.nbytes,
                      (uint32_t*)(uint32_t __attribute__((address_space(1)))*)(uintptr_t)d.u32, d.rows_pad, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// Hamming2 on the matrix cores: every row as 768 FP4 (e2m1) values = 24 blocks of 32 = 384 bytes.  A two-bit cell (b0, b1) becomes
// the three values (s0, s1, s0 s1), s = 1 - 2 bit -- the vertices of a regular simplex: the dot product of two cells' triples is 3
// when the cells are equal and -1 when they differ, so   dot(row a, row b) = 4 (equal cells) - cells   and
//     NORM_HAMMING2(a, b) = (3 cells - dot) / 4            (cells = 4 nbytes; 3 cells <= 732 values, i.e. nbytes <= 61: AKAZE's 61).
// Value v < 3 cells: cell v / 3, plane v % 3 (any fixed order would do: both operands of the MFMA use the same one).
// The 36 values left over carry what turns the accumulator itself into the kernel's top-2 key (knn2_hamming2_fp4_kernel):
//   736..744 (block 23): the row's 32-row tile index (row >> 5) & 255: bit j < 7 as the value {.5, 1, 1, 1, 1, 2, 4}[j], bit 7 as 4.0 twice;
//   732..735 (block 22) and 745..767 (block 23): 6.0 on rows past the set's end, 0 on real rows (the query side holds 6.0 there; the
//   kernel also reads the top one of them as the pad row's block scale).
#define H4_ROW_BYTES 384
#ifndef H4_WAVES
#define H4_WAVES 8             // waves per workgroup of knn2_hamming2_fp4_kernel (4 or 8)
#endif
#define H4_MAX_NBYTES 61
__device__ __forceinline__ void prep_hamming_fp4_body(const uint8_t* __restrict__ src, size_t ld, int rows, int nbytes,
                                                      uint32_t* __restrict__ dst, int rows_pad, size_t i)     // i: one dword = 8 values
{
    if (i >= (size_t)rows_pad * 96) return;
    const int row = (int)(i / 96), w = (int)(i % 96);
    const bool pad = row >= rows;
    const int ncell3 = 12 * nbytes;
    uint32_t out = 0;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int v = 8 * w + n;
        uint32_t code = 0;
        if (v < 732) {
            if (!pad && v < ncell3) {
                const int cell = v / 3, plane = v - 3 * cell;
                const uint32_t byte = src[(size_t)row * ld + (cell >> 2)];
                const uint32_t b0 = (byte >> (2 * (cell & 3))) & 1u, b1 = (byte >> (2 * (cell & 3) + 1)) & 1u;
                const uint32_t bit = plane == 0 ? b0 : (plane == 1 ? b1 : (b0 ^ b1));
                code = 0x2u | (bit << 3);                    // +1.0 / -1.0
            }
        } else if (v >= 736 && v < 745) {
            const int j = v - 736 < 7 ? v - 736 : 7;
            const uint32_t bcode = j == 0 ? 1u : (j < 5 ? 2u : (j == 5 ? 4u : 6u));      // .5, 1, 1, 1, 1, 2, 4, (4, 4)
            if ((((unsigned)row >> 5) >> j) & 1u) code = bcode;
        } else {
            code = pad ? 7u : 0u;                            // 6.0
        }
        out |= code << (4 * n);
    }
    dst[i] = out;
}
__global__ void prep_hamming_fp4_kernel(const uint8_t* __restrict__ src, size_t ld, int rows, int nbytes,
                                        uint32_t* __restrict__ dst, int rows_pad)
{
    prep_hamming_fp4_body(src, ld, rows, nbytes, dst, rows_pad, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void prep_hamming_fp4_batched_kernel(const PrepHamDesc* __restrict__ tbl)
{
    const PrepHamDesc d = tbl[blockIdx.y];
    if (!d.f4) return;
    prep_hamming_fp4_body((const uint8_t*)(const uint8_t __attribute__((address_space(1)))*)(uintptr_t)d.src, d.ld, d.rows, d.nbytes,
                          (uint32_t*)(uint32_t __attribute__((address_space(1)))*)(uintptr_t)d.f4, d.rows_pad, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// per-pair descriptor for the batched kernels
// ------------------------------------------------------------------------------------------------
struct PairDesc {
    const void* q; const void* t;          // int8 (mfma path) / u32 rows (hamming)
    const float* qf; const float* tf;      // float rows (exact path)
    const int32_t* qn; const int32_t* tn;  // query: sum a^2; train: sum b^2 + 2 sum b (biased int8 rows a, b)
    int nq, nt, nq_pad, nt_pad, dim;
    int nchunks, chunk_rows;               // train chunking (chunk_rows multiple of 128)
    long long part_off;                    // offset (entries of two keys) into the partial buffer, [row][chunk]
    long long out_off;                     // row offset into idx2 / dist2
    long long list_off;                    // offset into the rescore row list
    size_t ldq, ldt;                       // float row strides (elements)
};

// top-2 merge of sorted pairs (a1<=a2), (b1<=b2)
__device__ __forceinline__ void merge2(long long& a1, long long& a2, long long b1, long long b2)
{
    const long long lo = a1 < b1 ? a1 : b1;
    const long long hi = a1 < b1 ? b1 : a1;
    const long long m2 = a2 < b2 ? a2 : b2;
    a1 = lo; a2 = hi < m2 ? hi : m2;
}
__device__ __forceinline__ long long shfl_xor_ll(long long v, int off)
{
    int lo = (int)(v & 0xffffffffLL), hi = (int)(v >> 32);
    lo = __shfl_xor(lo, off); hi = __shfl_xor(hi, off);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// ------------------------------------------------------------------------------------------------
// int8 MFMA kNN-2.  grid = (query blocks of 128, chunks, pairs), block = 256 (4 waves x 32 query rows).
// LDS: two buffers of 128 train rows x DP bytes, 16-byte chunks XOR-swizzled so that the 16 lanes of a
// ds_read_b128 group (16 different rows, same k-chunk) hit 16 different slots of the 256-byte bank row.
// Partial output: for each (query row, chunk): two 64-bit keys (d2 << 32 | train index), ascending.
// ------------------------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256, KNN_WGS_PER_CU) void knn2_i8_kernel(const PairDesc* __restrict__ pairs, long long* __restrict__ part, int n_pairs)
{
    constexpr int DP = 32 * KS;          // bytes per row
    constexpr int CH = DP / 16;          // 16-byte chunks per row
    constexpr int TROWS = KNN_STAGE_ROWS, TILES = TROWS / 32;     // train rows staged per barrier
    constexpr int PASSES = (TROWS * CH) / 256;
    constexpr int BUF_BYTES = TROWS * DP, NORM_OFF = 2 * BUF_BYTES;
    constexpr int STAGE_BYTES = 2 * BUF_BYTES + 2 * TROWS * 4, MERGE_BYTES = 4 * 32 * 33 * 8;
    __shared__ __attribute__((aligned(16))) unsigned char lds[STAGE_BYTES > MERGE_BYTES ? STAGE_BYTES : MERGE_BYTES];
    // XCD-aware work mapping (speed only, any mapping is correct): consecutive workgroup ids go round-robin to the 8 XCDs,
    // each with its own L2, so pair 8g + k is given to the workgroups with id = k (mod 8): one XCD streams one train
    // image instead of all eight fetching every image (TCC FETCH_SIZE of the C4 launch: 1.06 GB -> see profiles/).
    // The grid's z extent is padded to a multiple of 8.
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int per_pair = gridDim.x * gridDim.y, slot = lin >> 3;
    const int pair = (slot / per_pair) * 8 + (lin & 7), rest = slot % per_pair;
    if (pair >= n_pairs) return;
    const PairDesc pd = pairs[pair];
    const int qb = rest % gridDim.x, chunk = rest / gridDim.x;
    if (qb * 128 >= pd.nq_pad || chunk >= pd.nchunks) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: keep it (and every address built from it) in SGPRs
    const int l31 = lane & 31, half = lane >> 5;
    // pointers that come out of the PairDesc table are generic to the compiler: without the address-space casts the
    // train prefetch becomes flat_load, which also counts in lgkmcnt -- every LDS wait then waits for HBM as well
    typedef const int8_t __attribute__((address_space(1)))* gi8;
    typedef const int32_t __attribute__((address_space(1)))* gi32;
    typedef const v4i __attribute__((address_space(1)))* gv4;
    const gi8 Q = (gi8)(uintptr_t)pd.q;
    const gi8 T = (gi8)(uintptr_t)pd.t;
    const gi32 TN = (gi32)(uintptr_t)pd.tn;
    const int t_begin = chunk * pd.chunk_rows;
    int t_end = t_begin + pd.chunk_rows; if (t_end > pd.nt_pad) t_end = pd.nt_pad;
    const int nblocks = (t_end - t_begin) / TROWS;
    const int q0 = qb * 128 + wave * 32;

    // stationary operand: 32 query rows per wave, lane holds row l31, k bytes [32 ks + 16 half, +16), COMPLEMENTED:
    // ~a = -a - 1 stays in int8 range, and with S = sum (~a) b = -a.b - sum b the partial key
    //     (|b|^2 - 2 a.b) * 128 + slot = (|b|^2 + 2 sum b) * 128 + slot + (S << 8)
    // is one v_lshl_add_u32 per accumulator (the train-side term comes from the prep kernel).  Zero padding stays
    // neutral: padded query bytes become -1 but meet zero train bytes.
    v4i afrag[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        afrag[ks] = ~*(gv4)(Q + (size_t)(q0 + l31) * DP + 32 * ks + 16 * half);
        asm volatile("" : "+v"(afrag[ks]));      // opaque: hipcc otherwise re-derives the complement inside the loop
    }

    int best1[16], best2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { best1[i] = INT_MAX; best2[i] = INT_MAX; }

    // Staging is LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land at M0 + lane*16, no staging registers and no
    // ds_write): wave w's p-th instruction fills the 1 KB segment seg = 4p + w of the buffer, i.e. LDS chunk position
    // (row r = 8 seg + lane/8, slot = lane%8); the XOR swizzle is applied on the GLOBAL side -- the lane fetches chunk
    // c = slot ^ ((r >> 1) & (CH-1)) of row r -- so the operand reads below find chunk c of row r at slot c ^ ((r>>1)&(CH-1)).
    // (r >> 1) & (CH-1) does not depend on p, so pass p is a uniform +p*4096 on both sides.
    // Everything the loop addresses is a per-thread constant plus a compile-time offset: the VALU is the busiest unit of
    // this kernel (PMC: SQ_ACTIVE_INST_VALU 74 % of the wall time), so the loop spends it on the top-2 epilogue only.
    static_assert(DP == 128 || PASSES == 1 || (1024 / DP) % (2 * CH) == 0, "staging swizzle must repeat per pass");
    typedef const char __attribute__((address_space(1)))* gbytes;
    typedef char __attribute__((address_space(3)))* lbytes;
    int rd_off[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) rd_off[ks] = l31 * DP + 16 * ((2 * ks + half) ^ ((l31 >> 1) & (CH - 1)));
    const int seg_row = (wave * 1024 + lane * 16) / DP, seg_slot = ((wave * 1024 + lane * 16) % DP) / 16;
    const unsigned st_goff = (unsigned)(seg_row * DP + 16 * (seg_slot ^ ((seg_row >> 1) & (CH - 1))));     // unsigned: lets hipcc use the SGPR-base + 32-bit VGPR-offset form
    const int nrm_off = NORM_OFF + 4 * l31;

    auto g_stage = [&](auto bufc, int blk) {
        constexpr int buf = decltype(bufc)::value;
        const gbytes blk_base = (gbytes)(T + (size_t)(t_begin + blk * TROWS) * DP);      // wave-uniform
#pragma unroll
        for (int p = 0; p < PASSES; ++p)
        {
            unsigned long long pb = (unsigned long long)(uintptr_t)(blk_base + p * 4096);
            asm volatile("" : "+s"(pb));          // keep the per-pass base in SGPRs: one VGPR offset serves all passes
            __builtin_amdgcn_global_load_lds((gbytes)pb + st_goff, (lbytes)(lds + buf * BUF_BYTES + p * 4096 + wave * 1024), 16, 0, 0);
        }
        // train-side key terms (|b|^2 + 2 sum b), one dword per row
        if (wave < TROWS / 64)
        {
            unsigned long long nb = (unsigned long long)(uintptr_t)(TN + t_begin + blk * TROWS + wave * 64);
            asm volatile("" : "+s"(nb));
            __builtin_amdgcn_global_load_lds((gbytes)nb + (unsigned)(4 * lane), (lbytes)(lds + NORM_OFF + buf * (4 * TROWS) + wave * 256), 4, 0, 0);
        }
    };
    // one block of TROWS trains out of LDS buffer `buf`
    auto compute = [&](auto bufc, int blk) {
        constexpr int buf = decltype(bufc)::value;
        // One tile at a time: 4 fragments, KS MFMAs, then the epilogue on the finished accumulator.  Nothing overlaps inside
        // the wave on purpose -- this form needs <= 128 VGPRs, and FOUR resident waves per SIMD overlap each other's MFMA,
        // VALU and LDS phases better than the software-pipelined 180-register form did with two (the epilogue mix issues at
        // 2.0 ns per instruction per SIMD with 4 waves against 2.3 with 2, experiments/valu_bench.hip).  Round 3 re-tried it the way the
        // FP4 Hamming2 kernel does it -- tile t - 1's update dealt out behind the four MFMAs of tile t, two accumulator sets, 167
        // registers = three waves per SIMD: 0.850-0.861 ms against 0.874-0.877 in the same call (2 %): not kept.
#pragma unroll
        for (int tile = 0; tile < TILES; ++tile) {
            v4i bf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bf[ks] = *(const v4i*)(lds + rd_off[ks] + (buf * BUF_BYTES + tile * 32 * DP));
            const int nbt = (*(const int*)(lds + nrm_off + (buf * (4 * TROWS) + tile * 128)) << 7) + (blk * TILES + tile);
            v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[ks], bf[ks], acc, 0, 0, 0);
            // C[row = query (reg), col = train (lane&31)].  key = (|b|^2 - 2 a.b) * 128 + local tile index.
            // Three VALU ops per accumulator: v_lshl_add_u32, v_med3_i32 (second smallest of {best1 <= best2, key}), v_min_i32.
            // (Round 3 tried compare-and-skip -- a key only matters below the runner-up, ~2 / j of the time for the j-th train: key +
            // v_cmp + s_cbranch_vccnz to an out-of-line update per register, 2.2 instructions per distance -- and measured 1.20 ms
            // against 0.87: sixteen dependent compare -> branch waits per tile cost more than the sixteen instructions saved.)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = (int)(((unsigned)acc[i] << 8) + (unsigned)nbt);
                const int lo = best1[i] < best2[i] ? best1[i] : best2[i], hi = best1[i] < best2[i] ? best2[i] : best1[i];
                const int t = hi < key ? hi : key;
                best2[i] = lo > t ? lo : t;                                  // max(min(a,b), min(max(a,b), c)) = med3
                best1[i] = best1[i] < key ? best1[i] : key;
                asm volatile("" : "+v"(best1[i]), "+v"(best2[i]));          // pin here: LLVM otherwise sinks the whole epilogue below the barrier
            }
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

#if !defined(KNN_EXP) || !defined(SFMHIP_EXPERIMENTS)
#undef KNN_EXP
#define KNN_EXP 0              // timing experiments only (SFMHIP_EXPERIMENTS builds; wrong results): 1 = no staging / barriers after the first block
#endif
    if (nblocks > 0) g_stage(B0{}, 0);
    __syncthreads();
    for (int blk = 0; blk < nblocks; blk += 2) {
        if (blk + 1 < nblocks && !(KNN_EXP & 1)) g_stage(B1{}, blk + 1);
        compute(B0{}, blk);
        if (!(KNN_EXP & 1)) __syncthreads();
        if (blk + 1 >= nblocks) break;
        if (blk + 2 < nblocks && !(KNN_EXP & 1)) g_stage(B0{}, blk + 2);
        compute(B1{}, blk + 1);
        if (!(KNN_EXP & 1)) __syncthreads();
    }

    // Merge across the 32 lanes that share a query row, through LDS (the staging buffers are free after the last
    // barrier): every lane parks its 16 (best1, best2) pairs, then lane L scans half of row L>>1's 32 parked pairs
    // in ascending lane order with 32-bit compares -- (key, lane) ordering IS (distance, train index) ordering, and a
    // strict < keeps the lower lane on ties -- and only the two winners are widened to 64-bit global keys.
    // (The previous 5-level xor-shuffle merge on 64-bit keys was ~2000 instructions per wave, a third of the kernel.)
    {
        int2* wk = (int2*)lds + wave * (32 * 33);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
            wk[row * 33 + l31] = make_int2(best1[i], best2[i]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int row = lane >> 1, side = lane & 1;
        int m1 = INT_MAX, m2 = INT_MAX, i1 = 0, i2 = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int l = side * 16 + j;
            const int2 v = wk[row * 33 + l];
            const bool c1 = v.x < m1, c2 = v.x < m2;
            m2 = c1 ? m1 : (c2 ? v.x : m2); i2 = c1 ? i1 : (c2 ? l : i2);
            m1 = c1 ? v.x : m1;             i1 = c1 ? l : i1;
            const bool c3 = v.y < m2;        // v.y >= v.x: it can only displace the runner-up
            m2 = c3 ? v.y : m2;             i2 = c3 ? l : i2;
        }
        // fold the odd lane (upper 16 source lanes) into the even one; ties stay with the even lane's lower source lanes
        const int o1 = __shfl_xor(m1, 1), oi1 = __shfl_xor(i1, 1), o2 = __shfl_xor(m2, 1), oi2 = __shfl_xor(i2, 1);
        {
            const bool c1 = o1 < m1, c2 = o1 < m2;
            m2 = c1 ? m1 : (c2 ? o1 : m2); i2 = c1 ? i1 : (c2 ? oi1 : i2);
            m1 = c1 ? o1 : m1;             i1 = c1 ? oi1 : i1;
            const bool c3 = o2 < m2;
            m2 = c3 ? o2 : m2;             i2 = c3 ? oi2 : i2;
        }
        if (side == 0) {
            const int qrow = q0 + row;
            const int qn = ((const int32_t __attribute__((address_space(1)))*)(uintptr_t)pd.qn)[qrow];
            long long k1 = KEY_INVALID, k2 = KEY_INVALID;
            if (m1 != INT_MAX) k1 = ((long long)((m1 >> 7) + qn) << 32) | (unsigned int)(t_begin + (m1 & 127) * 32 + i1);
            if (m2 != INT_MAX) k2 = ((long long)((m2 >> 7) + qn) << 32) | (unsigned int)(t_begin + (m2 & 127) * 32 + i2);
            long long* o = part + 2 * (pd.part_off + (long long)qrow * pd.nchunks + chunk);
            o[0] = k1; o[1] = k2;
        }
    }
}

#define DISTMAT_WAVES 4

// Correctly rounded sqrtf for an integer-valued float in [0, 2^24): Markstein's fma correction on the reciprocal square root --
// y = v_rsq_f32(x) (<= 1 ulp), g = x y, h = y / 2, d = x - g g (exact sign through the fma), result g + d h.  Five VALU issue
// slots beside the transcendental (round 2's v_sqrt_f32 + neighbour test: eight); x = 0 goes through y = rsq(1).  Verified
// against sqrtf for every integer < 2^24 on the device (tests/test_match_gpu.py::test_exact_sqrt_all_integers,
// experiments/sqrt_variants.hip: the v_sqrt + rcp form of the same correction fails at 2^24 - 1).
__device__ __forceinline__ float sqrt_exact_int(float x)
{
    const float y = __builtin_amdgcn_rsqf(fmaxf(x, 1.0f));
    const float g = x * y, h = 0.5f * y;
    const float d = fmaf(-g, g, x);
    return fmaf(d, h, g);
}

// ------------------------------------------------------------------------------------------------
// materialised distance matrix on the int8 MFMA path (HBM-write-bound: 4 B out per 2*dim ops).
// Operands are swapped (A = trains from LDS, B = queries in registers) so that each lane owns one query
// column and 4 x 4 CONSECUTIVE trains: every store is a 16-byte store and four consecutive store
// instructions complete a 128-byte line of each of the wave's 32 query rows.
// grid = (query blocks of 128, train super-blocks of TB*128 rows), block = 256.
// ------------------------------------------------------------------------------------------------
template <int KS, int NW>          // NW waves per workgroup = 32 NW query rows against one 128-train block
__global__ __launch_bounds__(64 * NW) void distmat_i8_kernel(const int8_t* __restrict__ Q, const int32_t* __restrict__ qnorm,
                                                         const int8_t* __restrict__ T, const int32_t* __restrict__ tnorm,
                                                         int nq, int nq_pad, int nt, int nt_pad, int blocks_per_wg,
                                                         float* __restrict__ dist, size_t ldd, int vec_ok, int parity_mode, int exp_mode_arg)
{
#ifdef SFMHIP_EXPERIMENTS
    const int exp_mode = exp_mode_arg;      // timing experiments (SFMHIP_EXP_DISTMAT; results are wrong with it set)
#else
    constexpr int exp_mode = 0;             // release build: the experiment branches below fold away
    (void)exp_mode_arg;
#endif
    constexpr int DP = 32 * KS;
    constexpr int CH = DP / 16;
    constexpr int NT = 64 * NW, QROWS = 32 * NW;
    constexpr int PASSES = (128 * CH + NT - 1) / NT;
    // one 128-row train block + its norms + a per-wave output slab: 35 KB at KS = 4, four workgroups per CU (the double-buffered
    // form of round 2 took 51 KB = three; a workgroup works on ONE train block in the shipped configuration, so the second buffer
    // bought nothing, and the write stream wants as many waves with stores in flight as it can get, profiles/README.md round 3)
    __shared__ __attribute__((aligned(16))) unsigned char lds[128 * DP + 128 * 4 + NW * 32 * 36 * 4];
    float* stage_out = (float*)(lds + 128 * DP + 128 * 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    // XCD-banded work mapping (speed only): workgroups are dealt round-robin to the 8 XCDs, so workgroup L works on the
    // band of query blocks qb = 8 (slot / n_tb) + L % 8 and walks along it (tb = slot % n_tb, slot = L / 8): horizontally
    // adjacent tiles -- which share the 128-B lines straddling their common edge whenever the row stride is not a
    // multiple of 32 floats (the reference's dense 10000-column cv::Mat) -- pass through ONE L2 one after the other and
    // leave it as whole lines.  Pure-writer check (experiments/wbw4.hip): 103 -> 88 us at stride 10000, 82 -> 76 us aligned.
    // parity_mode (row stride = 64 B mod 128 B, the reference's dense 10000-column cv::Mat: odd rows start half-way into a
    // 128-B line): a workgroup takes 128 rows of ONE parity out of 256 consecutive ones and the odd ones shift their train
    // window 16 columns down, so that every 512-B row segment a wave stores starts on a line boundary and no line is
    // shared between two workgroups (93-103 us -> the aligned layout's 77-80 us at 10k x 10k, profiles/README.md).
    const int n_qb = parity_mode ? 2 * ((nq + 2 * QROWS - 1) / (2 * QROWS)) : (nq + QROWS - 1) / QROWS;
    const int n_tb = (nt_pad / 128 + (parity_mode ? 1 : 0) + blocks_per_wg - 1) / blocks_per_wg;
    const int L = blockIdx.x, slot = L >> 3;
    const int qb = (slot / n_tb) * 8 + (L & 7), tb = slot % n_tb;
    if (qb >= n_qb) return;
    const int qbase = parity_mode ? (qb >> 1) * (2 * QROWS) + (qb & 1) : qb * QROWS, qstep = parity_mode ? 2 : 1;      // row i of the tile: qbase + qstep * i
    const int q0i = wave * 32;
    const int t_begin = tb * blocks_per_wg * 128 - ((parity_mode && (qb & 1)) ? 16 : 0);
    int nblocks = (nt_pad - t_begin + 127) / 128; if (nblocks > blocks_per_wg) nblocks = blocks_per_wg;
    if (nblocks <= 0 || t_begin >= nt) return;

    const int qrow = qbase + qstep * (q0i + l31);
    const int qrow_ld = qrow < nq_pad ? qrow : nq_pad - 1;
    v4i qfrag[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        qfrag[ks] = (exp_mode & 32) ? v4i{ 1, 2, 3, 4 } : *(const v4i*)(Q + (size_t)qrow_ld * DP + 32 * ks + 16 * half);      // 32: no operand loads (experiment)
    // d^2 = (|q|^2 + |t|^2) - 2 q.t in float: every term an integer below 2^24, so the add and the fma are exact -- one cvt, half a
    // packed add and half a packed fma per element (the integer form: add, shift, subtract, cvt; an inline-asm v_mad_i32_i24 read the
    // MFMA result without the wait states the hazard recogniser gives real instructions and returned stale values)
    const float qn = (float)qnorm[qrow_ld];

    float* lds_norm = (float*)(lds + 128 * DP);
    v4i stage[PASSES];
    int stage_norm = 0;
    auto g_load = [&](int blk) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int cid = p * NT + tid, r = (cid / CH) & 127, c = cid % CH;
            int tr = t_begin + blk * 128 + r; tr = tr < 0 ? 0 : (tr < nt_pad ? tr : nt_pad - 1);       // shifted windows reach 16 rows past either end
            stage[p] = (exp_mode & 32) ? v4i{ 1, 2, 3, 4 } : *(const v4i*)(T + (size_t)tr * DP + 16 * c);
        }
        if (tid < 128) { int tr = t_begin + blk * 128 + tid; tr = tr < 0 ? 0 : (tr < nt_pad ? tr : nt_pad - 1); stage_norm = tnorm[tr]; }
    };
    auto l_store = [&]() {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int cid = p * NT + tid, r = cid / CH, c = cid % CH;
            if (cid < 128 * CH) *(v4i*)(lds + r * DP + 16 * (c ^ ((r >> 1) & (CH - 1)))) = stage[p];
        }
        if (tid < 128) lds_norm[tid] = (float)stage_norm;
    };

    g_load(0); l_store();
    __syncthreads();
    for (int blk = 0; blk < nblocks; ++blk) {
        if (blk + 1 < nblocks) g_load(blk + 1);
        // tile by tile: one 16-register accumulator live at a time (<= 128 VGPRs = four waves per SIMD; the other waves' stores and
        // epilogues cover this wave's MFMA latency, which round 2 covered with four chains in flight at three waves per SIMD)
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
            const int r = tile * 32 + l31;
            v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (!(exp_mode & 16))           // 16: store-only experiment (no matrix products)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int c = 2 * ks + half;
                const v4i a = *(const v4i*)(lds + r * DP + 16 * (c ^ ((r >> 1) & (CH - 1))));
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qfrag[ks], acc, 0, 0, 0);
            }
            // C[row = train (reg), col = query (lane&31)]: regs 4g..4g+3 <-> trains 8g + 4 half + {0,1,2,3}.
            // The tile goes through a per-wave LDS slab [32 queries][36 floats] so that every global store instruction
            // writes 8 query rows x 128 contiguous bytes (full cache lines) instead of 32 scattered 32-byte pieces.
            float* slab = stage_out + wave * (32 * 36);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int tl = tile * 32 + 8 * g + 4 * half;
                const float4 tn = *(const float4*)(lds_norm + tl);
                float4 o;
                o.x = fmaf(-2.0f, (float)acc[4 * g + 0], qn + tn.x); o.y = fmaf(-2.0f, (float)acc[4 * g + 1], qn + tn.y);
                o.z = fmaf(-2.0f, (float)acc[4 * g + 2], qn + tn.z); o.w = fmaf(-2.0f, (float)acc[4 * g + 3], qn + tn.w);
                if (exp_mode & 16) { o.x = tn.x; o.y = tn.y; o.z = tn.z; o.w = qn; }
                else if (exp_mode & 1) { }
                else if (exp_mode & 4) { o.x = sqrtf(o.x); o.y = sqrtf(o.y); o.z = sqrtf(o.z); o.w = sqrtf(o.w); }
                else { o.x = sqrt_exact_int(o.x); o.y = sqrt_exact_int(o.y); o.z = sqrt_exact_int(o.z); o.w = sqrt_exact_int(o.w); }
                *(float4*)(slab + l31 * 36 + 8 * g + 4 * half) = o;
            }
            const int tg0 = t_begin + blk * 128 + tile * 32;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = pass * 8 + (lane >> 3), col = (lane & 7) * 4;
                const float4 o = *(const float4*)(slab + row * 36 + col);
                const int qr = qbase + qstep * (q0i + row), tg = tg0 + col;
                if (qr < nq && tg >= 0 && !((exp_mode & 2) && o.x != -12345.0f)) {
                    float* dst = dist + (size_t)qr * ldd + tg;
                    if (vec_ok && tg + 3 < nt) {
                        { v4f ov = { o.x, o.y, o.z, o.w }; if (exp_mode & 8) __builtin_nontemporal_store(ov, (v4f*)dst); else *(v4f*)dst = ov; }      // plain stores: a 400 MB write stream measures 5.5 TB/s plain vs 4.9 nontemporal (experiments/wbw2.hip)
                    } else {
                        if (tg + 0 < nt) dst[0] = o.x;
                        if (tg + 1 < nt) dst[1] = o.y;
                        if (tg + 2 < nt) dst[2] = o.z;
                        if (tg + 3 < nt) dst[3] = o.w;
                    }
                }
            }
        }
        if (blk + 1 < nblocks) { __syncthreads(); l_store(); __syncthreads(); }      // (more than one block per workgroup: experiments only)
    }
}

__global__ void sqrt_check_kernel(int* __restrict__ mismatches)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // 2^24 values
    const float x = (float)i;
    if (__float_as_int(sqrt_exact_int(x)) != __float_as_int(sqrtf(x))) atomicAdd(mismatches, 1);
}

// ------------------------------------------------------------------------------------------------
// exact fp32 path (general float descriptors, and re-scoring of rows flagged by the merge).
// One block handles QR query rows against the trains of one chunk; thread = train row.
// Accumulation order = OpenCV SSE2 normL2Sqr_ (16 partial sums, mul then add, no FMA), then sqrtf (rn).
// key = (float bits of the distance << 32) | train index.
// row_list == nullptr: rows are blockIdx.x*QR + r.  Otherwise rows come from row_list[0..*row_count).
// ------------------------------------------------------------------------------------------------
template <int QR, bool ALIGNED, bool STORE_ALL>
__global__ __launch_bounds__(256) void knn2_exact_f32_kernel(const PairDesc* __restrict__ pairs, long long* __restrict__ part,
                                                             const int* __restrict__ row_list, const int* __restrict__ row_count,
                                                             float* __restrict__ dist_out, size_t ldd)
{
    extern __shared__ __attribute__((aligned(16))) float sm_q[];   // QR x dim query rows
    __shared__ long long red[4][QR][2];
    const PairDesc pd = pairs[blockIdx.z];
    const int dim = pd.dim, chunk = blockIdx.y;
    if (chunk >= pd.nchunks) return;
    // (address-space casts: pointers read out of the PairDesc table are generic to the compiler -> flat_load otherwise)
    typedef const float __attribute__((address_space(1)))* gf32;
    const gf32 QF = (gf32)(uintptr_t)pd.qf, TF = (gf32)(uintptr_t)pd.tf;
    const int total = row_list ? row_count[blockIdx.z] : pd.nq;
    const int tid = threadIdx.x;
    const int t_begin = chunk * pd.chunk_rows;
    int t_end = t_begin + pd.chunk_rows; if (t_end > pd.nt) t_end = pd.nt;
    // groups of QR rows, strided over the grid (the re-score launch uses a small grid: its list is normally empty)
    for (int grp = blockIdx.x; grp * QR < total; grp += gridDim.x) {
        int rows[QR];
#pragma unroll
        for (int r = 0; r < QR; ++r) {
            const int k = grp * QR + r;
            rows[r] = k < total ? (row_list ? row_list[pd.list_off + k] : k) : -1;
        }
        __syncthreads();
        for (int r = 0; r < QR; ++r) {
            const int row = rows[r] < 0 ? 0 : rows[r];
            for (int k = tid; k < dim; k += 256) sm_q[r * dim + k] = QF[(size_t)row * pd.ldq + k];
        }
        __syncthreads();

        long long k1[QR], k2[QR];
#pragma unroll
        for (int r = 0; r < QR; ++r) { k1[r] = KEY_INVALID; k2[r] = KEY_INVALID; }

        for (int j = t_begin + tid; j < t_end; j += 256) {
            const gf32 b = TF + (size_t)j * pd.ldt;
            float acc[QR][16];
#pragma unroll
            for (int r = 0; r < QR; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[r][i] = 0.0f;
            int k = 0;
            for (; k <= dim - 16; k += 16) {
                float bb[16];
                if (ALIGNED) {
                    typedef float f4v __attribute__((ext_vector_type(4)));
                    typedef const f4v __attribute__((address_space(1)))* gf4;
                    const f4v b0 = *(gf4)(b + k), b1 = *(gf4)(b + k + 4), b2 = *(gf4)(b + k + 8), b3 = *(gf4)(b + k + 12);
                    bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
                    bb[8] = b2.x; bb[9] = b2.y; bb[10] = b2.z; bb[11] = b2.w; bb[12] = b3.x; bb[13] = b3.y; bb[14] = b3.z; bb[15] = b3.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) bb[i] = b[k + i];
                }
#pragma unroll
                for (int r = 0; r < QR; ++r)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float t = sm_q[r * dim + k + i] - bb[i];
                        const float m = t * t;
                        acc[r][i] = m + acc[r][i];
                    }
            }
#pragma unroll
            for (int r = 0; r < QR; ++r) {
                float s[4];
#pragma unroll
                for (int l = 0; l < 4; ++l)
                    s[l] = ((acc[r][l] + acc[r][4 + l]) + acc[r][8 + l]) + acc[r][12 + l];
                float d = (s[0] + s[2]) + (s[1] + s[3]);
                for (int kk = k; kk < dim; ++kk) {
                    const float t = sm_q[r * dim + kk] - b[kk];
                    const float m = t * t;
                    d = d + m;
                }
                const float dist = sqrtf(d);
                if (STORE_ALL) {
                    if (rows[r] >= 0) dist_out[(size_t)rows[r] * ldd + j] = dist;
                } else {
                    const long long key = ((long long)__float_as_int(dist) << 32) | (unsigned int)j;
                    if (key < k2[r]) { if (key < k1[r]) { k2[r] = k1[r]; k1[r] = key; } else k2[r] = key; }
                }
            }
        }
        if (STORE_ALL) continue;
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int r = 0; r < QR; ++r) {
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const long long o1 = shfl_xor_ll(k1[r], off), o2 = shfl_xor_ll(k2[r], off);
                merge2(k1[r], k2[r], o1, o2);
            }
            if (lane == 0) { red[wave][r][0] = k1[r]; red[wave][r][1] = k2[r]; }
        }
        __syncthreads();
        if (tid < QR && rows[tid] >= 0) {
            long long a1 = red[0][tid][0], a2 = red[0][tid][1];
            for (int w = 1; w < 4; ++w) merge2(a1, a2, red[w][tid][0], red[w][tid][1]);
            long long* o = part + 2 * (pd.part_off + (long long)rows[tid] * pd.nchunks + chunk);
            o[0] = a1; o[1] = a2;
        }
    }
}

__device__ __forceinline__ int bcnt_acc(unsigned x, int acc)
{
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// ------------------------------------------------------------------------------------------------
// Hamming2 kNN-2 (cv::NORM_HAMMING2 on CV_8U rows: number of non-zero 2-bit cells of a^b).
// thread = query row (16 dwords in registers), train rows streamed through the scalar path
// (wave-uniform address -> s_load), chunked over blockIdx.y.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn2_hamming2_kernel(const PairDesc* __restrict__ pairs, long long* __restrict__ part)
{
    const PairDesc pd = pairs[blockIdx.z];
    const int chunk = blockIdx.y;
    if ((int)blockIdx.x * 256 >= pd.nq_pad || chunk >= pd.nchunks) return;
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int rrow = row < pd.nq_pad ? row : 0;
    // Train rows go through the scalar path: the address is wave-uniform, and a pointer in the constant address space
    // makes hipcc emit s_load_dwordx16 into SGPRs (a generic pointer out of the PairDesc table became 64-lane flat_load
    // broadcasts with a full vmcnt(0) wait per row).  Two rows per trip, the next pair requested before this one is used.
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    typedef const u4v __attribute__((address_space(1)))* gu4;
    typedef const u4v __attribute__((address_space(4)))* cu4;
    const gu4 Q = (gu4)(uintptr_t)pd.q;
    const cu4 T = (cu4)(uintptr_t)pd.t;
    const u4v q0 = Q[(size_t)rrow * 4 + 0], q1 = Q[(size_t)rrow * 4 + 1], q2 = Q[(size_t)rrow * 4 + 2], q3 = Q[(size_t)rrow * 4 + 3];
    const int t_begin = chunk * pd.chunk_rows;
    int t_end = t_begin + pd.chunk_rows; if (t_end > pd.nt) t_end = pd.nt;
    int best1 = INT_MAX, best2 = INT_MAX;
    // rows are [L0..L7 | H0..H7] (prep_hamming_kernel): q0,q1 = low-bit dwords, q2,q3 = high-bit dwords
    // popcount with accumulate (v_bcnt_u32_b32 d, x, acc) in two chains of four: hipcc turns a sum of __popc into eight plain counts
    // + three v_add3 (11 instructions for 8 dwords; this: 8 + 1), and the kernel runs at the VALU issue rate (profiles/README.md, round 3)
#define H2X(ql, tl, qh, th) (((ql) ^ (tl)) | ((qh) ^ (th)))
#define ROW_DIST(t0, t1, t2, t3)                                                                                                   \
    (bcnt_acc(H2X(q0.w, t0.w, q2.w, t2.w), bcnt_acc(H2X(q0.z, t0.z, q2.z, t2.z), bcnt_acc(H2X(q0.y, t0.y, q2.y, t2.y), bcnt_acc(H2X(q0.x, t0.x, q2.x, t2.x), 0)))) \
   + bcnt_acc(H2X(q1.w, t1.w, q3.w, t3.w), bcnt_acc(H2X(q1.z, t1.z, q3.z, t3.z), bcnt_acc(H2X(q1.y, t1.y, q3.y, t3.y), bcnt_acc(H2X(q1.x, t1.x, q3.x, t3.x), 0)))))
#define TOP2(d, jrel)                                                                                                              \
    do {                                                                                                                           \
        const int key = ((d) << 22) | (jrel);                                                                                      \
        const int lo = best1 < best2 ? best1 : best2, hi = best1 < best2 ? best2 : best1;                                          \
        const int t = hi < key ? hi : key;                                                                                         \
        best2 = lo > t ? lo : t;                            /* med3(best1, best2, key) */                                          \
        best1 = best1 < key ? best1 : key;                                                                                         \
    } while (0)
    // rows_pad is a multiple of 256 and pad rows are zero: reading one row pair past t_end stays inside the buffer
    int j = t_begin;
    u4v a0 = T[(size_t)j * 4 + 0], a1 = T[(size_t)j * 4 + 1], a2 = T[(size_t)j * 4 + 2], a3 = T[(size_t)j * 4 + 3];
    u4v b0 = T[(size_t)j * 4 + 4], b1 = T[(size_t)j * 4 + 5], b2 = T[(size_t)j * 4 + 6], b3 = T[(size_t)j * 4 + 7];
    for (; j + 2 <= t_end; j += 2) {
        const int jn = (j + 2 < pd.nt_pad - 1) ? j + 2 : j;      // next pair (clamped inside the padded buffer)
        const u4v c0 = T[(size_t)jn * 4 + 0], c1 = T[(size_t)jn * 4 + 1], c2 = T[(size_t)jn * 4 + 2], c3 = T[(size_t)jn * 4 + 3];
        const u4v e0 = T[(size_t)jn * 4 + 4], e1 = T[(size_t)jn * 4 + 5], e2 = T[(size_t)jn * 4 + 6], e3 = T[(size_t)jn * 4 + 7];
        const int da = ROW_DIST(a0, a1, a2, a3), db = ROW_DIST(b0, b1, b2, b3);
        TOP2(da, j - t_begin);
        TOP2(db, j + 1 - t_begin);
        a0 = c0; a1 = c1; a2 = c2; a3 = c3; b0 = e0; b1 = e1; b2 = e2; b3 = e3;
    }
    if (j < t_end) {
        const int da = ROW_DIST(a0, a1, a2, a3);
        TOP2(da, j - t_begin);
    }
#undef TOP2
#undef ROW_DIST
#undef H2X
    if (row < pd.nq_pad) {
        long long k1 = KEY_INVALID, k2 = KEY_INVALID;
        if (best1 != INT_MAX) k1 = ((long long)(best1 >> 22) << 32) | (unsigned int)(t_begin + (best1 & 0x3fffff));
        if (best2 != INT_MAX) k2 = ((long long)(best2 >> 22) << 32) | (unsigned int)(t_begin + (best2 & 0x3fffff));
        long long* o = part + 2 * (pd.part_off + (long long)row * pd.nchunks + chunk);
        o[0] = k1; o[1] = k2;
    }
}

// ------------------------------------------------------------------------------------------------
// Hamming2 kNN-2 on the matrix cores: v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 operands (rows from prep_hamming_fp4_kernel).
// grid = (query blocks of 64 NW rows, chunks, pairs [padded to 8]), block = NW waves x 64 query rows; NW = 8 (512 rows, one workgroup
// per CU) is what runs: a staged train row then serves 512 queries (half the L2 requests of NW = 4, which measures the same time).
//   * the wave's 64 query rows stay in registers for the whole chunk: 2 tiles x 12 K-steps x 4 dwords = 96 VGPRs, sign bits
//     flipped (the accumulator then holds MINUS the dot product), spare values replaced by the weights below;
//   * train rows stream through LDS by LDS-DMA, 64 rows (24 KB, back to back) per stage, two stages in flight; the 16-byte chunks of
//     a row are XOR-swizzled inside their group of 8 by (row >> 1) & 7, on the global side, so the 16 lanes of a ds_read_b128 group
//     (16 rows, 384 B apart) find 16 different bank slots;
//   * every train fragment read from LDS feeds TWO MFMAs (both query tiles): 64 B/clk/CU of LDS reads at the MFMA rate;
//   * block scales (E8M0, one per lane per instruction) make the accumulator the top-2 key itself: query blocks 0..22 carry 2^6,
//     so acc = -64 dot; block 23 (query scale 2^2) adds the train's 8-bit tile index (weights {.5,.5,1,2,4,4,4,4,4} x the row's
//     bit values).  Real rows: key + 192 cells = 256 distance + tile (< 2^16).  Rows past the end of a set carry 6.0 in 27 spare
//     values that meet 6.0 on the query side, and their last K-step is scaled by 2^8 on the train side (the scale is read off the
//     fragment itself): keys above 3e6, never selected while a real row is left.  Two VALU ops per distance (below).
//   * chunks are power-of-two sized and aligned, <= 8192 rows, so the 8-bit tile index is monotone inside a chunk.
// Partial output as knn2_hamming2_kernel: two keys (distance << 32 | train index) per (query row, chunk).
// ------------------------------------------------------------------------------------------------
typedef int   v8i  __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int NW>        // waves per workgroup (4 or 8), 64 query rows each
__global__ __launch_bounds__(64 * NW, 2) void knn2_hamming2_fp4_kernel(const PairDesc* __restrict__ pairs, long long* __restrict__ part, int n_pairs)
{
#ifdef __HIP_DEVICE_COMPILE__      // the host pass only needs the launch stub: instantiating a TEMPLATE body with gfx950 builtins there drops the stub
#ifndef H4_TROWS
#define H4_TROWS 64
#endif
    constexpr int RB = H4_ROW_BYTES, TROWS = H4_TROWS, BUF_BYTES = TROWS * RB, BUF_STRIDE = TROWS * 512, TILE_BYTES = 32 * RB, QB = 64 * NW;
    constexpr int NG = 12 * (TROWS / 32);                    // K-steps per stage
#if !defined(H4_EXP) || !defined(SFMHIP_EXPERIMENTS)
#undef H4_EXP
#define H4_EXP 0               // timing experiments only (SFMHIP_EXPERIMENTS builds; wrong results): 1 = no staging after the first stage, 2 = no barriers, 4 = no top-2 updates, 8 = no fragment reads after the first three, 32 = one stage per workgroup, 64 = no merge epilogue, 128 = no query loads
#endif

    constexpr int PASSES = BUF_BYTES / (1024 * NW);          // LDS-DMA instructions per wave and stage: 6 (4 waves) / 3 (8 waves)
    constexpr int MERGE_BYTES = NW * 32 * 33 * 8, STAGE_BYTES = BUF_STRIDE + BUF_BYTES;
    static_assert(BUF_BYTES % (1024 * NW) == 0 && BUF_BYTES <= BUF_STRIDE, "whole 1 KB pieces per wave");
    __shared__ __attribute__((aligned(16))) unsigned char lds[STAGE_BYTES > MERGE_BYTES ? STAGE_BYTES : MERGE_BYTES];
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int per_pair = gridDim.x * gridDim.y, slot = lin >> 3;
    const int pair = (slot / per_pair) * 8 + (lin & 7), rest = slot % per_pair;       // pair -> XCD, as knn2_i8_kernel
    if (pair >= n_pairs) return;
    const PairDesc pd = pairs[pair];
    const int qb = rest % gridDim.x, chunk = rest / gridDim.x;
    if (qb * QB >= pd.nq_pad || chunk >= pd.nchunks) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    typedef const char __attribute__((address_space(1)))* gbytes;
    typedef char __attribute__((address_space(3)))* lbytes;
    typedef const v4i __attribute__((address_space(1)))* gv4;
    const gbytes Q = (gbytes)(uintptr_t)pd.q;
    const gbytes T = (gbytes)(uintptr_t)pd.t;
    const int t_begin = chunk * pd.chunk_rows;
    int t_end = t_begin + pd.chunk_rows; if (t_end > pd.nt_pad) t_end = pd.nt_pad;
    const int nblocks = (H4_EXP & 32) ? 1 : (t_end - t_begin) / TROWS;        // 32: one stage only (what a workgroup costs before and after its loop)
    const int q0 = qb * QB + wave * 64;
    // sets are padded to 256 rows: with 512-row query blocks the upper waves of the last block may have no rows; they still stage and
    // meet the barriers, on the block's first rows, and write nothing
    const bool has_rows = q0 < pd.nq_pad;
    const int q0r = has_rows ? q0 : qb * QB;

    // stationary operand
    v4i afrag[2][12];
#pragma unroll
    for (int at = 0; at < 2; ++at)
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            v4i a = (H4_EXP & 128) ? (v4i){ lane, s, at, 7 } : *(gv4)(Q + (size_t)(q0r + 32 * at + l31) * RB + 16 * (2 * s + half));       // 128: no query loads
            a ^= (v4i){ (int)0x88888888, (int)0x88888888, (int)0x88888888, (int)0x88888888 };
            if (s == 11) {
                if (half == 0) a[3] = (a[3] & 0x0000ffff) | 0x77770000;                                  // values 732..735: 6.0
                else a = (v4i){ 0x66664211, 0x77777776, 0x77777777, 0x77777777 };                        // tile-bit weights (9 values), then 6.0
            }
            asm volatile("" : "+v"(a));
            afrag[at][s] = a;
        }
    // block scales, one register each: query side 2^6 on the even blocks (lanes 0..31) and 2^2 on the odd ones (lanes 32..63), train
    // side 2^0 / 2^4 -- so every data block is scaled by 2^6 -- except in the last K-step, where the train side is 2^0 on both (block 23
    // = 2^2) or 2^8 on a pad row
    const int sa = half ? 129 : 133;
    const int sb = half ? 131 : 127;
    float neg_inf = -__builtin_inff();
    asm volatile("" : "+s"(neg_inf));                      // an SGPR operand, not a literal per instruction

    float best1[2][16], best2[2][16];
#pragma unroll
    for (int at = 0; at < 2; ++at)
#pragma unroll
        for (int i = 0; i < 16; ++i) { best1[at][i] = 1.0e9f; best2[at][i] = 1.0e9f; }

    // staging: the buffer is the stage's 64 rows back to back (24 KB); piece NW p + w (1 KB, one LDS-DMA instruction of wave w) covers
    // LDS bytes [1024 (NW p + w), +1024): the lane's 16 bytes are slot (pos % 384) / 16 of row pos / 384, which holds chunk
    // slot ^ ((row >> 1) & 7) (inside its group of 8) of that row.
    // Per-lane source offsets: three registers at most (4 waves: pieces 12 apart are 32 rows apart and repeat the pattern).  The
    // kernel sits close to the 256-register limit, and ONE spilled fragment once made the loop wait for its reload with
    // s_waitcnt vmcnt(0) -- i.e. for the LDS-DMA pieces just issued, every stage: 22 % of the kernel.
    unsigned g_off[PASSES < 3 ? PASSES : 3];
#pragma unroll
    for (int p = 0; p < (PASSES < 3 ? PASSES : 3); ++p) {
        const int pos = (NW * p + wave) * 1024 + lane * 16, r = pos / RB, sl = (pos % RB) / 16;
        g_off[p] = (unsigned)(r * RB + 16 * ((sl & ~7) | ((sl ^ (r >> 1)) & 7)));
    }
    auto g_stage = [&](int buf, int blk) {                 // buf: 0 / 1, wave-uniform
        const gbytes blk_base = (gbytes)(T + (size_t)(t_begin + blk * TROWS) * RB);       // wave-uniform
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            unsigned long long pb = (unsigned long long)(uintptr_t)(blk_base + (p / 3) * (8 * NW * RB));      // three pieces per wave = 8 NW rows
            asm volatile("" : "+s"(pb));
            __builtin_amdgcn_global_load_lds((gbytes)pb + g_off[p % 3], (lbytes)(lds + buf * BUF_STRIDE + (NW * p + wave) * 1024), 16, 0, 0);
        }
    };
    // operand reads: lane (row l31 of the tile, K half): chunk 2s + half of its row, swizzled
    int rd_off[4];
    {
        const int base_lane = l31 * RB, k16 = 16 * ((l31 >> 1) & 7);
#pragma unroll
        for (int j = 0; j < 4; ++j) rd_off[j] = base_lane + ((32 * j + 16 * half) ^ k16);
    }
    // One stage = two tiles of 32 trains = 24 K-steps out of the buffer rd_off[] points into, as ONE software-pipelined stream:
    //   * train fragments are read H4_AHEAD K-steps ahead of their MFMAs, across the tile boundary;
    //   * the top-2 update of a finished tile (64 VALU ops per wave) is spread over K-steps 1..11 of the NEXT tile, six ops behind
    //     each MFMA pair -- an MFMA holds the SIMD's issue port for 8 of its 32 cycles, so they cost nothing -- which needs two
    //     accumulator sets: tile 0 of a stage fills set A while set B (tile 1 of the stage before) is consumed, and vice versa;
    //   * two VALU ops per distance, in place: runner-up = med3(best, runner-up, key), best = med3(best, key, -inf) = min
    //     (v_min_f32 through fminf() would canonicalise both inputs first).  asm, so that the results stay in their inputs'
    //     registers (hipcc renamed them into a second generation of best[] and spilled the stationary operand).  hipcc pads
    //     nothing for asm: a set is first read five MFMAs (>= 40 cycles) after its last write -- 11 wait states are required
    //     after an 8-pass MFMA -- and the flush after the loop pads explicitly.
    // The scheduling barriers pin this order (hipcc otherwise hoists all twelve reads of a tile, 48 registers, above the first MFMA).
    v16f accA0, accA1, accB0, accB1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { accB0[i] = 1.0e9f; accB1[i] = 1.0e9f; }
#define H4_TOP2(B1v, B2v, KEY) asm volatile("v_med3_f32 %1, %0, %1, %2\n\tv_med3_f32 %0, %0, %2, %3" : "+v"(B1v), "+v"(B2v) : "v"(KEY), "s"(neg_inf))
    auto top2_of = [&](v16f& p0, v16f& p1, int v) {          // value v = 0..31 of a finished tile's two accumulators
        if (v < 16) H4_TOP2(best1[0][v], best2[0][v], p0[v]);
        else if (v < 32) H4_TOP2(best1[1][v - 16], best2[1][v - 16], p1[v - 16]);
    };
    auto compute = [&]() {
        auto rd = [&](int g) { return *(const v4i*)(lds + rd_off[(g % 12) & 3] + ((g / 12) * TILE_BYTES + 128 * ((g % 12) >> 2))); };
#ifndef H4_AHEAD
#define H4_AHEAD 1             // K-steps a train fragment is read ahead of its MFMAs: 1, 2 and 3 measured the same (1.36-1.37 ms), 1 leaves 8 registers of margin
#endif
        v4i bq[H4_AHEAD + 1];
#pragma unroll
        for (int g = 0; g < H4_AHEAD; ++g) bq[g] = rd(g);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int tile = (g / 12) & 1, s = g % 12;     // accumulator set by tile parity
            if (g + H4_AHEAD < NG && !((H4_EXP & 8) && g >= 1)) bq[(g + H4_AHEAD) % (H4_AHEAD + 1)] = rd(g + H4_AHEAD);
            const v4i b4 = bq[g % (H4_AHEAD + 1)];
            const v8i b8 = { b4[0], b4[1], b4[2], b4[3], 0, 0, 0, 0 };
            const v8i a0 = { afrag[0][s][0], afrag[0][s][1], afrag[0][s][2], afrag[0][s][3], 0, 0, 0, 0 };
            const v8i a1 = { afrag[1][s][0], afrag[1][s][1], afrag[1][s][2], afrag[1][s][3], 0, 0, 0, 0 };
            v16f& c0 = tile == 0 ? accA0 : accB0;
            v16f& c1 = tile == 0 ? accA1 : accB1;
            v16f& p0 = tile == 0 ? accB0 : accA0;
            v16f& p1 = tile == 0 ? accB1 : accA1;
            int sbv = sb;
            if (s == 11) sbv = 127 + ((b4[3] >> 27) & 8);       // rows past the end: the top value of the fragment is 6.0 (bit 30 set) -> 2^8
            if (s == 0) {
                v16f z;
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = 0.0f;
                // the pending set is still being read: the first MFMA pair of a tile writes the OTHER set
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b8, z, 4, 4, 0, sa, 0, sb);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b8, z, 4, 4, 0, sa, 0, sb);
            } else {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b8, c0, 4, 4, 0, sa, 0, sbv);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b8, c1, 4, 4, 0, sa, 0, sbv);
                if (!(H4_EXP & 4)) { top2_of(p0, p1, 3 * (s - 1)); top2_of(p0, p1, 3 * (s - 1) + 1); top2_of(p0, p1, 3 * (s - 1) + 2); }
            }
            // pin the pair here: set B's second chain has no reader before the next trip of the loop, and LLVM sank all twelve of
            // its MFMAs (and their twelve train fragments: 48 registers) below the stage's last step
            asm volatile("" : "+v"(c0), "+v"(c1));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // one loop body for both buffers (the buffer is a run-time offset: an XOR on the four read addresses per stage) -- with the two
    // buffers as separate code paths hipcc moved best[] between register sets at the joins and spilled the stationary operand
    g_stage(0, 0);
    __syncthreads();
    for (int blk = 0; blk < nblocks; ++blk) {
        if (blk + 1 < nblocks && !(H4_EXP & 1)) g_stage((blk + 1) & 1, blk + 1);
        compute();
#pragma unroll
        for (int j = 0; j < 4; ++j) rd_off[j] ^= BUF_STRIDE;
        if (!(H4_EXP & 2)) __syncthreads();
    }
    if (H4_EXP & 2) __syncthreads();
    // flush: the last tile's accumulators (set B)
    asm volatile("s_nop 11");
#pragma unroll
    for (int v = 0; v < 32; ++v) top2_of(accB0, accB1, v);
#undef H4_TOP2

    // merge across the 32 lanes that share a query row (as knn2_i8_kernel), one query tile at a time, on integer keys
    // K = key + 192 cells = 256 distance + tile for a real train (< 2^16); a pad row or the initial value is above 2^20
    // The margin between the two, spelled out (advisor, round 3): the largest real key is 256 * (4 * H4_MAX_NBYTES) + 255 = 62,719;
    // a pad row's spare values (6.0 on both sides, the train side's last K-step under a 2^8 block scale) put its key above 3e6 (see the
    // kernel's header); block 23's 23 + 4 products alone give 27 * 36 * 2^10 = 995,328.  Change H4_ROW_BYTES, the number of spare values
    // or the 2^8 scale and the k_pad = 2^20 test below no longer separates the two: the asserts pin the ingredients, and
    // tests/test_match_gpu.py::test_knn2_hamming2_matrix_core_kernel_pad_rows_never_beat_the_worst_real_row the outcome.
    static_assert(256 * (4 * H4_MAX_NBYTES) + 255 < (1 << 16), "real Hamming2 keys must stay below 2^16");
    static_assert(27 * 36 * 1024 + 768 >= (1 << 20) - (1 << 16), "pad-row keys must clear every real key by the tested margin");
    const int off_k = 768 * pd.dim, k_pad = 1 << 20;
    const int win_base = t_begin & ~8191;
    if (H4_EXP & 64) {            // 64: no merge (one store per lane keeps the registers alive)
        float sum = 0.0f;
#pragma unroll
        for (int at = 0; at < 2; ++at)
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += best1[at][i] + best2[at][i];
        if (sum == 12345.0f) part[2 * pd.part_off + tid] = 1;
        return;
    }
    int2* wk = (int2*)lds + wave * (32 * 33);
#pragma unroll
    for (int at = 0; at < 2; ++at) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
            wk[row * 33 + l31] = make_int2((int)best1[at][i] + off_k, (int)best2[at][i] + off_k);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int row = lane >> 1, side = lane & 1;
        int m1 = INT_MAX, m2 = INT_MAX, i1 = 0, i2 = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int l = side * 16 + j;
            const int2 v = wk[row * 33 + l];
            const bool c1 = v.x < m1, c2 = v.x < m2;
            m2 = c1 ? m1 : (c2 ? v.x : m2); i2 = c1 ? i1 : (c2 ? l : i2);
            m1 = c1 ? v.x : m1;             i1 = c1 ? l : i1;
            const bool c3 = v.y < m2;
            m2 = c3 ? v.y : m2;             i2 = c3 ? l : i2;
        }
        const int o1 = __shfl_xor(m1, 1), oi1 = __shfl_xor(i1, 1), o2 = __shfl_xor(m2, 1), oi2 = __shfl_xor(i2, 1);
        {
            const bool c1 = o1 < m1, c2 = o1 < m2;
            m2 = c1 ? m1 : (c2 ? o1 : m2); i2 = c1 ? i1 : (c2 ? oi1 : i2);
            m1 = c1 ? o1 : m1;             i1 = c1 ? oi1 : i1;
            const bool c3 = o2 < m2;
            m2 = c3 ? o2 : m2;             i2 = c3 ? oi2 : i2;
        }
        if (side == 0 && has_rows) {
            const int qrow = q0 + 32 * at + row;
            long long k1 = KEY_INVALID, k2 = KEY_INVALID;
            if (m1 < k_pad) k1 = ((long long)(m1 >> 8) << 32) | (unsigned int)(win_base + (m1 & 255) * 32 + i1);
            if (m2 < k_pad) k2 = ((long long)(m2 >> 8) << 32) | (unsigned int)(win_base + (m2 & 255) * 32 + i2);
            long long* o = part + 2 * (pd.part_off + (long long)qrow * pd.nchunks + chunk);
            o[0] = k1; o[1] = k2;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// merge of the per-chunk partial top-2 into idx2 / dist2.
// MODE 0: keys hold integer d^2 (int8 path): dist = sqrtf(float(d^2)); rows with 2nd d^2 >= 2^22 are
//         appended to the rescore list (their float distances may tie where the integers do not).
// MODE 1: keys hold float bits (exact path).  MODE 2: keys hold the integer Hamming distance.
// With row_list != nullptr only the listed rows are merged (after re-scoring).
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void merge_kernel(const PairDesc* __restrict__ pairs, const long long* __restrict__ part,
                             int32_t* __restrict__ idx2, float* __restrict__ dist2,
                             int* __restrict__ row_list, int* __restrict__ row_count, int use_list)
{
    const PairDesc pd = pairs[blockIdx.y];
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (use_list) {
        if (row >= row_count[blockIdx.y]) return;
        row = row_list[pd.list_off + row];
    } else if (row >= pd.nq) return;
    const long long* p = part + 2 * (pd.part_off + (long long)row * pd.nchunks);
    long long a1 = p[0], a2 = p[1];
    for (int c = 1; c < pd.nchunks; ++c) merge2(a1, a2, p[2 * c], p[2 * c + 1]);
    int i1 = (int)(a1 & 0xffffffffLL), i2 = (int)(a2 & 0xffffffffLL);
    const int h1 = (int)(a1 >> 32), h2 = (int)(a2 >> 32);
    const bool v1 = a1 != KEY_INVALID && i1 < pd.nt, v2 = a2 != KEY_INVALID && i2 < pd.nt;
    float d1, d2;
    if (MODE == 0) { d1 = sqrtf((float)h1); d2 = sqrtf((float)h2); }
    else if (MODE == 1) { d1 = __int_as_float(h1); d2 = __int_as_float(h2); }
    else { d1 = (float)h1; d2 = (float)h2; }
    const float missing = (MODE == 2) ? 2147483648.0f : FLT_MAX;
    if (!v1) { i1 = -1; d1 = missing; }
    if (!v2) { i2 = -1; d2 = missing; }
    const long long o = pd.out_off + row;
    idx2[2 * o] = i1; idx2[2 * o + 1] = i2;
    dist2[2 * o] = d1; dist2[2 * o + 1] = d2;
    if (MODE == 0 && row_list && v2 && h2 >= RESCORE_D2) {
        const int k = atomicAdd(&row_count[blockIdx.y], 1);
        row_list[pd.list_off + k] = row;
    }
}

// ------------------------------------------------------------------------------------------------
// ratio tail of match_features on the device (NViewReconstuct.cpp:880-908), one block per pair.
// fp64 compare for the ratio test, float gate: the same IEEE operations as the host code.
// Ordered compaction: matches keep query order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ratio_tail_kernel(const PairDesc* __restrict__ pairs,
                                                          const int32_t* __restrict__ idx2, const float* __restrict__ dist2,
                                                          double ratio, float floor_, float mult,
                                                          sfm_dmatch* __restrict__ matches, int max_per_pair, int32_t* __restrict__ counts)
{
    __shared__ int s_min;
    __shared__ int s_wave[16];
    __shared__ int s_running;
    const PairDesc pd = pairs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = pd.out_off;
    if (tid == 0) { s_min = __float_as_int(FLT_MAX); s_running = 0; }
    __syncthreads();
    int local = __float_as_int(FLT_MAX);
    for (int i = tid; i < pd.nq; i += 1024) {
        if (idx2[2 * (base + i)] < 0 || idx2[2 * (base + i) + 1] < 0) continue;
        const float d0 = dist2[2 * (base + i)], d1 = dist2[2 * (base + i) + 1];
        if ((double)d0 > ratio * (double)d1) continue;
        const int b = __float_as_int(d0);      // d0 >= 0: integer order == float order
        local = b < local ? b : local;
    }
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(local, off); local = o < local ? o : local; }
    if (lane == 0) atomicMin(&s_min, local);
    __syncthreads();
    const float min_dist = __int_as_float(s_min);
    const float gate = mult * (min_dist > floor_ ? min_dist : floor_);
    sfm_dmatch* out = matches + (size_t)blockIdx.x * max_per_pair;
    for (int start = 0; start < pd.nq; start += 1024) {
        const int i = start + tid;
        bool keep = false; float d0 = 0.0f; int ti = -1;
        if (i < pd.nq) {
            ti = idx2[2 * (base + i)];
            const int t2 = idx2[2 * (base + i) + 1];
            d0 = dist2[2 * (base + i)];
            const float d1 = dist2[2 * (base + i) + 1];
            keep = ti >= 0 && t2 >= 0 && !((double)d0 > ratio * (double)d1 || d0 > gate);
        }
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, total = 0;
        for (int w = 0; w < 16; ++w) { const int c = s_wave[w]; if (w < wave) woff += c; total += c; }
        const int run = s_running;
        if (keep) {
            const int k = run + woff + before;
            if (k < max_per_pair) { sfm_dmatch m; m.queryIdx = i; m.trainIdx = ti; m.imgIdx = 0; m.distance = d0; out[k] = m; }
        }
        __syncthreads();
        if (tid == 0) s_running = run + total;
        __syncthreads();
    }
    if (tid == 0) counts[blockIdx.x] = s_running < max_per_pair ? s_running : max_per_pair;
}

// ================================================================================================
// host side
// ================================================================================================
static int descset_alloc_common(sfmhip_ctx* ctx, int kind, int rows, int dim, sfmhip_descset** out)
{
    sfmhip_descset* s = new sfmhip_descset();
    s->ctx = ctx; s->kind = kind; s->rows = rows; s->dim = dim;
    s->rows_pad = round_up(rows > 0 ? rows : 1, KNN_BLOCK_ROWS);
    *out = s;
    return SFMHIP_OK;
}

static int descset_flag_slot(sfmhip_ctx* ctx, sfmhip_descset* s)
{
    if (!ctx->d_flagpool) SFM_HIP_TRY(ctx, hipMalloc((void**)&ctx->d_flagpool, sfmhip_ctx::FLAG_SLOTS * sizeof(int)));
    if (!ctx->flag_free.empty()) { s->flag_slot = ctx->flag_free.back(); ctx->flag_free.pop_back(); }
    else if (ctx->flag_next < sfmhip_ctx::FLAG_SLOTS) s->flag_slot = ctx->flag_next++;
    if (s->flag_slot >= 0) { s->d_flag = ctx->d_flagpool + s->flag_slot; return SFMHIP_OK; }
    void* q = nullptr;
    int rc = sfm_pool_get(ctx, 256, &q); if (rc) return rc;
    s->d_flag = (int*)q;
    return SFMHIP_OK;
}

// The preparation kernels leave "not every value is an integer in [0, 255]" in the sets' flags; whoever needs the verdict
// (path selection of a launch, sfmhip_descset_info) resolves all pending sets of its batch with one copy and one sync --
// a chain of N images prepared from host rows costs one round trip, not N.
static int descsets_resolve(sfmhip_ctx* ctx, sfmhip_descset* const* sets, int n)
{
    int lo = INT_MAX, hi = -1; bool any = false;
    for (int i = 0; i < n; ++i) {
        sfmhip_descset* s = sets[i];
        if (!s || !s->exact_pending) continue;
        any = true;
        if (s->flag_slot >= 0) { lo = std::min(lo, s->flag_slot); hi = std::max(hi, s->flag_slot); }
    }
    if (!any) return SFMHIP_OK;
    std::vector<int> host;
    if (hi >= 0) {
        host.resize((size_t)(hi - lo + 1));
        SFM_HIP_TRY(ctx, hipMemcpyAsync(host.data(), ctx->d_flagpool + lo, host.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    }
    SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; ++i) {
        sfmhip_descset* s = sets[i];
        if (!s || !s->exact_pending) continue;
        int flag = 0;
        if (s->flag_slot >= 0) flag = host[(size_t)(s->flag_slot - lo)];
        else SFM_HIP_TRY(ctx, hipMemcpy(&flag, s->d_flag, sizeof(int), hipMemcpyDeviceToHost));
        s->exact_u8 = flag == 0; s->exact_pending = false;
    }
    return SFMHIP_OK;
}

static int descset_prepare_l2(sfmhip_ctx* ctx, sfmhip_descset* s)
{
    s->dim_pad = s->dim <= 32 ? 32 : (s->dim <= 64 ? 64 : round_up(s->dim, 128));   // int8 row bytes: 32, 64 or 128
    const bool mfma_ok = s->dim_pad <= 128;    // d^2 <= 128*255^2 < 2^23 keeps the packed keys exact
    if (!mfma_ok) { s->exact_u8 = 0; return SFMHIP_OK; }
    void* q = nullptr;
    int rc = sfm_pool_get(ctx, (size_t)s->rows_pad * s->dim_pad, &q); if (rc) return rc; s->d_i8 = (int8_t*)q;
    rc = sfm_pool_get(ctx, 2 * (size_t)s->rows_pad * sizeof(int32_t), &q); if (rc) return rc; s->d_norm = (int32_t*)q;
    rc = descset_flag_slot(ctx, s); if (rc) return rc;
    SFM_HIP_TRY(ctx, hipMemsetAsync(s->d_flag, 0, sizeof(int), ctx->stream));
    const int waves_per_block = 4;
    const bool fast = prep_l2_fast(s->d_f32, s->ld, s->dim, s->dim_pad);
    const int rows_per_wave = fast ? 1024 / s->dim : 256 / s->dim_pad;
    hipLaunchKernelGGL(prep_l2_kernel, dim3(ceil_div(s->rows_pad, waves_per_block * rows_per_wave)), dim3(64 * waves_per_block), 0, ctx->stream,
                       s->d_f32, s->ld, s->rows, s->dim, s->dim_pad, s->d_i8, s->d_norm, s->d_flag, s->rows_pad, fast ? 1 : 0);
    SFM_HIP_TRY(ctx, hipGetLastError());
    s->exact_pending = true;
    return SFMHIP_OK;
}

// dim floats -> dim bytes; false when a value is not an integer in [0, 255] (NaN included).  SSE2: what every x86-64 has.
static inline bool l2_row_to_u8(const float* __restrict__ src, uint8_t* __restrict__ dst, int dim)
{
    int k = 0;
    __m128i bad = _mm_setzero_si128();
    for (; k + 16 <= dim; k += 16) {
        const __m128 a0 = _mm_loadu_ps(src + k), a1 = _mm_loadu_ps(src + k + 4), a2 = _mm_loadu_ps(src + k + 8), a3 = _mm_loadu_ps(src + k + 12);
        const __m128i i0 = _mm_cvtps_epi32(a0), i1 = _mm_cvtps_epi32(a1), i2 = _mm_cvtps_epi32(a2), i3 = _mm_cvtps_epi32(a3);
        // out of [0, 255] (NaN converts to 0x80000000) or not an integer
        const __m128i range = _mm_or_si128(_mm_or_si128(i0, i1), _mm_or_si128(i2, i3));
        const __m128 ne = _mm_or_ps(_mm_or_ps(_mm_cmpneq_ps(_mm_cvtepi32_ps(i0), a0), _mm_cmpneq_ps(_mm_cvtepi32_ps(i1), a1)),
                                    _mm_or_ps(_mm_cmpneq_ps(_mm_cvtepi32_ps(i2), a2), _mm_cmpneq_ps(_mm_cvtepi32_ps(i3), a3)));
        bad = _mm_or_si128(bad, _mm_or_si128(_mm_andnot_si128(_mm_set1_epi32(255), range), _mm_castps_si128(ne)));
        _mm_storeu_si128((__m128i*)(dst + k), _mm_packus_epi16(_mm_packs_epi32(i0, i1), _mm_packs_epi32(i2, i3)));
    }
    bool ok = _mm_movemask_epi8(_mm_cmpeq_epi32(bad, _mm_setzero_si128())) == 0xffff;
    for (; k < dim; ++k) {
        const float v = src[k];
        const int q = (v >= 0.0f && v <= 255.0f) ? (int)v : -1;
        if (q < 0 || (float)q != v) { ok = false; dst[k] = 0; } else dst[k] = (uint8_t)q;
    }
    return ok;
}

static int descset_create_l2_host_f32(sfmhip_ctx* ctx, const float* desc, int rows, int dim, size_t ld, sfmhip_descset** out);
static void descsets_destroy_all(sfmhip_descset** out, int n)
{
    for (int i = 0; i < n; ++i) if (out[i]) { sfmhip_descset_destroy(out[i]); out[i] = nullptr; }
}

extern "C" {

// Many images from host matrices in one call (the chain of match_features_for_all, NViewReconstuct.cpp:850-871 / :1369):
// one pass of the staging threads over all rows, one transfer stream, one preparation launch.  Integer-valued rows in [0, 255]
// (what cv::SIFT emits) cross PCIe as bytes: a quarter of the float rows' size; an image with any other value goes the
// float way of sfmhip_descset_create_l2_host (and takes the exact kernels later), image by image.
int sfmhip_descsets_create_l2_host(sfmhip_ctx* ctx, const float* const* desc, const int32_t* rows, int dim, const size_t* ld, int n, sfmhip_descset** out)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_descsets_create_l2_host");
    SFM_ARG_CHECK(ctx, ctx && out && (n == 0 || (desc && rows)) && n >= 0 && dim > 0);
    for (int i = 0; i < n; ++i) { out[i] = nullptr; SFM_ARG_CHECK(ctx, rows[i] >= 0 && (desc[i] || rows[i] == 0) && (!ld || ld[i] >= (size_t)dim)); }
    if (n == 0) return SFMHIP_OK;
    const bool bytes_ok = dim == 32 || dim == 64 || dim == 128;        // the int8 copy has no padding columns and a lane takes 16 values
    if (!bytes_ok) {
        for (int i = 0; i < n; ++i) {
            const int rc = descset_create_l2_host_f32(ctx, desc[i], rows[i], dim, ld ? ld[i] : (size_t)dim, &out[i]);
            if (rc) { descsets_destroy_all(out, n); return rc; }
        }
        return SFMHIP_OK;
    }
    std::vector<long long> first((size_t)n + 1, 0);                   // first row of image i in the concatenation of all images
    for (int i = 0; i < n; ++i) first[i + 1] = first[i] + rows[i];
    const long long total = first[n];
    uint8_t* d_u8 = nullptr;
    { void* q = nullptr; int rc = sfm_pool_get(ctx, (size_t)std::max<long long>(total, 1) * dim, &q); if (rc) return rc; d_u8 = (uint8_t*)q; }
    std::vector<int> bad((size_t)n, 0);
    int rc = sfm_upload_produced(ctx, d_u8, (size_t)total * dim, (size_t)dim, [&](char* piece, size_t off, size_t nb, int t, int nt) {
        const long long g0 = (long long)(off / dim), cnt = (long long)(nb / dim);
        long long r = g0 + cnt * t / nt;
        const long long re = g0 + cnt * (t + 1) / nt;
        int img = (int)(std::upper_bound(first.begin(), first.end(), r) - first.begin()) - 1;
        for (; r < re; ++r) {
            while (r >= first[img + 1]) ++img;
            const float* src = desc[img] + (size_t)(r - first[img]) * (ld ? ld[img] : (size_t)dim);
            if (!l2_row_to_u8(src, (uint8_t*)piece + (size_t)(r - g0) * dim, dim)) __atomic_store_n(&bad[img], 1, __ATOMIC_RELAXED);
        }
    });
    if (rc) { sfm_pool_put(ctx, d_u8); return rc; }
    std::vector<PrepU8Desc> tbl;
    int max_pad = 0;
    for (int i = 0; i < n && rc == SFMHIP_OK; ++i) {
        if (bad[i]) { rc = descset_create_l2_host_f32(ctx, desc[i], rows[i], dim, ld ? ld[i] : (size_t)dim, &out[i]); continue; }
        sfmhip_descset* s = nullptr;
        descset_alloc_common(ctx, SFMHIP_DESC_L2_F32, rows[i], dim, &s);
        out[i] = s;
        s->dim_pad = dim; s->ld = dim; s->owns_f32 = true; s->exact_u8 = 1; s->exact_pending = false;
        void* q = nullptr;
        rc = sfm_pool_get(ctx, (size_t)std::max(rows[i], 1) * dim * sizeof(float), &q); if (rc) break; s->d_f32 = (float*)q;
        rc = sfm_pool_get(ctx, (size_t)s->rows_pad * dim, &q); if (rc) break; s->d_i8 = (int8_t*)q;
        rc = sfm_pool_get(ctx, 2 * (size_t)s->rows_pad * sizeof(int32_t), &q); if (rc) break; s->d_norm = (int32_t*)q;
        rc = descset_flag_slot(ctx, s); if (rc) break;          // (sfmhip_descset_refresh re-derives the verdict there)
        PrepU8Desc d; d.src = d_u8 + (size_t)first[i] * dim; d.rows = rows[i]; d.dim = dim; d.rows_pad = s->rows_pad; d.dst = s->d_i8; d.norm = s->d_norm; d.f32 = (float*)s->d_f32;
        tbl.push_back(d);
        max_pad = std::max(max_pad, s->rows_pad);
    }
    if (rc == SFMHIP_OK && !tbl.empty()) {
        void* d_tbl = nullptr;
        rc = sfm_scratch2(ctx, tbl.size() * sizeof(PrepU8Desc), &d_tbl);
        if (rc == SFMHIP_OK) {
            hipError_t e = sfm_upload(ctx, d_tbl, tbl.data(), tbl.size() * sizeof(PrepU8Desc)) == SFMHIP_OK ? hipSuccess : hipErrorUnknown;     // (consumed on return)
            if (e == hipSuccess) {
                const int rows_per_block = 4 * (1024 / dim);
                hipLaunchKernelGGL(prep_l2_u8_batched_kernel, dim3(ceil_div(max_pad, rows_per_block), (unsigned)tbl.size()), dim3(256), 0, ctx->stream, (const PrepU8Desc*)d_tbl);
                e = hipGetLastError();
            }
            if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); rc = SFMHIP_E_HIP; }
        }
    }
    sfm_pool_put(ctx, d_u8);          // read only by the launch just enqueued (stream-ordered reuse)
    if (rc) descsets_destroy_all(out, n);
    return rc;
}

// The Hamming2 twin: all images' byte rows in one transfer, their two re-encodings in two launches (one each) instead of two per image.
int sfmhip_descsets_create_hamming2_host(sfmhip_ctx* ctx, const uint8_t* const* desc, const int32_t* rows, int nbytes, const size_t* ld, int n, sfmhip_descset** out)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_descsets_create_hamming2_host");
    SFM_ARG_CHECK(ctx, ctx && out && (n == 0 || (desc && rows)) && n >= 0 && nbytes > 0 && nbytes <= 64);
    for (int i = 0; i < n; ++i) { out[i] = nullptr; SFM_ARG_CHECK(ctx, rows[i] >= 0 && (desc[i] || rows[i] == 0) && (!ld || ld[i] >= (size_t)nbytes)); }
    if (n == 0) return SFMHIP_OK;
    std::vector<long long> first((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) first[i + 1] = first[i] + rows[i];
    const long long total = first[n];
    uint8_t* d_u8 = nullptr;
    { void* q = nullptr; int rc = sfm_pool_get(ctx, (size_t)std::max<long long>(total, 1) * nbytes, &q); if (rc) return rc; d_u8 = (uint8_t*)q; }
    int rc = sfm_upload_produced(ctx, d_u8, (size_t)total * nbytes, (size_t)nbytes, [&](char* piece, size_t off, size_t nb, int t, int nt) {
        const long long g0 = (long long)(off / nbytes), cnt = (long long)(nb / nbytes);
        long long r = g0 + cnt * t / nt;
        const long long re = g0 + cnt * (t + 1) / nt;
        int img = (int)(std::upper_bound(first.begin(), first.end(), r) - first.begin()) - 1;
        while (r < re) {
            while (r >= first[img + 1]) ++img;
            const long long run = std::min(re, first[img + 1]) - r;          // rows of this image in the share
            const size_t l = ld ? ld[img] : (size_t)nbytes;
            const uint8_t* src = desc[img] + (size_t)(r - first[img]) * l;
            uint8_t* dst = (uint8_t*)piece + (size_t)(r - g0) * nbytes;
            if (l == (size_t)nbytes) memcpy(dst, src, (size_t)run * nbytes);
            else for (long long k = 0; k < run; ++k) memcpy(dst + (size_t)k * nbytes, src + (size_t)k * l, (size_t)nbytes);
            r += run;
        }
    });
    if (rc) { sfm_pool_put(ctx, d_u8); return rc; }
    std::vector<PrepHamDesc> tbl((size_t)n);
    int max_pad = 0; bool any_f4 = false;
    for (int i = 0; i < n && rc == SFMHIP_OK; ++i) {
        sfmhip_descset* s = nullptr;
        descset_alloc_common(ctx, SFMHIP_DESC_HAMMING2_U8, rows[i], nbytes, &s);
        out[i] = s;
        void* q = nullptr;
        rc = sfm_pool_get(ctx, (size_t)s->rows_pad * 64, &q); if (rc) break; s->d_u32 = (uint32_t*)q;
        if (nbytes <= H4_MAX_NBYTES) { rc = sfm_pool_get(ctx, (size_t)s->rows_pad * H4_ROW_BYTES, &q); if (rc) break; s->d_f4 = (uint32_t*)q; any_f4 = true; }
        PrepHamDesc& d = tbl[i];
        d.src = d_u8 + (size_t)first[i] * nbytes; d.ld = (size_t)nbytes; d.rows = rows[i]; d.nbytes = nbytes; d.rows_pad = s->rows_pad; d.u32 = s->d_u32; d.f4 = s->d_f4;
        max_pad = std::max(max_pad, s->rows_pad);
    }
    if (rc == SFMHIP_OK) {
        void* d_tbl = nullptr;
        rc = sfm_scratch2(ctx, tbl.size() * sizeof(PrepHamDesc), &d_tbl);
        if (rc == SFMHIP_OK) {
            hipError_t e = sfm_upload(ctx, d_tbl, tbl.data(), tbl.size() * sizeof(PrepHamDesc)) == SFMHIP_OK ? hipSuccess : hipErrorUnknown;
            if (e == hipSuccess) {
                hipLaunchKernelGGL(prep_hamming_batched_kernel, dim3((unsigned)(((size_t)max_pad * 8 + 255) / 256), (unsigned)n), dim3(256), 0, ctx->stream, (const PrepHamDesc*)d_tbl);
                if (any_f4)
                    hipLaunchKernelGGL(prep_hamming_fp4_batched_kernel, dim3((unsigned)(((size_t)max_pad * 96 + 255) / 256), (unsigned)n), dim3(256), 0, ctx->stream, (const PrepHamDesc*)d_tbl);
                e = hipGetLastError();
            }
            if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); rc = SFMHIP_E_HIP; }
        }
    }
    sfm_pool_put(ctx, d_u8);
    if (rc) descsets_destroy_all(out, n);
    return rc;
}

int sfmhip_descset_create_l2_dev(sfmhip_ctx* ctx, const float* d_desc, int rows, int dim, size_t ld, sfmhip_descset** out)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_ARG_CHECK(ctx, ctx && out);
    SFM_ARG_CHECK(ctx, d_desc && rows >= 0 && dim > 0 && ld >= (size_t)dim);
    sfmhip_descset* s = nullptr;
    descset_alloc_common(ctx, SFMHIP_DESC_L2_F32, rows, dim, &s);
    s->d_f32 = d_desc; s->ld = ld; s->owns_f32 = false;
    const int rc = descset_prepare_l2(ctx, s);
    if (rc != SFMHIP_OK) { sfmhip_descset_destroy(s); return rc; }
    *out = s;
    return SFMHIP_OK;
}

int sfmhip_descset_create_l2_host(sfmhip_ctx* ctx, const float* desc, int rows, int dim, size_t ld, sfmhip_descset** out)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_ARG_CHECK(ctx, ctx && out);
    SFM_ARG_CHECK(ctx, (desc || rows == 0) && rows >= 0 && dim > 0 && ld >= (size_t)dim);
    // integer-valued rows cross PCIe as bytes (sfmhip_descsets_create_l2_host); anything else as floats (below)
    if (dim == 32 || dim == 64 || dim == 128) return sfmhip_descsets_create_l2_host(ctx, &desc, &rows, dim, &ld, 1, out);
    return descset_create_l2_host_f32(ctx, desc, rows, dim, ld, out);
}

}  // extern "C"

static int descset_create_l2_host_f32(sfmhip_ctx* ctx, const float* desc, int rows, int dim, size_t ld, sfmhip_descset** out)
{
    float* d = nullptr;
    const size_t nrow = rows > 0 ? rows : 1;
    { void* q = nullptr; int rc = sfm_pool_get(ctx, nrow * dim * sizeof(float), &q); if (rc) return rc; d = (float*)q; }
    if (rows > 0 && ld == (size_t)dim) {          // a dense cv::Mat (the reference's descriptors): through the pinned staging ring
        const int rc = sfm_upload(ctx, d, desc, (size_t)rows * dim * sizeof(float));
        if (rc) { sfm_pool_put(ctx, d); return rc; }
    } else if (rows > 0) {
        hipError_t e = hipMemcpy2DAsync(d, dim * sizeof(float), desc, ld * sizeof(float), dim * sizeof(float), rows,
                                        hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);       // the caller's rows must have been read when this returns
        if (e != hipSuccess) { sfm_pool_put(ctx, d); ctx->last_error = hipGetErrorString(e); return SFMHIP_E_HIP; }
    }
    sfmhip_descset* s = nullptr;
    descset_alloc_common(ctx, SFMHIP_DESC_L2_F32, rows, dim, &s);
    s->d_f32 = d; s->ld = dim; s->owns_f32 = true;
    const int rc = descset_prepare_l2(ctx, s);
    if (rc != SFMHIP_OK) { sfmhip_descset_destroy(s); return rc; }
    *out = s;
    return SFMHIP_OK;
}

extern "C" {

static int descset_prepare_hamming(sfmhip_ctx* ctx, sfmhip_descset* s, const uint8_t* d_src, size_t ld)
{
    { void* q = nullptr; int rc = sfm_pool_get(ctx, (size_t)s->rows_pad * 64, &q); if (rc) return rc; s->d_u32 = (uint32_t*)q; }
    const size_t n = (size_t)s->rows_pad * 64;
    hipLaunchKernelGGL(prep_hamming_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, ctx->stream,
                       d_src, ld, s->rows, s->dim, (uint32_t*)s->d_u32, s->rows_pad);
    SFM_HIP_TRY(ctx, hipGetLastError());
    if (s->dim <= H4_MAX_NBYTES) {
        { void* q = nullptr; int rc = sfm_pool_get(ctx, (size_t)s->rows_pad * H4_ROW_BYTES, &q); if (rc) return rc; s->d_f4 = (uint32_t*)q; }
        const size_t nw = (size_t)s->rows_pad * (H4_ROW_BYTES / 4);
        hipLaunchKernelGGL(prep_hamming_fp4_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, ctx->stream,
                           d_src, ld, s->rows, s->dim, s->d_f4, s->rows_pad);
        SFM_HIP_TRY(ctx, hipGetLastError());
    }
    return SFMHIP_OK;
}

int sfmhip_descset_create_hamming2_dev(sfmhip_ctx* ctx, const uint8_t* d_desc, int rows, int nbytes, size_t ld, sfmhip_descset** out)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_ARG_CHECK(ctx, ctx && out);
    SFM_ARG_CHECK(ctx, d_desc && rows >= 0 && nbytes > 0 && nbytes <= 64 && ld >= (size_t)nbytes);
    sfmhip_descset* s = nullptr;
    descset_alloc_common(ctx, SFMHIP_DESC_HAMMING2_U8, rows, nbytes, &s);
    const int rc = descset_prepare_hamming(ctx, s, d_desc, ld);
    if (rc != SFMHIP_OK) { sfmhip_descset_destroy(s); return rc; }
    *out = s;
    return SFMHIP_OK;
}

int sfmhip_descset_create_hamming2_host(sfmhip_ctx* ctx, const uint8_t* desc, int rows, int nbytes, size_t ld, sfmhip_descset** out)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_ARG_CHECK(ctx, ctx && out);
    SFM_ARG_CHECK(ctx, (desc || rows == 0) && rows >= 0 && nbytes > 0 && nbytes <= 64 && ld >= (size_t)nbytes);
    uint8_t* d = nullptr;
    const size_t nrow = rows > 0 ? rows : 1;
    { void* q = nullptr; int rc = sfm_pool_get(ctx, nrow * nbytes, &q); if (rc) return rc; d = (uint8_t*)q; }
    if (rows > 0 && ld == (size_t)nbytes) {
        const int rc = sfm_upload(ctx, d, desc, (size_t)rows * nbytes);
        if (rc) { sfm_pool_put(ctx, d); return rc; }
    } else if (rows > 0) {
        hipError_t e = hipMemcpy2DAsync(d, nbytes, desc, ld, nbytes, rows, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { sfm_pool_put(ctx, d); ctx->last_error = hipGetErrorString(e); return SFMHIP_E_HIP; }
    }
    sfmhip_descset* s = nullptr;
    descset_alloc_common(ctx, SFMHIP_DESC_HAMMING2_U8, rows, nbytes, &s);
    int rc = descset_prepare_hamming(ctx, s, d, nbytes);
    sfm_pool_put(ctx, d);          // the byte rows are only read by the re-encoding kernel just enqueued (stream-ordered reuse)
    if (rc != SFMHIP_OK) { sfmhip_descset_destroy(s); return rc; }
    *out = s;
    return SFMHIP_OK;
}

void sfmhip_descset_destroy(sfmhip_descset* s)
{
    SFM_DEVICE_GUARD(s ? s->ctx : nullptr);
    if (!s) return;
    // the blocks go back to the context's cache; whoever gets them next works on the same stream, behind this set's last launch
    sfmhip_ctx* ctx = s->ctx;
    if (s->owns_f32 && s->d_f32) sfm_pool_put(ctx, (void*)s->d_f32);
    if (s->d_i8) sfm_pool_put(ctx, s->d_i8);
    if (s->d_norm) sfm_pool_put(ctx, s->d_norm);
    if (s->d_u32) sfm_pool_put(ctx, s->d_u32);
    if (s->d_f4) sfm_pool_put(ctx, s->d_f4);
    if (s->flag_slot >= 0) ctx->flag_free.push_back(s->flag_slot);
    else if (s->d_flag) sfm_pool_put(ctx, s->d_flag);
    delete s;
}

// re-run the preparation pass on the (possibly rewritten) borrowed float rows; asynchronous, keeps exact_u8
int sfmhip_descset_refresh(sfmhip_descset* s)
{
    SFM_DEVICE_GUARD(s ? s->ctx : nullptr);
    if (!s || !s->ctx) return SFMHIP_E_ARG;
    sfmhip_ctx* ctx = s->ctx;
    if (s->kind != SFMHIP_DESC_L2_F32 || !s->d_i8) return SFMHIP_OK;
    SFM_HIP_TRY(ctx, hipMemsetAsync(s->d_flag, 0, sizeof(int), ctx->stream));
    const bool fast = prep_l2_fast(s->d_f32, s->ld, s->dim, s->dim_pad);
    hipLaunchKernelGGL(prep_l2_kernel, dim3(ceil_div(s->rows_pad, 4 * (fast ? 1024 / s->dim : 256 / s->dim_pad))), dim3(256), 0, ctx->stream,
                       s->d_f32, s->ld, s->rows, s->dim, s->dim_pad, s->d_i8, s->d_norm, s->d_flag, s->rows_pad, fast ? 1 : 0);
    SFM_HIP_TRY(ctx, hipGetLastError());
    return SFMHIP_OK;
}

// one launch for many images: what a per-frame "descriptors arrived" step costs when the float rows were rewritten
int sfmhip_descsets_refresh(sfmhip_ctx* ctx, sfmhip_descset* const* sets, int n)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_descsets_refresh");
    SFM_ARG_CHECK(ctx, ctx && (sets || n == 0) && n >= 0);
    std::vector<PrepDesc> tbl;
    int max_pad = 0, rows_per_block = 1 << 30;
    bool fast = true;
    for (int i = 0; i < n; ++i) {
        const sfmhip_descset* s = sets[i];
        SFM_ARG_CHECK(ctx, s != nullptr);
        if (s->kind != SFMHIP_DESC_L2_F32 || !s->d_i8) continue;
        fast = fast && prep_l2_fast(s->d_f32, s->ld, s->dim, s->dim_pad);
        PrepDesc d; d.src = s->d_f32; d.ld = s->ld; d.rows = s->rows; d.dim = s->dim; d.dim_pad = s->dim_pad; d.rows_pad = s->rows_pad;
        d.dst = s->d_i8; d.norm = s->d_norm; d.flag = s->d_flag;
        tbl.push_back(d);
        if (s->rows_pad > max_pad) max_pad = s->rows_pad;
    }
    if (tbl.empty()) return SFMHIP_OK;
    void* d_tbl = nullptr;
    int rc = sfm_scratch2(ctx, tbl.size() * sizeof(PrepDesc), &d_tbl); if (rc) return rc;
    SFM_HIP_TRY(ctx, hipMemcpyAsync(d_tbl, tbl.data(), tbl.size() * sizeof(PrepDesc), hipMemcpyHostToDevice, ctx->stream));
    // rows per 4-wave block: the smallest over the batch (16 values per lane when every image qualifies, else 4: >= 2 rows per wave)
    for (const PrepDesc& d : tbl) { const int r = 4 * (fast ? 1024 / d.dim : 256 / d.dim_pad); if (r < rows_per_block) rows_per_block = r; }
    hipLaunchKernelGGL(prep_l2_batched_kernel, dim3(ceil_div(max_pad, rows_per_block), (unsigned)tbl.size()), dim3(256), 0, ctx->stream, (const PrepDesc*)d_tbl, fast ? 1 : 0);
    SFM_HIP_TRY(ctx, hipGetLastError());
    return SFMHIP_OK;
}

int sfmhip_descset_info(sfmhip_descset* s, int* kind, int* rows, int* dim, int* exact_u8)
{
    SFM_DEVICE_GUARD(s ? s->ctx : nullptr);
    if (!s) return SFMHIP_E_ARG;
    if (kind) *kind = s->kind;
    if (rows) *rows = s->rows;
    if (dim) *dim = s->dim;
    if (exact_u8) { sfmhip_descset* one[1] = { s }; int rc = descsets_resolve(s->ctx, one, 1); if (rc) return rc; *exact_u8 = s->exact_u8; }
    return SFMHIP_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// batched kNN-2 driver
// ------------------------------------------------------------------------------------------------
struct KnnPlan {
    std::vector<PairDesc> pd;
    long long part_entries = 0, list_entries = 0, out_rows = 0;
    int max_qpad = 0, max_chunks = 0, max_nq = 0;
    int path = 0;    // 1 exact f32, 2 int8 mfma, 3 hamming (VALU), 4 hamming (FP4 MFMA)
    int ks = 0, dim = 0;
    bool aligned = true;
};

static int plan_pairs(sfmhip_ctx* ctx, sfmhip_descset* const* sets, int n_sets, const int32_t* pairs, int n_pairs,
                      int force_path, KnnPlan& P)
{
    SFM_ARG_CHECK(ctx, sets && pairs && n_pairs > 0 && n_sets > 0);
    { int rc = descsets_resolve(ctx, sets, n_sets); if (rc) return rc; }
    bool all_exact = true; int kind = 0, dim = 0;
    for (int p = 0; p < n_pairs; ++p) {
        const int a = pairs[2 * p], b = pairs[2 * p + 1];
        SFM_ARG_CHECK(ctx, a >= 0 && a < n_sets && b >= 0 && b < n_sets && sets[a] && sets[b]);
        const sfmhip_descset* q = sets[a]; const sfmhip_descset* t = sets[b];
        if (p == 0) { kind = q->kind; dim = q->dim; }
        SFM_ARG_CHECK(ctx, q->kind == kind && t->kind == kind && q->dim == dim && t->dim == dim);
        all_exact = all_exact && q->exact_u8 && t->exact_u8;
    }
    P.dim = dim;
    if (kind == SFMHIP_DESC_HAMMING2_U8) {
        // the matrix-core kernel whenever the rows fit its 768-value encoding (nbytes <= 61: AKAZE's 61-byte rows do)
        SFM_ARG_CHECK(ctx, force_path == 0 || force_path == 3 || (force_path == 4 && dim <= H4_MAX_NBYTES));
        P.path = (force_path == 3 || dim > H4_MAX_NBYTES) ? 3 : 4;
    } else {
        SFM_ARG_CHECK(ctx, force_path >= 0 && force_path <= 2);
        if (force_path == 2) { SFM_ARG_CHECK(ctx, all_exact); P.path = 2; }
        else if (force_path == 1) P.path = 1;
        else P.path = all_exact ? 2 : 1;
        P.ks = sets[pairs[0]]->dim_pad / 32;
    }
    // chunking: enough workgroups to fill the chip, chunks of whole 128-row blocks, <= 4096 rows (7-bit tile index)
    long long qblocks_total = 0;
    const int qgran = (P.path == 4) ? 64 * H4_WAVES : (P.path == 3 ? 256 : (P.path == 1 ? 4 : 128));
    for (int p = 0; p < n_pairs; ++p) qblocks_total += ceil_div(sets[pairs[2 * p]]->rows_pad, qgran);
    const long long target_wgs = 4LL * ctx->num_cus;
    P.pd.resize(n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
        const sfmhip_descset* q = sets[pairs[2 * p]]; const sfmhip_descset* t = sets[pairs[2 * p + 1]];
        PairDesc& d = P.pd[p];
        memset(&d, 0, sizeof d);
        d.nq = q->rows; d.nt = t->rows; d.nq_pad = q->rows_pad; d.nt_pad = t->rows_pad; d.dim = dim;
        if (P.path == 3) { d.q = q->d_u32; d.t = t->d_u32; }
        else if (P.path == 4) { d.q = q->d_f4; d.t = t->d_f4; }
        else { d.q = q->d_i8; d.t = t->d_i8; d.qn = q->d_norm; d.tn = t->d_norm + t->rows_pad; d.qf = q->d_f32; d.tf = t->d_f32; d.ldq = q->ld; d.ldt = t->ld; }
        const int brows = (P.path == 2) ? KNN_STAGE_ROWS : 128;               // rows per staged block of the kernel that runs
        const int tblocks = d.nt_pad / brows, max_cb = (P.path == 4 ? 8192 : 4096) / brows;   // <= 4096 train rows per chunk (7-bit tile index; 8 bits on path 4)
        int nch = (int)((target_wgs + qblocks_total - 1) / (qblocks_total > 0 ? qblocks_total : 1));
        if (P.path == 1) nch = 1 > nch ? 1 : (nch > 8 ? 8 : nch);
        if (nch < 1) nch = 1;
        if (nch > tblocks) nch = tblocks;
        if (nch < ceil_div(tblocks, max_cb)) nch = ceil_div(tblocks, max_cb);
        int cb = ceil_div(tblocks, nch);      // balanced chunks
        if (P.path == 4) {          // power-of-two chunks (aligned windows of <= 8192 rows: the tile index in the rows is (row >> 5) & 255)
            int c2 = 2; while (c2 < cb && c2 * 2 * brows <= 8192) c2 *= 2;
            cb = c2;
        }
        nch = ceil_div(tblocks, cb);
        d.nchunks = nch; d.chunk_rows = cb * brows;
        d.part_off = P.part_entries; P.part_entries += (long long)d.nq_pad * nch;
        d.out_off = P.out_rows; P.out_rows += d.nq;
        d.list_off = P.list_entries; P.list_entries += d.nq > 0 ? d.nq : 1;
        if (d.nq_pad > P.max_qpad) P.max_qpad = d.nq_pad;
        if (d.nq > P.max_nq) P.max_nq = d.nq;
        if (nch > P.max_chunks) P.max_chunks = nch;
        if (P.path != 3) {
            if ((d.ldq % 4) || (d.ldt % 4) || ((uintptr_t)d.qf % 16) || ((uintptr_t)d.tf % 16)) P.aligned = false;
        }
    }
    return SFMHIP_OK;
}

struct KnnWork { PairDesc* d_pd; long long* d_part; int* d_list; int* d_count; };

static int knn_workspace(sfmhip_ctx* ctx, const KnnPlan& P, int n_pairs, KnnWork& W)
{
    const size_t b_pd = (sizeof(PairDesc) * n_pairs + 255) / 256 * 256;
    const size_t b_part = ((size_t)P.part_entries * 16 + 255) / 256 * 256;
    const size_t b_list = ((size_t)P.list_entries * 4 + 255) / 256 * 256;
    const size_t b_cnt = ((size_t)n_pairs * 4 + 255) / 256 * 256;
    void* base = nullptr;
    int rc = sfm_scratch(ctx, b_pd + b_part + b_list + b_cnt, &base);
    if (rc != SFMHIP_OK) return rc;
    char* p = (char*)base;
    W.d_pd = (PairDesc*)p; p += b_pd;
    W.d_part = (long long*)p; p += b_part;
    W.d_list = (int*)p; p += b_list;
    W.d_count = (int*)p;
    SFM_HIP_TRY(ctx, hipMemcpyAsync(W.d_pd, P.pd.data(), sizeof(PairDesc) * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    SFM_HIP_TRY(ctx, hipMemsetAsync(W.d_count, 0, b_cnt, ctx->stream));
    return SFMHIP_OK;
}

template <bool ALIGNED>
static void launch_exact(sfmhip_ctx* ctx, const KnnPlan& P, const KnnWork& W, int n_pairs, bool use_list)
{
    constexpr int QR = 4;
    int gx = ceil_div(P.max_nq > 0 ? P.max_nq : 1, QR);
    if (use_list && gx > 8) gx = 8;          // the re-score list is normally empty: a few blocks per (chunk, pair) loop over it
    const dim3 grid(gx, P.max_chunks, n_pairs);
    const size_t shm = (size_t)QR * P.dim * sizeof(float);
    hipLaunchKernelGGL((knn2_exact_f32_kernel<QR, ALIGNED, false>), grid, dim3(256), shm, ctx->stream,
                       W.d_pd, W.d_part, use_list ? W.d_list : (const int*)nullptr, W.d_count, (float*)nullptr, (size_t)0);
}

// enqueue the kNN-2 of all pairs; results in d_idx2 / d_dist2 (rows concatenated in pair order)
static int knn2_pairs_enqueue(sfmhip_ctx* ctx, const KnnPlan& P, const KnnWork& W, int n_pairs, int32_t* d_idx2, float* d_dist2)
{
    if (P.max_nq == 0) return SFMHIP_OK;
    const dim3 mgrid(ceil_div(P.max_nq, 256), n_pairs);
    hipEvent_t* tev = (ctx->timing && ctx->timing_used < sfmhip_ctx::TIMING_SLOTS) ? ctx->tev[ctx->timing_used++] : nullptr;
    if (tev) (void)hipEventRecord(tev[0], ctx->stream);
    if (P.path == 2) {
        const dim3 grid(P.max_qpad / 128, P.max_chunks, round_up(n_pairs, 8));       // z padded: see the kernel's XCD mapping
        switch (P.ks) {
            case 1: hipLaunchKernelGGL(knn2_i8_kernel<1>, grid, dim3(256), 0, ctx->stream, W.d_pd, W.d_part, n_pairs); break;
            case 2: hipLaunchKernelGGL(knn2_i8_kernel<2>, grid, dim3(256), 0, ctx->stream, W.d_pd, W.d_part, n_pairs); break;
            case 4: hipLaunchKernelGGL(knn2_i8_kernel<4>, grid, dim3(256), 0, ctx->stream, W.d_pd, W.d_part, n_pairs); break;
            default: ctx->last_error = "int8 path: dim > 128"; return SFMHIP_E_ARG;
        }
        SFM_HIP_TRY(ctx, hipGetLastError());
        if (tev) (void)hipEventRecord(tev[1], ctx->stream);
        hipLaunchKernelGGL(merge_kernel<0>, mgrid, dim3(256), 0, ctx->stream, W.d_pd, W.d_part, d_idx2, d_dist2, W.d_list, W.d_count, 0);
        // rows whose float distances may tie although the integers differ: exact re-score (normally none)
        if (P.aligned) launch_exact<true>(ctx, P, W, n_pairs, true); else launch_exact<false>(ctx, P, W, n_pairs, true);
        hipLaunchKernelGGL(merge_kernel<1>, mgrid, dim3(256), 0, ctx->stream, W.d_pd, W.d_part, d_idx2, d_dist2, W.d_list, W.d_count, 1);
    } else if (P.path == 1) {
        if (P.aligned) launch_exact<true>(ctx, P, W, n_pairs, false); else launch_exact<false>(ctx, P, W, n_pairs, false);
        if (tev) (void)hipEventRecord(tev[1], ctx->stream);
        hipLaunchKernelGGL(merge_kernel<1>, mgrid, dim3(256), 0, ctx->stream, W.d_pd, W.d_part, d_idx2, d_dist2, (int*)nullptr, W.d_count, 0);
    } else if (P.path == 4) {
        const dim3 grid(ceil_div(P.max_qpad, 64 * H4_WAVES), P.max_chunks, round_up(n_pairs, 8));
        hipLaunchKernelGGL(knn2_hamming2_fp4_kernel<H4_WAVES>, grid, dim3(64 * H4_WAVES), 0, ctx->stream, W.d_pd, W.d_part, n_pairs);
        if (tev) (void)hipEventRecord(tev[1], ctx->stream);
        hipLaunchKernelGGL(merge_kernel<2>, mgrid, dim3(256), 0, ctx->stream, W.d_pd, W.d_part, d_idx2, d_dist2, (int*)nullptr, W.d_count, 0);
    } else {
        const dim3 grid(ceil_div(P.max_qpad, 256), P.max_chunks, n_pairs);
        hipLaunchKernelGGL(knn2_hamming2_kernel, grid, dim3(256), 0, ctx->stream, W.d_pd, W.d_part);
        if (tev) (void)hipEventRecord(tev[1], ctx->stream);
        hipLaunchKernelGGL(merge_kernel<2>, mgrid, dim3(256), 0, ctx->stream, W.d_pd, W.d_part, d_idx2, d_dist2, (int*)nullptr, W.d_count, 0);
    }
    if (tev) (void)hipEventRecord(tev[2], ctx->stream);
    SFM_HIP_TRY(ctx, hipGetLastError());
    return SFMHIP_OK;
}

extern "C" {

int sfmhip_knn2_dev(sfmhip_ctx* ctx, const sfmhip_descset* query, const sfmhip_descset* train,
                    int32_t* d_idx2, float* d_dist2, int force_path)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_knn2_dev");
    SFM_ARG_CHECK(ctx, ctx && query && train && d_idx2 && d_dist2);
    sfmhip_descset* sets[2] = { (sfmhip_descset*)query, (sfmhip_descset*)train };
    const int32_t pr[2] = { 0, 1 };
    KnnPlan P; KnnWork W;
    int rc = plan_pairs(ctx, sets, 2, pr, 1, force_path, P); if (rc) return rc;
    rc = knn_workspace(ctx, P, 1, W); if (rc) return rc;
    return knn2_pairs_enqueue(ctx, P, W, 1, d_idx2, d_dist2);
}

int sfmhip_match_pairs_dev(sfmhip_ctx* ctx, sfmhip_descset* const* sets, int n_sets,
                           const int32_t* pairs, int n_pairs, double ratio, float floor_, float mult,
                           sfm_dmatch* d_matches, int max_per_pair, int32_t* d_counts)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_match_pairs_dev");
    SFM_ARG_CHECK(ctx, ctx && d_matches && d_counts && max_per_pair > 0);
    if (n_pairs == 0) return SFMHIP_OK;
    KnnPlan P; KnnWork W;
    int rc = plan_pairs(ctx, sets, n_sets, pairs, n_pairs, 0, P); if (rc) return rc;
    SFM_ARG_CHECK(ctx, max_per_pair >= P.max_nq);
    rc = knn_workspace(ctx, P, n_pairs, W); if (rc) return rc;
    void* tmp = nullptr;
    const size_t rows = (size_t)(P.out_rows > 0 ? P.out_rows : 1);
    rc = sfm_scratch2(ctx, rows * 16, &tmp); if (rc) return rc;
    int32_t* d_idx2 = (int32_t*)tmp; float* d_dist2 = (float*)((char*)tmp + rows * 8);
    rc = knn2_pairs_enqueue(ctx, P, W, n_pairs, d_idx2, d_dist2); if (rc) return rc;
    hipLaunchKernelGGL(ratio_tail_kernel, dim3(n_pairs), dim3(1024), 0, ctx->stream, W.d_pd, d_idx2, d_dist2,
                       ratio, floor_, mult, d_matches, max_per_pair, d_counts);
    SFM_HIP_TRY(ctx, hipGetLastError());
    return SFMHIP_OK;
}

int sfmhip_match_pairs(sfmhip_ctx* ctx, sfmhip_descset* const* sets, int n_sets,
                       const int32_t* pairs, int n_pairs, double ratio, float floor_, float mult,
                       sfm_dmatch* matches_out, int max_per_pair, int32_t* counts_out)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_match_pairs");
    SFM_ARG_CHECK(ctx, ctx && matches_out && counts_out && max_per_pair > 0 && n_pairs >= 0);
    if (n_pairs == 0) return SFMHIP_OK;
    // the ratio-tail kernel writes the lists and the counts straight into pinned host staging memory (only the surviving
    // matches cross PCIe, no per-pair copies, no allocation per call); they are handed over with plain memcpys
    void* stage = nullptr;
    const size_t b_m = (size_t)n_pairs * max_per_pair * sizeof(sfm_dmatch), b_c = (size_t)n_pairs * sizeof(int32_t);
    int rc = sfm_pinned(ctx, b_m + b_c, &stage); if (rc) return rc;
    sfm_dmatch* h_m = (sfm_dmatch*)stage; int32_t* h_c = (int32_t*)((char*)stage + b_m);
    rc = sfmhip_match_pairs_dev(ctx, sets, n_sets, pairs, n_pairs, ratio, floor_, mult, h_m, max_per_pair, h_c);
    if (rc) return rc;
    SFM_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int p = 0; p < n_pairs; ++p) {
        counts_out[p] = h_c[p];
        if (h_c[p] > 0) memcpy(matches_out + (size_t)p * max_per_pair, h_m + (size_t)p * max_per_pair, (size_t)h_c[p] * sizeof(sfm_dmatch));
    }
    return SFMHIP_OK;
}

static int knn2_host_common(sfmhip_ctx* ctx, sfmhip_descset* qs, sfmhip_descset* ts, int nq, int32_t* idx2, float* dist2)
{
    int32_t* d_idx = nullptr; float* d_dist = nullptr;
    const size_t n = nq > 0 ? nq : 1;
    SFM_HIP_TRY(ctx, hipMalloc((void**)&d_idx, n * 8));
    hipError_t e = hipMalloc((void**)&d_dist, n * 8);
    if (e != hipSuccess) { (void)hipFree(d_idx); ctx->last_error = hipGetErrorString(e); return SFMHIP_E_HIP; }
    int rc = sfmhip_knn2_dev(ctx, qs, ts, d_idx, d_dist, 0);
    if (rc == SFMHIP_OK && nq > 0) {
        e = hipMemcpyAsync(idx2, d_idx, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dist2, d_dist, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); rc = SFMHIP_E_HIP; }
    }
    (void)hipFree(d_idx); (void)hipFree(d_dist);
    return rc;
}

int sfmhip_knn2_l2_f32(sfmhip_ctx* ctx, const float* q, int nq, const float* t, int nt, int dim,
                       size_t ldq, size_t ldt, int32_t* idx2, float* dist2)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_knn2_l2_f32");
    SFM_ARG_CHECK(ctx, ctx && idx2 && dist2 && nq >= 0 && nt >= 0);
    sfmhip_descset *qs = nullptr, *ts = nullptr;
    int rc = sfmhip_descset_create_l2_host(ctx, q, nq, dim, ldq, &qs); if (rc) return rc;
    rc = sfmhip_descset_create_l2_host(ctx, t, nt, dim, ldt, &ts);
    if (rc == SFMHIP_OK) rc = knn2_host_common(ctx, qs, ts, nq, idx2, dist2);
    sfmhip_descset_destroy(qs); sfmhip_descset_destroy(ts);
    return rc;
}

int sfmhip_knn2_hamming2_u8(sfmhip_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int nbytes,
                            size_t ldq, size_t ldt, int32_t* idx2, float* dist2)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_knn2_hamming2_u8");
    SFM_ARG_CHECK(ctx, ctx && idx2 && dist2 && nq >= 0 && nt >= 0);
    sfmhip_descset *qs = nullptr, *ts = nullptr;
    int rc = sfmhip_descset_create_hamming2_host(ctx, q, nq, nbytes, ldq, &qs); if (rc) return rc;
    rc = sfmhip_descset_create_hamming2_host(ctx, t, nt, nbytes, ldt, &ts);
    if (rc == SFMHIP_OK) rc = knn2_host_common(ctx, qs, ts, nq, idx2, dist2);
    sfmhip_descset_destroy(qs); sfmhip_descset_destroy(ts);
    return rc;
}

static int match_features_common(sfmhip_ctx* ctx, sfmhip_descset* qs, sfmhip_descset* ts, int nq, sfm_dmatch* out, int* n_out)
{
    sfmhip_descset* sets[2] = { qs, ts };
    const int32_t pr[2] = { 0, 1 };
    int32_t cnt = 0;
    if (nq == 0) { *n_out = 0; return SFMHIP_OK; }
    // NViewReconstuct.cpp:884,900-901: ratio 0.6 (double), gate 5 * max(min_dist, 10.0f)
    const int rc = sfmhip_match_pairs(ctx, sets, 2, pr, 1, 0.6, 10.0f, 5.0f, out, nq, &cnt);
    *n_out = cnt;
    return rc;
}

int sfmhip_match_features_l2(sfmhip_ctx* ctx, const float* q, int nq, const float* t, int nt, int dim,
                             size_t ldq, size_t ldt, sfm_dmatch* out, int* n_out)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_match_features_l2");
    SFM_ARG_CHECK(ctx, ctx && out && n_out && nq >= 0 && nt >= 0);
    sfmhip_descset *qs = nullptr, *ts = nullptr;
    int rc = sfmhip_descset_create_l2_host(ctx, q, nq, dim, ldq, &qs); if (rc) return rc;
    rc = sfmhip_descset_create_l2_host(ctx, t, nt, dim, ldt, &ts);
    if (rc == SFMHIP_OK) rc = match_features_common(ctx, qs, ts, nq, out, n_out);
    sfmhip_descset_destroy(qs); sfmhip_descset_destroy(ts);
    return rc;
}

int sfmhip_match_features_hamming2(sfmhip_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int nbytes,
                                   size_t ldq, size_t ldt, sfm_dmatch* out, int* n_out)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_match_features_hamming2");
    SFM_ARG_CHECK(ctx, ctx && out && n_out && nq >= 0 && nt >= 0);
    sfmhip_descset *qs = nullptr, *ts = nullptr;
    int rc = sfmhip_descset_create_hamming2_host(ctx, q, nq, nbytes, ldq, &qs); if (rc) return rc;
    rc = sfmhip_descset_create_hamming2_host(ctx, t, nt, nbytes, ldt, &ts);
    if (rc == SFMHIP_OK) rc = match_features_common(ctx, qs, ts, nq, out, n_out);
    sfmhip_descset_destroy(qs); sfmhip_descset_destroy(ts);
    return rc;
}

// self-test (sfmhip.h): number of integers in [0, 2^24) whose sqrt_exact_int differs from sqrtf
int sfmhip_selftest_exact_sqrt(sfmhip_ctx* ctx, int* mismatches)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_ARG_CHECK(ctx, ctx && mismatches);
    int* d = nullptr;
    SFM_HIP_TRY(ctx, hipMalloc((void**)&d, sizeof(int)));
    SFM_HIP_TRY(ctx, hipMemsetAsync(d, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(sqrt_check_kernel, dim3((1 << 24) / 256), dim3(256), 0, ctx->stream, d);
    hipError_t e = hipMemcpyAsync(mismatches, d, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return SFMHIP_E_HIP; }
    return SFMHIP_OK;
}

int sfmhip_l2_distance_matrix_dev(sfmhip_ctx* ctx, const sfmhip_descset* query, const sfmhip_descset* train,
                                  float* d_dist, size_t ld, int force_path)
{
    SFM_DEVICE_GUARD(ctx);
    SFM_RANGE("sfmhip_l2_distance_matrix_dev");
    SFM_ARG_CHECK(ctx, ctx && query && train && d_dist);
    SFM_ARG_CHECK(ctx, query->kind == SFMHIP_DESC_L2_F32 && train->kind == SFMHIP_DESC_L2_F32 && query->dim == train->dim);
    SFM_ARG_CHECK(ctx, ld >= (size_t)train->rows);
    if (query->rows == 0 || train->rows == 0) return SFMHIP_OK;
    { sfmhip_descset* two[2] = { const_cast<sfmhip_descset*>(query), const_cast<sfmhip_descset*>(train) }; int rc = descsets_resolve(ctx, two, 2); if (rc) return rc; }
    const bool exact = query->exact_u8 && train->exact_u8;
    SFM_ARG_CHECK(ctx, !(force_path == 2 && !exact));
    if (exact && force_path != 1) {
        int exp_mode = 0, bpw = 1;   // 128 trains per workgroup (in-process A/B on MI355X, aligned output: 1: 84 us, 2: 89 us, 4: 93 us, 8: 109 us)
#ifdef SFMHIP_EXPERIMENTS
        if (const char* em = getenv("SFMHIP_EXP_DISTMAT")) exp_mode = atoi(em);
        if (const char* eb = getenv("SFMHIP_EXP_BPW")) bpw = atoi(eb);
#endif
        const int vec_ok = (ld % 4 == 0) && ((uintptr_t)d_dist % 16 == 0);
        // rows alternate between line-aligned and half-a-line-off (ld = 16 mod 32 floats, e.g. the reference's 10000-column matrix)
        int parity = vec_ok && (ld % 32 == 16) && ((uintptr_t)d_dist % 128 == 0);
#ifdef SFMHIP_EXPERIMENTS
        if (getenv("SFMHIP_EXP_NO_PARITY")) parity = 0;
#endif
        int nw = DISTMAT_WAVES;          // waves per workgroup: 32 nw query rows share one 128-train block
#ifdef SFMHIP_EXPERIMENTS
        if (const char* e = getenv("SFMHIP_EXP_DM_NW")) nw = atoi(e) == 8 ? 8 : 4;
#endif
        const int qrows = 32 * nw;
        const int n_qb = parity ? 2 * ceil_div(query->rows, 2 * qrows) : ceil_div(query->rows, qrows), n_tb = ceil_div(train->rows_pad / 128 + (parity ? 1 : 0), bpw);
        const dim3 grid((unsigned)(8 * ceil_div(n_qb, 8) * n_tb));     // decoded in the kernel (XCD-banded mapping)
        const int ks = query->dim_pad / 32;
#define DM_LAUNCH(K, W) hipLaunchKernelGGL((distmat_i8_kernel<K, W>), grid, dim3(64 * W), 0, ctx->stream, query->d_i8, query->d_norm, \
                                        train->d_i8, train->d_norm, query->rows, query->rows_pad, train->rows, train->rows_pad, bpw, d_dist, ld, vec_ok, parity, exp_mode)
#ifdef SFMHIP_EXPERIMENTS
        // eight waves (256 query rows) per train block halve the train re-reads; measured level with four on the reference's stride and
        // 2 us slower on aligned rows (profiles/README.md, round 3): experiments builds only
        if (nw == 8) { if (ks == 1) DM_LAUNCH(1, 8); else if (ks == 2) DM_LAUNCH(2, 8); else DM_LAUNCH(4, 8); }
        else
#endif
        { if (ks == 1) DM_LAUNCH(1, 4); else if (ks == 2) DM_LAUNCH(2, 4); else DM_LAUNCH(4, 4); }
#undef DM_LAUNCH
        SFM_HIP_TRY(ctx, hipGetLastError());
        return SFMHIP_OK;
    }
    // exact fp32 path, all distances stored
    sfmhip_descset* sets[2] = { (sfmhip_descset*)query, (sfmhip_descset*)train };
    const int32_t pr[2] = { 0, 1 };
    KnnPlan P; KnnWork W;
    int rc = plan_pairs(ctx, sets, 2, pr, 1, 1, P); if (rc) return rc;
    rc = knn_workspace(ctx, P, 1, W); if (rc) return rc;
    constexpr int QR = 4;
    const dim3 grid(ceil_div(P.max_nq, QR), P.max_chunks, 1);
    const size_t shm = (size_t)QR * P.dim * sizeof(float);
    if (P.aligned)
        hipLaunchKernelGGL((knn2_exact_f32_kernel<QR, true, true>), grid, dim3(256), shm, ctx->stream, W.d_pd, W.d_part,
                           (const int*)nullptr, W.d_count, d_dist, ld);
    else
        hipLaunchKernelGGL((knn2_exact_f32_kernel<QR, false, true>), grid, dim3(256), shm, ctx->stream, W.d_pd, W.d_part,
                           (const int*)nullptr, W.d_count, d_dist, ld);
    SFM_HIP_TRY(ctx, hipGetLastError());
    return SFMHIP_OK;
}

}  // extern "C"
