/*
 * oracle/orc_match.c -- CPU restatement of match_features (TEST INFRASTRUCTURE, see orc.h).
 *
 * Follows NView:873-913 (live, NORM_HAMMING2) and TwoViewReconstruct.cpp:156-194 (NORM_L2 twin).
 * The kNN itself is cv::BFMatcher::knnMatch -> cv::batchDistance, OpenCV 4.4.0, not in
 * /root/reference; restated from its published source [3P]:
 *   - NORM_L2 / CV_32F: dist = std::sqrt(hal::normL2Sqr_(a, b, n)) in float32; normL2Sqr_ with the
 *     SSE2 universal intrinsics (128-bit, 4 lanes, the x64 baseline of opencv_world440): four
 *     4-lane accumulators, element j -> accumulator (j/4)%4 lane j%4, v_muladd = mul then add
 *     (no FMA on SSE2), then (d0+d1+d2+d3) lane-wise left to right, then
 *     v_reduce_sum = (l0+l2)+(l1+l3), then a scalar tail.  (An AVX2 build would sum in another
 *     order; for integer-valued descriptors such as OpenCV SIFT's every order gives the same bits.)
 *   - NORM_HAMMING2 / CV_8U: sum over bytes of popCountTable2[a^b] (number of non-zero 2-bit cells),
 *     int32, converted to float by BFMatcher::knnMatchImpl.
 *   - K=2 selection: per query row, train rows in ascending order, insert when d < dist[K-1],
 *     shifting entries with dist > d: stable => ties keep the lower train index.  Initial
 *     idx = -1, dist = FLT_MAX (INT_MAX for the integer metric).
 * parity unpinned (no golden vectors in the reference); cross-checked vs numpy in tests/.
 */
#include "orc.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <string.h>

static float normL2Sqr_sse(const float* a, const float* b, int n)
{
    float acc[4][4];
    memset(acc, 0, sizeof acc);
    int j = 0;
    for (; j <= n - 16; j += 16)
        for (int v = 0; v < 4; ++v)
            for (int l = 0; l < 4; ++l) {
                float t = a[j + 4 * v + l] - b[j + 4 * v + l];
                float m = t * t;              /* two roundings: compiled with -ffp-contract=off */
                acc[v][l] = m + acc[v][l];
            }
    float s[4];
    for (int l = 0; l < 4; ++l) s[l] = ((acc[0][l] + acc[1][l]) + acc[2][l]) + acc[3][l];
    float d = (s[0] + s[2]) + (s[1] + s[3]);
    for (; j < n; ++j) {
        float t = a[j] - b[j];
        float m = t * t;
        d = d + m;
    }
    return d;
}

static inline void knn2_insert_f(float d, int j, float dist[2], int32_t idx[2])
{
    if (d < dist[1]) {
        if (dist[0] > d) { dist[1] = dist[0]; idx[1] = idx[0]; dist[0] = d; idx[0] = j; }
        else             { dist[1] = d; idx[1] = j; }
    }
}

void orc_knn2_l2_f32(const float* q, int nq, const float* t, int nt, int dim,
                     size_t ldq, size_t ldt, int32_t* idx2, float* dist2)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nq; ++i) {
        float dist[2] = { FLT_MAX, FLT_MAX };
        int32_t idx[2] = { -1, -1 };
        const float* a = q + (size_t)i * ldq;
        for (int j = 0; j < nt; ++j) {
            float d = sqrtf(normL2Sqr_sse(a, t + (size_t)j * ldt, dim));
            knn2_insert_f(d, j, dist, idx);
        }
        idx2[2 * i] = idx[0]; idx2[2 * i + 1] = idx[1];
        dist2[2 * i] = dist[0]; dist2[2 * i + 1] = dist[1];
    }
}

void orc_l2_distance_matrix_f32(const float* q, int nq, const float* t, int nt, int dim,
                                size_t ldq, size_t ldt, float* dist, size_t ldd)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nq; ++i)
        for (int j = 0; j < nt; ++j)
            dist[(size_t)i * ldd + j] = sqrtf(normL2Sqr_sse(q + (size_t)i * ldq, t + (size_t)j * ldt, dim));
}

static unsigned char g_tab2[256];
static int g_tab2_ready = 0;
static void init_tab2(void)
{
    for (int x = 0; x < 256; ++x) {
        int c = 0;
        for (int k = 0; k < 4; ++k) c += ((x >> (2 * k)) & 3) != 0;
        g_tab2[x] = (unsigned char)c;
    }
    g_tab2_ready = 1;
}

void orc_knn2_hamming2_u8(const uint8_t* q, int nq, const uint8_t* t, int nt, int nbytes,
                          size_t ldq, size_t ldt, int32_t* idx2, float* dist2)
{
    if (!g_tab2_ready) init_tab2();
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nq; ++i) {
        int dist[2] = { INT_MAX, INT_MAX };
        int32_t idx[2] = { -1, -1 };
        const uint8_t* a = q + (size_t)i * ldq;
        for (int j = 0; j < nt; ++j) {
            const uint8_t* b = t + (size_t)j * ldt;
            int d = 0;
            for (int k = 0; k < nbytes; ++k) d += g_tab2[a[k] ^ b[k]];
            if (d < dist[1]) {
                if (dist[0] > d) { dist[1] = dist[0]; idx[1] = idx[0]; dist[0] = d; idx[0] = j; }
                else             { dist[1] = d; idx[1] = j; }
            }
        }
        idx2[2 * i] = idx[0]; idx2[2 * i + 1] = idx[1];
        /* knnMatchImpl: dist.convertTo(temp, CV_32F) [3P]; INT_MAX -> 2147483648.f */
        dist2[2 * i] = (float)dist[0]; dist2[2 * i + 1] = (float)dist[1];
    }
}

/* NView:880-908.  `float > 0.6 * float` is evaluated in double (NView:884, 900); the absolute
 * gate `5 * max(min_dist, 10.0f)` stays in float (int 5 -> float, NView:901).  Rows with fewer
 * than two neighbours are skipped (the reference indexes knn_matches[i][1] unconditionally). */
int orc_ratio_filter(const int32_t* idx2, const float* dist2, int nq,
                     double ratio, float floor_, float mult, orc_dmatch* out)
{
    float min_dist = FLT_MAX;
    for (int i = 0; i < nq; ++i) {
        if (idx2[2 * i] < 0 || idx2[2 * i + 1] < 0) continue;
        float d0 = dist2[2 * i], d1 = dist2[2 * i + 1];
        if ((double)d0 > ratio * (double)d1) continue;
        if (d0 < min_dist) min_dist = d0;
    }
    int n = 0;
    float gate = mult * (min_dist > floor_ ? min_dist : floor_);   /* std::max(min_dist, 10.0f) */
    for (int i = 0; i < nq; ++i) {
        if (idx2[2 * i] < 0 || idx2[2 * i + 1] < 0) continue;
        float d0 = dist2[2 * i], d1 = dist2[2 * i + 1];
        if ((double)d0 > ratio * (double)d1 || d0 > gate) continue;
        out[n].queryIdx = i; out[n].trainIdx = idx2[2 * i]; out[n].imgIdx = 0; out[n].distance = d0;
        ++n;
    }
    return n;
}
