/*
 * oracle/matcher.c -- CPU ORACLE (test infrastructure only; see mso.h header).
 * Hamming distance, rotation histogram, and the matcher policies on flat arrays.
 */
#include "mso.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define THR_LOW 50u      /* keyframe_matcher.hpp:10 / match_base.h:13 */
#define THR_HIGH 100u    /* keyframe_matcher.hpp:11 */
#define MAX_DIST 256u    /* keyframe_matcher.hpp:12 */

/* H1: openvslam/match_base.h:18-39 -- SWAR popcount over 8 words, restated verbatim in meaning */
unsigned mso_hamming256(const uint32_t *a, const uint32_t *b) {
    unsigned dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t v = a[i] ^ b[i];
        v -= ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (((v + (v >> 4)) & 0x0F0F0F0Fu) * 0x01010101u) >> 24;
    }
    return dist;
}

/* best / second-best scan, update rule of keyframe_matcher.cpp:106-112 */
void mso_hamming_best2(const uint32_t *q, int nq, const uint32_t *t, int nt,
                       const int32_t *qb, const int32_t *tb, const uint8_t *t_valid,
                       int32_t *best_idx, uint16_t *best_dist, uint16_t *second_dist) {
    for (int i = 0; i < nq; ++i) {
        unsigned best = MAX_DIST, second = MAX_DIST; int bi = -1;
        for (int j = 0; j < nt; ++j) {
            if (t_valid && !t_valid[j]) continue;
            if (qb && tb && qb[i] != tb[j]) continue;
            const unsigned d = mso_hamming256(q + 8 * (size_t)i, t + 8 * (size_t)j);
            if (d < best) { second = best; best = d; bi = j; }
            else if (d < second) second = d;
        }
        best_idx[i] = bi; best_dist[i] = (uint16_t)best; second_dist[i] = (uint16_t)second;
    }
}

/* A1: openvslam/match_angle_checker.h:72-83 (binning), :108-134 (invalid = all but the 3 largest bins) */
static int angle_bin(float delta) {
    if (delta < 0.0) delta = (float)(delta + 360.0);
    if (360.0 <= delta) delta = (float)(delta - 360.0);
    const float inv_len = 1.0f / 30;
    return (int)lrintf(delta * inv_len);
}

int mso_angle_check(const float *delta_angle, const int32_t *ids, int n, int32_t *invalid) {
    int count[30] = {0}, bin_of[n > 0 ? n : 1];
    for (int i = 0; i < n; ++i) { int b = angle_bin(delta_angle[i]); if (b < 0 || b >= 30) b = 29; bin_of[i] = b; count[b]++; }
    int order[30];
    for (int i = 0; i < 30; ++i) order[i] = i;
    for (int i = 1; i < 30; ++i) {                 /* stable insertion sort: size desc, bin asc on ties */
        int k = order[i], j = i - 1;
        while (j >= 0 && count[order[j]] < count[k]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = k;
    }
    int valid_bin[30] = {0};
    for (int i = 0; i < 3; ++i) valid_bin[order[i]] = 1;
    int m = 0;
    for (int b = 0; b < 30; ++b) {
        if (valid_bin[b]) continue;
        for (int i = 0; i < n; ++i) if (bin_of[i] == b) invalid[m++] = ids[i];
    }
    return m;
}

/* M1: keyframe_matcher.cpp:50-158 */
int mso_match_loop_closure(const uint32_t *desc1, const float *angle1, const uint8_t *usable1, int n1, const mso_bow *bow1,
                           const uint32_t *desc2, const float *angle2, const uint8_t *usable2, int n2, const mso_bow *bow2,
                           float lowe_ratio, int check_orientation, int32_t *matched) {
    int num = 0;
    for (int i = 0; i < n1; ++i) matched[i] = -1;
    uint8_t *used2 = (uint8_t *)calloc(n2 > 0 ? n2 : 1, 1);
    float *dang = (float *)malloc(sizeof(float) * (n1 > 0 ? n1 : 1));
    int32_t *did = (int32_t *)malloc(sizeof(int32_t) * (n1 > 0 ? n1 : 1));
    int nd = 0, a = 0, b = 0;
    while (a < bow1->n_nodes && b < bow2->n_nodes) {                 /* ordered-map merge :70-147 */
        const int32_t ida = bow1->node_id[a], idb = bow2->node_id[b];
        if (ida < idb) { ++a; continue; }                            /* lower_bound() steps to the same place */
        if (idb < ida) { ++b; continue; }
        for (int p = bow1->node_start[a]; p < bow1->node_start[a + 1]; ++p) {
            const int i1 = bow1->kp_idx[p];
            if (!usable1[i1]) continue;                               /* :79-84 */
            unsigned best = MAX_DIST, second = MAX_DIST; int bi = -1;
            for (int r = bow2->node_start[b]; r < bow2->node_start[b + 1]; ++r) {
                const int i2 = bow2->kp_idx[r];
                if (!usable2[i2]) continue;                           /* :94-96 */
                if (used2[i2]) continue;                              /* :98-100 */
                const unsigned d = mso_hamming256(desc1 + 8 * (size_t)i1, desc2 + 8 * (size_t)i2);
                if (d < best) { second = best; best = d; bi = i2; }
                else if (d < second) second = d;
            }
            if (THR_LOW < best) continue;                             /* :115 */
            if (lowe_ratio * second < (float)best) continue;          /* :120 */
            matched[i1] = bi; used2[bi] = 1; ++num;                   /* :126-130 */
            if (check_orientation) { dang[nd] = angle1[i1] - angle2[bi]; did[nd] = i1; ++nd; }
        }
        ++a; ++b;
    }
    if (check_orientation) {                                          /* :149-155 */
        int32_t *inv = (int32_t *)malloc(sizeof(int32_t) * (nd > 0 ? nd : 1));
        const int m = mso_angle_check(dang, did, nd, inv);
        for (int i = 0; i < m; ++i) { matched[inv[i]] = -1; --num; }
        free(inv);
    }
    free(used2); free(dang); free(did);
    return num;
}

/* openvslam/essential_solver.cc:149-162 */
void mso_create_E21(const double *R1, const double *t1, const double *R2, const double *t2, double *E) {
    double R21[9], t21[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += R2[3 * i + k] * R1[3 * j + k];     /* R2 * R1^T */
            R21[3 * i + j] = s;
        }
    for (int i = 0; i < 3; ++i) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += -R21[3 * i + k] * t1[k];
        t21[i] = s + t2[i];
    }
    const double S[9] = {0, -t21[2], t21[1], t21[2], 0, -t21[0], -t21[1], t21[0], 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += S[3 * i + k] * R21[3 * k + j];
            E[3 * i + j] = s;
        }
}

/* keyframe_matcher.cpp:23-44 */
static int epipolar_ok(const double *b1, const double *b2, const double *E, float scale1, float thr_deg) {
    double n[3];
    for (int i = 0; i < 3; ++i) n[i] = E[3 * i] * b2[0] + E[3 * i + 1] * b2[1] + E[3 * i + 2] * b2[2];
    const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    const double cr = (n[0] * b1[0] + n[1] * b1[1] + n[2] * b1[2]) / nn;
    const double res = M_PI / 2.0 - fabs(acos(cr));
    const double thr = thr_deg * M_PI / 180.0;
    return res < thr * scale1;
}

/* M2: keyframe_matcher.cpp:160-293 */
int mso_match_triangulation(const uint32_t *desc1, const float *angle1, const int32_t *octave1, const double *bearing1,
                            const uint8_t *usable1, int n1, const mso_bow *bow1,
                            const uint32_t *desc2, const float *angle2, const double *bearing2,
                            const uint8_t *usable2, int n2, const mso_bow *bow2,
                            const double *E12, const float *scale_factors, float residual_deg_thr,
                            int check_orientation, int32_t *matched) {
    int num = 0;
    for (int i = 0; i < n1; ++i) matched[i] = -1;
    uint8_t *used2 = (uint8_t *)calloc(n2 > 0 ? n2 : 1, 1);
    float *dang = (float *)malloc(sizeof(float) * (n1 > 0 ? n1 : 1));
    int32_t *did = (int32_t *)malloc(sizeof(int32_t) * (n1 > 0 ? n1 : 1));
    int nd = 0, a = 0, b = 0;
    while (a < bow1->n_nodes && b < bow2->n_nodes) {
        const int32_t ida = bow1->node_id[a], idb = bow2->node_id[b];
        if (ida < idb) { ++a; continue; }
        if (idb < ida) { ++b; continue; }
        for (int p = bow1->node_start[a]; p < bow1->node_start[a + 1]; ++p) {
            const int i1 = bow1->kp_idx[p];
            if (!usable1[i1]) continue;                               /* :205-209 */
            unsigned best = THR_LOW; int bi = -1;                     /* :213-214 */
            for (int r = bow2->node_start[b]; r < bow2->node_start[b + 1]; ++r) {
                const int i2 = bow2->kp_idx[r];
                if (!usable2[i2]) continue;                           /* :217-221 */
                if (used2[i2]) continue;                              /* :224-226 */
                const unsigned d = mso_hamming256(desc1 + 8 * (size_t)i1, desc2 + 8 * (size_t)i2);
                if (d > THR_LOW || d > best) continue;                /* :231 (ties: LAST wins) */
                if (epipolar_ok(bearing1 + 3 * (size_t)i1, bearing2 + 3 * (size_t)i2, E12,
                                scale_factors[octave1[i1]], residual_deg_thr)) { bi = i2; best = d; }
            }
            if (bi < 0) continue;
            used2[bi] = 1; matched[i1] = bi; ++num;                   /* :249-251 */
            if (check_orientation) { dang[nd] = angle1[i1] - angle2[bi]; did[nd] = i1; ++nd; }
        }
        ++a; ++b;
    }
    if (check_orientation) {                                          /* :271-277 */
        int32_t *inv = (int32_t *)malloc(sizeof(int32_t) * (nd > 0 ? nd : 1));
        const int m = mso_angle_check(dang, did, nd, inv);
        for (int i = 0; i < m; ++i) { matched[inv[i]] = -1; --num; }
        free(inv);
    }
    free(used2); free(dang); free(did);
    return num;
}

/* M3/M4/M5 scoring core: keyframe_matcher.cpp:356-378 (best+second with octaves), :482-494, :600-623 */
int mso_best2_candidates(const uint32_t *qdesc, const uint32_t *tdesc, const int32_t *cand, int ncand,
                         const uint8_t *skip, const int32_t *t_octave,
                         unsigned *best, unsigned *second, int *best_oct, int *second_oct) {
    unsigned b = MAX_DIST, s = MAX_DIST; int bl = -1, sl = -1, bi = -1;
    for (int c = 0; c < ncand; ++c) {
        const int j = cand[c];
        if (skip && skip[j]) continue;
        const unsigned d = mso_hamming256(qdesc, tdesc + 8 * (size_t)j);
        const int lvl = t_octave ? t_octave[j] : 0;
        if (d < b) { s = b; b = d; sl = bl; bl = lvl; bi = j; }
        else if (d < s) { sl = lvl; s = d; }
    }
    *best = b; *second = s; *best_oct = bl; *second_oct = sl;
    return bi;
}

/* N4: MapPoint::updateDescriptor (map_point.cpp:75-116): the observation whose MEDIAN Hamming distance to all
 * observations (itself included, distance 0) is smallest; median = sorted[(unsigned)(0.5*(n-1))]; first index wins
 * ties; an index is accepted only when its median is < MAX_HAMMING_DIST, otherwise index 0 stays. -1 when n == 0. */
static int cmp_u(const void *a, const void *b) { unsigned x = *(const unsigned *)a, y = *(const unsigned *)b; return (x > y) - (x < y); }
int mso_descriptor_medoid(const uint32_t *desc, int n) {
    if (n <= 0) return -1;
    unsigned *m = (unsigned *)malloc(sizeof(unsigned) * (size_t)n * n), *row = (unsigned *)malloc(sizeof(unsigned) * n);
    for (int i = 0; i < n; ++i) {
        m[(size_t)i * n + i] = 0;
        for (int j = i + 1; j < n; ++j) m[(size_t)i * n + j] = m[(size_t)j * n + i] = mso_hamming256(desc + 8 * (size_t)i, desc + 8 * (size_t)j);
    }
    unsigned best = MAX_DIST; int best_idx = 0;
    for (int i = 0; i < n; ++i) {
        memcpy(row, m + (size_t)i * n, sizeof(unsigned) * n);
        qsort(row, n, sizeof(unsigned), cmp_u);
        const unsigned med = row[(unsigned)(0.5 * (n - 1))];
        if (med < best) { best = med; best_idx = i; }
    }
    free(m); free(row);
    return best_idx;
}

/* N2: FeatureSearch::getFeaturesAround (feature_search.cpp:33-48) on arrays ALREADY sorted by y (stable: ties keep index order;
 * the reference's std::sort leaves them unspecified): lower_bound on y - r, walk while y <= y + r, keep dx*dx + dy*dy < r*r.
 * Writes sorted positions; returns how many. */
int mso_features_around(const float *sx, const float *sy, int n, float x, float y, float r, int32_t *out_pos) {
    const float ylo = y - r, yhi = y + r, r2 = r * r;
    int lo = 0, hi = n, cnt = 0;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sy[mid] < ylo) lo = mid + 1; else hi = mid; }
    for (int p = lo; p < n && sy[p] <= yhi; ++p) {
        const float dx = x - sx[p], dy = y - sy[p];
        const float a = dx * dx, b = dy * dy;              /* separate roundings: no fused multiply-add */
        if (a + b < r2) out_pos[cnt++] = p;
    }
    return cnt;
}
