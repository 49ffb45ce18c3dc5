/*
 * oracle/mso.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's hot path (AaltoML/SLAM-module: image pyramid, ORB
 * extraction, Hamming matching, local bundle adjustment).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product (slam-module_amd/) never
 * links or calls it and fails loudly when its HIP library is missing.
 *
 * Every function cites the reference file:line it restates.  Pinning status:
 *   pinned by the reference's own data  : ORB pattern table (oracle/_ref dump of orb_point_pairs.h)
 *   pinned by in-tree source text only  : ic_angle, steered BRIEF bit packing, util::cos/sin,
 *                                         Hamming distance, angle histogram, matcher accept rules,
 *                                         pyramid geometry, keypoint quotas, BA problem construction
 *   PARITY UNPINNED vs the reference    : cv::resize / cv::GaussianBlur pixels, the corner detector
 *                                         (external tracker::FeatureDetector), cv::fastAtan2, the g2o
 *                                         LM iterates, the DBoW2 tree descent (bow.c) -- arithmetic lives in un-vendored, un-pinned
 *                                         third-party code; restated here from their published
 *                                         algorithms and that restatement is the spec.
 */
#ifndef MSO_H
#define MSO_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSO_MAX_LEVELS 16
#define MSO_PATCH_RADIUS 19      /* static_settings.hpp:14 ORB_PATCH_RADIUS */
#define MSO_HALF_PATCH 15        /* static_settings.hpp:15-16 ORB_FAST_PATCH_SIZE/2 */

/* ---- S1/S2: pyramid geometry (static_settings.cpp:9-60, image_pyramid.cpp:76-78) ---- */
void mso_scale_factors(int levels, float f, float *out);
void mso_level_sigma_sq(int levels, float f, float *out);
void mso_level_quotas(int levels, float f, int max_kpts, int *out);
void mso_level_sizes(int levels, float f, int w0, int h0, int *w, int *h);
void mso_umax(int *umax16);   /* orb_extractor.cpp:174-186 */

/* ---- P1/P2: pyramid pixels (image_pyramid.cpp:75-85; OpenCV 8U semantics restated) ---- */
void mso_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride,
                          uint8_t *dst, int dw, int dh, int dstride);
void mso_gauss7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride);
extern const int mso_gauss7_q8[7];

/* ---- D1: corner detector (feature_detector.cpp:68-134; detector core is this build's own) ---- */
/* FAST-9/16 score of one pixel (0 = not a corner at threshold 0). */
int mso_fast_score(const uint8_t *img, int stride, int x, int y);
/* Detect on one level: FAST score > threshold, 3x3 strict NMS, top-`quota` by
 * key = ((255-score)<<24 | y*w+x) ascending, then the 19 px border filter of
 * feature_detector.cpp:106-123.  With min_dist >= 2 the best min(4*quota, 4096) corners are walked in key order and a
 * corner is kept only if no already kept corner lies closer than min_dist (squared distance < min_dist^2), until `quota`
 * are kept (the maxTracks / minDistance contract of the external detector call, feature_detector.cpp:97-98).
 * Returns count; xs/ys sized >= quota. */
int mso_detect_level(const uint8_t *img, int w, int h, int stride, int threshold, int quota, int min_dist,
                     int *xs, int *ys, int *scores);
/* feature_detector.cpp:79-82: per-level minimum distance from gfttMinDistance */
int mso_level_min_dist(float gftt_min_distance, int w, int h);

/* ---- O1/O2: orientation + descriptor (orb_extractor.cpp:245-352) ---- */
float mso_fast_atan2(float y, float x);               /* cv::fastAtan2 restated */
float mso_cos(float v);                               /* openvslam/trigonometric.h:25-42 */
float mso_sin(float v);                               /* openvslam/trigonometric.h:44-46 */
float mso_ic_angle(const uint8_t *img, int stride, int x, int y);
void mso_orb_descriptor(const uint8_t *blur, int stride, int x, int y, float angle_deg, uint32_t desc[8]);
extern const int8_t mso_orb_pattern[1024];

/* ---- O3: whole extractor (orb_extractor.cpp:73-164) ---- */
typedef struct {
    int levels;            /* orbScaleLevels */
    float scale_factor;    /* orbScaleFactor */
    int max_kpts;          /* maxKeypoints */
    int lk_track_level;    /* orbLkTrackLevel */
    int fast_threshold;    /* detector threshold (this build's detector) */
    float min_distance;    /* tracker.gfttMinDistance (0 = no suppression) */
} mso_orb_config;

typedef struct {
    int n;                 /* keypoints written */
    float *x, *y;          /* level-0 coords */
    float *angle;          /* degrees [0,360) */
    int32_t *octave;
    uint32_t *desc;        /* 8 words / keypoint */
    int32_t *track_id;     /* -1 for detected points */
} mso_keypoints;

/* pyramid buffers are allocated by the call and returned for inspection (free with mso_free_pyramid). */
typedef struct {
    int levels;
    int w[MSO_MAX_LEVELS], h[MSO_MAX_LEVELS];
    uint8_t *img[MSO_MAX_LEVELS];
    uint8_t *blur[MSO_MAX_LEVELS];
} mso_pyramid;

int mso_build_pyramid(const mso_orb_config *cfg, const uint8_t *img, int w, int h, int stride, mso_pyramid *out);
void mso_free_pyramid(mso_pyramid *p);

/* valid_mask: optional w*h u8 (level-0), 0 = camera-invalid pixel (camera.isValidPixel stand-in). */
int mso_orb_extract(const mso_orb_config *cfg, const uint8_t *img, int w, int h, int stride,
                    const uint8_t *valid_mask,
                    const float *track_xy, const int32_t *track_id, int n_tracks,
                    mso_keypoints *out, int capacity);

/* ---- H1: Hamming distance (openvslam/match_base.h:18-39) ---- */
unsigned mso_hamming256(const uint32_t *a, const uint32_t *b);

/* best / second-best of each query over the targets (no greedy state):
 * strict '<' updates => lowest index wins ties (keyframe_matcher.cpp:106-112).
 * q_bucket/t_bucket: optional bucket ids (compare only equal buckets); t_valid optional mask. */
void mso_hamming_best2(const uint32_t *q, int nq, const uint32_t *t, int nt,
                       const int32_t *q_bucket, const int32_t *t_bucket, const uint8_t *t_valid,
                       int32_t *best_idx, uint16_t *best_dist, uint16_t *second_dist);

/* ---- A1: rotation histogram (openvslam/match_angle_checker.h:60-134) ---- */
/* Returns number of invalid entries; invalid[] receives their ids (bin order, insertion order).
 * Ties between bin sizes at the top-3 boundary are broken towards the LOWER bin index here
 * (std::sort in the reference leaves that order unspecified). */
int mso_angle_check(const float *delta_angle, const int32_t *ids, int n, int32_t *invalid);

/* ---- M1: matchForLoopClosures (keyframe_matcher.cpp:50-158) on flat arrays ---- */
/* bow CSR: node ids ascending (std::map order), idx lists per node in insertion order. */
typedef struct {
    int n_nodes;
    const int32_t *node_id;     /* [n_nodes] strictly ascending */
    const int32_t *node_start;  /* [n_nodes+1] */
    const int32_t *kp_idx;      /* keypoint indices, concatenated */
} mso_bow;

int mso_match_loop_closure(const uint32_t *desc1, const float *angle1, const uint8_t *usable1, int n1, const mso_bow *bow1,
                           const uint32_t *desc2, const float *angle2, const uint8_t *usable2, int n2, const mso_bow *bow2,
                           float lowe_ratio, int check_orientation, int32_t *matched /* [n1], -1 = none */);

/* ---- M2: matchForTriangulationDBoW (keyframe_matcher.cpp:160-293) on flat arrays ---- */
/* usable = keypoint has NO map point yet.  E = create_E_21 (essential_solver.cc:157-162) row-major 3x3. */
int mso_match_triangulation(const uint32_t *desc1, const float *angle1, const int32_t *octave1, const double *bearing1,
                            const uint8_t *usable1, int n1, const mso_bow *bow1,
                            const uint32_t *desc2, const float *angle2, const double *bearing2,
                            const uint8_t *usable2, int n2, const mso_bow *bow2,
                            const double *E12, const float *scale_factors, float residual_deg_thr,
                            int check_orientation, int32_t *matched /* [n1] */);
void mso_create_E21(const double *R1w, const double *t1w, const double *R2w, const double *t2w, double *E /* 3x3 row-major */);

/* ---- M3/M4/M5 scoring cores (keyframe_matcher.cpp:349-386, :479-499, :600-627) ---- */
/* One query descriptor against an explicit candidate list; `skip` marks already-bound targets.
 * Returns best index (or -1); writes best/second distance and the octave of the best and second. */
int mso_best2_candidates(const uint32_t *qdesc, const uint32_t *tdesc, const int32_t *cand, int ncand,
                         const uint8_t *skip, const int32_t *t_octave,
                         unsigned *best, unsigned *second, int *best_oct, int *second_oct);

/* ---- N2: FeatureSearch::getFeaturesAround (feature_search.cpp:33-48) on arrays sorted by y; writes sorted positions ---- */
int mso_features_around(const float *sx, const float *sy, int n, float x, float y, float r, int32_t *out_pos);

/* ---- N4: MapPoint::updateDescriptor (map_point.cpp:75-116): index of the median-Hamming medoid, -1 when n == 0 ---- */
int mso_descriptor_medoid(const uint32_t *desc /* [n][8] */, int n);

/* ---- N3: DBoW2 vocabulary-tree descent behind BowIndex::transform (bow_index.cpp:59-93); see bow.c (PARITY UNPINNED) ---- */
void mso_bow_transform(int n_nodes, const int32_t *parent, const uint32_t *node_desc, const double *node_weight,
                       const int32_t *node_word, int depth_levels, const uint32_t *desc, int n, int levels_up,
                       int32_t *word, double *weight, int32_t *node);
int mso_bow_assemble(const int32_t *word, const double *weight, const int32_t *node, int n,
                     int32_t *out_words, double *out_values, int32_t *fv_nodes, int32_t *fv_start, int32_t *fv_feat, int *n_fv);

/* ---- B1-B6: bundle adjustment (bundle_adjuster.cpp:43-111, :141-394; g2o semantics restated in ba.c) ---- */
typedef struct {
    int n_pose, n_point, n_obs, n_edge;
    double *pose;               /* [n_pose*7] qx,qy,qz,qw,tx,ty,tz  world->camera (g2o SE3Quat); updated in place */
    const uint8_t *pose_fixed;  /* [n_pose] */
    double *point;              /* [n_point*3]; updated in place */
    const uint8_t *point_fixed; /* [n_point] or NULL */
    const int32_t *obs_pose, *obs_point;   /* [n_obs] */
    const double *obs_uv;       /* [n_obs*2] bearing.xy / bearing.z (bundle_adjuster.cpp:52) */
    const double *obs_info;     /* [n_obs]  focal^2 / sigma^2_octave (bundle_adjuster.cpp:51) */
    double huber_delta;         /* sqrt(5.991) (bundle_adjuster.cpp:56); <= 0 disables the kernel */
    const int32_t *edge_i, *edge_j;        /* EdgeSE3Expmap vertices 0 and 1 */
    const double *edge_meas;    /* [n_edge*7] */
    const double *edge_info;    /* [n_edge*36] row-major */
    int max_iters;
} mso_ba_problem;

typedef struct { int iters, trials_total, stop_reason; double lambda, chi2_init, chi2_final; } mso_ba_stats;

/* full_system = 1 solves the undamped-ordering-free dense (poses+points) system, 0 the Schur complement. */
int mso_ba_solve(mso_ba_problem *P, double *chi2_per_obs, mso_ba_stats *st, int flags);      /* flags: see ba.c */
void mso_se3_exp(const double *update6, double *pose7);
void mso_se3_log(const double *pose7, double *out6);
/* test hooks: residual + Jacobians of the two edge types */
void mso_ba_proj_edge(const double *pose7, const double *X, const double *uv, double *e2, double *Jp12, double *Jl6);
void mso_ba_pose_edge(const double *Ti, const double *Tj, const double *M, double *e6, double *Ji36, double *Jj36);
void mso_se3_mul(const double *a, const double *b, double *r);

/* ---- synthetic inputs (integer-only) ---- */
void mso_synth_frame(uint8_t *img, int w, int h, uint32_t seed, int shift_x, int shift_y);

#ifdef __cplusplus
}
#endif
#endif
