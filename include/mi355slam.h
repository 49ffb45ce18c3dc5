/*
 * mi355slam.h -- C ABI of libmi355slam.so: the MI355X (gfx950) implementation of the
 * AaltoML/SLAM-module hot path (image pyramid + ORB extraction, Hamming matching, local BA).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Each entry point
 * names the reference interface it replaces (file:line relative to the reference tree).
 * Host-side C++ shims with the reference's own signatures live in slam-module_amd/host/.
 *
 * Conventions
 *   - every function returns MS_OK (0) or a negative ms_status; ms_last_error() gives the text.
 *     Nothing throws across the ABI (the reference itself has no exceptions / error codes:
 *     assert + bool/empty returns, e.g. bundle_adjuster.cpp:239,411).
 *   - a ms_ctx owns one HIP stream on one device; objects created from it are single-threaded
 *     (the reference calls this path from one backend thread, mapper.cpp:229-279).  Distinct
 *     contexts may be used concurrently from different threads.
 *   - "device pointer" arguments must be memory of the context's device; "host pointer"
 *     arguments are ordinary memory (pinned memory makes the copies asynchronous).
 *   - there is NO CPU fallback: without a usable gfx950 device ms_ctx_create() fails.
 */
#ifndef MI355SLAM_H
#define MI355SLAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MS_MAX_LEVELS 16
#define MS_ORB_PATCH_RADIUS 19       /* static_settings.hpp:14 */
#define MS_HAMMING_THR_LOW 50        /* keyframe_matcher.hpp:10 */
#define MS_HAMMING_THR_HIGH 100      /* keyframe_matcher.hpp:11 */
#define MS_HAMMING_MAX 256           /* keyframe_matcher.hpp:12 */

typedef enum {
    MS_OK = 0,
    MS_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
    MS_ERR_NO_DEVICE = -2,    /* no gfx950 device, or HIP runtime failure at init */
    MS_ERR_HIP = -3,          /* a HIP call failed; see ms_last_error */
    MS_ERR_CAPACITY = -4,     /* a fixed capacity given at create time was exceeded */
    MS_ERR_NUMERIC = -5,      /* BA: non-finite state */
    MS_ERR_TOO_LATE = -6      /* ms_prepare_process: the GPU runtime of this process is already up; the call can no longer have an effect */
} ms_status;

typedef struct ms_ctx ms_ctx;
typedef struct ms_orb ms_orb;

/* ---------------------------------------------------------------------------------------------
 * Context
 * ------------------------------------------------------------------------------------------- */
/* Optional, once per process, BEFORE the first HIP call of the process (ms_ctx_create included): tells the runtime how many contexts (= sequences, each
 * with its own stream) will drive the same GPU at the same time.  The HIP runtime maps streams onto a handful of hardware queues (4 by default), and kernels
 * of streams that share a queue run one after the other -- a 1.8 ms local-BA launch of one sequence then holds up the 10 us front-end kernels of another:
 * eight sequences on one GPU measured 4 300 frames/s with 4 queues, 5 600 with 8 and 8 400 - 8 700 with 12 ... 32 (the process holds more streams than its sequences' -- a context's
 * download stream, the application's own --, and with exactly one queue per context two sequences share one; tools/hw_queue_sweep.sh).  Sets GPU_MAX_HW_QUEUES to max(4, min(2 * concurrent_contexts, 32)) unless the variable is already set (the caller's own setting wins: MS_OK, nothing changed).  Once the runtime is
 * initialised -- by this library or by anybody else in the process: the kernel driver's device node is open -- the variable has been read and the call returns
 * MS_ERR_TOO_LATE without touching the environment, so a host that calls it late learns that it runs on the default number of queues.  (The reference has no
 * counterpart: its back end is one CPU thread per sequence, mapper.cpp:268-269.) */
int ms_prepare_process(int concurrent_contexts);
int ms_ctx_create(int device, ms_ctx **out);
void ms_ctx_destroy(ms_ctx *ctx);
int ms_ctx_sync(ms_ctx *ctx);                 /* hipStreamSynchronize on the context stream */
void *ms_ctx_stream(ms_ctx *ctx);             /* the hipStream_t, for event timing by the caller */
const char *ms_last_error(const ms_ctx *ctx); /* never NULL; valid until the next call on ctx */

/* Stage ranges in a profiler's marker trace (rocprofv3 --marker-trace): with on = 1 the entry points below bracket their work with roctx ranges named
 * like the reference's own timers where it has them (mapper_helpers.cpp:1044 "poseBundleAdjust", :1080 "localBundleAdjust", :1193 "Bow index
 * transform") and by stage elsewhere ("pyramid", "detect", "describe" inside ms_orb_extract; "match"; "ms_ba_create", "ms_ba_solve",
 * "ms_ba_download" -- the host mirrors wrap those in "localBundleAdjust" / "poseBundleAdjust" / "globalBundleAdjust").  The ranges mark where
 * the work is ENQUEUED on the calling thread.  Process-wide, off by default (also switched on by MS_TRACE_RANGES=1 in the environment); the roctx
 * library is looked up at run time, MS_ERR_INVALID if there is none.  ms_trace_range_push / pop let a caller (the host mirrors do) add its own. */
int ms_set_trace_ranges(int on);
void ms_trace_range_push(const char *name);
void ms_trace_range_pop(void);
const char *ms_version(void);
/* HIP-event timing on the context stream (bench.py measures the hot path with these). */
int ms_timer_start(ms_ctx *ctx);
int ms_timer_stop_ms(ms_ctx *ctx, float *ms); /* records, synchronises, returns elapsed ms */
/* 1024 general event slots (created on first use): mark = hipEventRecord on the context stream; elapsed synchronises on slot b. */
int ms_event_mark(ms_ctx *ctx, int slot);
int ms_event_elapsed_ms(ms_ctx *ctx, int slot_a, int slot_b, float *ms);
/* plain device memory helpers so a C caller needs no HIP headers */
int ms_dev_alloc(ms_ctx *ctx, size_t bytes, void **out);
int ms_dev_free(ms_ctx *ctx, void *p);
/* Page-locked host memory for frames going in and results coming out: copies from / to it are truly asynchronous (the copy engines read it
 * directly), which is what lets ms_orb_extract overlap a batch's copies with its kernels and ms_dev_download_async run under the next batch. */
int ms_host_alloc(ms_ctx *ctx, size_t bytes, void **out);
int ms_host_free(ms_ctx *ctx, void *p);
int ms_dev_upload(ms_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int ms_dev_download(ms_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* The same copy without waiting for it: it runs on a stream of its own, after everything enqueued on the context stream so far, while the context
 * stream goes on (the next batch's kernels and copies run under it).  `dst` should be pinned host memory (pageable memory is staged by the runtime and
 * the call then waits).  What the copy READS must stay intact until it is done: ms_orb_extract orders itself after the downloads issued so far (it
 * is the call that overwrites the extractor's outputs, and everything enqueued after it is ordered behind it); other writers of the source are the
 * caller's business.  ms_dev_download_wait blocks the host until every asynchronous download has arrived (ms_ctx_sync does, too). */
int ms_dev_download_async(ms_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int ms_dev_download_wait(ms_ctx *ctx);

/* ---------------------------------------------------------------------------------------------
 * S1/S2: pyramid geometry -- replaces StaticSettings (static_settings.cpp:9-60) and the level
 * sizing of image_pyramid.cpp:76-78.  Host-only arithmetic, exposed so callers size buffers.
 * ------------------------------------------------------------------------------------------- */
int ms_scale_factors(int levels, float scale_factor, float *out);
int ms_level_sigma_sq(int levels, float scale_factor, float *out);
int ms_level_quotas(int levels, float scale_factor, int max_kpts, int32_t *out);
int ms_level_sizes(int levels, float scale_factor, int width, int height, int32_t *w, int32_t *h);

/* ---------------------------------------------------------------------------------------------
 * ORB extractor -- replaces OrbExtractor::build / detectAndExtract (orb_extractor.hpp:11-30,
 * orb_extractor.cpp:73-164), ImagePyramid (image_pyramid.hpp:16-30, image_pyramid.cpp:68-86) and
 * FeatureDetector (feature_detector.hpp:15-24, feature_detector.cpp:68-134).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t width, height;     /* frame size (fixed per extractor, like the lazily-built pyramid) */
    int32_t levels;            /* parameters.slam.orbScaleLevels */
    float scale_factor;        /* parameters.slam.orbScaleFactor */
    int32_t max_kpts;          /* parameters.slam.maxKeypoints */
    int32_t lk_track_level;    /* parameters.slam.orbLkTrackLevel */
    int32_t fast_threshold;    /* this build's detector: FAST-9/16 threshold (score > threshold) */
    int32_t max_tracks;        /* capacity for tracker features per frame */
    int32_t max_batch;         /* frames per ms_orb_extract call (>= 1) */
    float min_distance;        /* parameters.tracker.gfttMinDistance (feature_detector.cpp:79-82): keypoints of a level keep
                                  floor(min_distance * min(w,h)/720 * 0.8 + 0.5) pixels apart; 0 = no suppression */
} ms_orb_config;

/* Per-frame outputs, structure-of-arrays, `capacity` = max_tracks + max(max_kpts, sum of the per-level quotas) slots per frame
 * (ms_orb_capacity; the quotas are rounded level by level, static_settings.cpp:52, and may add up to a few more than max_kpts).
 * Frame f's keypoint i is element f*capacity + i of each array (desc: 8 words per slot).
 * Order inside a frame: tracker features first, then detected points level-major
 * (orb_extractor.cpp:136-162); each level ordered by (FAST score desc, y*w+x asc). */
typedef struct {
    int32_t capacity;
    int32_t *count;      /* [batch] */
    float *x, *y;        /* level-0 pixel coordinates (KeyPoint::pt, key_point.hpp:14) */
    float *angle;        /* degrees [0,360) (KeyPoint::angle) */
    int32_t *octave;     /* KeyPoint::octave */
    uint32_t *desc;      /* KeyPoint::descriptor, std::array<uint32_t,8> (key_point.hpp:19-20) */
    int32_t *track_id;   /* keyPointTrackIds: tracker id or -1 (orb_extractor.cpp:122,161) */
} ms_keypoints;

int ms_orb_create(ms_ctx *ctx, const ms_orb_config *cfg, ms_orb **out);
void ms_orb_destroy(ms_orb *orb);

/* Optional camera validity mask (stand-in for tracker::Camera::isValidPixel, orb_extractor.cpp:101,
 * :231): width*height bytes in host memory, 0 = invalid.  NULL clears it (all pixels valid).  The mask is sampled at the
 * rounded level-0 position of a keypoint; the reference evaluates the camera model at the sub-pixel position, so the two can
 * differ at the rim of the valid region.  host/mi355slam/orb_extractor.hpp's predicate overload of detectAndExtract applies the
 * model itself to the output coordinates and reproduces the reference exactly. */
int ms_orb_set_valid_mask(ms_orb *orb, const uint8_t *mask_host);

/* detectAndExtract for a batch of frames.
 *   images        : frame f starts at images + f*frame_stride, rows `row_stride` bytes apart.
 *   images_on_device != 0: `images` is device memory and is used IN PLACE as pyramid level 0
 *                   (needs a 16-byte aligned base and row_stride % 16 == 0, else it is copied on
 *                   the device); it must stay valid until the call's work has completed.
 *   track_xy      : optional [n_frames][max_tracks][2] level-0 coords (host), track_id [n_frames][max_tracks],
 *                   n_tracks [n_frames].  NULL = no tracker features.
 * Asynchronous on the context stream; results are read with ms_orb_download / ms_orb_device_view
 * after ms_ctx_sync (ms_orb_download synchronises itself). */
int ms_orb_extract(ms_orb *orb, const uint8_t *images, int images_on_device, int n_frames,
                   size_t frame_stride, size_t row_stride,
                   const float *track_xy, const int32_t *track_id, const int32_t *n_tracks);

/* Device-resident results of the last ms_orb_extract (pointers are device memory owned by orb). */
int ms_orb_device_view(ms_orb *orb, ms_keypoints *view);
/* Copy frame `frame`'s keypoints to caller-owned host arrays (each sized >= capacity); returns count in *n. */
int ms_orb_download(ms_orb *orb, int frame, float *x, float *y, float *angle, int32_t *octave,
                    uint32_t *desc, int32_t *track_id, int32_t *n);
int ms_orb_capacity(const ms_orb *orb);

/* N4, serialization half: KeyPoint::serialize (key_point.hpp:22-25) writes `ar(pt.x, pt.y, angle, octave, octave, bearing, descriptor)`
 * for every element of KeyframeShared::keyPoints (keyframe.hpp:80-93).  ms_keypoints_pack lays the first n keypoints of frame `frame`
 * of a device SoA view (ms_orb_device_view, or any ms_keypoints of device arrays) out in exactly that field order, 76 bytes per keypoint with no
 * padding -- x, y, angle as f32; octave as i32 TWICE (the reference's own quirk); bearing as 3 x f64 (`bearing`: device array [n*3] =
 * KeyPoint::bearing, which the caller computes, keyframe.cpp:55-68; NULL writes zeros); descriptor as 8 x u32 -- on the device, and
 * copies the records to `records_host` with one transfer (synchronises).  ms_keypoints_unpack is the inverse on the host (any output may
 * be NULL; MS_ERR_INVALID if the two octave copies of a record differ).  What is NOT known here: the archive framing around the records
 * (cereal's size tag of the vector and the Eigen::Vector3d serializer live in the parent project's util/serialization.hpp, not in the
 * reference tree); the record body above is what a binary archive writes for arithmetic fields and std::array<uint32_t, 8>. */
#define MS_KEYPOINT_RECORD_BYTES 76
int ms_keypoints_pack(ms_ctx *ctx, const ms_keypoints *view, int frame, int n, const double *bearing, uint8_t *records_host);
int ms_keypoints_unpack(const uint8_t *records, int n, float *x, float *y, float *angle, int32_t *octave, double *bearing, uint32_t *desc);

/* Per-kernel timing of ms_orb_extract with HIP events on the context stream (off by default).
 * Stage order: 0 resize (all levels), 1 blur, 2 fast, 3 select, 4 tracks, 5 describe.
 * ms_orb_stage_ms synchronises and returns the durations of the LAST ms_orb_extract call. */
#define MS_ORB_STAGES 6
int ms_orb_set_profiling(ms_orb *orb, int enable);
int ms_orb_stage_ms(ms_orb *orb, float *ms /* [MS_ORB_STAGES] */);
/* The same for an earlier profiled call: calls_back = 0 is the last one, up to 127 back (each profiled call records into the next set of a ring), so a
 * run of calls can be enqueued without a host wait in between and read afterwards. */
int ms_orb_stage_ms_back(ms_orb *orb, int calls_back, float *ms /* [MS_ORB_STAGES] */);
/* ImagePyramid::getLevel / getBlurredLevel (image_pyramid.hpp:24-25): copy one level of one frame
 * of the last batch to host, tightly packed w*h bytes (debug / parity testing).
 * LIFETIME of device inputs: frames handed to ms_orb_extract in device memory (16-byte aligned base and strides) are used IN PLACE as
 * level 0 -- they are not copied.  Level 0, and every BLURRED level (the blurred pyramid is produced on demand, by k_blur over the levels
 * of the last batch), therefore read the caller's buffer at the time of THIS call: it must still hold the frames of the last
 * ms_orb_extract, unchanged and not freed, until the next ms_orb_extract or until the caller stops asking for level 0 / blurred levels.
 * Host inputs and unaligned device inputs are copied into the extractor's own memory and carry no such requirement. */
int ms_orb_level_size(const ms_orb *orb, int level, int32_t *w, int32_t *h);
int ms_orb_download_level(ms_orb *orb, int frame, int level, int blurred, uint8_t *dst_host);
/* FeatureDetector::detect (feature_detector.cpp:20-28): per-level detector output of the last batch
 * BEFORE orientation: integer level coordinates + FAST score, `*n` points (<= quota of the level). */
int ms_orb_download_detections(ms_orb *orb, int frame, int level, int32_t *x, int32_t *y, int32_t *score, int32_t *n);

/* ---------------------------------------------------------------------------------------------
 * Descriptor matching -- the scoring core of keyframe_matcher.cpp (compute_descriptor_distance_32,
 * openvslam/match_base.h:18-39) as device primitives.
 * ------------------------------------------------------------------------------------------- */
/* Per-context choice of the kernel behind the UNMASKED searches: 0 = automatic (the i8 matrix-core kernel), 1 = the popcount
 * kernel (v_xor / v_bcnt, wave reductions) that the masked searches always use.  Both give identical results; the switch exists to
 * cross-check one against the other and to time them side by side (bench.py reports both). */
int ms_hamming_set_path(ms_ctx *ctx, int path);

/* Brute-force best / second-best of every query against every target, for `n_pairs` independent
 * (query set, target set) pairs laid out back to back: pair p uses q + p*nq*8 and t + p*nt*8.
 * Update rule of keyframe_matcher.cpp:106-112 (strict '<': lowest index wins ties).
 * Optional masks (device pointers or NULL): q_bucket/t_bucket [n_pairs*nq]/[n_pairs*nt] -- only equal
 * bucket ids are compared (DBoW2 node gate, keyframe_matcher.cpp:74); t_valid [n_pairs*nt] -- 0 skips the
 * target (map-point gates, keyframe_matcher.cpp:94-100).
 * Outputs (device): best_idx (-1 if none), best_dist, second_dist (256 if none). All device pointers. */
int ms_hamming_best2(ms_ctx *ctx, const uint32_t *q, int nq, const uint32_t *t, int nt, int n_pairs,
                     const int32_t *q_bucket, const int32_t *t_bucket, const uint8_t *t_valid,
                     int32_t *best_idx, uint16_t *best_dist, uint16_t *second_dist);

/* Same search over SETS of descriptors kept in two pools (e.g. the per-frame outputs of ms_orb_extract,
 * ms_keypoints.desc with stride = capacity and count = ms_keypoints.count): set s of a pool starts at
 * pool + s*stride*8 words and holds count[s] rows (count == NULL: stride rows).  Pair p compares set
 * pair_q[p] of the query pool with set pair_t[p] of the target pool (NULL: set p).  Outputs are
 * [n_pairs * q_stride]; rows beyond a set's count get (-1, 256, 256).  All pointers are device memory. */
int ms_hamming_best2_sets(ms_ctx *ctx, const uint32_t *q_pool, int q_stride, const int32_t *q_count,
                          const uint32_t *t_pool, int t_stride, const int32_t *t_count,
                          const int32_t *pair_q, const int32_t *pair_t, int n_pairs,
                          int32_t *best_idx, uint16_t *best_dist, uint16_t *second_dist);

/* Accept rule of matchForLoopClosures without the greedy state (keyframe_matcher.cpp:115-122):
 * match[i] = best_idx if best <= max_dist and NOT (ratio * second < (float)best), else -1.  Device pointers.
 * Arithmetic of the product: float32 (`lowe_ratio` is a float here, the product is one rounded float32 multiply).  The reference writes
 * `parameters.loopClosureFeatureMatchLoweRatio * second_best_hamm_dist < static_cast<float>(best_hamm_dist)` (:120); the parameter's type lives in
 * the parent project (not in the tree).  If it is a double there, the reference's product is exact in double and the two can differ for a ratio
 * that float32 cannot represent, exactly at the boundary ratio * second == best (0.75, 0.5, 0.8125 ... are exact either way; 0.8 or 0.7 are not).
 * The same holds for ms_match_loop_closure's lowe_ratio. */
int ms_ratio_test(ms_ctx *ctx, const int32_t *best_idx, const uint16_t *best_dist, const uint16_t *second_dist,
                  int n, float lowe_ratio, int max_dist, int32_t *match);

/* Scoring core of the projection-guided matchers searchByProjection / replaceDuplication / findMatchesTranformedMps
 * (keyframe_matcher.cpp:349-378, :479-494, :600-623): query i (a map point's descriptor) against ITS OWN candidate list
 * cand_idx[cand_start[i] .. cand_start[i+1]) (keypoint indices from the radius query, Keyframe::getFeaturesAround).
 * t_skip marks targets to ignore (already bound features, :358-360); t_octave gives KeyPoint::octave so the caller can apply
 * the same-level ratio rule (:382-386).  Outputs follow the reference's sequential scan exactly: best = first minimum,
 * second = next smallest; *_octave are -1 when absent.  All pointers are device memory. */
int ms_hamming_candidates(ms_ctx *ctx, const uint32_t *q_desc, int nq, const uint32_t *t_desc,
                          const int32_t *cand_start, const int32_t *cand_idx, const uint8_t *t_skip, const int32_t *t_octave,
                          int32_t *best_idx, uint16_t *best_dist, uint16_t *second_dist, int32_t *best_octave, int32_t *second_octave,
                          int32_t *second_idx /* may be NULL */);

/* Representative descriptor of many map points at once (MapPoint::updateDescriptor, map_point.cpp:75-116): for map point p
 * with observations obs_idx[obs_start[p] .. obs_start[p+1]) (indices into desc_pool, 8 words each), the observation whose
 * median Hamming distance to all of the point's observations is smallest (median = sorted[(n-1)/2], self distance 0
 * included; lowest index wins ties; only a median < 256 replaces index 0).  best_local[p] = position in the point's list,
 * best_pool[p] = the pool index (either may be NULL); both -1 for a point without observations (the reference returns
 * early, :86).  max_obs >= the longest list, at most MS_MEDOID_MAX_OBS (a longer list yields -2
 * for that point).  All pointers are device memory. */
#define MS_MEDOID_MAX_OBS 256
int ms_descriptor_medoid(ms_ctx *ctx, const uint32_t *desc_pool, const int32_t *obs_start, const int32_t *obs_idx, int n_points,
                         int max_obs, int32_t *best_local, int32_t *best_pool);

/* ---- N3: vocabulary-tree descent behind BowIndex::transform (bow_index.cpp:59-93) -------------------------------------
 * The reference hands every keypoint descriptor to DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(features,
 * bowVector, featureVector, levelsup = 4) (bow_index.cpp:86-92; DBoW2 is an external library, not in the reference tree).
 * ms_bow_vocab_create takes the loaded vocabulary the way DBoW2 stores it -- HOST arrays: parent[i] = parent node id
 * (node 0 = root, parent[0] ignored, every parent id smaller than its children's ids, as DBoW2's loaders and k-means builder
 * produce), node_desc[i] = the node's 256-bit descriptor, node_weight[i] = its weight, node_word[i] = its word id for a leaf
 * (-1 for inner nodes), depth_levels = the vocabulary's L -- and keeps a device copy laid out for the descent.
 * ms_bow_transform walks n descriptors (DEVICE memory, 8 words each, 16-byte aligned) down the tree: at every level the child
 * with the smallest Hamming distance, the first child (lowest node id) on ties; word[i] / weight[i] are the reached leaf's,
 * node[i] is the node passed at level depth_levels - levels_up (0 = root when that level is <= 0; the leaf's own id when the
 * leaf lies above that level, where DBoW2 leaves the value unset).  word = -1, weight = 0 for a vocabulary without words.
 * weight / node may be NULL.  The BowVector / FeatureVector maps are assembled from these arrays by the host mirror
 * (mi355slam::BowIndex::transform) in feature order, so the sums match the reference's. */
typedef struct ms_bow_vocab ms_bow_vocab;
int ms_bow_vocab_create(ms_ctx *ctx, int n_nodes, const int32_t *parent, const uint32_t *node_desc, const double *node_weight,
                        const int32_t *node_word, int depth_levels, ms_bow_vocab **out);
void ms_bow_vocab_destroy(ms_bow_vocab *vocab);
int ms_bow_transform(ms_ctx *ctx, const ms_bow_vocab *vocab, const uint32_t *desc, int n, int levels_up,
                     int32_t *word, double *weight, int32_t *node);

/* FeatureSearch (feature_search.{hpp,cpp}): the keyframe's keypoints sorted by y.  Host helper; std::stable_sort, so points
 * with equal y keep index order (the reference's std::sort leaves that order unspecified).  sorted_idx[p] = keypoint index. */
int ms_feature_search_sort(const float *x, const float *y, int n, float *sorted_x, float *sorted_y, int32_t *sorted_idx);

/* getFeaturesAround + the candidate scan of searchByProjection / replaceDuplication / findMatchesTranformedMps in one launch
 * (feature_search.cpp:33-48 + keyframe_matcher.cpp:349-378, :479-494, :600-623): query i = (projected position, search radius,
 * map-point descriptor, optional octave window [min, max] as in :611).  Candidates are the keypoints with
 * y in [qy - r, qy + r] and dx*dx + dy*dy < r*r (float32), in sorted order; t_skip / t_octave are indexed by KEYPOINT index, as are
 * the returned best / second indices.  Outputs as ms_hamming_candidates; n_candidates[i] = size of the reference's output vector
 * (before skip / octave filtering), may be NULL.  All pointers are device memory. */
int ms_projection_candidates(ms_ctx *ctx, const float *sorted_x, const float *sorted_y, const int32_t *sorted_idx, int n_kp,
                             const uint32_t *t_desc, const int32_t *t_octave, const uint8_t *t_skip,
                             const float *q_x, const float *q_y, const float *q_radius, const int32_t *q_min_octave, const int32_t *q_max_octave,
                             const uint32_t *q_desc, int nq,
                             int32_t *best_idx, uint16_t *best_dist, uint16_t *second_dist, int32_t *best_octave, int32_t *second_octave,
                             int32_t *second_idx, int32_t *n_candidates);

/* The same two scans returning, per query, the FOUR best candidates in the order of (distance, position in the scan) instead of (best, second):
 * top_idx / top_octave [nq*4] (keypoint index / its octave, -1 where the list is shorter), top_dist [nq*4] (256 there), n_scored [nq] = candidates
 * that were scored (inside the radius, not skipped, inside the octave window).  searchByProjection binds keypoints while it walks its map points
 * (keyframe_matcher.cpp:356-360,:388-389): with the list a host replay finds the best and second best among the keypoints still free without going
 * back to the device -- exact whenever two list entries are still free or n_scored <= 4 (the list is the whole candidate set); otherwise that one
 * query is scored again against the current mask (rare: three of its four best must have been taken by earlier map points of the same call). */
int ms_hamming_candidates_topk(ms_ctx *ctx, const uint32_t *q_desc, int nq, const uint32_t *t_desc, const int32_t *cand_start, const int32_t *cand_idx,
                               const uint8_t *t_skip, const int32_t *t_octave, int32_t *top_idx, uint16_t *top_dist, int32_t *top_octave, int32_t *n_scored);
int ms_projection_topk(ms_ctx *ctx, const float *sorted_x, const float *sorted_y, const int32_t *sorted_idx, int n_kp,
                       const uint32_t *t_desc, const int32_t *t_octave, const uint8_t *t_skip,
                       const float *q_x, const float *q_y, const float *q_radius, const int32_t *q_min_octave, const int32_t *q_max_octave,
                       const uint32_t *q_desc, int nq, int32_t *top_idx, uint16_t *top_dist, int32_t *top_octave, int32_t *n_scored, int32_t *n_candidates);

/* Rotation-consistency histogram (openvslam/match_angle_checker.h:60-134), host arithmetic: 30 bins of
 * cvRound(delta/30), everything outside the 3 fullest bins is invalid (ties between bins go to the lower bin).
 * Writes the ids of invalid entries (bin order, then insertion order) and returns their count. */
int ms_angle_check(const float *delta_angle, const int32_t *ids, int n, int32_t *invalid_ids);

/* Bag-of-words buckets of one keyframe in CSR form (DBoW2::FeatureVector is an ordered std::map
 * node id -> keypoint indices; keyframe_matcher.cpp:65-76).  Arrays are DEVICE pointers. */
typedef struct {
    int32_t n_nodes;
    const int32_t *node_id;      /* [n_nodes] strictly ascending */
    const int32_t *node_start;   /* [n_nodes+1] */
    const int32_t *kp_idx;       /* [node_start[n_nodes]] */
} ms_bow;

/* One keyframe's matching inputs (device pointers). */
typedef struct {
    int32_t n;                   /* keypoints */
    const uint32_t *desc;        /* [n*8] */
    const float *angle;          /* [n] degrees */
    const int32_t *octave;       /* [n] (M2 only) */
    const double *bearing;       /* [n*3] KeyPoint::bearing (M2 only) */
    const uint8_t *usable;       /* [n] M1: has a (triangulated) map point (keyframe_matcher.cpp:79-84,:94-96);
                                        M2: has NO map point (keyframe_matcher.cpp:205-221) */
    ms_bow bow;
} ms_match_frame;

/* matchForLoopClosures (keyframe_matcher.hpp:33-40, keyframe_matcher.cpp:50-158): exact greedy
 * semantics (targets consumed in BoW-node / keypoint order), rotation histogram included.
 * Batched: pair p matches kf1[p] against kf2[p]; matched[p] is a device array [kf1[p].n] (-1 = none);
 * n_matches [n_pairs] device.  `pairs1/pairs2` are HOST arrays of structs holding device pointers. */
int ms_match_loop_closure(ms_ctx *ctx, const ms_match_frame *pairs1, const ms_match_frame *pairs2, int n_pairs,
                          float lowe_ratio, int check_orientation, int32_t *const *matched, int32_t *n_matches);

/* Execution path of the two greedy matchers: 0 (default) = one wavefront per shared vocabulary node, all nodes and pairs side by side
 * (valid because a DBoW2 FeatureVector names each keypoint in exactly one node, so the greedy order only matters inside a node; a pair
 * whose node lists break that property is detected on the device and redone sequentially); 1 = one wavefront per pair walking the
 * nodes in order; 2 = as 0, but ONE workgroup walks the whole work list of large nodes (a test setting: it makes a workgroup reuse its
 * staging memory across nodes).  All give the reference's result bit for bit. */
int ms_match_set_path(ms_ctx *ctx, int path);

/* matchForTriangulationDBoW (keyframe_matcher.hpp:53, keyframe_matcher.cpp:160-293).
 * E12 [n_pairs*9] device, row-major essential matrices (create_E_21, essential_solver.cc:157-162);
 * scale_factors [levels] device; residual_deg_thr = epipolarCheckThresholdDegrees. */
int ms_match_triangulation(ms_ctx *ctx, const ms_match_frame *pairs1, const ms_match_frame *pairs2, int n_pairs,
                           const double *E12, const float *scale_factors, float residual_deg_thr,
                           int check_orientation, int32_t *const *matched, int32_t *n_matches);

/* ---------------------------------------------------------------------------------------------
 * Bundle adjustment -- replaces the g2o optimisation inside localBundleAdjust / poseBundleAdjust /
 * globalBundleAdjust (bundle_adjuster.hpp:30-51; bundle_adjuster.cpp:149-154, :322-323, :372-373,
 * :482-483, :577-578): EdgeSE3ProjectXYZ + EdgeSE3Expmap residuals, Huber kernel, Levenberg-Marquardt.
 * The host wrapper builds the problem exactly as bundle_adjuster.cpp:156-319 builds the g2o graph.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_pose, n_point, n_obs, n_pose_edge;
    const double *pose;          /* [n_pose*7]  qx,qy,qz,qw,tx,ty,tz  world->camera (g2o::SE3Quat of poseCW, :250) */
    const uint8_t *pose_fixed;   /* [n_pose]    vertex->setFixed (:252, :351) */
    const double *point;         /* [n_point*3] MapPoint::position (:266) */
    const uint8_t *point_fixed;  /* [n_point] or NULL (pose-only BA fixes every point, :465) */
    const int32_t *obs_pose;     /* [n_obs] index of the keyframe vertex of each EdgeSE3ProjectXYZ */
    const int32_t *obs_point;    /* [n_obs] index of its map-point vertex */
    const double *obs_uv;        /* [n_obs*2] measurement bearing.xy / bearing.z (:52) */
    const double *obs_info;      /* [n_obs]   information = focal^2 / levelSigmaSq[octave] times I2 (:51-53) */
    double huber_delta;          /* sqrt(5.991) (:56); <= 0 disables the robust kernel */
    const int32_t *edge_i;       /* [n_pose_edge] EdgeSE3Expmap vertex 0 (:76, :99, :355) */
    const int32_t *edge_j;       /* [n_pose_edge] vertex 1 */
    const double *edge_meas;     /* [n_pose_edge*7] measurement SE3 */
    const double *edge_info;     /* [n_pose_edge*36] 6x6 information, row-major, (rotation, translation) order */
    int32_t max_iters;           /* optimizer.optimize(iterations) */
} ms_ba_problem;                 /* all pointers are HOST memory, read during ms_ba_create only */

typedef struct {
    int32_t iterations;          /* LM iterations run */
    int32_t trials;              /* damped solves, including rejected ones */
    int32_t stopped_early;       /* 1 if g2o's Terminate condition ended the run */
    double final_lambda;
    double chi2_initial, chi2_final;   /* activeRobustChi2 before / after */
    double phase_cycles[8];            /* shader cycles of the problem's first workgroup: chi2 eval, linearise, Schur, Cholesky+back-subst,
                                          points+update, total, then inside the fused Schur pass: tile init, wait for the slowest wave
                                          (record-based path: Hinv / Y / S init, the same + rhs) */
} ms_ba_result;

typedef struct ms_ba ms_ba;

/* Upload `n` independent problems (different sizes allowed) and build their index structures.  The handle's device memory is ONE block;
 * ms_ba_destroy hands it back to the context (up to four blocks, 1 GiB in total: a block that would push the kept total past that --
 * a global-BA sized handle -- is freed instead), and the next ms_ba_create on that context takes the smallest kept block that is large
 * enough and at most 8 times the request instead of allocating: a window per keyframe (create, solve, download, destroy) allocates nothing
 * after warm-up, and a small window never sits on a large block.  If the device is out of memory the kept blocks are freed and the
 * allocation is tried once more.  Kept blocks are released by ms_ctx_destroy.
 * Two shapes are recognised here and solved by kernels of their own (same LM schedule, same arithmetic per edge; results agree with the general
 * kernel to rounding): every problem has ONE free pose and only fixed points, <= 8 SE3 edges at the free pose -- poseBundleAdjust, bundle_adjuster.cpp:396-491;
 * every problem has ONE free pose and at least one free point, <= 8 SE3 edges at the free pose -- stage 1 of localBundleAdjust,
 * bundle_adjuster.cpp:251-252,:322-333 (for this one ms_ba_set_team picks the team of the special kernel: 0 = automatic, up to 8 workgroups).
 * Environment switches for comparisons: MS_BA_NO_POSE_KERNEL=1 / MS_BA_NO_ONE_POSE_KERNEL=1 keep the general kernel. */
int ms_ba_create(ms_ctx *ctx, const ms_ba_problem *problems, int n, ms_ba **out);
void ms_ba_destroy(ms_ba *ba);
/* Run the full LM schedule of every problem from its initial estimates, one workgroup per problem,
 * entirely on the device; asynchronous on the context stream, repeatable. */
/* Workgroups (CUs) that share ONE problem in the next ms_ba_solve: 0 = automatic (by problem size, and only while
 * problems x workgroups fits the chip: a batch of >= #CUs problems always runs one workgroup per problem), 1 = the whole
 * Levenberg-Marquardt loop in one workgroup, up to 64.  Results agree to rounding: a team adds its sums with LDS and global fp64
 * atomics, in no fixed order (run-to-run differences stay below 1e-9 on poses and points; the reference's own order is
 * unspecified).  The index structures are built at create time for the regime the automatic rule will pick; forcing the other one
 * afterwards (a team on a handle created for a chip-filling batch, or one workgroup on a single window's handle) is correct but
 * slower.  A single local-BA window is ~10x faster with a team (1.7 ms against 21 ms for the 50-keyframe window of the bench). */
int ms_ba_set_team(ms_ba *ba, int workgroups_per_problem);
/* Of a team, the workgroups that share the distributed Cholesky factorisation of a system with more than 176 free poses
 * (0 = automatic: one per 16 row tiles a panel touches, so a banded trajectory is factored by one workgroup without team barriers
 * and a densely coupled map by many).  Has no effect on smaller systems. */
int ms_ba_set_factor_team(ms_ba *ba, int workgroups);
int ms_ba_solve(ms_ba *ba);
/* Chains two solves on the device: the state `src`'s last solve left (poses, points) becomes the INITIAL state of `dst`'s problems --
 * the step between stage 1 and stage 2 of localBundleAdjust (bundle_adjuster.cpp:335-373: same vertices, every keyframe unfixed, one more
 * edge against a fixed copy of the just-optimised pose) without a download / upload in between.  Both handles hold the same number of
 * problems on the same context; problem i of dst has the points of problem i of src and at least its poses; each pose dst has beyond
 * them takes the value of src's pose extra_pose_src[i] (HOST array, one entry per problem; may be NULL when the pose counts agree).
 * Asynchronous on the context stream, ordered after src's solve. */
int ms_ba_copy_state(ms_ba *dst, const ms_ba *src, const int32_t *extra_pose_src);
/* Team launches (more than one workgroup per problem) synchronise their workgroups with spin barriers, which need every workgroup
 * of the launch resident: problems x team <= CUs is enforced per launch, and the team launches of one process are admitted per device
 * so that their workgroups together fit the CUs (a launch waits, on the device, for as many older ones of other contexts as it takes),
 * so two contexts -- the front end's poseBundleAdjust beside the back
 * end's localBundleAdjust, mapper.cpp:379-390 vs :268-269 -- may solve at the same time.  If a barrier still gives up (no progress
 * for ~1 s: CUs held by another PROCESS), ms_ba_download repeats the solve with one workgroup per problem before it returns;
 * ms_ba_team_fallbacks counts those repeats.  Before the FIRST problem of a team launch is handed out every problem's marker is looked at, so
 * no result of a launch that is going to be repeated is ever returned.  A barrier gives up after 2 s WITHOUT PROGRESS (arrival counter and
 * the team's heartbeat both still), not after a fixed number of polls: a long single-workgroup phase is not mistaken for a lost team.
 *
 * Process-wide state: team launches are admitted per DEVICE across all contexts of the process (the sum of their workgroups must fit the CUs,
 * or two half-resident teams would wait for each other).  That list -- one event per running team launch, guarded by a mutex -- is the one
 * piece of global mutable state in the library; its events live until the process ends.  An event query that fails with anything but "not
 * ready" retires the entry, is counted (ms_ba_admission_errors) and its text is kept for ms_last_error of the context that saw it.  ms_ba_debug_fail_team_barriers(ba, 1) makes every team barrier of the following
 * launches give up at once (test hook for that path). */
int ms_ba_team_fallbacks(const ms_ba *ba);
/* Event-query failures the team admission list has seen in this process (0 in a healthy run; see above). */
int ms_ba_admission_errors(void);
int ms_ba_debug_fail_team_barriers(ms_ba *ba, int on);
/* Allocations the library has made so far on the host or the device (handle objects, growth of its host scratch, device blocks, pinned staging, events), process-wide.
 * The per-keyframe path -- ms_ba_create / solve / download / destroy of windows of a steady size -- leaves it unchanged after warm-up (the reference mallocs
 * per vertex and edge, bundle_adjuster.cpp:55,73,247,265,279).  tests/host_shim_smoke.cpp holds 20 consecutive windows against it. */
long long ms_debug_host_allocs(void);
/* Test hook: the first `first_trials` damped trials of the following solves count as rejected whatever their gain (state restored, lambda *= nu, nu *= 2): ten of them
 * in one iteration drive g2o's Terminate path (OptimizationAlgorithmLevenberg, _maxTrialsAfterFailure = 10) deterministically.  General solver only. */
int ms_ba_debug_force_reject(ms_ba *ba, int first_trials);
/* Results of problem i (synchronises): poses [n_pose*7], points [n_point*3], per-observation chi2
 * (what the outlier rule chi2 > 5.991 of :376-388 reads).  They are evaluated at the state that is RETURNED, i.e. the last accepted one.  g2o's edge->chi2() is the
 * error of the last computeActiveErrors(): when optimize() ends on rejected trials (Terminate after ten failures in a row, or a trial without gain) the vertices
 * are restored but the edges keep the rejected trial's errors, and bundle_adjuster.cpp:378 reads those.  The two differ by the last rejected step -- after ten
 * rejections lambda has grown by 2^55, so by about 2^-55 of a Gauss-Newton step; oracle/ba.c restates g2o's value under flag bit 1, and
 * tests/test_gpu_ba.py::test_ten_rejected_trials_terminate_like_the_oracle holds the two against each other.  Any output pointer may be NULL.  The status is read first: on
 * MS_ERR_NUMERIC (non-finite state) none of the caller's arrays is written (res, when given, is filled). */
int ms_ba_download(ms_ba *ba, int i, double *pose, double *point, double *chi2_per_obs, ms_ba_result *res);
/* create + solve + download + destroy for one problem. */
int ms_ba_solve_host(ms_ctx *ctx, const ms_ba_problem *problem, double *pose_out, double *point_out,
                     double *chi2_per_obs, ms_ba_result *res);

#ifdef __cplusplus
}
#endif
#endif /* MI355SLAM_H */
