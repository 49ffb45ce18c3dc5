// keyframe_matcher.hpp -- host mirror of the free functions of keyframe_matcher.hpp:33-91.
//
// The reference's Keyframe / MapDB carry the graph; the device only needs flat per-keyframe arrays, so a
// KeyframeFeatures view is built once per keyframe (INTEGRATION.md shows the 15 lines that fill it from
// kf.shared->keyPoints, kf.mapPoints and kf.shared->bowFeatureVec) and uploaded by DeviceKeyframe.
#pragma once
#include <algorithm>
#include <cstring>
#include <map>
#include <utility>
#include "common.hpp"

namespace mi355slam {

constexpr unsigned HAMMING_DIST_THR_LOW = MS_HAMMING_THR_LOW, HAMMING_DIST_THR_HIGH = MS_HAMMING_THR_HIGH, MAX_HAMMING_DIST = MS_HAMMING_MAX;

struct KeyframeFeatures {
    const KeyPointVector *keyPoints = nullptr;                              // kf.shared->keyPoints
    std::vector<std::uint8_t> usable;                                        // per keypoint: the matcher-specific map-point gate
    std::map<unsigned, std::vector<unsigned>> bowFeatureVec;                  // DBoW2::FeatureVector (ordered node -> keypoint indices)
};

// Device-resident copy of one keyframe's matcher inputs.
class DeviceKeyframe {
public:
    DeviceKeyframe(Context &ctx, const KeyframeFeatures &kf) : ctx_(ctx) {
        const auto &kps = *kf.keyPoints;
        const std::size_t n = kps.size();
        std::vector<std::uint32_t> desc(8 * n); std::vector<float> ang(n); std::vector<std::int32_t> oct(n); std::vector<double> bear(3 * n);
        for (std::size_t i = 0; i < n; ++i) {
            for (int k = 0; k < 8; ++k) desc[8 * i + k] = kps[i].descriptor[k];
            ang[i] = kps[i].angle; oct[i] = kps[i].octave;
            for (int k = 0; k < 3; ++k) bear[3 * i + k] = kps[i].bearing[k];
        }
        std::vector<std::int32_t> node_id, node_start{0}, kp_idx;
        for (const auto &kv : kf.bowFeatureVec) {                             // std::map iterates node ids ascending
            node_id.push_back((std::int32_t)kv.first);
            for (unsigned i : kv.second) kp_idx.push_back((std::int32_t)i);
            node_start.push_back((std::int32_t)kp_idx.size());
        }
        f_.n = (std::int32_t)n;
        f_.desc = up(desc); f_.angle = up(ang); f_.octave = up(oct); f_.bearing = up(bear); f_.usable = up(kf.usable);
        f_.bow.n_nodes = (std::int32_t)node_id.size(); f_.bow.node_id = up(node_id); f_.bow.node_start = up(node_start); f_.bow.kp_idx = up(kp_idx);
        // FeatureSearch::create (feature_search.cpp:22-30): the keypoints sorted by y, for the radius queries of M3-M5
        std::vector<float> x(n), y(n), sx(n), sy(n); std::vector<std::int32_t> si(n);
        for (std::size_t i = 0; i < n; ++i) { x[i] = kps[i].pt.x; y[i] = kps[i].pt.y; }
        ctx_.check(ms_feature_search_sort(x.data(), y.data(), (int)n, sx.data(), sy.data(), si.data()), "ms_feature_search_sort");
        sx_ = up(sx); sy_ = up(sy); si_ = up(si);
        // host copies of what a single query's scan needs (searchByProjectionCore settles the rare query whose top-4 list ran out right here,
        // instead of a round trip to the device): 48 bytes per keypoint
        hsx_ = std::move(sx); hsy_ = std::move(sy); hsi_ = std::move(si); hdesc_ = std::move(desc); hoct_ = std::move(oct);
    }
    const std::vector<float> &hostSortedX() const { return hsx_; }
    const std::vector<float> &hostSortedY() const { return hsy_; }
    const std::vector<std::int32_t> &hostSortedIndex() const { return hsi_; }
    const std::uint32_t *hostDescriptor(std::size_t i) const { return hdesc_.data() + 8 * i; }
    int hostOctave(std::size_t i) const { return hoct_[i]; }
    const float *sortedX() const { return sx_; }
    const float *sortedY() const { return sy_; }
    const std::int32_t *sortedIndex() const { return si_; }
    ~DeviceKeyframe() { for (void *p : owned_) ms_dev_free(ctx_.get(), p); }
    DeviceKeyframe(const DeviceKeyframe &) = delete;
    const ms_match_frame &frame() const { return f_; }
private:
    template <typename T> const T *up(const std::vector<T> &v) {
        void *d = nullptr;
        ctx_.check(ms_dev_alloc(ctx_.get(), v.size() * sizeof(T) + 16, &d), "ms_dev_alloc");
        owned_.push_back(d);
        if (!v.empty()) ctx_.check(ms_dev_upload(ctx_.get(), d, v.data(), v.size() * sizeof(T)), "ms_dev_upload");
        return static_cast<const T *>(d);
    }
    Context &ctx_;
    ms_match_frame f_{};
    const float *sx_ = nullptr, *sy_ = nullptr;
    const std::int32_t *si_ = nullptr;
    std::vector<float> hsx_, hsy_;
    std::vector<std::int32_t> hsi_, hoct_;
    std::vector<std::uint32_t> hdesc_;
    std::vector<void *> owned_;
};

namespace detail {
inline std::size_t pad16(std::size_t n) { return (n + 15) / 16 * 16; }

inline unsigned run_greedy(Context &ctx, bool triangulation, const DeviceKeyframe &kf1, const DeviceKeyframe &kf2, std::vector<int> &out,
                           float ratio, const double *E12, const std::vector<float> *scaleFactors, float thrDeg) {
    // one workspace block: [match count (16 B)][matched: n1 ints][E: 9 doubles][scale factors]; one upload (M2 only), one call, one download
    const std::size_t n1 = (std::size_t)kf1.frame().n;
    const std::size_t oM = 16, oE = oM + pad16(4 * n1), oS = oE + 80, total = oS + pad16(4 * (scaleFactors ? scaleFactors->size() : 0));
    unsigned char *ws = ctx.workspace(total);
    std::int32_t *mptr = reinterpret_cast<std::int32_t *>(ws + oM);
    ms_match_frame f1 = kf1.frame(), f2 = kf2.frame();
    int rc;
    if (triangulation) {
        std::vector<unsigned char> &st = ctx.staging();
        st.assign(total - oE, 0);
        std::memcpy(st.data(), E12, 72);
        std::memcpy(st.data() + 80, scaleFactors->data(), 4 * scaleFactors->size());
        ctx.check(ms_dev_upload(ctx.get(), ws + oE, st.data(), st.size()), "ms_dev_upload");
        rc = ms_match_triangulation(ctx.get(), &f1, &f2, 1, reinterpret_cast<const double *>(ws + oE), reinterpret_cast<const float *>(ws + oS), thrDeg, 1, &mptr,
                                    reinterpret_cast<std::int32_t *>(ws));
    } else {
        rc = ms_match_loop_closure(ctx.get(), &f1, &f2, 1, ratio, 1, &mptr, reinterpret_cast<std::int32_t *>(ws));
    }
    ctx.check(rc, "greedy matcher");
    std::vector<unsigned char> &st = ctx.staging();
    st.resize(oM + 4 * n1);
    ctx.check(ms_dev_download(ctx.get(), st.data(), ws, st.size()), "ms_dev_download");
    std::int32_t num = 0;
    std::memcpy(&num, st.data(), 4);
    out.assign(n1, -1);
    if (n1) std::memcpy(out.data(), st.data() + oM, 4 * n1);
    return (unsigned)num;
}
}  // namespace detail

// matchForLoopClosures (keyframe_matcher.hpp:33-40, keyframe_matcher.cpp:50-158).
// usable1 = keypoint has a map point (and it is TRIANGULATED when requireTringulationForLoopClosures, :79-84);
// usable2 = keypoint has a TRIANGULATED map point (:94-96).
inline unsigned matchForLoopClosures(Context &ctx, const DeviceKeyframe &kf1, const DeviceKeyframe &kf2,
                                     std::vector<int> &matchedMapPoints, const Parameters &parameters) {
    return detail::run_greedy(ctx, false, kf1, kf2, matchedMapPoints, parameters.loopClosureFeatureMatchLoweRatio, nullptr, nullptr, 0.f);
}

// matchForTriangulationDBoW (keyframe_matcher.hpp:53, keyframe_matcher.cpp:160-293).  usable = keypoint has NO map point.
// E12 = create_E_21(kf2.R, kf2.t, kf1.R, kf1.t) (keyframe_matcher.cpp:171-175), row-major.
inline std::vector<std::pair<int, int>> matchForTriangulationDBoW(Context &ctx, const DeviceKeyframe &kf1, const DeviceKeyframe &kf2,
                                                                  const double E12[9], const StaticSettings &settings) {
    std::vector<int> m;
    detail::run_greedy(ctx, true, kf1, kf2, m, 0.f, E12, &settings.scaleFactors, settings.parameters.epipolarCheckThresholdDegrees);
    std::vector<std::pair<int, int>> matches;                               // ascending idx_1 (:279-292)
    for (std::size_t i = 0; i < m.size(); ++i) if (m[i] >= 0) matches.emplace_back((int)i, m[i]);
    return matches;
}

// ---- projection-guided matchers (M3-M5) ---------------------------------------------------------------------------
// The caller keeps what touches the map graph: reprojection, distance / viewing-angle gates, predictScaleLevel, the
// radius query Keyframe::getFeaturesAround and all map mutation.  What it hands over per surviving map point is its
// descriptor and the candidate keypoint indices; the Hamming scoring runs on the GPU for all map points at once.
struct ProjectionQuery {
    KeyPoint::Descriptor descriptor;            // mp.descriptor
    std::vector<std::int32_t> candidates;       // indices from kf.getFeaturesAround(...) (keyframe_matcher.cpp:340-344, :473, :596)
};

// The same scan with the radius query done on the device: the reprojected position and the search radius instead of a candidate list
// (kf.getFeaturesAround(reprojection, radius, indices), keyframe_matcher.cpp:340-344 / :470-473 / :596).
struct RadiusQuery {
    KeyPoint::Descriptor descriptor;
    float x = 0, y = 0, radius = 0;
    std::int32_t minOctave = -0x7fffffff, maxOctave = 0x7fffffff;      // findMatchesTranformedMps keeps [pred - 1, pred] (:611)
};

// Per query the four best candidates in (distance, scan position) order -- ms_hamming_candidates_topk / ms_projection_topk -- and how many candidates
// were scored in all: with nScored <= 4 the list IS the candidate set.
struct CandidateLists {
    std::vector<std::int32_t> idx, octave, nScored;      // [4 n], [4 n], [n]
    std::vector<std::uint16_t> dist;                     // [4 n]
};

namespace detail {
// uploads the queries (one block), runs the scan, downloads the lists (one block)
inline CandidateLists score_candidates(Context &ctx, const DeviceKeyframe &kf, const std::vector<ProjectionQuery> &qs,
                                       const std::vector<std::uint8_t> *skip, std::size_t first = 0, std::size_t count = ~std::size_t(0)) {
    count = std::min(count, qs.size() - first);
    CandidateLists out;
    out.idx.assign(4 * count, -1); out.octave.assign(4 * count, -1); out.nScored.assign(count, 0); out.dist.assign(4 * count, MS_HAMMING_MAX);
    if (count == 0) return out;
    std::size_t nc = 0;
    for (std::size_t i = 0; i < count; ++i) nc += qs[first + i].candidates.size();
    const std::size_t nk = skip ? skip->size() : 0;
    const std::size_t oD = 0, oS = oD + 32 * count, oI = oS + pad16(4 * (count + 1)), oK = oI + pad16(4 * nc), inBytes = oK + pad16(nk);
    const std::size_t oTi = inBytes, oTo = oTi + 16 * count, oN = oTo + 16 * count, oTd = oN + pad16(4 * count), total = oTd + pad16(8 * count);
    std::vector<unsigned char> &st = ctx.staging();
    st.assign(inBytes, 0);
    std::int32_t *start = reinterpret_cast<std::int32_t *>(st.data() + oS), *idx = reinterpret_cast<std::int32_t *>(st.data() + oI);
    std::size_t at = 0;
    for (std::size_t i = 0; i < count; ++i) {
        const ProjectionQuery &q = qs[first + i];
        std::memcpy(st.data() + oD + 32 * i, q.descriptor.data(), 32);
        start[i] = (std::int32_t)at;
        if (!q.candidates.empty()) std::memcpy(idx + at, q.candidates.data(), 4 * q.candidates.size());
        at += q.candidates.size();
    }
    start[count] = (std::int32_t)at;
    if (nk) std::memcpy(st.data() + oK, skip->data(), nk);
    unsigned char *ws = ctx.workspace(total);
    ctx.check(ms_dev_upload(ctx.get(), ws, st.data(), inBytes), "ms_dev_upload");
    const ms_match_frame &f = kf.frame();
    ctx.check(ms_hamming_candidates_topk(ctx.get(), reinterpret_cast<const std::uint32_t *>(ws + oD), (int)count, f.desc, reinterpret_cast<const std::int32_t *>(ws + oS),
                                         reinterpret_cast<const std::int32_t *>(ws + oI), nk ? ws + oK : nullptr, f.octave, reinterpret_cast<std::int32_t *>(ws + oTi),
                                         reinterpret_cast<std::uint16_t *>(ws + oTd), reinterpret_cast<std::int32_t *>(ws + oTo), reinterpret_cast<std::int32_t *>(ws + oN)),
              "ms_hamming_candidates_topk");
    st.resize(total - oTi);
    ctx.check(ms_dev_download(ctx.get(), st.data(), ws + oTi, total - oTi), "ms_dev_download");
    std::memcpy(out.idx.data(), st.data(), 16 * count); std::memcpy(out.octave.data(), st.data() + (oTo - oTi), 16 * count);
    std::memcpy(out.nScored.data(), st.data() + (oN - oTi), 4 * count); std::memcpy(out.dist.data(), st.data() + (oTd - oTi), 8 * count);
    return out;
}

inline CandidateLists score_candidates(Context &ctx, const DeviceKeyframe &kf, const std::vector<RadiusQuery> &qs,
                                       const std::vector<std::uint8_t> *skip, std::size_t first = 0, std::size_t count = ~std::size_t(0)) {
    count = std::min(count, qs.size() - first);
    CandidateLists out;
    out.idx.assign(4 * count, -1); out.octave.assign(4 * count, -1); out.nScored.assign(count, 0); out.dist.assign(4 * count, MS_HAMMING_MAX);
    if (count == 0) return out;
    const std::size_t nk = skip ? skip->size() : 0, sec = pad16(4 * count);
    const std::size_t oD = 0, oX = 32 * count, oY = oX + sec, oR = oY + sec, oLo = oR + sec, oHi = oLo + sec, oK = oHi + sec, inBytes = oK + pad16(nk);
    const std::size_t oTi = inBytes, oTo = oTi + 16 * count, oN = oTo + 16 * count, oTd = oN + sec, total = oTd + pad16(8 * count);
    std::vector<unsigned char> &st = ctx.staging();
    st.assign(inBytes, 0);
    float *x = reinterpret_cast<float *>(st.data() + oX), *y = reinterpret_cast<float *>(st.data() + oY), *r = reinterpret_cast<float *>(st.data() + oR);
    std::int32_t *lo = reinterpret_cast<std::int32_t *>(st.data() + oLo), *hi = reinterpret_cast<std::int32_t *>(st.data() + oHi);
    for (std::size_t i = 0; i < count; ++i) {
        const RadiusQuery &q = qs[first + i];
        std::memcpy(st.data() + oD + 32 * i, q.descriptor.data(), 32);
        x[i] = q.x; y[i] = q.y; r[i] = q.radius; lo[i] = q.minOctave; hi[i] = q.maxOctave;
    }
    if (nk) std::memcpy(st.data() + oK, skip->data(), nk);
    unsigned char *ws = ctx.workspace(total);
    ctx.check(ms_dev_upload(ctx.get(), ws, st.data(), inBytes), "ms_dev_upload");
    const ms_match_frame &f = kf.frame();
    ctx.check(ms_projection_topk(ctx.get(), kf.sortedX(), kf.sortedY(), kf.sortedIndex(), f.n, f.desc, f.octave, nk ? ws + oK : nullptr,
                                 reinterpret_cast<const float *>(ws + oX), reinterpret_cast<const float *>(ws + oY), reinterpret_cast<const float *>(ws + oR),
                                 reinterpret_cast<const std::int32_t *>(ws + oLo), reinterpret_cast<const std::int32_t *>(ws + oHi),
                                 reinterpret_cast<const std::uint32_t *>(ws + oD), (int)count, reinterpret_cast<std::int32_t *>(ws + oTi),
                                 reinterpret_cast<std::uint16_t *>(ws + oTd), reinterpret_cast<std::int32_t *>(ws + oTo), reinterpret_cast<std::int32_t *>(ws + oN), nullptr),
              "ms_projection_topk");
    st.resize(total - oTi);
    ctx.check(ms_dev_download(ctx.get(), st.data(), ws + oTi, total - oTi), "ms_dev_download");
    std::memcpy(out.idx.data(), st.data(), 16 * count); std::memcpy(out.octave.data(), st.data() + (oTo - oTi), 16 * count);
    std::memcpy(out.nScored.data(), st.data() + (oN - oTi), 4 * count); std::memcpy(out.dist.data(), st.data() + (oTd - oTi), 8 * count);
    return out;
}
}  // namespace detail

namespace detail {
struct Best2 { int best = -1, bestDist = 256, bestDist2 = 256, bestLevel = -1, bestLevel2 = -1; };
inline int hamming256(const std::uint32_t *a, const std::uint32_t *b) { int d = 0; for (int k = 0; k < 8; ++k) d += __builtin_popcount(a[k] ^ b[k]); return d; }
inline void best2_update(Best2 &b, int idx, int dist, int level) {                        // keyframe_matcher.cpp:369-377
    if (dist < b.bestDist) { b.bestDist2 = b.bestDist; b.bestDist = dist; b.bestLevel2 = b.bestLevel; b.bestLevel = level; b.best = idx; }
    else if (dist < b.bestDist2) { b.bestLevel2 = level; b.bestDist2 = dist; }
}
// one query's whole scan on the host, against the CURRENT mask (the reference's loop :356-378 as it stands); a handful of queries per keyframe get here
inline Best2 rescan_on_host(const DeviceKeyframe &kf, const ProjectionQuery &q, const std::vector<std::uint8_t> &bound) {
    Best2 b;
    for (std::int32_t j : q.candidates)
        if (!bound[(std::size_t)j]) best2_update(b, j, hamming256(q.descriptor.data(), kf.hostDescriptor((std::size_t)j)), kf.hostOctave((std::size_t)j));
    return b;
}
inline Best2 rescan_on_host(const DeviceKeyframe &kf, const RadiusQuery &q, const std::vector<std::uint8_t> &bound) {
    // FeatureSearch::getFeaturesAround (feature_search.cpp:33-48) in float32, every operation rounded on its own (a product or sum formed in double and
    // rounded to float IS the float32 result, so the compiler's contraction setting cannot change it) -- the same arithmetic as k_projection_candidates
    const std::vector<float> &sx = kf.hostSortedX(), &sy = kf.hostSortedY();
    const float ylo = q.y - q.radius, yhi = q.y + q.radius, r2 = (float)((double)q.radius * (double)q.radius);
    Best2 b;
    for (std::size_t pos = (std::size_t)(std::lower_bound(sy.begin(), sy.end(), ylo) - sy.begin()); pos < sy.size() && sy[pos] <= yhi; ++pos) {
        const float dx = q.x - sx[pos], dy = q.y - sy[pos];
        const float dx2 = (float)((double)dx * (double)dx), dy2 = (float)((double)dy * (double)dy);
        if (!((float)((double)dx2 + (double)dy2) < r2)) continue;
        const std::size_t j = (std::size_t)kf.hostSortedIndex()[pos];
        if (bound[j] || kf.hostOctave(j) < q.minOctave || kf.hostOctave(j) > q.maxOctave) continue;
        best2_update(b, (int)j, hamming256(q.descriptor.data(), kf.hostDescriptor(j)), kf.hostOctave(j));
    }
    return b;
}
}  // namespace detail

// Scoring + accept rule of searchByProjection (keyframe_matcher.cpp:349-389).  `bound[k]` != 0 marks keypoints that already
// carry an observed map point (:358-360); it is updated as matches are accepted, in query order, exactly like the
// reference's loop.  ONE launch scores every query against the initial mask and returns its four best candidates; the replay below
// walks the queries in order and takes, per query, the first two list entries that no earlier query of this call has bound -- the
// best and second best of the reference's scan over the keypoints still free (the scan order is the list's tie-break).  A query whose
// list is too short for that (more than four candidates in all, and fewer than two of its four best still free: three of its four best
// went to earlier map points of the same call) is scanned again on the host against the current mask (detail::rescan_on_host).
// Returns, per query, the matched keypoint index or -1; the caller performs addObservation (:388-389).
// `Query` is ProjectionQuery (candidate lists from the host's getFeaturesAround) or RadiusQuery (radius query on the device).
template <class Query>
inline std::vector<int> searchByProjectionCore(Context &ctx, const DeviceKeyframe &kf, const std::vector<Query> &queries,
                                               std::vector<std::uint8_t> &bound, unsigned *rescored = nullptr) {
    std::vector<int> match(queries.size(), -1);
    const CandidateLists s = detail::score_candidates(ctx, kf, queries, &bound);
    std::vector<std::uint8_t> taken(bound.size(), 0);            // bound during this call
    unsigned again = 0;
    for (std::size_t i = 0; i < queries.size(); ++i) {
        int best = -1, bestDist = 256, bestDist2 = 256, bestLevel = -1, bestLevel2 = -1, found = 0;
        auto walk = [&](const CandidateLists &l, std::size_t q) {
            best = -1; bestDist = 256; bestDist2 = 256; bestLevel = -1; bestLevel2 = -1; found = 0;
            for (int e = 0; e < 4 && found < 2; ++e) {
                const int j = l.idx[4 * q + e];
                if (j < 0) break;
                if (taken[(std::size_t)j]) continue;
                if (found == 0) { best = j; bestDist = l.dist[4 * q + e]; bestLevel = l.octave[4 * q + e]; }
                else { bestDist2 = l.dist[4 * q + e]; bestLevel2 = l.octave[4 * q + e]; }
                ++found;
            }
        };
        walk(s, i);
        if (found < 2 && s.nScored[i] > 4) {                       // the list ran out before the candidate set did
            const detail::Best2 r = detail::rescan_on_host(kf, queries[i], bound);               // `bound` = initial mask + everything taken so far
            best = r.best; bestDist = r.bestDist; bestDist2 = r.bestDist2; bestLevel = r.bestLevel; bestLevel2 = r.bestLevel2;
            ++again;
        }
        if (best == -1) continue;                                                          // :380
        if (bestDist <= (int)HAMMING_DIST_THR_HIGH) {                                        // :382-383
            if (bestLevel == bestLevel2 && bestDist > 0.8 * bestDist2) continue;             // :385-386
            match[i] = best; bound[(std::size_t)best] = 1; taken[(std::size_t)best] = 1;
        }
    }
    if (rescored) *rescored = again;
    return match;
}

// Scoring of replaceDuplication (keyframe_matcher.cpp:479-499: best only, accept <= 50) and findMatchesTranformedMps
// (:600-627: accept <= 100; the caller pre-filters candidates by octave, :611).  No greedy state in the scoring itself.
template <class Query>
inline std::vector<int> bestCandidateCore(Context &ctx, const DeviceKeyframe &kf, const std::vector<Query> &queries, unsigned maxDist) {
    const CandidateLists s = detail::score_candidates(ctx, kf, queries, nullptr);
    std::vector<int> match(queries.size(), -1);
    for (std::size_t i = 0; i < queries.size(); ++i) if (s.idx[4 * i] >= 0 && s.dist[4 * i] <= maxDist) match[i] = s.idx[4 * i];
    return match;
}

// MapPoint::updateDescriptor (map_point.cpp:75-116) for many map points in one launch.  observations[p] lists, for map
// point p, the descriptors of its observing keypoints (the caller gathers kf.shared->keyPoints.at(obs.second.v).descriptor
// for keyframes with hasFeatureDescriptors(), :78-84, in the order of the observations map).  Returns the position of the
// chosen descriptor in that list, or -1 for an empty list (the reference then leaves `descriptor` untouched, :86).
inline std::vector<int> updateDescriptors(Context &ctx, const std::vector<std::vector<KeyPoint::Descriptor>> &observations) {
    const std::size_t n = observations.size();
    std::vector<int> best(n, -1);
    if (n == 0) return best;
    std::vector<std::uint32_t> pool;
    std::vector<std::int32_t> start(n + 1, 0), idx;
    std::size_t longest = 0;
    for (std::size_t p = 0; p < n; ++p) {
        for (const KeyPoint::Descriptor &d : observations[p]) { idx.push_back((std::int32_t)(pool.size() / 8)); pool.insert(pool.end(), d.begin(), d.end()); }
        start[p + 1] = (std::int32_t)idx.size();
        longest = std::max(longest, observations[p].size());
    }
    std::vector<void *> bufs;
    auto up = [&](const void *src, std::size_t bytes) { void *d = nullptr; ctx.check(ms_dev_alloc(ctx.get(), bytes + 32, &d), "ms_dev_alloc"); bufs.push_back(d);
                                                         if (bytes && src) { ctx.check(ms_dev_upload(ctx.get(), d, src, bytes), "ms_dev_upload"); }
                                                         return d; };
    void *dp = up(pool.data(), pool.size() * 4), *ds = up(start.data(), start.size() * 4), *di = up(idx.data(), idx.size() * 4);
    void *out = up(nullptr, 4 * n);
    const int rc = ms_descriptor_medoid(ctx.get(), static_cast<const std::uint32_t *>(dp), static_cast<const std::int32_t *>(ds),
                                        static_cast<const std::int32_t *>(di), (int)n, (int)longest, static_cast<std::int32_t *>(out), nullptr);
    if (rc == MS_OK) ctx.check(ms_dev_download(ctx.get(), best.data(), out, 4 * n), "download");
    for (void *p : bufs) ms_dev_free(ctx.get(), p);
    ctx.check(rc, "ms_descriptor_medoid");
    return best;
}

// ---- M5: findMatchesTranformedMps + matchMapPointsSim3 (keyframe_matcher.cpp:552-686) ------------------------------------
// One map point of keyframe A as findMatchesTranformedMps sees it after the part that needs the map graph and the camera model
// (:572-596): `usable` = has a map point (mpId != -1), TRIANGULATED, reprojectToImage succeeded, the viewing distance |R X + t| lies in
// [minViewingDistance, maxViewingDistance]; (x, y) = the reprojection into keyframe B; predScaleLevel = mp.predictScaleLevel(...).
struct Sim3Projection {
    bool usable = false;
    KeyPoint::Descriptor descriptor{};          // mp.descriptor
    float x = 0, y = 0;
    int predScaleLevel = 0;
};

// findMatchesTranformedMps (:552-631): matches[iA] = index of the best keypoint of kfB (radius margin * scaleFactors[pred], octave in
// [pred - 1, pred], strictly smaller distance wins = first of the radius query's order, accepted at <= HAMMING_DIST_THR_HIGH) or -1.
inline std::vector<int> findMatchesTranformedMps(Context &ctx, const std::vector<Sim3Projection> &mpsA, const std::vector<bool> &alreadyMatchedInA,
                                                 const DeviceKeyframe &kfB, float margin, const StaticSettings &settings) {
    std::vector<int> matchesAtoB(mpsA.size(), -1);
    std::vector<RadiusQuery> qs;
    std::vector<std::size_t> owner;
    for (std::size_t indA = 0; indA < mpsA.size(); ++indA) {
        if (alreadyMatchedInA.at(indA) || !mpsA[indA].usable) continue;                        // :566-590
        const Sim3Projection &m = mpsA[indA];
        RadiusQuery q;
        q.descriptor = m.descriptor; q.x = m.x; q.y = m.y;
        q.radius = margin * settings.scaleFactors.at((std::size_t)m.predScaleLevel);           // :596
        q.minOctave = m.predScaleLevel - 1; q.maxOctave = m.predScaleLevel;                    // :611
        qs.push_back(q); owner.push_back(indA);
    }
    const std::vector<int> best = bestCandidateCore(ctx, kfB, qs, HAMMING_DIST_THR_HIGH);     // :600-627
    for (std::size_t i = 0; i < qs.size(); ++i) matchesAtoB[owner[i]] = best[i];
    return matchesAtoB;
}

// matchMapPointsSim3 (:633-686) on keypoint indices: `matches` holds (index in kf1, index in kf2) pairs -- the reference keeps
// (MpId, MpId) and looks the indices up through mapPoints.at(id).observations.at(kf.id) (:645-648); the caller does that lookup and the
// reverse one (kf.mapPoints.at(index)) for the pairs appended here.  mps1in2 = kf1's map points projected into kf2 with
// transform12^-1 * kf1.poseCW (:650), mps2in1 = kf2's into kf1 with transform12 * kf2.poseCW (:661).  Returns the number added.
inline unsigned matchMapPointsSim3(Context &ctx, const DeviceKeyframe &kf1, const DeviceKeyframe &kf2, const std::vector<Sim3Projection> &mps1in2,
                                   const std::vector<Sim3Projection> &mps2in1, std::vector<std::pair<int, int>> &matches, const StaticSettings &settings) {
    constexpr float margin = 7.5;                                                              // :641
    // keypoints that already carry a match are left out of both searches (:645-648)
    std::vector<bool> taken1(mps1in2.size(), false), taken2(mps2in1.size(), false);
    for (const std::pair<int, int> &m : matches) { taken1.at((std::size_t)m.first) = true; taken2.at((std::size_t)m.second) = true; }
    const std::vector<int> fwd = findMatchesTranformedMps(ctx, mps1in2, taken1, kf2, margin, settings);      // kf1 keypoint -> kf2 keypoint or -1
    const std::vector<int> bwd = findMatchesTranformedMps(ctx, mps2in1, taken2, kf1, margin, settings);      // kf2 keypoint -> kf1 keypoint or -1
    // a pair is kept when each side names the other (:672-685), in ascending kf1 index
    const std::size_t before = matches.size();
    for (std::size_t i1 = 0; i1 < fwd.size(); ++i1)
        if (fwd[i1] >= 0 && bwd.at((std::size_t)fwd[i1]) == (int)i1) matches.emplace_back((int)i1, fwd[i1]);
    return (unsigned)(matches.size() - before);
}

// create_E_21 (openvslam/essential_solver.cc:157-162), row-major 3x3
inline void create_E_21(const double R1w[9], const double t1w[3], const double R2w[9], const double t2w[3], double E[9]) {
    double R21[9], t21[3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += R2w[3 * i + k] * R1w[3 * j + k]; R21[3 * i + j] = s; }
    for (int i = 0; i < 3; ++i) { double s = 0; for (int k = 0; k < 3; ++k) s += -R21[3 * i + k] * t1w[k]; t21[i] = s + t2w[i]; }
    const double S[9] = {0, -t21[2], t21[1], t21[2], 0, -t21[0], -t21[1], t21[0], 0};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += S[3 * i + k] * R21[3 * k + j]; E[3 * i + j] = s; }
}

}  // namespace mi355slam
