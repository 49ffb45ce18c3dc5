// Compile + link check of the host-side C++ mirrors against libmi355slam.so; when a GPU is present it also runs one
// extraction, one triangulation match and one two-stage local BA end to end (used by tests/test_host_shims.py).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <cmath>
#include "mi355slam/orb_extractor.hpp"
#include "mi355slam/keyframe_matcher.hpp"
#include "mi355slam/bundle_adjuster.hpp"
#include "mi355slam/bow_index.hpp"

using namespace mi355slam;

int main(int argc, char **argv) {
    Parameters params;
    params.maxTracks = 16;
    StaticSettings settings(params);
    if (settings.maxNumberOfKeypointsPerLevel().front() != 434) { std::printf("quota mismatch\n"); return 2; }
    {   // VocabularyTree::loadFromTextFile: DBoW2's text format ("k L scoring weighting", then "parent isLeaf 32 bytes weight" per node)
        const char *tmp = std::getenv("TMPDIR");
        const std::string path = std::string(tmp ? tmp : "/tmp") + "/mi355slam_vocab_test.txt";
        {
            std::ofstream f(path);
            f << "3 2 0 0\n";
            int id = 0;
            for (int p = 0; p < 3; ++p) {                     // three inner nodes under the root, then three leaves under each (ids 1..3, 4..12)
                f << 0 << " " << 0; for (int b = 0; b < 32; ++b) f << " " << ((b * 7 + p) & 255); f << " " << 0.0 << "\n"; ++id;
            }
            for (int p = 1; p <= 3; ++p) for (int c = 0; c < 3; ++c) {
                f << p << " " << 1; for (int b = 0; b < 32; ++b) f << " " << ((b * 13 + 5 * p + c) & 255); f << " " << (0.5 * p + c) << "\n"; ++id;
            }
        }
        const VocabularyTree T = VocabularyTree::loadFromTextFile(path);
        std::remove(path.c_str());
        bool ok = T.size() == 13 && T.branchingFactor == 3 && T.depthLevels == 2 && T.parent[5] == 1 && T.parent[12] == 3 && T.wordId[3] == -1 && T.wordId[4] == 0 &&
                  T.wordId[12] == 8 && T.weight[7] == 1.0 && T.weight[12] == 3.5;
        // byte b of node 4 (parent 1, child 0) is (13 b + 5) & 255; word k of the descriptor holds bytes 4k .. 4k+3, lowest first
        const std::uint32_t w2 = T.descriptor[8 * 4 + 2];
        ok = ok && w2 == (std::uint32_t)(((13 * 8 + 5) & 255) | (((13 * 9 + 5) & 255) << 8) | (((13 * 10 + 5) & 255) << 16) | (((13 * 11 + 5) & 255) << 24));
        if (!ok) { std::printf("vocabulary text file mismatch\n"); return 11; }
    }
    if (argc > 1 && std::string(argv[1]) == "--no-gpu") { std::printf("link ok\n"); return 0; }
    Context ctx(0);
    const int W = 320, H = 240;
    std::vector<std::uint8_t> img((std::size_t)W * H);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) img[(std::size_t)y * W + x] = (std::uint8_t)(((x / 16 + y / 16) & 1) ? 200 : 40);
    auto extractor = OrbExtractor::build(ctx, settings);
    KeyPointVector kps; std::vector<int> ids;
    extractor->detectAndExtract(ImageView{img.data(), W, H, (std::size_t)W, false}, nullptr, {TrackPoint{100.f, 100.f, 7}}, kps, ids);
    std::printf("keypoints %zu (first track id %d)\n", kps.size(), ids.empty() ? -99 : ids[0]);
    if (kps.size() < 50 || ids[0] != 7) return 3;
    {   // camera validity with the model itself (sub-pixel positions, orb_extractor.cpp:101,:231) next to the rasterised mask
        auto valid = [&](float x, float y) { const float dx = x - 160.25f, dy = y - 119.75f; return dx * dx + dy * dy < 95.5f * 95.5f; };
        KeyPointVector exact, rast; std::vector<int> exactIds, rastIds;
        extractor->detectAndExtract(ImageView{img.data(), W, H, (std::size_t)W, false}, valid, {TrackPoint{100.f, 100.f, 7}, TrackPoint{20.f, 30.f, 8}}, exact, exactIds);
        std::size_t want = 0; bool same = true;
        KeyPointVector all; std::vector<int> allIds;
        extractor->detectAndExtract(ImageView{img.data(), W, H, (std::size_t)W, false}, nullptr, {TrackPoint{100.f, 100.f, 7}, TrackPoint{20.f, 30.f, 8}}, all, allIds);
        for (std::size_t i = 0; i < all.size(); ++i) {
            if (!valid(all[i].pt.x, all[i].pt.y)) continue;
            same = same && want < exact.size() && exact[want].pt.x == all[i].pt.x && exact[want].pt.y == all[i].pt.y && exact[want].descriptor == all[i].descriptor && exactIds[want] == allIds[i];
            ++want;
        }
        std::vector<std::uint8_t> mask((std::size_t)W * H);
        for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) mask[(std::size_t)y * W + x] = valid((float)x, (float)y);
        extractor->detectAndExtract(ImageView{img.data(), W, H, (std::size_t)W, false}, mask.data(), {TrackPoint{100.f, 100.f, 7}, TrackPoint{20.f, 30.f, 8}}, rast, rastIds);
        {   // a mask edited IN PLACE (same pointer) must be seen: contents are hashed per call; with a version number the caller says when
            KeyPointVector k2, k3, k4; std::vector<int> i2, i3, i4;
            std::fill(mask.begin(), mask.end(), (std::uint8_t)1);                       // now everything is valid again, same pointer
            extractor->detectAndExtract(ImageView{img.data(), W, H, (std::size_t)W, false}, mask.data(), {TrackPoint{100.f, 100.f, 7}, TrackPoint{20.f, 30.f, 8}}, k2, i2);
            if (k2.size() != all.size()) { std::printf("in-place mask edit was ignored: %zu vs %zu\n", k2.size(), all.size()); return 15; }
            for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) mask[(std::size_t)y * W + x] = valid((float)x, (float)y);
            extractor->detectAndExtract(ImageView{img.data(), W, H, (std::size_t)W, false}, mask.data(), 1, {TrackPoint{100.f, 100.f, 7}, TrackPoint{20.f, 30.f, 8}}, k3, i3);
            std::fill(mask.begin(), mask.end(), (std::uint8_t)1);
            extractor->detectAndExtract(ImageView{img.data(), W, H, (std::size_t)W, false}, mask.data(), 1, {TrackPoint{100.f, 100.f, 7}, TrackPoint{20.f, 30.f, 8}}, k4, i4);   // same version: by contract NOT re-read
            if (k3.size() != rast.size() || k4.size() != rast.size()) { std::printf("mask version handling: %zu %zu vs %zu\n", k3.size(), k4.size(), rast.size()); return 16; }
        }
        extractor->detectAndExtract(ImageView{img.data(), W, H, (std::size_t)W, false}, nullptr, {TrackPoint{100.f, 100.f, 7}}, kps, ids);   // back to the unmasked state
        std::printf("validity: %zu of %zu keypoints inside (rasterised mask keeps %zu)\n", exact.size(), all.size(), rast.size());
        if (!same || want != exact.size() || exact.size() >= all.size() || exact.empty() || exactIds[0] != 7 || std::find(exactIds.begin(), exactIds.end(), 8) != exactIds.end()) return 12;
    }
    // match the frame against itself through the triangulation matcher (identity geometry -> epipolar residual is 0/0 guarded by E != 0)
    for (auto &kp : kps) { kp.bearing = {(kp.pt.x - W / 2) / 300.0, (kp.pt.y - H / 2) / 300.0, 1.0}; }
    KeyframeFeatures f; f.keyPoints = &kps; f.usable.assign(kps.size(), 1);
    for (unsigned i = 0; i < kps.size(); ++i) f.bowFeatureVec[kps[i].descriptor[0] % 10].push_back(i);
    DeviceKeyframe d1(ctx, f), d2(ctx, f);
    const double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t1[3] = {0, 0, 0}, t2[3] = {0.1, 0, 0};
    double E[9]; create_E_21(R, t2, R, t1, E);
    auto matches = matchForTriangulationDBoW(ctx, d1, d2, E, settings);
    std::printf("triangulation matches %zu\n", matches.size());
    // searchByProjection scoring: GPU batch + greedy replay vs the reference's sequential loop restated on the CPU
    {
        std::vector<ProjectionQuery> qs;
        unsigned rng = 12345u;
        auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
        for (int i = 0; i < 400; ++i) {
            ProjectionQuery q;
            q.descriptor = kps[rnd() % kps.size()].descriptor;
            q.descriptor[rnd() % 8] ^= (1u << (rnd() % 32)) | (1u << (rnd() % 32));
            const int nc = rnd() % 40;                                  // heavy overlap between queries -> greedy conflicts
            for (int c = 0; c < nc; ++c) q.candidates.push_back((int)(rnd() % std::min<std::size_t>(kps.size(), 120)));
            std::sort(q.candidates.begin(), q.candidates.end()); q.candidates.erase(std::unique(q.candidates.begin(), q.candidates.end()), q.candidates.end());
            qs.push_back(q);
        }
        std::vector<std::uint8_t> bound(kps.size(), 0), bound_ref;
        for (std::size_t k = 0; k < bound.size(); k += 7) bound[k] = 1;
        bound_ref = bound;
        auto popc = [](const KeyPoint::Descriptor &a, const KeyPoint::Descriptor &b) { int d = 0; for (int k = 0; k < 8; ++k) d += __builtin_popcount(a[k] ^ b[k]); return d; };
        std::vector<int> want(qs.size(), -1);
        for (std::size_t i = 0; i < qs.size(); ++i) {                   // keyframe_matcher.cpp:349-389
            int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
            for (int idx : qs[i].candidates) {
                if (bound_ref[idx]) continue;
                const int dist = popc(qs[i].descriptor, kps[idx].descriptor), level = kps[idx].octave;
                if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = level; bestIdx = idx; }
                else if (dist < bestDist2) { bestLevel2 = level; bestDist2 = dist; }
            }
            if (bestIdx == -1) continue;
            if (bestDist <= 100) { if (bestLevel == bestLevel2 && bestDist > 0.8 * bestDist2) continue; want[i] = bestIdx; bound_ref[bestIdx] = 1; }
        }
        std::vector<int> got = searchByProjectionCore(ctx, d1, qs, bound);
        int nm = 0; for (int m : want) nm += m >= 0;
        std::printf("searchByProjection: %d matches of %zu queries\n", nm, qs.size());
        if (got != want || bound != bound_ref || nm < 20) { std::printf("searchByProjectionCore mismatch\n"); return 5; }
        std::vector<int> dup = bestCandidateCore(ctx, d1, qs, HAMMING_DIST_THR_LOW);
        for (std::size_t i = 0; i < qs.size(); ++i) {
            int bd = 256, bi = -1;
            for (int idx : qs[i].candidates) { const int d = popc(qs[i].descriptor, kps[idx].descriptor); if (d < bd) { bd = d; bi = idx; } }
            if ((bi >= 0 && bd <= 50 ? bi : -1) != dup[i]) { std::printf("bestCandidateCore mismatch at %zu\n", i); return 6; }
        }
    }
    // the same matcher with the radius query on the device: reprojections + radii in, getFeaturesAround restated on the CPU as the check
    {
        unsigned rng = 777u;
        auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
        auto popc = [](const KeyPoint::Descriptor &a, const KeyPoint::Descriptor &b) { int d = 0; for (int k = 0; k < 8; ++k) d += __builtin_popcount(a[k] ^ b[k]); return d; };
        std::vector<std::size_t> order(kps.size());
        for (std::size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](std::size_t a, std::size_t b) { return kps[a].pt.y < kps[b].pt.y; });   // feature_search.cpp:26-29
        std::vector<RadiusQuery> rq;
        for (int i = 0; i < 300; ++i) {
            const KeyPoint &k = kps[rnd() % kps.size()];
            RadiusQuery q; q.descriptor = k.descriptor; q.descriptor[rnd() % 8] ^= 1u << (rnd() % 32);
            q.x = k.pt.x + (float)(rnd() % 9) - 4.f; q.y = k.pt.y + (float)(rnd() % 9) - 4.f; q.radius = 3.f + (float)(rnd() % 40);
            rq.push_back(q);
        }
        std::vector<std::uint8_t> bound(kps.size(), 0), bound_ref;
        for (std::size_t k = 0; k < bound.size(); k += 5) bound[k] = 1;
        bound_ref = bound;
        std::vector<int> want(rq.size(), -1);
        for (std::size_t i = 0; i < rq.size(); ++i) {
            int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
            for (std::size_t o : order) {                                  // feature_search.cpp:33-48 (lower_bound + walk == filter in sorted order)
                const float dx = rq[i].x - kps[o].pt.x, dy = rq[i].y - kps[o].pt.y;
                if (kps[o].pt.y < rq[i].y - rq[i].radius || !(kps[o].pt.y <= rq[i].y + rq[i].radius) || !(dx * dx + dy * dy < rq[i].radius * rq[i].radius)) continue;
                if (bound_ref[o]) continue;
                const int dist = popc(rq[i].descriptor, kps[o].descriptor), level = kps[o].octave;
                if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = level; bestIdx = (int)o; }
                else if (dist < bestDist2) { bestLevel2 = level; bestDist2 = dist; }
            }
            if (bestIdx == -1) continue;
            if (bestDist <= 100) { if (bestLevel == bestLevel2 && bestDist > 0.8 * bestDist2) continue; want[i] = bestIdx; bound_ref[bestIdx] = 1; }
        }
        std::vector<int> got = searchByProjectionCore(ctx, d1, rq, bound);
        int nm = 0; for (int m : want) nm += m >= 0;
        std::printf("searchByProjection (device radius query): %d matches of %zu queries\n", nm, rq.size());
        if (got != want || bound != bound_ref || nm < 20) { std::printf("radius searchByProjectionCore mismatch\n"); return 8; }
    }
    // searchByProjection at the size a keyframe has (mapper_helpers.cpp:231-269): 2000 keypoints on 1280 x 720, 2000 map points that reproject near
    // "their" keypoint with a 15 px radius, a third of them sharing their best keypoint with another map point (greedy conflicts); timed, and
    // checked against the sequential restatement
    {
        unsigned rng = 4711u;
        auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
        auto popc = [](const KeyPoint::Descriptor &a, const KeyPoint::Descriptor &b) { int d = 0; for (int k = 0; k < 8; ++k) d += __builtin_popcount(a[k] ^ b[k]); return d; };
        KeyPointVector big(2000);
        for (auto &k : big) {
            k.pt = {(float)(rnd() % 12800) / 10.f, (float)(rnd() % 7200) / 10.f}; k.angle = 0; k.octave = (int)(rnd() % 8);
            for (int w = 0; w < 8; ++w) k.descriptor[w] = (rnd() << 8) ^ rnd();
        }
        KeyframeFeatures fb; fb.keyPoints = &big; fb.usable.assign(big.size(), 1);
        DeviceKeyframe dbig(ctx, fb);
        std::vector<std::size_t> order(big.size());
        for (std::size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](std::size_t a, std::size_t b) { return big[a].pt.y < big[b].pt.y; });
        std::vector<RadiusQuery> rq;
        for (int i = 0; i < 2000; ++i) {
            const KeyPoint &k = big[i % 3 == 0 ? (std::size_t)(rnd() % 600) : (std::size_t)i];      // a third of the map points aim at one of 600 keypoints
            RadiusQuery q; q.descriptor = k.descriptor;
            for (int f = 0; f < 12; ++f) q.descriptor[rnd() % 8] ^= 1u << (rnd() % 32);
            q.x = k.pt.x + (float)(rnd() % 7) - 3.f; q.y = k.pt.y + (float)(rnd() % 7) - 3.f; q.radius = 15.f;
            rq.push_back(q);
        }
        std::vector<std::uint8_t> bound(big.size(), 0), bound_ref;
        for (std::size_t k = 0; k < bound.size(); k += 9) bound[k] = 1;
        bound_ref = bound;
        const std::vector<std::uint8_t> bound0 = bound;
        std::vector<int> want(rq.size(), -1);
        int conflicts = 0;
        {
            std::vector<int> firstChoice(rq.size(), -1);
            for (std::size_t i = 0; i < rq.size(); ++i) {
                int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1, free0 = -1, free0d = 256;
                for (std::size_t o : order) {
                    const float dx = rq[i].x - big[o].pt.x, dy = rq[i].y - big[o].pt.y;
                    if (big[o].pt.y < rq[i].y - rq[i].radius || !(big[o].pt.y <= rq[i].y + rq[i].radius) || !(dx * dx + dy * dy < rq[i].radius * rq[i].radius)) continue;
                    const int dist = popc(rq[i].descriptor, big[o].descriptor), level = big[o].octave;
                    if (!bound0[o] && dist < free0d) { free0d = dist; free0 = (int)o; }
                    if (bound_ref[o]) continue;
                    if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = level; bestIdx = (int)o; }
                    else if (dist < bestDist2) { bestLevel2 = level; bestDist2 = dist; }
                }
                if (free0 >= 0 && free0 != bestIdx) ++conflicts;          // the keypoint it would have taken on an untouched keyframe is gone
                if (bestIdx == -1) continue;
                if (bestDist <= 100) { if (bestLevel == bestLevel2 && bestDist > 0.8 * bestDist2) continue; want[i] = bestIdx; bound_ref[bestIdx] = 1; }
            }
        }
        unsigned rescored = 0;
        std::vector<std::uint8_t> b1 = bound0;
        std::vector<int> got = searchByProjectionCore(ctx, dbig, rq, b1, &rescored);       // warm-up (workspace allocation)
        const auto t0 = std::chrono::steady_clock::now();
        const int reps = 10;
        for (int r = 0; r < reps; ++r) { b1 = bound0; got = searchByProjectionCore(ctx, dbig, rq, b1, &rescored); }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
        int nm = 0; for (int m : want) nm += m >= 0;
        std::printf("searchByProjection 2000 map points x 2000 keypoints, radius 15 px: %d matches, %d queries lost their first choice to an earlier one, %u scored again, %.3f ms per call\n",
                    nm, conflicts, rescored, ms);
        if (got != want || b1 != bound_ref || nm < 1000 || conflicts < 300) { std::printf("searchByProjectionCore (keyframe size) mismatch\n"); return 17; }
    }
    // M5: matchMapPointsSim3 on two keyframes whose map points see each other (kf2 = kf1 shifted by (3, -2) px with a few bits flipped),
    // against a sequential restatement of keyframe_matcher.cpp:552-686 (radius query in y-sorted order, octave window, <= 100, seeds, mutual check)
    {
        unsigned rng = 4242u;
        auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
        auto popc = [](const KeyPoint::Descriptor &a, const KeyPoint::Descriptor &b) { int d = 0; for (int k = 0; k < 8; ++k) d += __builtin_popcount(a[k] ^ b[k]); return d; };
        KeyPointVector kps2 = kps;
        for (auto &k : kps2) { k.pt.x += 3.f; k.pt.y -= 2.f; for (int f = 0; f < 6; ++f) k.descriptor[rnd() % 8] ^= 1u << (rnd() % 32); if (rnd() % 7 == 0) k.octave += 1; }
        std::rotate(kps2.begin(), kps2.begin() + 17, kps2.end());                 // indices differ between the keyframes
        KeyframeFeatures f2; f2.keyPoints = &kps2; f2.usable.assign(kps2.size(), 1);
        DeviceKeyframe dA(ctx, f), dB(ctx, f2);
        auto project = [&](const KeyPointVector &from, float dx, float dy) {
            std::vector<Sim3Projection> out(from.size());
            for (std::size_t i = 0; i < from.size(); ++i) {
                Sim3Projection &m = out[i];
                m.usable = rnd() % 9 != 0;                                        // some points fail the map / camera gates
                m.descriptor = from[i].descriptor; m.x = from[i].pt.x + dx + 0.25f * (float)(rnd() % 5); m.y = from[i].pt.y + dy - 0.25f * (float)(rnd() % 5);
                m.predScaleLevel = std::min(7, from[i].octave + (int)(rnd() % 2));
            }
            return out;
        };
        const std::vector<Sim3Projection> p12 = project(kps, 3.f, -2.f), p21 = project(kps2, -3.f, 2.f);
        std::vector<std::pair<int, int>> seed;
        for (int i = 0; i < 12; ++i) { const int a = (int)((i * 37 + 5) % kps.size()); seed.emplace_back(a, (int)((a + kps.size() - 17) % kps.size())); }
        auto reference = [&](const std::vector<Sim3Projection> &mps, const std::vector<bool> &already, const KeyPointVector &target) {
            std::vector<std::size_t> order(target.size());
            for (std::size_t i = 0; i < order.size(); ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](std::size_t a, std::size_t b) { return target[a].pt.y < target[b].pt.y; });
            std::vector<int> out(mps.size(), -1);
            for (std::size_t a = 0; a < mps.size(); ++a) {
                if (already[a] || !mps[a].usable) continue;
                const float r = 7.5f * settings.scaleFactors[(std::size_t)mps[a].predScaleLevel];
                unsigned bestD = 256; int bestI = -1;
                for (std::size_t o : order) {
                    const float dx = mps[a].x - target[o].pt.x, dy = mps[a].y - target[o].pt.y;
                    if (target[o].pt.y < mps[a].y - r || !(target[o].pt.y <= mps[a].y + r) || !(dx * dx + dy * dy < r * r)) continue;
                    if (target[o].octave < mps[a].predScaleLevel - 1 || target[o].octave > mps[a].predScaleLevel) continue;
                    const unsigned d = (unsigned)popc(mps[a].descriptor, target[o].descriptor);
                    if (d < bestD) { bestD = d; bestI = (int)o; }
                }
                if (bestD <= 100) out[a] = bestI;
            }
            return out;
        };
        std::vector<bool> a1(kps.size(), false), a2(kps2.size(), false);
        for (auto &m : seed) { a1[(std::size_t)m.first] = true; a2[(std::size_t)m.second] = true; }
        const std::vector<int> w12 = reference(p12, a1, kps2), w21 = reference(p21, a2, kps);
        std::vector<std::pair<int, int>> want = seed, got = seed;
        for (std::size_t i = 0; i < w12.size(); ++i) if (w12[i] >= 0 && w21[(std::size_t)w12[i]] == (int)i) want.emplace_back((int)i, w12[i]);
        const unsigned added = matchMapPointsSim3(ctx, dA, dB, p12, p21, got, settings);
        int one_way = 0; for (int m : w12) one_way += m >= 0;
        std::printf("matchMapPointsSim3: %u mutual matches added to %zu seeds (%d one-way candidates)\n", added, seed.size(), one_way);
        if (got != want || added + seed.size() != got.size() || added < 30 || (int)added >= one_way) { std::printf("matchMapPointsSim3 mismatch\n"); return 12; }
        if (findMatchesTranformedMps(ctx, p12, a1, dB, 7.5f, settings) != w12) { std::printf("findMatchesTranformedMps mismatch\n"); return 13; }
    }
    {   auto popc = [](const KeyPoint::Descriptor &a, const KeyPoint::Descriptor &b) { unsigned d = 0; for (int k = 0; k < 8; ++k) d += __builtin_popcount(a[k] ^ b[k]); return d; };
        // updateDescriptor for a batch of map points vs the plain median-of-row rule (map_point.cpp:75-116)
        std::vector<std::vector<KeyPoint::Descriptor>> obs(40);
        for (std::size_t p = 0; p < obs.size(); ++p)
            for (std::size_t k = 0; k < p % 9; ++k) obs[p].push_back(kps[(7 * p + 13 * k) % kps.size()].descriptor);
        std::vector<int> got = updateDescriptors(ctx, obs);
        for (std::size_t p = 0; p < obs.size(); ++p) {
            int want = -1; unsigned bestMed = 256; const std::size_t n = obs[p].size();
            if (n) want = 0;
            for (std::size_t i = 0; i < n; ++i) {
                std::vector<unsigned> row; for (std::size_t j = 0; j < n; ++j) row.push_back(popc(obs[p][i], obs[p][j]));
                std::sort(row.begin(), row.end());
                const unsigned med = row[(unsigned)(0.5 * (n - 1))];
                if (med < bestMed) { bestMed = med; want = (int)i; }
            }
            if (got[p] != want) { std::printf("updateDescriptors mismatch at %zu: %d vs %d\n", p, got[p], want); return 7; }
        }
    }
    // BowIndex::transform through the device descent vs DBoW2's transform restated on std::maps (bow_index.cpp:59-93)
    {
        unsigned rng = 4242u;
        auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
        auto popc = [](const std::uint32_t *a, const std::uint32_t *b) { int d = 0; for (int k = 0; k < 8; ++k) d += __builtin_popcount(a[k] ^ b[k]); return d; };
        VocabularyTree T; T.branchingFactor = 6; T.depthLevels = 3;
        T.parent = {0}; T.wordId = {-1}; T.weight = {0.0}; T.descriptor.assign(8, 0u);
        std::vector<int> frontier{0};
        int words = 0;
        for (int lv = 1; lv <= 3; ++lv) {
            std::vector<int> next;
            for (int p : frontier) for (int c = 0; c < 6; ++c) {
                const KeyPoint &k = kps[rnd() % kps.size()];
                T.parent.push_back(p); T.wordId.push_back(lv == 3 ? words++ : -1); T.weight.push_back(lv == 3 ? (rnd() % 20 == 0 ? 0.0 : 0.25 + (rnd() % 1000) / 128.0) : 0.0);
                for (int w = 0; w < 8; ++w) T.descriptor.push_back(k.descriptor[w]);
                next.push_back((int)T.parent.size() - 1);
            }
            frontier = next;
        }
        BowIndex index(ctx, T);
        BowVector bv; FeatureVector fv;
        index.transform(kps, bv, fv);
        std::vector<std::vector<int>> children(T.size());
        for (std::size_t i = 1; i < T.size(); ++i) children[T.parent[i]].push_back((int)i);
        BowVector wantV; FeatureVector wantF;
        const int nidLevel = T.depthLevels - 4;
        for (std::size_t i = 0; i < kps.size(); ++i) {
            int fin = 0, level = 0, nid = nidLevel <= 0 ? 0 : -1;
            do {
                ++level;
                const auto &nodes = children[fin];
                fin = nodes[0];
                int best = popc(kps[i].descriptor.data(), &T.descriptor[8 * fin]);
                for (std::size_t c = 1; c < nodes.size(); ++c) { const int d = popc(kps[i].descriptor.data(), &T.descriptor[8 * nodes[c]]); if (d < best) { best = d; fin = nodes[c]; } }
                if (level == nidLevel) nid = fin;
            } while (!children[fin].empty());
            if (T.weight[fin] > 0) { wantV[(unsigned)T.wordId[fin]] += T.weight[fin]; wantF[(unsigned)nid].push_back((unsigned)i); }
        }
        double norm = 0; for (auto &kv : wantV) norm += std::fabs(kv.second);
        for (auto &kv : wantV) kv.second /= norm;
        std::printf("BowIndex::transform: %zu words, %zu nodes for %zu keypoints\n", bv.size(), fv.size(), kps.size());
        if (bv != wantV || fv != wantF || bv.size() < 20) { std::printf("BowIndex::transform mismatch\n"); return 9; }
    }
    // a tiny two-stage local BA: 3 cameras on a line looking at 30 points
    BaWindow w; w.currentKeyframe = 2;
    for (int i = 0; i < 3; ++i) w.poses.push_back({0, 0, 0, 1, -0.2 * i, 0, 0});
    for (int l = 0; l < 30; ++l) w.points.push_back({-1.0 + 0.07 * l, 0.3 * std::sin(l), 5.0 + 0.05 * l});
    for (int l = 0; l < 30; ++l) for (int i = 0; i < 3; ++i) {
        const double X = w.points[l][0] - 0.2 * i, Y = w.points[l][1], Z = w.points[l][2];
        w.obsPose.push_back(i); w.obsPoint.push_back(l); w.obsUv.push_back({X / Z + 1e-3 * std::sin(l + i), Y / Z}); w.obsInfo.push_back(250000.0);
    }
    for (int i = 1; i < 3; ++i) { w.edgeI.push_back(i); w.edgeJ.push_back(i - 1); w.edgeMeas.push_back({0, 0, 0, 1, 0.2, 0, 0});
        std::array<double, 36> info{}; for (int k = 0; k < 6; ++k) info[7 * k] = 1e4; w.edgeInfo.push_back(info); }
    WorkspaceBA workspace(true);
    std::vector<std::string> table;
    workspace.baStats = BaStats(true, [&](const char *l) { table.emplace_back(l); });
    BaOutcome o = localBundleAdjust(ctx, w, 50, params, true, &workspace);
    {   // BaStats / WorkspaceBA (ba_stats.hpp:9-84): LOCAL after the two-stage run, NEIGHBOR when stage 2 is skipped, NONE for an empty frame
        BaWindow w1 = w;
        localBundleAdjust(ctx, w1, 50, params, false, &workspace);
        workspace.baStats.update(BaStats::Ba::POSE);
        bool ok = workspace.baStats.frameCount(BaStats::Ba::LOCAL) == 1 && workspace.baStats.frameCount(BaStats::Ba::NEIGHBOR) == 1 && workspace.baStats.frameCount(BaStats::Ba::POSE) == 1;
        workspace.baStats.finishFrame();
        ok = ok && table.size() == 8 && table[1] == "TYPE   \tNUM\tTOTAL" && table[3] == "pose     \t1\t1" && table[4] == "neighbor \t1\t1" && table[5] == "local    \t1\t1" &&
             table[7] == "TOTAL    \t3\t3" && workspace.baStats.frameCount(BaStats::Ba::LOCAL) == 0;
        table.clear();
        workspace.baStats.finishFrame();                                       // a frame without any BA counts as NONE
        ok = ok && table[2] == "none     \t1\t1" && table[7] == "TOTAL    \t1\t4" && workspace.baStats.totalCount(BaStats::Ba::NONE) == 1;
        BaStats off(false); off.update(BaStats::Ba::LOCAL); off.finishFrame();
        if (!ok || off.totalCount(BaStats::Ba::LOCAL) != 0) { std::printf("BaStats mismatch\n"); return 14; }
    }
    std::printf("BA stage1 chi2 %.3f -> %.3f, stage2 -> %.3f, iterations %d/%d\n", o.stage1.chi2_initial, o.stage1.chi2_final, o.stage2.chi2_final,
                o.stage1.iterations, o.stage2.iterations);
    if (!(o.stage2.chi2_final <= o.stage1.chi2_initial)) return 4;
    {   // poseBundleAdjust (bundle_adjuster.cpp:396-491) through the mirror: the current keyframe knocked off its place, every point fixed -> the pose-only kernel
        BaWindow wp = w;
        const std::array<double, 7> before = wp.poses[2];
        wp.poses[2][4] += 0.03; wp.poses[2][5] -= 0.02;
        const auto pts = wp.points; const auto p0 = wp.poses[0], p1 = wp.poses[1];
        ms_ba_result pr{};
        const bool ran = poseBundleAdjust(ctx, wp, 10, &pr);
        const double moved = std::fabs(wp.poses[2][4] - before[4]) + std::fabs(wp.poses[2][5] - before[5]);
        std::printf("poseBundleAdjust: chi2 %.3f -> %.3f in %d iterations, pose back within %.2e of where it was\n", pr.chi2_initial, pr.chi2_final, pr.iterations, moved);
        if (!ran || !(pr.chi2_final < 0.05 * pr.chi2_initial) || moved > 2e-3 || wp.points != pts || wp.poses[0] != p0 || wp.poses[1] != p1) return 18;
        BaWindow empty; empty.poses.push_back({0, 0, 0, 1, 0, 0, 0});
        if (poseBundleAdjust(ctx, empty, 10)) return 19;                           // nothing to adjust: false, like :410-412
    }
    {   // nothing is allocated on the per-keyframe path after warm-up.  24 consecutive sliding windows (12 keyframes x 360 points, 6 views each, every window
        // with its own noise and its newest keyframe free in stage 1) through the two-stage mirror; the library's own count of allocations -- handle objects, growth of
        // its host scratch, device blocks, pinned staging, events (ms_debug_host_allocs) -- must stand still from the fifth window on.
        unsigned rng = 7u;
        auto uni = [&]() { rng = rng * 1664525u + 1013904223u; return (rng >> 8) / 16777216.0; };
        auto make = [&](int shift) {
            BaWindow s; s.currentKeyframe = 11;
            for (int i = 0; i < 12; ++i) s.poses.push_back({0, 0, 0, 1, -0.2 * (i + shift) + 0.004 * (uni() - 0.5), 0.004 * (uni() - 0.5), 0});
            for (int l = 0; l < 360; ++l) {
                const int s0 = l % 7;
                const double X = 0.2 * (s0 + shift) + 0.6 + 1.2 * (uni() - 0.5), Y = 1.6 * (uni() - 0.5), Z = 4.0 + 4.0 * uni();
                s.points.push_back({X + 0.02 * (uni() - 0.5), Y + 0.02 * (uni() - 0.5), Z + 0.05 * (uni() - 0.5)});
                for (int i = s0; i < s0 + 6; ++i) {
                    s.obsPose.push_back(i); s.obsPoint.push_back(l);
                    s.obsUv.push_back({(X - 0.2 * (i + shift)) / Z + 1e-3 * (uni() - 0.5), Y / Z + 1e-3 * (uni() - 0.5)}); s.obsInfo.push_back(250000.0);
                }
            }
            for (int i = 1; i < 12; ++i) { s.edgeI.push_back(i); s.edgeJ.push_back(i - 1); s.edgeMeas.push_back({0, 0, 0, 1, 0.2, 0, 0});
                std::array<double, 36> info{}; for (int k = 0; k < 6; ++k) info[7 * k] = 1e4; s.edgeInfo.push_back(info); }
            return s;
        };
        long long at_warm = 0;
        double worst = 0;
        for (int k = 0; k < 24; ++k) {
            BaWindow s = make(k);
            if (k == 4) at_warm = ms_debug_host_allocs();
            const BaOutcome bo = localBundleAdjust(ctx, s, 12, params, true, nullptr);
            worst = std::max(worst, bo.stage2.chi2_final / std::max(bo.stage1.chi2_initial, 1e-30));
        }
        const long long after = ms_debug_host_allocs();
        std::printf("24 sliding windows: library allocations after warm-up %lld (total so far %lld), worst chi2 ratio %.3g\n", after - at_warm, after, worst);
        if (after != at_warm || !(worst < 1.0)) return 21;
        // ... and the per-frame path beside it (mapper_helpers.cpp:1043-1050, :1079-1081): a poseBundleAdjust per frame, a window on every 5th -- the staging block, the
        // handles' result blocks and their objects go round between the two kinds of problem; nothing new after the first keyframes
        long long at_warm2 = 0;
        for (int f = 0; f < 60; ++f) {
            if (f == 20) at_warm2 = ms_debug_host_allocs();
            BaWindow s = make(f);
            BaWindow sp = s;                                              // the newest keyframe against the window's points (all fixed), knocked off its place
            sp.poses[11][4] += 0.01; sp.poses[11][5] -= 0.01;
            ms_ba_result pr{};
            if (!poseBundleAdjust(ctx, sp, 10, &pr) || !(pr.chi2_final < pr.chi2_initial)) return 22;
            if (f % 5 == 0) { const BaOutcome bo = localBundleAdjust(ctx, s, 12, params, true, nullptr); if (!(bo.stage2.chi2_final <= bo.stage1.chi2_initial)) return 23; }
        }
        const long long after2 = ms_debug_host_allocs();
        std::printf("60 frames (a pose adjustment each, a window on every 5th): library allocations after warm-up %lld\n", after2 - at_warm2);
        if (after2 != at_warm2) return 24;
    }
    // globalBundleAdjust-sized map: 200 keyframes on a line (the current one fixed), 1500 points each seen by 8 consecutive keyframes
    {
        BaWindow g; g.currentKeyframe = 199;
        unsigned rng = 99u;
        auto uni = [&]() { rng = rng * 1664525u + 1013904223u; return (rng >> 8) / 16777216.0; };
        for (int i = 0; i < 200; ++i) g.poses.push_back({0, 0, 0, 1, -0.2 * i + (i == 199 ? 0.0 : 0.01 * (uni() - 0.5)), 0.004 * (uni() - 0.5), 0});
        for (int l = 0; l < 1500; ++l) {
            const int s0 = (int)(uni() * 192);
            const double X = 0.2 * s0 + 0.7 + 1.5 * (uni() - 0.5), Y = 2.0 * (uni() - 0.5), Z = 4.0 + 5.0 * uni();
            g.points.push_back({X + 0.02 * (uni() - 0.5), Y + 0.02 * (uni() - 0.5), Z + 0.02 * (uni() - 0.5)});
            for (int i = s0; i < s0 + 8; ++i) {
                g.obsPose.push_back(i); g.obsPoint.push_back(l); g.obsUv.push_back({(X - 0.2 * i) / Z + 2e-3 * (uni() - 0.5), Y / Z + 2e-3 * (uni() - 0.5)}); g.obsInfo.push_back(250000.0);
            }
        }
        for (int i = 1; i < 200; ++i) { g.edgeI.push_back(i); g.edgeJ.push_back(i - 1); g.edgeMeas.push_back({0, 0, 0, 1, 0.2, 0, 0});
            std::array<double, 36> info{}; for (int k = 0; k < 6; ++k) info[7 * k] = 1e3; g.edgeInfo.push_back(info); }
        BaOutcome go = globalBundleAdjust(ctx, g, 10);
        int nOut = 0; for (auto o : go.outlier) nOut += o;
        std::printf("global BA (199 free keyframes): chi2 %.1f -> %.1f in %d iterations, %d of %zu observations above the threshold\n",
                    go.stage1.chi2_initial, go.stage1.chi2_final, go.stage1.iterations, nOut, go.outlier.size());
        if (!(go.stage1.chi2_final < 0.2 * go.stage1.chi2_initial) || nOut > (int)go.outlier.size() / 20) return 10;
    }
    std::printf("host shims ok\n");
    return 0;
}
