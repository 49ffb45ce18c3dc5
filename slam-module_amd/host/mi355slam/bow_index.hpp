// bow_index.hpp -- host mirror of BowIndex::transform (bow_index.hpp:36-64, bow_index.cpp:59-93).
//
// The reference's BowIndex owns a DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> loaded from a file and calls its batch
// transform with levelsUp = 4.  DBoW2 is an external library (not in the reference tree); the mirror keeps the vocabulary as
// the flat arrays DBoW2 holds per node, sends the tree descent to the device (ms_bow_transform) and assembles the two ordered
// maps on the host in feature order, exactly as DBoW2's batch transform does: v[word] += weight and fv[node].push_back(i) for
// weight > 0, then BowVector::normalize(L1).  add / remove / getBowSimilar of the reference are inverted-index bookkeeping on
// the map graph and stay in the reference's own class; only `transform` is replaced.
#pragma once
#include <cmath>
#include <fstream>
#include <map>
#include <sstream>
#include "common.hpp"

namespace mi355slam {

using BowVector = std::map<unsigned, double>;                        // DBoW2::BowVector: WordId -> WordValue
using FeatureVector = std::map<unsigned, std::vector<unsigned>>;     // DBoW2::FeatureVector: NodeId -> feature indices

// The vocabulary as DBoW2 stores it (TemplatedVocabulary::m_nodes): node 0 is the root.
struct VocabularyTree {
    int branchingFactor = 0, depthLevels = 0;                        // m_k, m_L
    std::vector<std::int32_t> parent, wordId;                        // wordId = -1 for inner nodes
    std::vector<std::uint32_t> descriptor;                           // 8 words per node
    std::vector<double> weight;
    std::size_t size() const { return parent.size(); }

    // DBoW2's text vocabulary format (the one ORB-SLAM's vocabularies ship in; TemplatedVocabulary::loadFromTextFile): a
    // header line "k L scoring weighting", then one line per node "parent isLeaf b0 .. b31 weight" with the descriptor as 32
    // decimal bytes.  Nodes get ids in file order starting at 1, words get ids in the order their leaves appear.
    static VocabularyTree loadFromTextFile(const std::string &path) {
        std::ifstream f(path);
        if (!f) throw std::runtime_error("mi355slam: cannot open vocabulary " + path);
        VocabularyTree v;
        std::string line;
        std::getline(f, line);
        int scoring = 0, weighting = 0;
        { std::stringstream ss(line); ss >> v.branchingFactor >> v.depthLevels >> scoring >> weighting; }
        if (v.branchingFactor < 0 || v.branchingFactor > 20 || v.depthLevels < 1 || v.depthLevels > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3)
            throw std::runtime_error("mi355slam: vocabulary header out of range in " + path);
        v.parent.push_back(0); v.wordId.push_back(-1); v.weight.push_back(0.0); v.descriptor.resize(8, 0u);
        int words = 0;
        while (std::getline(f, line)) {
            if (line.empty()) continue;
            std::stringstream ss(line);
            int pid = 0, leaf = 0;
            ss >> pid >> leaf;
            std::uint8_t bytes[32];
            for (int k = 0; k < 32; ++k) { int b = 0; ss >> b; bytes[k] = (std::uint8_t)b; }
            double w = 0.0;
            ss >> w;
            if (!ss) throw std::runtime_error("mi355slam: malformed vocabulary line in " + path);
            v.parent.push_back(pid); v.weight.push_back(w); v.wordId.push_back(leaf > 0 ? words++ : -1);
            for (int k = 0; k < 8; ++k)
                v.descriptor.push_back((std::uint32_t)bytes[4 * k] | ((std::uint32_t)bytes[4 * k + 1] << 8) | ((std::uint32_t)bytes[4 * k + 2] << 16) | ((std::uint32_t)bytes[4 * k + 3] << 24));
        }
        return v;
    }
};

class BowIndex {
public:
    BowIndex(Context &ctx, const VocabularyTree &tree) : ctx_(ctx) {
        ctx_.check(ms_bow_vocab_create(ctx_.get(), (int)tree.size(), tree.parent.data(), tree.descriptor.data(), tree.weight.data(), tree.wordId.data(),
                                       tree.depthLevels, &vocab_), "ms_bow_vocab_create");
    }
    ~BowIndex() { ms_bow_vocab_destroy(vocab_); for (void *p : {d_desc_, d_word_, d_weight_, d_node_}) if (p) ms_dev_free(ctx_.get(), p); }
    BowIndex(const BowIndex &) = delete;

    // bow_index.cpp:59-93
    void transform(const KeyPointVector &keypoints, BowVector &bowVector, FeatureVector &bowFeatureVector) {
        const int levelsUp = 4;                                                   // bow_index.cpp:85
        const std::size_t n = keypoints.size();
        bowVector.clear(); bowFeatureVector.clear();
        if (n == 0) return;
        reserve(n);
        host_desc_.resize(8 * n);
        for (std::size_t i = 0; i < n; ++i) for (int k = 0; k < 8; ++k) host_desc_[8 * i + k] = keypoints[i].descriptor[k];
        ctx_.check(ms_dev_upload(ctx_.get(), d_desc_, host_desc_.data(), 32 * n), "ms_dev_upload");
        ctx_.check(ms_bow_transform(ctx_.get(), vocab_, static_cast<const std::uint32_t *>(d_desc_), (int)n, levelsUp,
                                    static_cast<std::int32_t *>(d_word_), static_cast<double *>(d_weight_), static_cast<std::int32_t *>(d_node_)), "ms_bow_transform");
        word_.resize(n); weight_.resize(n); node_.resize(n);
        ctx_.check(ms_dev_download(ctx_.get(), word_.data(), d_word_, 4 * n), "ms_dev_download");
        ctx_.check(ms_dev_download(ctx_.get(), weight_.data(), d_weight_, 8 * n), "ms_dev_download");
        ctx_.check(ms_dev_download(ctx_.get(), node_.data(), d_node_, 4 * n), "ms_dev_download");
        assemble(word_, weight_, node_, bowVector, bowFeatureVector);
    }

    // DBoW2's batch transform after the per-feature descents (TF_IDF / TF weighting, L1 scoring): host only
    static void assemble(const std::vector<std::int32_t> &word, const std::vector<double> &weight, const std::vector<std::int32_t> &node,
                         BowVector &v, FeatureVector &fv) {
        for (std::size_t i = 0; i < word.size(); ++i) {
            if (!(weight[i] > 0)) continue;                                       // stop words
            v[(unsigned)word[i]] += weight[i];                                    // BowVector::addWeight
            fv[(unsigned)node[i]].push_back((unsigned)i);                         // FeatureVector::addFeature
        }
        double norm = 0.0;
        for (const auto &kv : v) norm += std::fabs(kv.second);                    // BowVector::normalize(L1)
        if (norm > 0.0) for (auto &kv : v) kv.second /= norm;
    }

private:
    void reserve(std::size_t n) {
        if (n <= cap_) return;
        for (void **p : {&d_desc_, &d_word_, &d_weight_, &d_node_}) if (*p) { ms_dev_free(ctx_.get(), *p); *p = nullptr; }
        cap_ = n + n / 2 + 64;
        ctx_.check(ms_dev_alloc(ctx_.get(), 32 * cap_, &d_desc_), "ms_dev_alloc");
        ctx_.check(ms_dev_alloc(ctx_.get(), 4 * cap_, &d_word_), "ms_dev_alloc");
        ctx_.check(ms_dev_alloc(ctx_.get(), 8 * cap_, &d_weight_), "ms_dev_alloc");
        ctx_.check(ms_dev_alloc(ctx_.get(), 4 * cap_, &d_node_), "ms_dev_alloc");
    }
    Context &ctx_;
    ms_bow_vocab *vocab_ = nullptr;
    void *d_desc_ = nullptr, *d_word_ = nullptr, *d_weight_ = nullptr, *d_node_ = nullptr;
    std::size_t cap_ = 0;
    std::vector<std::uint32_t> host_desc_;
    std::vector<std::int32_t> word_, node_;
    std::vector<double> weight_;
};

}  // namespace mi355slam
