/*
 * oracle/bow.c -- CPU ORACLE (test infrastructure, NOT product code): vocabulary-tree descent of a binary descriptor.
 *
 * N3 of SURVEY.md 8(f): BowIndex::transform (bow_index.cpp:59-93) hands every keypoint descriptor to
 * DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(features, bowVector, featureVector, levelsup = 4)
 * (bow_index.cpp:86-92).  DBoW2 is a third-party dependency that is NOT in the reference tree (bow_index.hpp:8-9 includes
 * <DBoW2/FORB.h>, <DBoW2/TemplatedVocabulary.h>; no version is pinned in the tree and no vocabulary file ships with it), so
 * this file restates DBoW2's published algorithm (dorian3d/DBoW2, TemplatedVocabulary.h: the per-feature transform and the
 * batch transform; FORB.cpp: distance) and is PARITY UNPINNED against the reference: there are no golden vectors for it.
 *
 *   per feature:  start at the root; at every level take the child with the smallest Hamming distance to the feature, the
 *                 FIRST child winning ties (strict `d < best_d`, children visited in the order they were attached = ascending
 *                 node id); remember the node reached at level L - levelsup; stop at a node without children; report its
 *                 word id and weight.
 *   batch:        v[word] += weight and fv[node].push_back(i) for every feature i, in feature order, when weight > 0;
 *                 then L1 normalisation of v (the scoring object of the ORB vocabularies: L1_NORM, weighting TF_IDF).
 */
#include <stdlib.h>
#include <string.h>
#include "mso.h"

static unsigned bow_distance(const uint32_t *a, const uint32_t *b) {    /* FORB::distance: bit count of the xor, as a double there */
    unsigned d = 0;
    for (int k = 0; k < 8; ++k) d += (unsigned)__builtin_popcount(a[k] ^ b[k]);
    return d;
}

/* Descent of n descriptors.  The tree is given the way DBoW2 stores it: parent id per node (node 0 = root), children of a
 * node in ascending node id.  word[i] = -1 and weight[i] = 0 for an empty vocabulary (a root without children; DBoW2's batch
 * transform returns before descending, `if (empty()) return`).  node[i]: the node at level depth_levels - levels_up, 0 when
 * that level is <= 0; when the leaf lies ABOVE that level DBoW2 leaves *nid unset -- reported here as the leaf's own id. */
void mso_bow_transform(int n_nodes, const int32_t *parent, const uint32_t *node_desc, const double *node_weight,
                       const int32_t *node_word, int depth_levels, const uint32_t *desc, int n, int levels_up,
                       int32_t *word, double *weight, int32_t *node) {
    int *n_child = (int *)calloc((size_t)n_nodes + 1, sizeof(int)), *first = (int *)calloc((size_t)n_nodes + 1, sizeof(int));
    int *child = (int *)calloc((size_t)n_nodes + 1, sizeof(int)), *fill = (int *)calloc((size_t)n_nodes + 1, sizeof(int));
    for (int i = 1; i < n_nodes; ++i) n_child[parent[i]]++;
    for (int i = 1; i < n_nodes; ++i) first[i] = first[i - 1] + n_child[i - 1];
    for (int i = 1; i < n_nodes; ++i) child[first[parent[i]] + fill[parent[i]]++] = i;      /* ascending id = attach order */
    const int nid_level = depth_levels - levels_up;
    for (int i = 0; i < n; ++i) {
        const uint32_t *f = desc + 8 * (size_t)i;
        if (n_nodes < 2 || n_child[0] == 0) { word[i] = -1; if (weight) weight[i] = 0.0; if (node) node[i] = 0; continue; }
        int final_id = 0, level = 0, nid = -1;
        if (nid_level <= 0) nid = 0;
        do {
            ++level;
            const int *c = child + first[final_id];
            const int nc = n_child[final_id];
            final_id = c[0];
            unsigned best = bow_distance(f, node_desc + 8 * (size_t)final_id);
            for (int k = 1; k < nc; ++k) {
                const unsigned d = bow_distance(f, node_desc + 8 * (size_t)c[k]);
                if (d < best) { best = d; final_id = c[k]; }
            }
            if (level == nid_level) nid = final_id;
        } while (n_child[final_id] != 0);
        if (nid < 0) nid = final_id;
        word[i] = node_word[final_id];
        if (weight) weight[i] = node_weight[final_id];
        if (node) node[i] = nid;
    }
    free(n_child); free(first); free(child); free(fill);
}

/* Batch assembly on sorted arrays (std::map iteration order): out_words / out_values = the BowVector after L1 normalisation,
 * returns its size; fv_nodes / fv_start / fv_feat = the FeatureVector as CSR over ascending node ids (features of a node in
 * ascending feature index), *n_fv its size.  All outputs sized >= n (fv_start n + 1). */
int mso_bow_assemble(const int32_t *word, const double *weight, const int32_t *node, int n,
                     int32_t *out_words, double *out_values, int32_t *fv_nodes, int32_t *fv_start, int32_t *fv_feat, int *n_fv) {
    int nv = 0, nf = 0;
    int32_t *fnode = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    for (int i = 0; i < n; ++i) {
        if (!(weight[i] > 0)) continue;                       /* stop words */
        int lo = 0;
        while (lo < nv && out_words[lo] < word[i]) ++lo;     /* BowVector::addWeight: lower_bound, add or insert */
        if (lo < nv && out_words[lo] == word[i]) out_values[lo] += weight[i];
        else {
            memmove(out_words + lo + 1, out_words + lo, sizeof(int32_t) * (size_t)(nv - lo));
            memmove(out_values + lo + 1, out_values + lo, sizeof(double) * (size_t)(nv - lo));
            out_words[lo] = word[i]; out_values[lo] = weight[i]; ++nv;
        }
        fnode[i] = node[i];
    }
    double norm = 0.0;                                        /* BowVector::normalize(L1): sum of |v| in word order */
    for (int k = 0; k < nv; ++k) norm += out_values[k] < 0 ? -out_values[k] : out_values[k];
    if (norm > 0.0) for (int k = 0; k < nv; ++k) out_values[k] /= norm;
    /* FeatureVector: distinct nodes ascending, members in feature order */
    for (int i = 0; i < n; ++i) {
        if (!(weight[i] > 0)) continue;
        int lo = 0;
        while (lo < nf && fv_nodes[lo] < fnode[i]) ++lo;
        if (lo < nf && fv_nodes[lo] == fnode[i]) continue;
        memmove(fv_nodes + lo + 1, fv_nodes + lo, sizeof(int32_t) * (size_t)(nf - lo));
        fv_nodes[lo] = fnode[i]; ++nf;
    }
    int at = 0;
    for (int k = 0; k < nf; ++k) {
        fv_start[k] = at;
        for (int i = 0; i < n; ++i) if (weight[i] > 0 && fnode[i] == fv_nodes[k]) fv_feat[at++] = i;
    }
    fv_start[nf] = at;
    *n_fv = nf;
    free(fnode);
    return nv;
}
