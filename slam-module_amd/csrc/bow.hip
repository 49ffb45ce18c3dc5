// bow.hip -- N3: vocabulary-tree descent of ORB descriptors on the device.
//
// BowIndex::transform (bow_index.cpp:59-93) copies every keypoint descriptor into a cv::Mat and calls
// DBoW2::TemplatedVocabulary<FORB>::transform(features, bowVector, featureVector, levelsup = 4) (:86-92), which walks each
// descriptor down the k-ary tree by smallest Hamming distance.  Here the walk is one kernel over all descriptors of a batch:
// 16 lanes per descriptor, one child per lane (32 B = two dwordx4 loads, the 16 lanes of a group read one contiguous run of
// 512 B because the tree is re-laid out with every node's children adjacent), xor + v_bcnt, and a 4-step DPP rotate-min over
// the group on the key (distance << 16 | child position): the FIRST child wins ties exactly like the strict `d < best_d` scan.
// The word/weight/node triple per descriptor is all the device produces; the two std::maps (BowVector with its L1
// normalisation, FeatureVector) are assembled from it by the host mirror in feature order, which keeps the floating-point sums
// identical to the reference's.
#include "ms_internal.h"

namespace {

struct BowTree {
    const int32_t *first_child;    // [n] internal index of the first child (children are contiguous)
    const int32_t *n_children;     // [n]
    const uint4 *desc;             // [n][2]
    const int32_t *orig_id;        // [n] node id of the caller's numbering
    const int32_t *word;           // [n] word id (-1 for inner nodes)
    const double *weight;          // [n]
    int depth_levels, max_depth;
};

__device__ __forceinline__ uint32_t group_min16(uint32_t v) {      // min over each row of 16 lanes, result in every lane
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121 /*row_ror:1*/, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122 /*row_ror:2*/, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124 /*row_ror:4*/, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128 /*row_ror:8*/, 0xf, 0xf, false));
    return v;
}

__global__ __launch_bounds__(256) void k_bow_descend(BowTree T, const uint4 *__restrict__ desc, int n, int levels_up,
                                                     int32_t *__restrict__ out_word, double *__restrict__ out_weight, int32_t *__restrict__ out_node) {
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const bool live = i < n;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    if (live) { q0 = desc[2 * (size_t)i]; q1 = desc[2 * (size_t)i + 1]; }
    const int nid_level = T.depth_levels - levels_up;
    int cur = 0, level = 0, nid = nid_level <= 0 ? 0 : -1;
    int nc = live ? T.n_children[0] : 0;
    const bool empty = nc == 0;
    // every group walks until its node has no children; max_depth bounds the loop for any tree the host accepted
    for (int step = 0; step < T.max_depth && __ballot(nc > 0) != 0; ++step) {
        if (nc > 0) {
            const int first = T.first_child[cur];
            uint32_t best = 0xFFFFFFFFu;
            for (int c0 = 0; c0 < nc; c0 += 16) {
                const int c = c0 + sub;
                uint32_t key = 0xFFFFFFFFu;
                if (c < nc) {
                    const uint4 a = T.desc[2 * (size_t)(first + c)], b = T.desc[2 * (size_t)(first + c) + 1];
                    const uint32_t d = __popc(a.x ^ q0.x) + __popc(a.y ^ q0.y) + __popc(a.z ^ q0.z) + __popc(a.w ^ q0.w) +
                                       __popc(b.x ^ q1.x) + __popc(b.y ^ q1.y) + __popc(b.z ^ q1.z) + __popc(b.w ^ q1.w);
                    key = (d << 16) | (uint32_t)c;
                }
                best = min(best, key);
            }
            best = group_min16(best);
            cur = first + (int)(best & 0xFFFFu);
            ++level;
            if (level == nid_level) nid = cur;
            nc = T.n_children[cur];
        }
    }
    if (live && sub == 0) {
        if (empty) { out_word[i] = -1; if (out_weight) out_weight[i] = 0.0; if (out_node) out_node[i] = 0; return; }
        out_word[i] = T.word[cur];
        if (out_weight) out_weight[i] = T.weight[cur];
        if (out_node) out_node[i] = nid < 0 ? T.orig_id[cur] : (nid == 0 ? 0 : T.orig_id[nid]);
    }
}

}  // namespace

struct ms_bow_vocab {
    ms_ctx *ctx = nullptr;
    void *slab = nullptr;
    BowTree tree{};
    int n_nodes = 0, max_children = 0;
};

extern "C" {

int ms_bow_vocab_create(ms_ctx *c, int n_nodes, const int32_t *parent, const uint32_t *node_desc, const double *node_weight,
                        const int32_t *node_word, int depth_levels, ms_bow_vocab **out) {
    if (!c || !out || n_nodes < 1 || !parent || !node_desc || !node_weight || !node_word || depth_levels < 0) return MS_ERR_INVALID;
    // children in ascending node id (the order DBoW2 attaches them while loading / building); parents must precede children
    std::vector<int32_t> n_child((size_t)n_nodes, 0), first((size_t)n_nodes, 0), fill((size_t)n_nodes, 0), child((size_t)n_nodes, 0);
    for (int i = 1; i < n_nodes; ++i) {
        if (parent[i] < 0 || parent[i] >= i) return ms_fail(c, MS_ERR_INVALID, "bow vocabulary: parent of node %d is %d (must be an earlier node)", i, parent[i]);
        n_child[(size_t)parent[i]]++;
    }
    int max_children = 0;
    for (int i = 0; i < n_nodes; ++i) max_children = n_child[(size_t)i] > max_children ? n_child[(size_t)i] : max_children;
    if (max_children > 65535) return ms_fail(c, MS_ERR_CAPACITY, "bow vocabulary: a node has %d children (max 65535)", max_children);
    for (int i = 1; i < n_nodes; ++i) first[(size_t)i] = first[(size_t)i - 1] + n_child[(size_t)i - 1];
    for (int i = 1; i < n_nodes; ++i) child[(size_t)(first[(size_t)parent[i]] + fill[(size_t)parent[i]]++)] = i;
    // breadth-first renumbering: internal index -> original id, children of a node adjacent
    std::vector<int32_t> orig((size_t)n_nodes), inner_first((size_t)n_nodes, 0), inner_nc((size_t)n_nodes, 0), depth((size_t)n_nodes, 0);
    orig[0] = 0;
    int tail = 1, max_depth = 0;
    for (int head = 0; head < tail; ++head) {
        const int o = orig[(size_t)head];
        inner_first[(size_t)head] = tail; inner_nc[(size_t)head] = n_child[(size_t)o];
        for (int k = 0; k < n_child[(size_t)o]; ++k) { orig[(size_t)tail] = child[(size_t)(first[(size_t)o] + k)]; depth[(size_t)tail] = depth[(size_t)head] + 1; ++tail; }
        max_depth = depth[(size_t)head] > max_depth ? depth[(size_t)head] : max_depth;
    }
    if (tail != n_nodes) return ms_fail(c, MS_ERR_INVALID, "bow vocabulary: %d of %d nodes are reachable from the root", tail, n_nodes);
    const size_t N = (size_t)n_nodes;
    const size_t o_first = 0, o_nc = ms_align_up(o_first + 4 * N, 256), o_desc = ms_align_up(o_nc + 4 * N, 256), o_orig = ms_align_up(o_desc + 32 * N, 256),
                 o_word = ms_align_up(o_orig + 4 * N, 256), o_weight = ms_align_up(o_word + 4 * N, 256), total = o_weight + 8 * N;
    std::vector<uint8_t> host(total, 0);
    for (size_t k = 0; k < N; ++k) {
        const size_t o = (size_t)orig[k];
        reinterpret_cast<int32_t *>(host.data() + o_first)[k] = inner_first[k];
        reinterpret_cast<int32_t *>(host.data() + o_nc)[k] = inner_nc[k];
        for (int w = 0; w < 8; ++w) reinterpret_cast<uint32_t *>(host.data() + o_desc)[8 * k + (size_t)w] = node_desc[8 * o + (size_t)w];
        reinterpret_cast<int32_t *>(host.data() + o_orig)[k] = (int32_t)o;
        reinterpret_cast<int32_t *>(host.data() + o_word)[k] = node_word[o];
        reinterpret_cast<double *>(host.data() + o_weight)[k] = node_weight[o];
    }
    MS_HIP(c, hipSetDevice(c->device));
    ms_bow_vocab *v = new ms_bow_vocab();
    v->ctx = c; v->n_nodes = n_nodes; v->max_children = max_children;
    hipError_t e = hipMalloc(&v->slab, total);
    if (e == hipSuccess) e = hipMemcpy(v->slab, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) { if (v->slab) (void)hipFree(v->slab); delete v; return ms_fail(c, MS_ERR_HIP, "bow vocabulary upload failed: %s", hipGetErrorString(e)); }
    char *b = static_cast<char *>(v->slab);
    v->tree.first_child = reinterpret_cast<const int32_t *>(b + o_first);
    v->tree.n_children = reinterpret_cast<const int32_t *>(b + o_nc);
    v->tree.desc = reinterpret_cast<const uint4 *>(b + o_desc);
    v->tree.orig_id = reinterpret_cast<const int32_t *>(b + o_orig);
    v->tree.word = reinterpret_cast<const int32_t *>(b + o_word);
    v->tree.weight = reinterpret_cast<const double *>(b + o_weight);
    v->tree.depth_levels = depth_levels; v->tree.max_depth = max_depth + 1;
    *out = v;
    return MS_OK;
}

void ms_bow_vocab_destroy(ms_bow_vocab *v) {
    if (!v) return;
    if (v->slab) (void)hipFree(v->slab);
    delete v;
}

int ms_bow_transform(ms_ctx *c, const ms_bow_vocab *v, const uint32_t *desc, int n, int levels_up, int32_t *word, double *weight, int32_t *node) {
    MsRange range("Bow index transform");     // the reference's timer name, mapper_helpers.cpp:1193
    if (!c || !v || n < 0 || (n > 0 && (!desc || !word))) return MS_ERR_INVALID;
    if (n == 0) return MS_OK;
    if (reinterpret_cast<uintptr_t>(desc) % 16) return ms_fail(c, MS_ERR_INVALID, "bow transform: descriptors must be 16-byte aligned");
    MS_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_bow_descend, dim3(ms_div_up(n, 16)), dim3(256), 0, c->stream, v->tree, reinterpret_cast<const uint4 *>(desc), n, levels_up, word, weight, node);
    MS_KERNEL_CHECK(c, "k_bow_descend");
    return MS_OK;
}

}  // extern "C"
