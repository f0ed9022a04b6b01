// pt_bvh.cpp — host-side BVH build, the stand-in for `e_BVH::unanimated(8, ..)` (bvh.rs:17-21,
// triangle_mesh.rs:44).  The reference's builder lives in an un-vendored crate, so
// this is the build's own: binned SAH over triangle centroids, leaves of at most `max_leaf`
// triangles, flattened into the two-child node format of include/pt.h.
//
// Two properties the extend kernel relies on:
//  * depth bound: the tree never gets deeper than kMaxDepth, so the per-lane LDS stack of
//    ptd::kStackDepth entries cannot overflow (when the SAH split would leave too little depth
//    budget for the remaining triangles the builder switches to median splits);
//  * conservative boxes: child boxes are stored as f32 and padded outward by ~2^-21 of the scene's
//    largest coordinate, nine orders of magnitude more than the rounding error of the f64 slab
//    test, so a box never culls a triangle that triangle_mesh.rs:101-126 would accept.
#include "pt_host.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <thread>

namespace pth {

namespace {

struct Ref {
    float lo[3], hi[3];
    float c[3];
    uint32_t id;
};
struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    }
    void grow(const float* l, const float* h) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
    }
    void grow_pt(const float* p) {
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.f)) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct TNode {  // temporary tree
    Box box;
    int left = -1, right = -1;
    uint32_t first = 0, count = 0;
};

constexpr int kBins = 16;
constexpr int kMaxDepth = 30;  // inner-node levels; stack needs <= depth entries (< ptd::kStackDepth = 32)

// fork-join over [0, n) in `threads` contiguous chunks (chunk t on thread t; the caller runs chunk 0)
template <class F>
void parallel_chunks(size_t n, uint32_t threads, F fn) {
    threads = (uint32_t)std::max<size_t>(1, std::min<size_t>(threads, n / 4096 + 1));
    auto bounds = [&](uint32_t t) { return n * t / threads; };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < threads; ++t) pool.emplace_back([&, t] { fn(bounds(t), bounds(t + 1), t); });
    fn(bounds(0), bounds(1), 0u);
    for (auto& t : pool) t.join();
}

struct Pending {  // a subtree the top-level pass left for a worker thread
    int node;
    uint32_t first, count;
    int depth;
};

struct Builder {
    std::vector<Ref>& refs;
    uint32_t max_leaf;
    std::vector<TNode> nodes;
    uint32_t max_depth = 0;
    // top-level pass of a parallel build: ranges of at most `cut` references are not split but
    // recorded in `pending` (their node holds only first/count until the subtree is spliced in)
    uint32_t cut = 0;
    std::vector<Pending>* pending = nullptr;
    uint32_t threads = 1;            // the top-level pass also splits its big ranges with this many threads
    std::vector<Ref>* scratch = nullptr;  // n references, for the out-of-place parallel partition
    static constexpr uint32_t kParallelRange = 1u << 17;

    Builder(std::vector<Ref>& r, uint32_t ml) : refs(r), max_leaf(ml) {}

    static int ceil_log2(uint32_t x) {
        int l = 0;
        while ((1u << l) < x) ++l;
        return l;
    }

    int build(uint32_t first, uint32_t count, int depth) {
        int idx = (int)nodes.size();
        nodes.emplace_back();
        if (pending && count <= cut && count > max_leaf) {
            nodes[idx].first = first;
            nodes[idx].count = count;
            pending->push_back({idx, first, count, depth});
            return idx;
        }
        const uint32_t par = (threads > 1 && count >= kParallelRange) ? threads : 1;
        Box box, cbox;
        box.reset();
        cbox.reset();
        {
            std::vector<Box> pb(par), pc(par);
            parallel_chunks(count, par, [&](size_t b0, size_t b1, uint32_t t) {
                Box bx, cx;
                bx.reset();
                cx.reset();
                for (size_t i = first + b0; i < first + b1; ++i) {
                    bx.grow(refs[i].lo, refs[i].hi);
                    cx.grow_pt(refs[i].c);
                }
                pb[t] = bx;
                pc[t] = cx;
            });
            for (uint32_t t = 0; t < par; ++t) {  // unused chunks keep the reset (empty) box
                if (pb[t].lo[0] <= pb[t].hi[0]) { box.grow(pb[t].lo, pb[t].hi); cbox.grow(pc[t].lo, pc[t].hi); }
            }
        }
        nodes[idx].box = box;
        nodes[idx].first = first;
        nodes[idx].count = count;
        max_depth = std::max(max_depth, (uint32_t)depth);
        if (count <= max_leaf) {
            // canonical leaf: triangles by id, so the output does not depend on how the ranges
            // above were permuted (serial std::partition vs. the stable parallel one)
            std::sort(refs.begin() + first, refs.begin() + first + count, [](const Ref& a, const Ref& b) { return a.id < b.id; });
            return idx;
        }

        uint32_t mid = 0;
        bool have_split = false;
        // levels still needed if we split evenly from here on
        int need = ceil_log2((count + max_leaf - 1) / max_leaf);
        bool budget_ok = depth + need + 2 < kMaxDepth;
        if (budget_ok) {
            float best_cost = INFINITY;
            int best_axis = -1, best_bin = -1;
            for (int axis = 0; axis < 3; ++axis) {
                float cmin = cbox.lo[axis], cmax = cbox.hi[axis];
                if (!(cmax > cmin)) continue;
                float scale = (float)kBins / (cmax - cmin);
                Box bb[kBins];
                uint32_t bc[kBins];
                for (int b = 0; b < kBins; ++b) { bb[b].reset(); bc[b] = 0; }
                if (par == 1) {
                    for (uint32_t i = first; i < first + count; ++i) {
                        int b = std::min(kBins - 1, (int)((refs[i].c[axis] - cmin) * scale));
                        bb[b].grow(refs[i].lo, refs[i].hi);
                        bc[b]++;
                    }
                } else {  // per-thread bins, merged: min/max and integer counts are order-independent
                    struct Bins { Box bb[kBins]; uint32_t bc[kBins]; };
                    std::vector<Bins> pbins(par);
                    for (auto& pbn : pbins) for (int b = 0; b < kBins; ++b) { pbn.bb[b].reset(); pbn.bc[b] = 0; }
                    parallel_chunks(count, par, [&](size_t b0, size_t b1, uint32_t t) {
                        Bins& my = pbins[t];
                        for (size_t i = first + b0; i < first + b1; ++i) {
                            int b = std::min(kBins - 1, (int)((refs[i].c[axis] - cmin) * scale));
                            my.bb[b].grow(refs[i].lo, refs[i].hi);
                            my.bc[b]++;
                        }
                    });
                    for (auto& pbn : pbins)
                        for (int b = 0; b < kBins; ++b)
                            if (pbn.bc[b]) { bb[b].grow(pbn.bb[b].lo, pbn.bb[b].hi); bc[b] += pbn.bc[b]; }
                }
                float right_area[kBins];
                uint32_t right_cnt[kBins];
                Box acc;
                acc.reset();
                uint32_t cnt = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(bb[b].lo, bb[b].hi);
                    cnt += bc[b];
                    right_area[b] = acc.half_area();
                    right_cnt[b] = cnt;
                }
                acc.reset();
                cnt = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(bb[b].lo, bb[b].hi);
                    cnt += bc[b];
                    if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = acc.half_area() * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                float cmin = cbox.lo[best_axis], cmax = cbox.hi[best_axis];
                float scale = (float)kBins / (cmax - cmin);
                auto is_left = [&](const Ref& r) {
                    int b = std::min(kBins - 1, (int)((r.c[best_axis] - cmin) * scale));
                    return b <= best_bin;
                };
                if (par == 1 || !scratch) {
                    auto it = std::partition(refs.begin() + first, refs.begin() + first + count, is_left);
                    mid = (uint32_t)(it - refs.begin());
                } else {  // stable out-of-place partition: count per chunk, prefix, scatter, copy back
                    std::vector<size_t> nl(par + 1, 0);
                    parallel_chunks(count, par, [&](size_t b0, size_t b1, uint32_t t) {
                        size_t c = 0;
                        for (size_t i = first + b0; i < first + b1; ++i) c += is_left(refs[i]);
                        nl[t + 1] = c;
                    });
                    for (uint32_t t = 0; t < par; ++t) nl[t + 1] += nl[t];
                    const size_t n_left = nl[par];
                    Ref* tmp = scratch->data() + first;
                    parallel_chunks(count, par, [&](size_t b0, size_t b1, uint32_t t) {
                        size_t l = nl[t], r = n_left + (b0 - nl[t]);
                        for (size_t i = first + b0; i < first + b1; ++i) {
                            if (is_left(refs[i])) tmp[l++] = refs[i]; else tmp[r++] = refs[i];
                        }
                    });
                    parallel_chunks(count, par, [&](size_t b0, size_t b1, uint32_t) {
                        std::memcpy(&refs[first + b0], tmp + b0, (b1 - b0) * sizeof(Ref));
                    });
                    mid = first + (uint32_t)n_left;
                }
                have_split = mid > first && mid < first + count;
            }
        }
        if (!have_split) {  // median split on the widest centroid axis (also the depth-budget fallback)
            int axis = 0;
            for (int a = 1; a < 3; ++a)
                if (cbox.hi[a] - cbox.lo[a] > cbox.hi[axis] - cbox.lo[axis]) axis = a;
            mid = first + count / 2;
            std::nth_element(refs.begin() + first, refs.begin() + mid, refs.begin() + first + count,
                             [&](const Ref& a, const Ref& b) { return a.c[axis] < b.c[axis] || (a.c[axis] == b.c[axis] && a.id < b.id); });
        }
        int l = build(first, mid - first, depth + 1);
        int r = build(mid, first + count - mid, depth + 1);
        nodes[idx].left = l;
        nodes[idx].right = r;
        return idx;
    }
};

inline float round_down(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafterf(f, -INFINITY);
    return f;
}
inline float round_up(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafterf(f, INFINITY);
    return f;
}

}  // namespace

int bvh_build(const float* verts, uint32_t n_verts, const uint32_t* tris, uint32_t n_tris, uint32_t max_leaf,
              std::vector<pt_bvh_node>& out_nodes, std::vector<uint32_t>& out_order, uint32_t& out_depth, uint32_t threads) {
    if (max_leaf < 1 || max_leaf > 8) return PT_ERR_INVALID;
    if (n_tris >= (1u << 28)) return PT_ERR_UNSUPPORTED;  // leaf code keeps `first` in 28 bits
    out_nodes.clear();
    out_order.clear();
    out_depth = 0;
    if (n_tris == 0) return PT_OK;
    std::vector<Ref> refs(n_tris);
    uint32_t n_threads = threads ? std::min(64u, threads) : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (n_tris < (1u << 16)) n_threads = 1;
    double max_abs = 0.;
    {
        std::vector<double> pmax(n_threads, 0.);
        std::vector<int> bad(n_threads, 0);
        parallel_chunks(n_tris, n_threads, [&](size_t b0, size_t b1, uint32_t t) {
            double m = 0.;
            for (size_t i = b0; i < b1; ++i) {
                Ref& r = refs[i];
                for (int a = 0; a < 3; ++a) { r.lo[a] = INFINITY; r.hi[a] = -INFINITY; }
                for (int k = 0; k < 3; ++k) {
                    uint32_t v = tris[3 * i + k];
                    if (v >= n_verts) { bad[t] = 1; return; }
                    for (int a = 0; a < 3; ++a) {
                        float x = verts[3 * (size_t)v + a];
                        if (!std::isfinite(x)) { bad[t] = 1; return; }
                        r.lo[a] = std::min(r.lo[a], x);
                        r.hi[a] = std::max(r.hi[a], x);
                        m = std::max(m, (double)std::fabs(x));
                    }
                }
                for (int a = 0; a < 3; ++a) r.c[a] = 0.5f * r.lo[a] + 0.5f * r.hi[a];
                r.id = (uint32_t)i;
            }
            pmax[t] = m;
        });
        for (uint32_t t = 0; t < n_threads; ++t) {
            if (bad[t]) return PT_ERR_INVALID;
            max_abs = std::max(max_abs, pmax[t]);
        }
    }
    Builder b(refs, max_leaf);
    if (n_threads > 1) {
        std::vector<Pending> pending;
        std::vector<Ref> scratch(n_tris);
        b.threads = n_threads;
        b.scratch = &scratch;
        b.cut = std::max(n_tris / (8 * n_threads), 4096u);
        b.pending = &pending;
        b.build(0, n_tris, 0);
        std::vector<Builder> subs;
        subs.reserve(pending.size());
        for (size_t i = 0; i < pending.size(); ++i) subs.emplace_back(refs, max_leaf);
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < pending.size();) {
                subs[i].nodes.reserve((size_t)pending[i].count * 2 / max_leaf + 16);
                subs[i].build(pending[i].first, pending[i].count, pending[i].depth);
            }
        };
        std::vector<std::thread> pool;
        for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
        std::vector<size_t> off(pending.size() + 1, b.nodes.size());
        for (size_t i = 0; i < pending.size(); ++i) off[i + 1] = off[i] + subs[i].nodes.size();
        b.nodes.resize(off[pending.size()]);
        next = 0;
        auto splice = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < pending.size();) {
                const int o = (int)off[i];
                for (size_t k = 0; k < subs[i].nodes.size(); ++k) {
                    TNode n = subs[i].nodes[k];
                    if (n.left >= 0) { n.left += o; n.right += o; }
                    b.nodes[off[i] + k] = n;
                }
                b.nodes[pending[i].node] = b.nodes[off[i]];  // the subtree's root takes the placeholder's place
                std::vector<TNode>().swap(subs[i].nodes);
            }
        };
        pool.clear();
        for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back(splice);
        splice();
        for (auto& t : pool) t.join();
        for (size_t i = 0; i < pending.size(); ++i) b.max_depth = std::max(b.max_depth, subs[i].max_depth);
    } else {
        b.nodes.reserve((size_t)n_tris * 2 / max_leaf + 16);
        b.build(0, n_tris, 0);
    }
    out_order.resize(n_tris);
    for (uint32_t i = 0; i < n_tris; ++i) out_order[i] = refs[i].id;

    const double pad = std::max(max_abs, 1e-30) * (1.0 / 2097152.0);  // 2^-21 of the largest |coordinate|
    auto leaf_code = [](const TNode& n) { return ~(int32_t)((n.first << 3) | (n.count - 1)); };
    auto put_box = [&](pt_bvh_node& dst, int child, const Box& bx) {
        for (int a = 0; a < 3; ++a) {
            dst.bounds[child * 6 + a] = round_down((double)bx.lo[a] - pad);
            dst.bounds[child * 6 + 3 + a] = round_up((double)bx.hi[a] + pad);
        }
    };
    // flatten: inner temp nodes get consecutive output indices in depth-first (pre-)order
    std::vector<int> out_index(b.nodes.size(), -1);
    uint32_t n_inner = 0;
    {
        std::vector<int> dfs{0};
        while (!dfs.empty()) {
            int i = dfs.back();
            dfs.pop_back();
            if (b.nodes[i].left < 0) continue;
            out_index[i] = (int)n_inner++;
            dfs.push_back(b.nodes[i].right);
            dfs.push_back(b.nodes[i].left);
        }
    }
    if (n_inner == 0) {  // the root is a leaf: one node with a single child
        pt_bvh_node node;
        std::memset(&node, 0, sizeof node);
        put_box(node, 0, b.nodes[0].box);
        node.child[0] = leaf_code(b.nodes[0]);
        for (int a = 0; a < 6; ++a) node.bounds[6 + a] = 0.f;
        node.child[1] = PT_BVH_EMPTY;
        out_nodes.push_back(node);
        out_depth = 1;
        return PT_OK;
    }
    out_nodes.resize(n_inner);
    for (size_t i = 0; i < b.nodes.size(); ++i) {
        const TNode& n = b.nodes[i];
        if (n.left < 0 || out_index[i] < 0) continue;  // leaf, or the unused copy of a spliced subtree root
        pt_bvh_node node;
        std::memset(&node, 0, sizeof node);
        const TNode& l = b.nodes[n.left];
        const TNode& r = b.nodes[n.right];
        put_box(node, 0, l.box);
        put_box(node, 1, r.box);
        node.child[0] = l.left >= 0 ? out_index[n.left] : leaf_code(l);
        node.child[1] = r.left >= 0 ? out_index[n.right] : leaf_code(r);
        out_nodes[out_index[i]] = node;
    }
    out_depth = b.max_depth;
    return PT_OK;
}

// A caller's tree (pt_scene_desc::bvh_nodes) holds whatever boxes its builder made - tight ones, as a
// rule.  The traversal kernels test boxes in f32 and rely on every box standing `pad` = 2^-21 of the
// scene's largest |coordinate| clear of its triangles (pt_traverse.hip.h slab32), which is what the
// library's own builder stores; a caller's boxes are therefore widened by that pad here, rounded
// outward.  Widening an already padded tree again costs a few more opened boxes, nothing else.
void bvh_widen(std::vector<pt_bvh_node>& nodes, const float* verts, uint32_t n_verts) {
    double max_abs = 0.;
    for (size_t i = 0; i < (size_t)n_verts * 3; ++i)
        if (std::isfinite(verts[i])) max_abs = std::max(max_abs, (double)std::fabs(verts[i]));
    const double pad = std::max(max_abs, 1e-30) * (1.0 / 2097152.0);
    for (auto& n : nodes)
        for (int c = 0; c < 2; ++c) {
            if (n.child[c] == PT_BVH_EMPTY) continue;
            for (int a = 0; a < 3; ++a) {
                n.bounds[c * 6 + a] = round_down((double)n.bounds[c * 6 + a] - pad);
                n.bounds[c * 6 + 3 + a] = round_up((double)n.bounds[c * 6 + 3 + a] + pad);
            }
        }
}

// ---- four-wide nodes for deep trees ------------------------------------------------------------------
// A ray in a deep tree is a chain of dependent node fetches, each a trip to L2 / HBM.  Collapsing the
// validated two-child tree into nodes of up to four children halves that chain: a node absorbs the
// inner child with the largest box (by half-area) until it has four children.  Leaf references and
// boxes are taken over unchanged, so a ray opens the same leaves.
//
// Stack bound.  Descending into one of k children leaves k - 1 on the stack, so the traversal needs
// max over root-to-leaf paths of sum(k_i - 1) entries.  The kernels hold `budget` entries (16 in LDS +
// 48 in private memory) and the collapse makes that bound a property of the tree: a child subtree of
// two-child height h can always be emitted as it is with h entries, so a node may grow to k children
// only while every child's height fits the entries left, h(child) <= b - (k - 1); the children then
// inherit b - (k - 1).  By induction every subtree needs at most its b; the root's is `budget` and
// bvh_validate has bounded the height by 32.  The actual need is returned for the statistics.
int bvh_collapse4(const std::vector<pt_bvh_node>& nodes2, uint32_t budget, std::vector<pt_bvh_node4>& out, uint32_t& need_out,
                  uint32_t top) {
    out.clear();
    need_out = 0;
    if (nodes2.empty()) return PT_OK;
    const size_t n2 = nodes2.size();
    // two-child height of every inner node (leaves 0), by an explicit post-order walk
    std::vector<uint8_t> h2(n2, 0);
    {
        std::vector<std::pair<int, int>> st;  // (node, next child to visit)
        st.push_back({0, 0});
        while (!st.empty()) {
            auto& [i, c] = st.back();
            if (c < 2) {
                int32_t ch = nodes2[i].child[c++];
                if (ch >= 0 && ch != PT_BVH_EMPTY) st.push_back({ch, 0});
            } else {
                uint8_t h = 0;
                for (int k = 0; k < 2; ++k) {
                    int32_t ch = nodes2[i].child[k];
                    if (ch >= 0 && ch != PT_BVH_EMPTY) h = std::max(h, h2[ch]);
                }
                h2[i] = (uint8_t)(h + 1);
                st.pop_back();
            }
        }
    }
    if (h2[0] > budget) return PT_ERR_UNSUPPORTED;
    struct Kid { int32_t ref; float lo[3], hi[3]; };
    auto height = [&](int32_t ref) -> uint32_t { return (ref >= 0 && ref != PT_BVH_EMPTY) ? h2[ref] : 0u; };
    auto area = [](const Kid& k) {
        float dx = k.hi[0] - k.lo[0], dy = k.hi[1] - k.lo[1], dz = k.hi[2] - k.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto kids_of = [&](int32_t node, Kid* dst) {
        int n = 0;
        for (int c = 0; c < 2; ++c) {
            if (nodes2[node].child[c] == PT_BVH_EMPTY) continue;
            Kid& k = dst[n++];
            k.ref = nodes2[node].child[c];
            for (int a = 0; a < 3; ++a) { k.lo[a] = nodes2[node].bounds[c * 6 + a]; k.hi[a] = nodes2[node].bounds[c * 6 + 3 + a]; }
        }
        return n;
    };
    struct Item { int32_t node2; uint32_t out; uint32_t b; };
    std::vector<Item> work;
    out.reserve(n2 / 2 + 16);
    out.emplace_back();
    work.push_back({0, 0u, budget});
    std::vector<uint32_t> need(1, 0);  // filled bottom-up afterwards
    std::vector<int32_t> parent(1, -1);
    while (!work.empty()) {
        Item it = work.back();
        work.pop_back();
        Kid kid[4];
        int k = kids_of(it.node2, kid);
        while (k < 4) {
            int best = -1;
            for (int c = 0; c < k; ++c) {
                if (kid[c].ref < 0 || kid[c].ref == PT_BVH_EMPTY) continue;  // a leaf stays
                if (best < 0 || area(kid[c]) > area(kid[best])) best = c;
            }
            if (best < 0) break;
            Kid two[2];
            int n_new = kids_of(kid[best].ref, two);
            const int k_new = k - 1 + n_new;
            if (k_new > 4) break;
            // every child of the grown node must still fit the entries left below it
            bool fits = (uint32_t)(k_new - 1) <= it.b;
            const uint32_t left = it.b - (uint32_t)(k_new - 1);
            for (int c = 0; c < k && fits; ++c) if (c != best && height(kid[c].ref) > left) fits = false;
            for (int c = 0; c < n_new && fits; ++c) if (height(two[c].ref) > left) fits = false;
            if (!fits) break;
            kid[best] = two[0];
            if (n_new == 2) kid[k++] = two[1];
        }
        pt_bvh_node4 n4;
        std::memset(&n4, 0, sizeof n4);
        for (int c = 0; c < 4; ++c) {
            n4.child[c] = PT_BVH_EMPTY;
            if (c >= k) continue;
            for (int a = 0; a < 3; ++a) { n4.lo[a][c] = kid[c].lo[a]; n4.hi[a][c] = kid[c].hi[a]; }
            if (kid[c].ref < 0) { n4.child[c] = kid[c].ref; continue; }
            const uint32_t idx = (uint32_t)out.size();
            out.emplace_back();
            parent.push_back((int32_t)it.out);
            n4.child[c] = (int32_t)idx;
            work.push_back({kid[c].ref, idx, it.b - (uint32_t)(k - 1)});
        }
        n4.count = (uint32_t)k;
        out[it.out] = n4;
    }
    // Numbering: the first `top` nodes in breadth-first order (the levels every ray starts with: the kernels keep
    // them in LDS), the subtrees below them depth-first, parent before children, a node's children side by side.
    if (top > 1 && out.size() > 1) {
        const uint32_t n = (uint32_t)out.size();
        std::vector<uint32_t> order;  // order[new] = old
        order.reserve(n);
        std::vector<uint8_t> placed(n, 0);
        order.push_back(0);
        placed[0] = 1;
        for (size_t head = 0; head < order.size() && order.size() < top; ++head)
            for (int c = 0; c < 4 && order.size() < top; ++c) {
                const int32_t ch = out[order[head]].child[c];
                if (ch >= 0 && ch != PT_BVH_EMPTY) { order.push_back((uint32_t)ch); placed[ch] = 1; }
            }
        const size_t n_top = order.size();
        std::vector<uint32_t> st;
        for (size_t t = 0; t < n_top; ++t) {  // the subtrees hanging below the top, in the order of their parents
            for (int c = 3; c >= 0; --c) {
                const int32_t ch = out[order[t]].child[c];
                if (ch >= 0 && ch != PT_BVH_EMPTY && !placed[ch]) st.push_back((uint32_t)ch);
            }
            while (!st.empty()) {
                const uint32_t i = st.back();
                st.pop_back();
                order.push_back(i);
                placed[i] = 1;
                for (int c = 3; c >= 0; --c) {
                    const int32_t ch = out[i].child[c];
                    if (ch >= 0 && ch != PT_BVH_EMPTY) st.push_back((uint32_t)ch);
                }
            }
        }
        std::vector<uint32_t> new_of(n);
        for (uint32_t k = 0; k < n; ++k) new_of[order[k]] = k;
        std::vector<pt_bvh_node4> re(n);
        for (uint32_t k = 0; k < n; ++k) {
            re[k] = out[order[k]];
            for (int c = 0; c < 4; ++c)
                if (re[k].child[c] >= 0 && re[k].child[c] != PT_BVH_EMPTY) re[k].child[c] = (int32_t)new_of[re[k].child[c]];
        }
        out.swap(re);
    }
    // actual stack need: children have larger indices than their parent
    need.assign(out.size(), 0);
    for (size_t i = out.size(); i-- > 0;) {
        uint32_t below = 0;
        for (int c = 0; c < 4; ++c) if (out[i].child[c] >= 0 && out[i].child[c] != PT_BVH_EMPTY) below = std::max(below, need[out[i].child[c]]);
        need[i] = (out[i].count ? out[i].count - 1 : 0) + below;
    }
    need_out = need[0];
    return need_out <= budget ? PT_OK : PT_ERR_UNSUPPORTED;
}

// Walks a BVH (the library's or a caller's) and checks everything the extend kernel assumes:
// child indices in range, no cycles (each inner node reached once), leaf runs inside
// [0, n_tris), every triangle referenced exactly once, depth within the LDS stack.
int bvh_validate(const pt_bvh_node* nodes, uint32_t n_nodes, const uint32_t* order, uint32_t n_tris,
                 uint32_t stack_depth, uint32_t& depth_out, std::string& why) {
    depth_out = 0;
    if (n_tris == 0) return PT_OK;
    if (n_nodes == 0) { why = "BVH has no nodes"; return PT_ERR_INVALID; }
    std::vector<uint8_t> seen_node(n_nodes, 0), seen_tri(n_tris, 0);
    for (uint32_t i = 0; i < n_tris; ++i) {
        if (order[i] >= n_tris) { why = "bvh_tri_order entry out of range"; return PT_ERR_INVALID; }
    }
    struct Item { int node; uint32_t depth; };
    std::vector<Item> st;
    st.push_back({0, 1});
    seen_node[0] = 1;
    uint64_t covered = 0;
    while (!st.empty()) {
        Item it = st.back();
        st.pop_back();
        depth_out = std::max(depth_out, it.depth);
        if (it.depth > stack_depth) { why = "BVH deeper than the traversal stack"; return PT_ERR_UNSUPPORTED; }
        for (int c = 0; c < 2; ++c) {
            int32_t ch = nodes[it.node].child[c];
            if (ch == PT_BVH_EMPTY) continue;
            if (ch >= 0) {
                if ((uint32_t)ch >= n_nodes) { why = "BVH child index out of range"; return PT_ERR_INVALID; }
                if (seen_node[ch]) { why = "BVH node referenced twice"; return PT_ERR_INVALID; }
                seen_node[ch] = 1;
                st.push_back({ch, it.depth + 1});
            } else {
                uint32_t code = (uint32_t)~ch;
                uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
                if ((uint64_t)first + cnt > n_tris) { why = "BVH leaf run out of range"; return PT_ERR_INVALID; }
                for (uint32_t k = first; k < first + cnt; ++k) {
                    if (seen_tri[k]) { why = "triangle slot in two leaves"; return PT_ERR_INVALID; }
                    seen_tri[k] = 1;
                }
                covered += cnt;
            }
        }
    }
    if (covered != n_tris) { why = "BVH leaves do not cover every triangle"; return PT_ERR_INVALID; }
    return PT_OK;
}

}  // namespace pth
