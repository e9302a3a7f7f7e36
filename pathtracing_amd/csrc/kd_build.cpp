// kd_build.cpp — host flattener: builds a mesh's KD-tree with the reference's
// topology and emits it directly in DFS preorder with skip links.
//
// Follows kd_tree_node::kd_tree_node (reference src/Pathtracing/scene.cpp:214-286),
// bounding_box::split (scene.cpp:31-56) and the bounds loop of mesh::compile
// (scene.cpp:123-137).  Differences in representation only: leaves keep triangle
// indices (the reference copies triangles by value, scene.cpp:222-223,259,264),
// interior nodes keep nothing (scene.cpp:280-284), and the pointer tree is replaced
// by preorder positions: node i's left child is i+1, `skip` is the first node after
// its subtree, which is all mesh::intersect_internal's recursion (scene.cpp:188-212)
// needs.  Host-only code, no GPU calls.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/ptk.h"

namespace {

struct Builder {
    const ptk_triangle* tris;
    std::vector<ptk_kd_node> nodes;
    std::vector<uint32_t> prims;
    uint32_t leaves = 0;
    uint32_t max_depth = 0;
};

void emit_leaf(Builder& b, size_t me, const std::vector<uint32_t>& idx)
{
    b.nodes[me].first_prim = static_cast<uint32_t>(b.prims.size());
    b.nodes[me].num_prims = static_cast<uint32_t>(idx.size());
    b.prims.insert(b.prims.end(), idx.begin(), idx.end());
    b.leaves++;
}

void build(Builder& b, std::vector<uint32_t>& idx, const float* bmin, const float* bmax, uint32_t depth)
{
    const size_t me = b.nodes.size();
    b.nodes.emplace_back();
    {
        ptk_kd_node& n = b.nodes[me];
        std::memcpy(n.bmin, bmin, sizeof(float) * 3);
        std::memcpy(n.bmax, bmax, sizeof(float) * 3);
        n.first_prim = static_cast<uint32_t>(b.prims.size());
        n.num_prims = 0;
        n.depth = depth;
    }
    if (depth > b.max_depth) b.max_depth = depth;

    const size_t n = idx.size();
    // Stop condition, scene.cpp:220
    if (n <= 1 || depth == PTK_KD_MAX_DEPTH) {
        emit_leaf(b, me, idx);
        b.nodes[me].skip = static_cast<uint32_t>(b.nodes.size());
        return;
    }

    // Round-robin axis, scene.cpp:230; split = fp32 running mean of all 3n vertex
    // coordinates in triangle order, scene.cpp:240-249 (sequential on purpose).
    const uint32_t ax = depth % 3;
    float median = 0;
    for (size_t i = 0; i < n; ++i) {
        const ptk_triangle& t = b.tris[idx[i]];
        median += t.v[0].position[ax];
        median += t.v[1].position[ax];
        median += t.v[2].position[ax];
    }
    median /= static_cast<float>(3 * n);

    float lmin[3], lmax[3], rmin[3], rmax[3];
    std::memcpy(lmin, bmin, 12);
    std::memcpy(lmax, bmax, 12);
    std::memcpy(rmin, bmin, 12);
    std::memcpy(rmax, bmax, 12);
    lmax[ax] = median; // scene.cpp:35-52
    rmin[ax] = median;

    std::vector<uint32_t> left, right;
    left.reserve(n);
    right.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        const ptk_triangle& t = b.tris[idx[i]];
        const float p0 = t.v[0].position[ax], p1 = t.v[1].position[ax], p2 = t.v[2].position[ax];
        if (p0 <= median || p1 <= median || p2 <= median) left.push_back(idx[i]);  // scene.cpp:257
        if (p0 >= median || p1 >= median || p2 >= median) right.push_back(idx[i]); // scene.cpp:262
    }

    // "too many in common" rule, scene.cpp:271 (compared in double like the reference)
    if (static_cast<double>(left.size() + right.size()) > 1.5 * static_cast<double>(n)) {
        emit_leaf(b, me, idx);
    } else {
        std::vector<uint32_t>().swap(idx); // parent's list is no longer needed
        if (!left.empty()) build(b, left, lmin, lmax, depth + 1);
        std::vector<uint32_t>().swap(left);
        if (!right.empty()) build(b, right, rmin, rmax, depth + 1);
    }
    b.nodes[me].skip = static_cast<uint32_t>(b.nodes.size());
}

// Triangle boxes for the pruned traversal (not in the reference): a leaf's box covers its
// whole triangles (which may stick out of the leaf's cut box), an interior node's box is
// the union of its children's.  Children follow their parent in preorder, so one reverse
// sweep fills every node.  Boxes are padded so that a hit position computed in fp32
// (rounded, and accepted by `bar >= 0` with rounded barycentrics) is always inside.
void fill_triangle_boxes(Builder& b)
{
    const float big = 3.402823466e+38f;
    for (size_t i = b.nodes.size(); i-- > 0;) {
        ptk_kd_node& n = b.nodes[i];
        float lo[3] = { big, big, big }, hi[3] = { -big, -big, -big };
        if (n.num_prims != 0 || i + 1 >= n.skip) {
            for (uint32_t k = 0; k < n.num_prims; ++k) {
                const ptk_triangle& t = b.tris[b.prims[n.first_prim + k]];
                for (int v = 0; v < 3; ++v)
                    for (int a = 0; a < 3; ++a) {
                        const float p = t.v[v].position[a];
                        if (p < lo[a]) lo[a] = p;
                        if (p > hi[a]) hi[a] = p;
                    }
            }
        } else {
            for (size_t c = i + 1; c < n.skip; c = b.nodes[c].skip)
                for (int a = 0; a < 3; ++a) {
                    if (b.nodes[c].tmin[a] < lo[a]) lo[a] = b.nodes[c].tmin[a];
                    if (b.nodes[c].tmax[a] > hi[a]) hi[a] = b.nodes[c].tmax[a];
                }
        }
        std::memcpy(n.tmin, lo, 12);
        std::memcpy(n.tmax, hi, 12);
    }
    for (ptk_kd_node& n : b.nodes)
        for (int a = 0; a < 3; ++a) {
            if (!(n.tmin[a] <= n.tmax[a])) continue; // empty leaf or NaN vertices: never pruned by this axis... kept as is
            const float ext = n.tmax[a] - n.tmin[a];
            const float mag = std::fmax(std::fabs(n.tmin[a]), std::fabs(n.tmax[a]));
            const float pad = 1e-4f * ext + 1e-5f * mag + 1e-30f;
            n.tmin[a] -= pad;
            n.tmax[a] += pad;
        }
}

} // namespace

// Internal C++ entries used by ptk.hip as well.
void ptk_kd_build_vectors_bounded(const ptk_triangle* triangles, uint64_t num_triangles, const float* bmin,
                                  const float* bmax, uint32_t root_depth, ptk_kd_info* info,
                                  std::vector<ptk_kd_node>& nodes, std::vector<uint32_t>& prims)
{
    Builder b;
    b.tris = triangles;
    std::vector<uint32_t> idx(num_triangles);
    for (uint64_t i = 0; i < num_triangles; ++i) idx[i] = static_cast<uint32_t>(i);
    build(b, idx, bmin, bmax, root_depth);
    fill_triangle_boxes(b);
    if (info) {
        info->num_nodes = static_cast<uint32_t>(b.nodes.size());
        info->num_leaves = b.leaves;
        info->num_leaf_prims = b.prims.size();
        info->max_depth = b.max_depth;
        std::memcpy(info->bmin, bmin, 12);
        std::memcpy(info->bmax, bmax, 12);
    }
    nodes.swap(b.nodes);
    prims.swap(b.prims);
}

void ptk_kd_build_vectors(const ptk_triangle* triangles, uint64_t num_triangles, ptk_kd_info* info,
                          std::vector<ptk_kd_node>& nodes, std::vector<uint32_t>& prims)
{
    // mesh::compile bounds: grown from the default-constructed (0,0,0)-(0,0,0) box
    // (scene.h:67, scene.cpp:125-136), so the origin is always inside.
    float bmin[3] = { 0.0f, 0.0f, 0.0f }, bmax[3] = { 0.0f, 0.0f, 0.0f };
    for (uint64_t i = 0; i < num_triangles; ++i)
        for (int j = 0; j < 3; ++j)
            for (int k = 0; k < 3; ++k) {
                const float p = triangles[i].v[j].position[k];
                bmin[k] = (p < bmin[k]) ? p : bmin[k];
                bmax[k] = (bmax[k] < p) ? p : bmax[k];
            }
    ptk_kd_build_vectors_bounded(triangles, num_triangles, bmin, bmax, 0, info, nodes, prims);
}

namespace {
int copy_out(const ptk_kd_info& local, const std::vector<ptk_kd_node>& n, const std::vector<uint32_t>& p,
             ptk_kd_info* info, ptk_kd_node* nodes, uint64_t nodes_cap, uint32_t* prims, uint64_t prims_cap)
{
    if (info) *info = local;
    if (nodes) {
        if (nodes_cap < n.size()) return PTK_ERR_INVALID;
        std::memcpy(nodes, n.data(), n.size() * sizeof(ptk_kd_node));
    }
    if (prims) {
        if (prims_cap < p.size()) return PTK_ERR_INVALID;
        std::memcpy(prims, p.data(), p.size() * sizeof(uint32_t));
    }
    return PTK_OK;
}
} // namespace

extern "C" int ptk_kd_build(const ptk_triangle* triangles, uint64_t num_triangles, ptk_kd_info* info,
                            ptk_kd_node* nodes, uint64_t nodes_cap, uint32_t* prims, uint64_t prims_cap)
{
    if (!triangles && num_triangles) return PTK_ERR_INVALID;
    if (num_triangles > 0xFFFFFFFFull / 4) return PTK_ERR_INVALID;
    std::vector<ptk_kd_node> n;
    std::vector<uint32_t> p;
    ptk_kd_info local;
    ptk_kd_build_vectors(triangles, num_triangles, &local, n, p);
    return copy_out(local, n, p, info, nodes, nodes_cap, prims, prims_cap);
}

extern "C" int ptk_kd_build_bounded(const ptk_triangle* triangles, uint64_t num_triangles, const float bounds[6],
                                    uint32_t root_depth, ptk_kd_info* info, ptk_kd_node* nodes, uint64_t nodes_cap,
                                    uint32_t* prims, uint64_t prims_cap)
{
    if ((!triangles && num_triangles) || !bounds) return PTK_ERR_INVALID;
    if (num_triangles > 0xFFFFFFFFull / 4 || root_depth > PTK_KD_MAX_DEPTH) return PTK_ERR_INVALID;
    std::vector<ptk_kd_node> n;
    std::vector<uint32_t> p;
    ptk_kd_info local;
    ptk_kd_build_vectors_bounded(triangles, num_triangles, bounds, bounds + 3, root_depth, &local, n, p);
    return copy_out(local, n, p, info, nodes, nodes_cap, prims, prims_cap);
}
