// scene_graph.cpp — host side of ptk_rt/scene_graph.hpp (reference
// src/Pathtracing/scene.cpp).  Scene construction, node transforms and the single-ray
// query API (shape::intersect, scene::cast_ray).  Rendering does not pass through here:
// rt::pathtracer::run flattens the scene and hands it to libptk.so (device_render.cpp).
//
// A mesh's KD-tree is the flattened preorder array produced by ptk_kd_build /
// ptk_kd_build_bounded (include/ptk.h, csrc/kd_build.cpp) — the same builder the GPU
// upload uses — and the reference's pointer tree is only a view derived from it.
#include "ptk_rt/scene_graph.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "ptk.h"
#include "ptk_rt/samplers.hpp"

namespace rt {

size_t object_id::s_next = 0;

// scene.cpp:18-23: the direction is re-normalised, so local t != world t.
ray operator*(const glm::mat4& m, const ray& r)
{
    return { glm::vec3(m * glm::vec4(r.origin, 1.0f)), glm::normalize(glm::vec3(m * glm::vec4(r.direction, 0.0f))) };
}

// ---------------------------------------------------------------- bounding_box

float bounding_box::surface() const // scene.cpp:25-28
{
    const float dx = max.x - min.x, dy = max.y - min.y, dz = max.z - min.z;
    return 2 * (dx * dy + dx * dz + dz * dy);
}

void bounding_box::split(axis ax, float value, bounding_box& left, bounding_box& right) const // scene.cpp:31-56
{
    const uint32_t k = static_cast<uint32_t>(ax);
    if (k > 2) return;
    left = *this;
    right = *this;
    left.max[int(k)] = value;
    right.min[int(k)] = value;
}

// scene.cpp:58-75, quirks included: six IEEE divides (a zero direction component gives
// +-inf / NaN), NaN-dropping fmin/fmax, and `!= -1` as the hit test.
bool bounding_box::intersect(const ray& r) const
{
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) {
        const float a = (min[k] - r.origin[k]) / r.direction[k];
        const float b = (max[k] - r.origin[k]) / r.direction[k];
        lo[k] = std::fmin(a, b);
        hi[k] = std::fmax(a, b);
    }
    const float t_in = std::fmax(std::fmax(lo[0], lo[1]), lo[2]);
    const float t_out = std::fmin(std::fmin(hi[0], hi[1]), hi[2]);
    const float t = (t_out < 0 || t_in > t_out) ? -1.0f : t_in;
    return t != -1.0f;
}

// ---------------------------------------------------------------- triangle

void triangle::update() // scene.cpp:91-107
{
    const glm::vec3 &a = vertices[0].position, &b = vertices[1].position, &c = vertices[2].position;
    m_edges = { b - a, c - a, c - b };
    m_face_normal = glm::normalize(glm::cross(b - a, c - b));
    m_d00 = glm::dot(m_edges[0], m_edges[0]);
    m_d01 = glm::dot(m_edges[0], m_edges[1]);
    m_d11 = glm::dot(m_edges[1], m_edges[1]);
    m_inv_den = 1.0f / (m_d00 * m_d11 - m_d01 * m_d01);
}

glm::vec3 triangle::baricentric(const glm::vec3& point) const // scene.cpp:78-89
{
    const glm::vec3 rel = point - vertices[0].position;
    const float p0 = glm::dot(rel, m_edges[0]);
    const float p1 = glm::dot(rel, m_edges[1]);
    const float v = (m_d11 * p0 - m_d01 * p1) * m_inv_den;
    const float w = (m_d00 * p1 - m_d01 * p0) * m_inv_den;
    return { 1.0f - v - w, v, w };
}

// ---------------------------------------------------------------- material

material::material() // scene.cpp:443-449
    : albedo(std::make_shared<color_sampler>(glm::vec3(1.0f))),
      emission(std::make_shared<color_sampler>(glm::vec3(0.0f))),
      roughness(std::make_shared<color_sampler>(glm::vec3(1.0f))),
      metallic(std::make_shared<color_sampler>(glm::vec3(0.0f)))
{
}

// ---------------------------------------------------------------- flattened KD-tree

struct mesh_accel {
    std::vector<ptk_kd_node> nodes; // DFS preorder with skip links
    std::vector<uint32_t> prims;    // leaf-ordered triangle indices
};

namespace {

static_assert(sizeof(vertex) == sizeof(ptk_vertex), "rt::vertex must match ptk_vertex");

std::vector<ptk_triangle> pack_triangles(const std::vector<triangle>& tris)
{
    std::vector<ptk_triangle> out(tris.size());
    for (size_t i = 0; i < tris.size(); ++i) std::memcpy(&out[i], tris[i].vertices.data(), sizeof(ptk_triangle));
    return out;
}

void build_flat(const std::vector<triangle>& tris, const bounding_box& bounds, uint32_t depth, mesh_accel& acc)
{
    const std::vector<ptk_triangle> packed = pack_triangles(tris);
    const float b[6] = { bounds.min.x, bounds.min.y, bounds.min.z, bounds.max.x, bounds.max.y, bounds.max.z };
    ptk_kd_info info{};
    if (ptk_kd_build_bounded(packed.data(), packed.size(), b, depth, &info, nullptr, 0, nullptr, 0) != PTK_OK) return;
    acc.nodes.resize(info.num_nodes);
    acc.prims.resize(info.num_leaf_prims);
    ptk_kd_build_bounded(packed.data(), packed.size(), b, depth, &info, acc.nodes.data(), acc.nodes.size(),
                         acc.prims.data(), acc.prims.size());
}

} // namespace

kd_tree_node::kd_tree_node(const std::vector<triangle>& triangles, const bounding_box& bounds, uint32_t depth)
{
    // scene.cpp:214-286 through the shared flattener, then unfolded into this node.
    mesh_accel acc;
    build_flat(triangles, bounds, depth, acc);
    if (!acc.nodes.empty()) assign_from_flat(acc, triangles, 0);
}

// Pointer-tree view of the flattened subtree rooted at slot `at`: the first child is the next preorder slot, a
// second child starts where the first one's skip link points.  A lone child (scene.cpp:280-284 makes a child only
// for a non-empty side) is the left one if it kept the parent's min on the split axis, else the right one
// (bounding_box::split, scene.cpp:31-56).
void kd_tree_node::assign_from_flat(const mesh_accel& acc, const std::vector<triangle>& tris, size_t at)
{
    const ptk_kd_node& n = acc.nodes[at];
    m_depth = n.depth;
    m_bounds = bounding_box({ n.bmin[0], n.bmin[1], n.bmin[2] }, { n.bmax[0], n.bmax[1], n.bmax[2] });
    m_triangles.clear();
    m_triangles.reserve(n.num_prims);
    for (uint32_t k = 0; k < n.num_prims; ++k) m_triangles.push_back(tris[acc.prims[n.first_prim + k]]);
    m_left.reset();
    m_right.reset();
    const size_t first = at + 1;
    if (n.num_prims != 0 || first >= n.skip) return; // leaf
    const size_t second = acc.nodes[first].skip;
    const uint32_t axis = n.depth % 3u;
    const bool lone_right = second >= n.skip && std::memcmp(&acc.nodes[first].bmin[axis], &n.bmin[axis], sizeof(float)) != 0;
    std::unique_ptr<kd_tree_node>& slot = lone_right ? m_right : m_left;
    slot = std::make_unique<kd_tree_node>();
    slot->assign_from_flat(acc, tris, first);
    if (second < n.skip) {
        m_right = std::make_unique<kd_tree_node>();
        m_right->assign_from_flat(acc, tris, second);
    }
}

// scene.cpp:288-293 is never called by the reference and walks the left child twice; this
// returns the depth of the deepest node below (what its name promises).
const uint32_t kd_tree_node::get_max_depth() const
{
    uint32_t d = m_depth;
    if (m_left) d = std::max(d, m_left->get_max_depth());
    if (m_right) d = std::max(d, m_right->get_max_depth());
    return d;
}

// ---------------------------------------------------------------- sphere

raycast_result sphere::intersect(const ray& r) const // scene.cpp:401-441
{
    raycast_result out;
    const float along = glm::dot(glm::vec3(0.0f, 0.0f, 0.0f) - r.origin, r.direction);
    const float miss2 = glm::dot(r.origin, r.origin) - along * along;
    if (miss2 > 1.0f) return out;
    const float half_chord = std::sqrt(1.0f - miss2);
    const float t_near = along - half_chord, t_far = along + half_chord;
    if (t_near < 0 && t_far < 0) return out;
    out.hit = true;
    out.position = r.origin + r.direction * (t_near >= 0.0f ? t_near : t_far);
    out.normal = glm::normalize(out.position);
    out.uv = { std::atan2(out.normal.x, out.normal.z) / glm::pi<float>() + 0.5f, out.normal.y * 0.5f + 0.5f };
    return out;
}

// ---------------------------------------------------------------- mesh

mesh::mesh() = default;
mesh::~mesh() = default;

triangle& mesh::add_triangle() { return m_triangles.emplace_back(); }

// scene.cpp:123-140.  Bounds keep growing from whatever they were (initially the
// (0,0,0)-(0,0,0) box of scene.h:67), so the origin is always inside.  The tree itself
// is built on first use: run() does not need a host tree (ptk_scene_upload builds the
// device copy), so the reference's compile-on-every-run (abstract_pathtracer.cpp:13)
// costs nothing here unless a caller actually queries the mesh on the host.
void mesh::compile()
{
    for (triangle& t : m_triangles) {
        t.update();
        for (const vertex& v : t.vertices) {
            m_bounds.min = glm::min(m_bounds.min, v.position);
            m_bounds.max = glm::max(m_bounds.max, v.position);
        }
    }
    std::lock_guard<std::mutex> g(m_lazy);
    m_accel.reset();
    m_tree.reset();
}

const mesh_accel& mesh::accel() const
{
    std::lock_guard<std::mutex> g(m_lazy);
    if (!m_accel) {
        m_accel = std::make_unique<mesh_accel>();
        build_flat(m_triangles, m_bounds, 0, *m_accel);
    }
    return *m_accel;
}

const std::unique_ptr<kd_tree_node>& mesh::get_kd_tree() const
{
    const mesh_accel& acc = accel();
    std::lock_guard<std::mutex> g(m_lazy);
    if (!m_tree && !acc.nodes.empty()) {
        m_tree = std::make_unique<kd_tree_node>();
        m_tree->assign_from_flat(acc, m_triangles, 0);
    }
    return m_tree;
}

raycast_result mesh::intersect_triangle(const ray& r, const triangle& t) const // scene.cpp:142-186
{
    raycast_result out;
    const glm::vec3& N = t.m_face_normal;
    const float height = glm::dot(r.origin - t.vertices[0].position, N);
    if (height < 0) return out; // one-sided: origin behind the plane
    const float facing = glm::dot(r.direction, N);
    if (facing >= 0) return out; // parallel or leaving
    const glm::vec3 p = r.origin + r.direction * (height / -facing);
    const glm::vec3 bar = t.baricentric(p);
    if (bar.x >= 0 && bar.y >= 0 && bar.z >= 0) {
        out.hit = true;
        out.position = p;
        out.normal = glm::normalize(t.vertices[0].normal * bar.x + t.vertices[1].normal * bar.y +
                                    t.vertices[2].normal * bar.z);
        out.uv = bar.x * t.vertices[0].uv + bar.y * t.vertices[1].uv + bar.z * t.vertices[2].uv;
    }
    return out;
}

// scene.cpp:115-121,188-212 on the flattened tree: "box hit -> own triangles, then left,
// then right" is "hit or leaf -> next slot, interior miss -> skip link"; same visiting
// order, same strict `<`, so ties resolve to the same triangle.
raycast_result mesh::intersect(const ray& r) const
{
    raycast_result best;
    if (m_triangles.empty()) return best;
    const mesh_accel& acc = accel();
    float best_d2 = std::numeric_limits<float>::max();
    const size_t count = acc.nodes.size();
    for (size_t i = 0; i < count;) {
        const ptk_kd_node& n = acc.nodes[i];
        const bounding_box box({ n.bmin[0], n.bmin[1], n.bmin[2] }, { n.bmax[0], n.bmax[1], n.bmax[2] });
        const bool inside = box.intersect(r);
        if (inside)
            for (uint32_t k = 0; k < n.num_prims; ++k) {
                const raycast_result c = intersect_triangle(r, m_triangles[acc.prims[n.first_prim + k]]);
                if (!c.hit) continue;
                const float d2 = glm::length2(r.origin - c.position);
                if (d2 < best_d2) {
                    best_d2 = d2;
                    best = c;
                }
            }
        i = (inside || n.num_prims != 0) ? i + 1 : n.skip;
    }
    return best;
}

// ---------------------------------------------------------------- scene_node

void scene_node::update_matrices() // scene.cpp:295-299
{
    m_inv_transform = glm::inverse(m_transform);
    m_normal_transform = glm::transpose(m_inv_transform);
}

// scene.cpp:301-329: every operation post-multiplies the current transform.
void scene_node::load_identity()
{
    m_transform = glm::identity<glm::mat4>();
    update_matrices();
}
void scene_node::multiply(const glm::mat4& mat)
{
    m_transform *= mat;
    update_matrices();
}
void scene_node::translate(const glm::vec3& t) { multiply(glm::translate(t)); }
void scene_node::rotate(const glm::vec3& axis, float angle) { multiply(glm::rotate(angle, axis)); }
void scene_node::scale(const glm::vec3& s) { multiply(glm::scale(s)); }

// ---------------------------------------------------------------- scene

scene::scene() : background(std::make_shared<color_sampler>(glm::vec3{ 0.0f, 0.0f, 0.0f })) {} // scene.cpp:350-353

void scene::compile() // scene.cpp:331-348
{
    m_light_sources.clear();
    for (const std::shared_ptr<scene_node>& n : nodes) {
        if (n->shape) n->shape->compile();
    }
    for (const std::shared_ptr<scene_node>& n : nodes) {
        const glm::vec3 e = n->material.emission->average();
        if (e.r + e.g + e.b > 0.0f) m_light_sources.push_back(n);
    }
}

// scene.cpp:355-399: every node is tested (there is no top-level structure); the closest
// hit by squared world distance wins, the first node on exact ties.
std::tuple<raycast_result, std::shared_ptr<scene_node>> scene::cast_ray(
    const ray& r, bool return_on_first_hit, const std::vector<std::shared_ptr<scene_node>>& avoid_nodes) const
{
    raycast_result best;
    std::shared_ptr<scene_node> best_node;
    float best_d2 = std::numeric_limits<float>::max();
    for (const std::shared_ptr<scene_node>& node : nodes) {
        if (!avoid_nodes.empty() && std::find(avoid_nodes.begin(), avoid_nodes.end(), node) != avoid_nodes.end()) continue;
        raycast_result c = node->shape->intersect(node->get_inverse_transform() * r);
        if (!c.hit) continue;
        c.position = glm::vec3(node->get_transform() * glm::vec4(c.position, 1.0f));
        c.normal = glm::normalize(glm::vec3(node->get_normal_transform() * glm::vec4(c.normal, 0.0f)));
        if (return_on_first_hit) return { c, node };
        const float d2 = glm::length2(c.position - r.origin);
        if (d2 < best_d2) {
            best_d2 = d2;
            best = c;
            best_node = node;
        }
    }
    return { best, std::move(best_node) };
}

} // namespace rt
