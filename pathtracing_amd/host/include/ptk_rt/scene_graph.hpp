// ptk_rt/scene_graph.hpp — host mirror of the reference's scene API
// (reference src/Pathtracing/scene.h:12-395): ray, camera, vertex, raycast_result,
// bounding_box, triangle, material, kd_tree_node, object_id, shape, sphere, mesh,
// scene_node, scene — same names, members and signatures.
//
// What differs is behind the interface: a mesh keeps its KD-tree as the flattened
// preorder array the GPU traverses (built by ptk_kd_build, include/ptk.h) and only
// materialises the reference's pointer tree if a caller asks for get_kd_tree().
// intersect()/cast_ray() are the single-ray host API; rendering goes through
// rt::pathtracer::run -> libptk.so (src/device_render.cpp).
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#include <glm/glm.hpp>
#include <glm/ext.hpp>

namespace rt {

class sampler_2d;
class sampler_3d;

enum class axis : uint32_t { X = 0, Y = 1, Z = 2 }; // scene.h:17

struct ray { // scene.h:19-23
    glm::vec3 origin;
    glm::vec3 direction;
};

ray operator*(const glm::mat4& m, const ray& r); // scene.h:25

struct camera { // scene.h:27-37
public:
    glm::vec3 position = { 0.0f, 0.0f, 0.0f };
    const glm::vec3& get_direction() const { return m_direction; }
    void set_direction(const glm::vec3& dir) { m_direction = glm::normalize(dir); }

private:
    glm::vec3 m_direction = { 0.0f, 0.0f, -1.0f };
};

struct vertex { // scene.h:40-45 (32 B; bit-compatible with ptk_vertex)
    glm::vec3 position = { 0.0f, 0.0f, 0.0f };
    glm::vec3 normal = { 0.0f, 0.0f, 0.0f };
    glm::vec2 uv = { 0.0f, 0.0f };
};

struct raycast_result { // scene.h:47-53
    bool hit = false;
    glm::vec3 position;
    glm::vec3 normal;
    glm::vec2 uv;
};

class bounding_box { // scene.h:58-90
public:
    glm::vec3 min, max;
    bounding_box() : min(), max() {}
    bounding_box(const glm::vec3& min, const glm::vec3& max) : min(min), max(max) {}
    void split(axis axis, float value, bounding_box& left, bounding_box& right) const;
    float surface() const;
    bool intersect(const ray& ray) const;
};

class triangle { // scene.h:95-127
public:
    std::array<vertex, 3> vertices;
    const glm::vec3& get_face_normal() const { return m_face_normal; }
    glm::vec3 baricentric(const glm::vec3& point) const;
    void update();

private:
    std::array<glm::vec3, 3> m_edges;
    glm::vec3 m_face_normal;
    float m_d00, m_d01, m_d11;
    float m_inv_den;
    friend class mesh;
};

struct material { // scene.h:133-140
    material();
    std::shared_ptr<sampler_2d> albedo;
    std::shared_ptr<sampler_2d> emission;
    std::shared_ptr<sampler_2d> roughness;
    std::shared_ptr<sampler_2d> metallic;
};

struct mesh_accel; // flattened KD-tree (src/scene_graph.cpp)

class kd_tree_node { // scene.h:145-192
public:
    kd_tree_node() {}
    kd_tree_node(const std::vector<triangle>& triangles, const bounding_box& bounds, uint32_t depth);
    const uint32_t get_max_depth() const;
    const std::vector<triangle>& get_triangles() const { return m_triangles; }
    const bounding_box& get_bounds() const { return m_bounds; }
    const std::unique_ptr<kd_tree_node>& get_left() const { return m_left; }
    const std::unique_ptr<kd_tree_node>& get_right() const { return m_right; }

private:
    uint32_t m_depth = 0;
    bounding_box m_bounds;
    std::vector<triangle> m_triangles;
    std::unique_ptr<kd_tree_node> m_left = nullptr;
    std::unique_ptr<kd_tree_node> m_right = nullptr;
    void assign_from_flat(const mesh_accel& acc, const std::vector<triangle>& tris, size_t at);
    friend class mesh;
};

class object_id { // scene.h:194-201
public:
    const size_t id;
    object_id() : id(s_next++) {}

private:
    static size_t s_next;
};

class shape { // scene.h:206-226
public:
    virtual ~shape() = default;
    virtual void compile() = 0;
    virtual raycast_result intersect(const ray& ray) const = 0;
    virtual const bounding_box& get_bounds() const = 0;
};

class sphere : public shape { // scene.h:231-239
public:
    void compile() override {}
    raycast_result intersect(const ray& ray) const override;
    const bounding_box& get_bounds() const override { return m_Bounds; }

private:
    bounding_box m_Bounds = bounding_box({ -1, -1, -1 }, { 1, 1, 1 });
};

class mesh : public shape, public object_id { // scene.h:244-279
public:
    mesh();
    ~mesh() override;

    const bounding_box& get_bounds() const override { return m_bounds; }
    const std::vector<triangle>& get_triangles() const { return m_triangles; }
    triangle& add_triangle();
    raycast_result intersect(const ray& ray) const override;
    void compile() override;
    const std::unique_ptr<kd_tree_node>& get_kd_tree() const;

private:
    raycast_result intersect_triangle(const ray& ray, const triangle& triangle) const;
    const mesh_accel& accel() const;

    bounding_box m_bounds;
    std::vector<triangle> m_triangles;
    mutable std::mutex m_lazy;
    mutable std::unique_ptr<mesh_accel> m_accel;  // built on first intersect()/get_kd_tree()
    mutable std::unique_ptr<kd_tree_node> m_tree; // pointer-tree view, built on first get_kd_tree()
};

class scene_node { // scene.h:284-350
private:
    glm::mat4 m_transform = glm::identity<glm::mat4>();
    glm::mat4 m_inv_transform = glm::identity<glm::mat4>();
    glm::mat4 m_normal_transform = glm::identity<glm::mat4>();
    void update_matrices();

public:
    rt::material material;
    std::shared_ptr<rt::shape> shape = nullptr;

    void load_identity();
    void translate(const glm::vec3& t);
    void rotate(const glm::vec3& axis, float angle);
    void scale(const glm::vec3& s);
    void multiply(const glm::mat4& mat);
    const glm::mat4& get_transform() const { return m_transform; }
    const glm::mat4& get_inverse_transform() const { return m_inv_transform; }
    const glm::mat4& get_normal_transform() const { return m_normal_transform; }
};

class scene { // scene.h:355-394
public:
    scene();

    rt::camera camera;
    std::shared_ptr<sampler_3d> background = nullptr;
    std::vector<std::shared_ptr<scene_node>> nodes;

    std::tuple<raycast_result, std::shared_ptr<scene_node>> cast_ray(
        const ray& ray, bool return_on_first_hit = false,
        const std::vector<std::shared_ptr<scene_node>>& avoid_nodes = {}) const;

    const std::vector<std::shared_ptr<scene_node>>& get_light_sources() const { return m_light_sources; }
    void compile();

private:
    std::vector<std::shared_ptr<scene_node>> m_light_sources;
};

} // namespace rt
