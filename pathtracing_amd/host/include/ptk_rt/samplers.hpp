// ptk_rt/samplers.hpp — host mirror of the reference's sampler API
// (reference src/Pathtracing/sampler.h:15-156): same class names, members and
// signatures, so callers written against the reference compile unchanged.
//
// These objects are the *inputs* of the GPU path: the device renderer reads their
// pixels / colours when it flattens a scene (src/device_render.cpp).  The sample()
// methods are the single-lookup host API the reference exposes (used by callers such
// as debug tracers); the render loop does not call them.
#pragma once
#include <cstdint>
#include <memory>
#include <string_view>
#include <vector>

#include <glm/glm.hpp>

namespace rt {

class sampler_2d { // sampler.h:15-29
public:
    virtual ~sampler_2d() = default;
    virtual glm::vec3 sample(const glm::vec2& uv) const = 0;
    virtual glm::vec3 average() const = 0;
};

class sampler_3d { // sampler.h:34-43
public:
    virtual ~sampler_3d() = default;
    virtual glm::vec3 sample(const glm::vec3& uvw) const = 0;
};

class color_sampler : public sampler_2d, public sampler_3d { // sampler.h:48-63
public:
    color_sampler(const glm::vec3& color) : m_color(color) {}
    glm::vec3 sample(const glm::vec2&) const override { return m_color; }
    glm::vec3 sample(const glm::vec3&) const override { return m_color; }
    glm::vec3 average() const override { return m_color; }

private:
    glm::vec3 m_color;
};

enum class sample_mode { linear, nearest }; // sampler.h:65-68

class image : public sampler_2d { // sampler.h:73-140
public:
    rt::sample_mode sample_mode = rt::sample_mode::linear;

    image();
    image(size_t width, size_t height);
    void resize(size_t width, size_t height);

    glm::vec3 average() const override;
    glm::vec3 sample(const glm::vec2& uv) const override;

    void set_pixel(size_t x, size_t y, const glm::vec3& color);
    glm::vec3 get_pixel(size_t x, size_t y) const;
    glm::vec3 get_pixel(const glm::uvec2 xy) const;
    size_t get_width() const { return m_width; }
    size_t get_height() const { return m_height; }

    // sampler.cpp:78-96.  Decodes through stb_image when <stb_image.h> is on the include
    // path at build time (the reference vendors it); Portable Float Maps (.pfm) are read
    // natively.  Rows are stored bottom-up like stbi's flipped load.
    void load(std::string_view file_name);
    void to_ldr();

    // Additive (not in the reference): contiguous RGB floats, row-major, for bulk transfer.
    const glm::vec3* data() const { return m_pixels.data(); }
    glm::vec3* data() { return m_pixels.data(); }

private:
    uint32_t m_width, m_height;
    std::vector<glm::vec3> m_pixels;
};

class equirectangular_map : public sampler_3d { // sampler.h:145-156
public:
    equirectangular_map(const std::shared_ptr<image>& image) : m_image(image) {}
    glm::vec3 sample(const glm::vec3& uvw) const override;
    // Additive: the wrapped image (the device renderer uploads it as a texture).
    const std::shared_ptr<image>& get_image() const { return m_image; }

private:
    std::shared_ptr<image> m_image;
};

} // namespace rt
