// samplers.cpp — host side of ptk_rt/samplers.hpp (reference src/Pathtracing/sampler.cpp).
// Single-lookup host API + image ingest.  The render loop samples on the GPU
// (image_sample / background_sample in csrc/ptk_device.h) from the pixels stored here.
#include "ptk_rt/samplers.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>

#if defined(__has_include)
#if __has_include(<stb_image.h>)
#define PTK_RT_HAVE_STB 1
#define STB_IMAGE_IMPLEMENTATION
#include <stb_image.h>
#endif
#endif

namespace rt {

image::image() : image(0, 0) {}

image::image(size_t width, size_t height)
    : m_width(static_cast<uint32_t>(width)), m_height(static_cast<uint32_t>(height)), m_pixels(width * height)
{
}

void image::resize(size_t width, size_t height)
{
    m_width = static_cast<uint32_t>(width);
    m_height = static_cast<uint32_t>(height);
    m_pixels.resize(width * height);
}

// sampler.cpp:28-35: column-major walk, one divide at the end.
glm::vec3 image::average() const
{
    glm::vec3 sum(0.0f);
    for (size_t x = 0; x < m_width; ++x)
        for (size_t y = 0; y < m_height; ++y) sum += m_pixels[y * m_width + x];
    return sum / float(m_width * m_height);
}

void image::set_pixel(size_t x, size_t y, const glm::vec3& color) { m_pixels[y * m_width + x] = color; }
glm::vec3 image::get_pixel(size_t x, size_t y) const { return m_pixels[y * m_width + x]; }
glm::vec3 image::get_pixel(const glm::uvec2 xy) const { return m_pixels[size_t(xy.y) * m_width + xy.x]; }

// sampler.cpp:36-61.  No half-texel offset; the interpolation weights use the
// *wrapped* integer taps, exactly like the device's image_sample.
glm::vec3 image::sample(const glm::vec2& uv) const
{
    const glm::vec2 w = glm::fract(uv);
    const float fx = w.x * m_width;
    const float fy = w.y * m_height;
    if (sample_mode == rt::sample_mode::nearest)
        return get_pixel(uint32_t(glm::round(fx)) % m_width, uint32_t(glm::round(fy)) % m_height);
    const uint32_t xa = uint32_t(glm::floor(fx)) % m_width, xb = uint32_t(glm::ceil(fx)) % m_width;
    const uint32_t ya = uint32_t(glm::floor(fy)) % m_height, yb = uint32_t(glm::ceil(fy)) % m_height;
    const float wx = fx - xa, wy = fy - ya;
    const glm::vec3 lo = glm::mix(get_pixel(xa, ya), get_pixel(xb, ya), wx);
    const glm::vec3 hi = glm::mix(get_pixel(xa, yb), get_pixel(xb, yb), wx);
    return glm::mix(lo, hi, wy);
}

namespace {

// Portable Float Map: "PF\n<w> <h>\n<scale>\n" + rows bottom-to-top, 3 floats per pixel;
// negative scale = little endian.  Its row order is the one image::load produces.
bool load_pfm(const std::string& path, uint32_t& w, uint32_t& h, std::vector<glm::vec3>& px)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string magic;
    long lw = 0, lh = 0;
    float scale = 0.0f;
    f >> magic >> lw >> lh >> scale;
    if (!f || magic != "PF" || lw <= 0 || lh <= 0 || scale == 0.0f) return false;
    f.get(); // the single whitespace byte after the header
    px.resize(size_t(lw) * size_t(lh));
    f.read(reinterpret_cast<char*>(px.data()), std::streamsize(px.size() * sizeof(glm::vec3)));
    if (!f) return false;
    if (scale > 0.0f) { // big-endian payload
        unsigned char* b = reinterpret_cast<unsigned char*>(px.data());
        for (size_t i = 0; i < px.size() * 3; ++i) std::reverse(b + i * 4, b + i * 4 + 4);
    }
    w = uint32_t(lw);
    h = uint32_t(lh);
    return true;
}

bool ends_with(std::string_view s, std::string_view suffix)
{
    return s.size() >= suffix.size() && s.substr(s.size() - suffix.size()) == suffix;
}

} // namespace

void image::load(std::string_view file_name)
{
    const std::string path(file_name);
    if (ends_with(file_name, ".pfm")) {
        if (!load_pfm(path, m_width, m_height, m_pixels)) std::fprintf(stderr, "[rt] Can't load image: %s\n", path.c_str());
        return;
    }
#ifdef PTK_RT_HAVE_STB
    int w = 0, h = 0, channels = 0;
    stbi_set_flip_vertically_on_load_thread(true);
    float* data = stbi_loadf(path.c_str(), &w, &h, &channels, 3);
    if (!data) {
        std::fprintf(stderr, "[rt] Can't load image: %s\n", path.c_str());
        return;
    }
    m_width = uint32_t(w);
    m_height = uint32_t(h);
    m_pixels.resize(size_t(w) * size_t(h));
    std::memcpy(m_pixels.data(), data, sizeof(float) * 3 * m_pixels.size());
    stbi_image_free(data);
#else
    std::fprintf(stderr, "[rt] Can't load image: %s (built without stb_image; only .pfm is decoded natively)\n",
                 path.c_str());
#endif
}

// sampler.cpp:98-113: only images whose brightest channel exceeds 1 are compressed.
void image::to_ldr()
{
    float peak = 0.0f;
    for (const glm::vec3& p : m_pixels) peak = std::max({ peak, p.x, p.y, p.z });
    if (!(peak > 1.0f)) return;
    for (glm::vec3& p : m_pixels) p = glm::vec3(1.0f) - glm::exp(-p);
}

// sampler.cpp:116-124
glm::vec3 equirectangular_map::sample(const glm::vec3& uvw) const
{
    const glm::vec3 n = glm::normalize(uvw);
    const float pi = glm::pi<float>();
    return m_image->sample({ std::atan2(n.x, n.z) / (2.0f * pi) + 0.5f, std::asin(n.y) / pi - 0.5f });
}

} // namespace rt
