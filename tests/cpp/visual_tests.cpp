// visual_tests.cpp — the reference's golden tests, written against the C++ host mirror (include/pfx.hpp) the way the
// reference writes them against its Rust types:
//   tests/visual_blend.rs:19-136   make_blend_test / blend_test! x25 / normal_half_opacity / hidden_layer_invisible
//   tests/visual_filters.rs:30-62,90-94,143-147,291  gaussian / box / median / pixelate goldens + sigma=0 identity
//   tests/gpu_pipelines.rs         GpuRenderer filter methods (here held to the CPU-path goldens, not loose bounds)
// Usage: visual_tests <golden.bin>   (golden.bin = records written by tests/test_gpu_cpp_host.py: name\0 w h rgba)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <string>

#include "../../include/pfx.hpp"

using namespace pfx;

static std::map<std::string, RgbaImage> g_golden;
static int g_failed = 0, g_run = 0;

static void load_goldens(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    while (f) {
        std::string name;
        std::getline(f, name, '\0');
        if (name.empty()) break;
        uint32_t wh[2];
        f.read((char*)wh, 8);
        RgbaImage img(wh[0], wh[1]);
        f.read((char*)img.data.data(), (std::streamsize)img.data.size());
        g_golden[name] = img;
    }
}

// assert_golden(category, name, actual): tolerance 0 (tests/common/mod.rs:211-263)
static void assert_golden(const std::string& category, const std::string& name, const RgbaImage& actual)
{
    ++g_run;
    auto it = g_golden.find(category + "/" + name);
    if (it == g_golden.end()) { std::printf("FAILED %s/%s: golden not found\n", category.c_str(), name.c_str()); ++g_failed; return; }
    if (!(actual == it->second)) {
        size_t bad = 0;
        for (size_t i = 0; i < actual.data.size() && i < it->second.data.size(); i += 4) bad += std::memcmp(&actual.data[i], &it->second.data[i], 4) != 0;
        std::printf("FAILED %s/%s: %zu px differ\n", category.c_str(), name.c_str(), bad);
        ++g_failed;
    }
}
static void assert_eq(const RgbaImage& a, const RgbaImage& b, const char* what)
{
    ++g_run;
    if (!(a == b)) { std::printf("FAILED %s\n", what); ++g_failed; }
}

// tests/common/mod.rs:272-307
static RgbaImage create_test_gradient(uint32_t w, uint32_t h)
{
    RgbaImage img(w, h);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t r = w > 1 ? (uint8_t)(x * 255 / (w - 1)) : 128, b = h > 1 ? (uint8_t)(y * 255 / (h - 1)) : 128;
            uint8_t* p = img.pixel(x, y);
            p[0] = r; p[1] = 255 - r; p[2] = b; p[3] = 255;
        }
    return img;
}
static RgbaImage create_test_checkerboard(uint32_t w, uint32_t h)
{
    RgbaImage img(w, h);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t v = ((x / 8 + y / 8) % 2 == 0) ? 255 : 0;
            uint8_t* p = img.pixel(x, y);
            p[0] = p[1] = p[2] = v; p[3] = 255;
        }
    return img;
}

// tests/visual_blend.rs:19-49
static RgbaImage make_blend_test(GpuRenderer& gpu, BlendMode mode)
{
    const uint32_t w = 64, h = 64;
    RgbaImage fg_img(w, h);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t* p = fg_img.pixel(x, y);
            p[0] = (uint8_t)(((float)x / (float)w) * 255.0f);
            p[1] = (uint8_t)(((float)y / (float)h) * 255.0f);
            p[2] = 128;
            p[3] = (uint8_t)(((float)(x + y) / (float)(w + h - 2)) * 200.0f + 55.0f);
        }
    CanvasState state(w, h);
    state.layers[0].pixels = create_test_checkerboard(w, h);
    Layer fg{"Foreground", fg_img};
    fg.blend_mode = mode;
    state.layers.push_back(fg);
    return state.composite(gpu);
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: visual_tests <golden.bin>\n"); return 2; }
    load_goldens(argv[1]);
    auto maybe = GpuRenderer::try_new(0);
    if (!maybe) { std::fprintf(stderr, "no GPU: %s\n", pfx_last_error(nullptr)); return 3; }
    GpuRenderer gpu = std::move(*maybe);

    // ---- visual_blend.rs: blend_test!(name, mode) x25
    static const char* names[25] = {"normal", "multiply", "screen", "additive", "reflect", "glow", "color_burn", "color_dodge", "overlay",
                                    "difference", "negation", "lighten", "darken", "xor", "overwrite", "hard_light", "soft_light", "exclusion",
                                    "subtract", "divide", "linear_burn", "vivid_light", "linear_light", "pin_light", "hard_mix"};
    for (int m = 0; m < 25; ++m) assert_golden("blend", names[m], make_blend_test(gpu, (BlendMode)m));
    { // normal_half_opacity (:89-106)
        CanvasState state(64, 64);
        state.layers[0].pixels = create_test_checkerboard(64, 64);
        Layer fg{"Foreground", create_test_gradient(64, 64)};
        fg.opacity = 0.5f;
        state.layers.push_back(fg);
        assert_golden("blend", "normal_half_opacity", state.composite(gpu));
    }
    { // hidden_layer_invisible (:110-136)
        CanvasState state(64, 64), bg_only(64, 64);
        state.layers[0].pixels = create_test_checkerboard(64, 64);
        bg_only.layers[0].pixels = create_test_checkerboard(64, 64);
        Layer fg{"Hidden", create_test_gradient(64, 64)};
        fg.visible = false;
        state.layers.push_back(fg);
        assert_eq(state.composite(gpu), bg_only.composite(gpu), "hidden layer should not contribute to composite");
    }

    // ---- visual_filters.rs
    const RgbaImage img = create_test_gradient(64, 64);
    assert_golden("filters", "gaussian_blur_s2", ops::parallel_gaussian_blur_pub(gpu, img, 2.0f));
    assert_golden("filters", "gaussian_blur_s5", ops::parallel_gaussian_blur_pub(gpu, img, 5.0f));
    assert_golden("filters", "box_blur_r3", ops::box_blur_core(gpu, img, 3.0f));
    assert_golden("filters", "median_r2", ops::median_core(gpu, img, 2));
    assert_golden("filters", "pixelate_8", ops::pixelate_core(gpu, img, 8));
    assert_eq(ops::parallel_gaussian_blur_pub(gpu, img, 0.0f), img, "sigma=0 should be identity");
    assert_eq(ops::pixelate_core(gpu, img, 1), ops::pixelate_core(gpu, img, 2), "block 1 behaves as block 2 (bs.max(2))");

    // ---- GpuRenderer filter methods (gpu_pipelines.rs surface), held to the CPU-path goldens
    {
        RgbaImage blurred(64, 64, gpu.blur_rgba(img.data, 64, 64, 2.0f));
        assert_golden("filters", "gaussian_blur_s2", blurred);
        RgbaImage inv(64, 64, gpu.invert_rgba(img.data, 64, 64)), inv2(64, 64, gpu.invert_rgba(inv.data, 64, 64));
        assert_eq(inv2, img, "invert twice is identity");
        RgbaImage hsl0(64, 64, gpu.hsl_rgba(img.data, 64, 64, 0.0f, 0.0f, 0.0f));
        assert_eq(hsl0, img, "hsl(0,0,0) is identity");
        RgbaImage bc0(64, 64, gpu.brightness_contrast_rgba(img.data, 64, 64, 0.0f, 0.0f));
        assert_eq(bc0, img, "brightness/contrast (0,0) is identity");
        ++g_run;
        if (gpu.median_rgba(img.data, 64, 64, 1000).has_value()) { std::printf("FAILED median_rgba beyond the device radius must be None\n"); ++g_failed; }
        gpu.ensure_layer_texture(0, 64, 64, img.data, 7);
        auto flat = gpu.composite(64, 64, {{0, 1.0f, true, 0}});
        ++g_run;
        if (!flat || !(RgbaImage(64, 64, *flat) == img)) { std::printf("FAILED single opaque layer composite\n"); ++g_failed; }
    }
    // ---- io_roundtrip.rs:148-200 roundtrip_pfe_multi_layer, :314-330 load through a path; then composite the loaded document
    {
        CanvasState state(64, 64);
        for (auto& b : state.layers[0].pixels.data) b = 255; // white background
        RgbaImage red(64, 64);
        for (size_t i = 0; i < red.data.size(); i += 4) { red.data[i] = 255; red.data[i + 1] = 0; red.data[i + 2] = 0; red.data[i + 3] = 128; }
        Layer l1{"Red", red};
        l1.opacity = 0.75f;
        state.layers.push_back(l1);
        Layer l2{"Gradient", create_test_gradient(64, 64)};
        l2.blend_mode = BlendMode::Multiply;
        state.layers.push_back(l2);
        state.active_layer_index = 2;
        const std::string path = std::string(argv[1]) + ".rt_multi.pfe";
        io::save_pfe(state, path);
        CanvasState loaded = io::load_pfe(path);
        ++g_run;
        if (loaded.layers.size() != 3 || loaded.layers[1].opacity != 0.75f || loaded.layers[1].name != "Red" || loaded.layers[2].name != "Gradient" ||
            loaded.layers[2].blend_mode != BlendMode::Multiply || loaded.active_layer_index != 2) { std::printf("FAILED pfe metadata round trip\n"); ++g_failed; }
        for (size_t i = 0; i < 3; ++i) assert_eq(loaded.layers[i].pixels, state.layers[i].pixels, "PFE layers should be pixel-exact");
        assert_eq(loaded.composite(gpu), state.composite(gpu), "a loaded project composites like the original");
        std::remove(path.c_str());
        ++g_run;
        try { io::load_pfe(path); std::printf("FAILED loading a missing project must fail\n"); ++g_failed; } catch (const Error&) {}
    }
    std::printf("%d checks, %d failed\n", g_run, g_failed);
    return g_failed ? 1 : 0;
}
