// rt_render_ppm: App3D::run's frame loop (src/app.rs:219-248) without the window, as a command-line program on top of the
// host mirror: builds one of the reference's demo scenes, attaches textures (PNG / JPEG / PPM files, in slot order), renders
// `frames` frames through Renderer::render -- the scene's update function runs before each frame with ts = frame / 60 --
// and writes the last frame as a binary PPM.
//
//   rt_render_ppm <scene1|scene2|scene3|scene4|tex4> <width> <height> <frames> <out.ppm> [texture files ...]
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "rt_host.h"

int main(int argc, char **argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s <scene> <width> <height> <frames> <out.ppm> [texture files ...]\n", argv[0]);
        return 2;
    }
    const std::string name = argv[1], out_path = argv[5];
    const long width = std::atol(argv[2]), height = std::atol(argv[3]), frames = std::atol(argv[4]);
    if (width <= 0 || height <= 0 || frames <= 0) { std::fprintf(stderr, "width, height and frames must be positive\n"); return 2; }

    rt::scenes::Built built;
    if (!rt::scenes::by_name(name, built)) { std::fprintf(stderr, "unknown scene %s\n", name.c_str()); return 2; }
    rt::Scene scene = built.scene;
    if (argc > 6) scene.textures.clear();   // the built scene only declares its texture slots; the files fill them in order
    for (int i = 6; i < argc; i++) {
        rt::Texture t;
        const std::string err = rt::load_image(argv[i], t);
        if (!err.empty()) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
        scene = scene.with_texture(t);
    }
    rt::Camera camera = built.camera;
    rt::CameraEvent resize; resize.kind = rt::CameraEvent::Resize; resize.w = (size_t)width; resize.h = (size_t)height;
    camera.update({resize}, 0.0f);

    rt::Renderer renderer;
    renderer.enable_accumulation = built.renderer_accumulation;
    std::vector<uint8_t> img((size_t)width * (size_t)height * 4u, 0);
    for (long f = 0; f < frames; f++) {
        bool updated = f == 0;
        if (scene.update_func && scene.update_func(scene, (float)f / 60.0f)) updated = true;
        const std::string err = renderer.render(scene, img, camera, updated, 200);
        if (!err.empty()) { std::fprintf(stderr, "render: %s\n", err.c_str()); return 1; }
    }

    FILE *fp = std::fopen(out_path.c_str(), "wb");
    if (!fp) { std::fprintf(stderr, "cannot write %s\n", out_path.c_str()); return 1; }
    std::fprintf(fp, "P6\n%ld %ld\n255\n", width, height);
    std::vector<uint8_t> rgb((size_t)width * (size_t)height * 3u);
    for (size_t p = 0; p < (size_t)width * (size_t)height; p++) { rgb[3 * p] = img[4 * p]; rgb[3 * p + 1] = img[4 * p + 1]; rgb[3 * p + 2] = img[4 * p + 2]; }
    std::fwrite(rgb.data(), 1, rgb.size(), fp);
    std::fclose(fp);
    std::printf("%s %ldx%ld, %ld frame(s): %llu scene upload(s), %llu object update(s) -> %s\n", name.c_str(), width, height, frames,
                (unsigned long long)renderer.full_uploads(), (unsigned long long)renderer.object_updates(), out_path.c_str());
    return 0;
}
