// rtx_example -- the reference's `cargo run --release --example <name> > image.ppm`:
// builds the example's scene, renders it on the current HIP device through
// RenderRunner::run (src/runner.rs:23-41) and prints the P3 image to stdout
// and the timers to stderr.
//
//   rtx_example <name> [--width W] [--spp N] [--depth D] [--seed S] [--scene-seed S]
//               [--partition bisect|midpoint|sah-equal:<buckets>|sah-per-surface]
//               [--build host|device] [--gpus N] [-o file.ppm]
//   rtx_example --list
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../raytracing_amd/host/examples.hpp"

using namespace raytracing;

static int usage(const char* msg) {
    if (msg) std::fprintf(stderr, "rtx_example: %s\n", msg);
    std::fprintf(stderr,
                 "usage: rtx_example <name> [--width W] [--spp N] [--depth D] [--seed S] [--scene-seed S]\n"
                 "                   [--partition bisect|midpoint|sah-equal:<buckets>|sah-per-surface] [--build host|device] [--gpus N] [-o file.ppm]\n"
                 "       rtx_example --list\n");
    return 2;
}

int main(int argc, char** argv) {
    if (argc < 2) return usage(nullptr);
    if (std::strcmp(argv[1], "--list") == 0) {
        for (const char* n : examples::NAMES) std::printf("%s\n", n);
        return 0;
    }
    const std::string name = argv[1];
    uint64_t scene_seed = 0xC0FFEEull;
    RenderRunner runner;
    long width = -1, spp = -1, depth = -1;
    const char* out_path = nullptr;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
        const char* v = nullptr;
        if (a == "--width" && (v = value())) width = std::atol(v);
        else if (a == "--spp" && (v = value())) spp = std::atol(v);
        else if (a == "--depth" && (v = value())) depth = std::atol(v);
        else if (a == "--seed" && (v = value())) runner.seed = std::strtoull(v, nullptr, 0);
        else if (a == "--scene-seed" && (v = value())) scene_seed = std::strtoull(v, nullptr, 0);
        else if (a == "-o" && (v = value())) out_path = v;
        else if (a == "--gpus" && (v = value())) runner.n_gpus = (int)std::atol(v);
        else if (a == "--build" && (v = value())) {
            if (std::strcmp(v, "device") == 0) runner.build_on_device = true;
            else if (std::strcmp(v, "host") != 0) return usage("--build takes host or device");
        } else if (a == "--partition" && (v = value())) {
            const std::string p = v;
            if (p == "bisect") runner.partition_strategy = PartitionBy::bisect();
            else if (p == "midpoint") runner.partition_strategy = PartitionBy::midpoint();
            else if (p == "sah-per-surface") runner.partition_strategy = PartitionBy::surface_area_heuristic(SAHBucketStrategy::per_surface());
            else if (p.rfind("sah-equal:", 0) == 0)
                runner.partition_strategy = PartitionBy::surface_area_heuristic(SAHBucketStrategy::equal_size((uint32_t)std::atol(p.c_str() + 10)));
            else return usage("unknown partition strategy");
        } else {
            return usage(("bad argument " + a).c_str());
        }
    }
    FILE* image_out = stdout;
    try {
        examples::Example e = examples::by_name(name, scene_seed);
        if (width > 0) e.camera.image_width = (uint32_t)width;
        if (spp > 0) e.camera.samples_per_pixel = (uint32_t)spp;
        if (depth > 0) e.camera.max_depth = (uint32_t)depth;
        runner.camera = e.camera;
        runner.keep_image = false;   // only the P3 text is wanted, as in the reference
        if (out_path && !(image_out = std::fopen(out_path, "wb"))) {
            std::fprintf(stderr, "rtx_example: cannot open %s\n", out_path);
            return 1;
        }
        runner.run(std::move(e.surfaces), image_out, stderr);
        if (out_path) std::fclose(image_out);
        return 0;
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "rtx_example: %s\n", ex.what());
        return 1;
    }
}
