// loader3d.cpp — the reference's CLI (examples/loader3d.rs:34-101) over the C ABI:
//   loader3d <scene_file> [--width W --height H] [--spp N --window X] [--max-depth D] [--gpus N] [--standins] [--ppm] [--times]
// --times prints one JSON line with the wall time of every stage of the run (what a drop-in for examples/loader3d.rs:34-101 costs end to end: bench.py's
// `drop_in_end_to_end` block).
// Loads the scene, creates the device scene once (nrays_scene_create), renders every camera with
// nrays_render — or, with --gpus N, with nrays_render_multi on N band owners (one per visible GPU, round-robin) —
// and writes the PNG named by the camera's `output`.  libnrays_hip.so is dlopen'ed so the front-end itself builds
// without ROCm.
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host.hpp"

using namespace nrays_host;

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "Usage: %s scene_file [--width W --height H --spp N --window X --max-depth D --gpus N --standins --ppm]\n", argv[0]); return 2; }
    std::string path = argv[1];
    long ow = 0, oh = 0, ospp = 0, maxd = 0, gpus = 1; double owin = -1.0; bool standins = false, ppm = false, times = false;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    double ms_parse = 0, ms_lib = 0, ms_create = 0, ms_render = 0, ms_gpu = 0, ms_save = 0; unsigned long long total_rays = 0;
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--width") ow = std::atol(val()); else if (a == "--height") oh = std::atol(val());
        else if (a == "--spp") ospp = std::atol(val()); else if (a == "--window") owin = std::atof(val());
        else if (a == "--max-depth") maxd = std::atol(val()); else if (a == "--gpus") gpus = std::atol(val()); else if (a == "--standins") standins = true; else if (a == "--ppm") ppm = true; else if (a == "--times") times = true;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    try {
        std::printf("Loading the scene.\n");
        LoadOptions lo; lo.allow_standins = standins;
        auto t_ = std::chrono::steady_clock::now();
        auto sc = load_scene_file(path, lo);
        ms_parse = since(t_);
        for (auto& w : sc->warnings) std::printf("%s\n", w.c_str());
        std::printf("Scene loaded. %zu lights, %zu objects, %zu cameras.\n", sc->lights.size(), sc->nodes.size(), sc->cameras.size());

        std::string self = argv[0]; size_t k = self.find_last_of('/');
        std::string lib = (k == std::string::npos ? std::string(".") : self.substr(0, k)) + "/libnrays_hip.so";
        t_ = std::chrono::steady_clock::now();
        void* h = dlopen(lib.c_str(), RTLD_NOW);
        if (!h) h = dlopen("libnrays_hip.so", RTLD_NOW);
        if (!h) { std::fprintf(stderr, "cannot load libnrays_hip.so: %s\n", dlerror()); return 1; }
        auto create = (int (*)(const NraysSceneDesc*, NraysScene**))dlsym(h, "nrays_scene_create");
        auto render = (int (*)(NraysScene*, const NraysRenderParams*, float*))dlsym(h, "nrays_render");
        auto render_rgb8 = (int (*)(NraysScene*, const NraysRenderParams*, uint8_t*))dlsym(h, "nrays_render_rgb8");
        auto destroy = (void (*)(NraysScene*))dlsym(h, "nrays_scene_destroy");
        auto last_error = (const char* (*)())dlsym(h, "nrays_last_error");
        auto get_stats = (int (*)(NraysScene*, NraysStats*))dlsym(h, "nrays_get_stats");
        auto comm_create_local = (int (*)(uint32_t, const int32_t*, NraysComm**))dlsym(h, "nrays_comm_create_local");
        auto comm_destroy = (void (*)(NraysComm*))dlsym(h, "nrays_comm_destroy");
        auto set_create = (int (*)(const NraysSceneDesc*, NraysComm*, NraysSceneSet**))dlsym(h, "nrays_scene_set_create");
        auto set_destroy = (void (*)(NraysSceneSet*))dlsym(h, "nrays_scene_set_destroy");
        auto render_multi = (int (*)(NraysSceneSet*, const NraysRenderParams*, float*))dlsym(h, "nrays_render_multi");
        auto multi_stats = (int (*)(NraysSceneSet*, NraysStats*))dlsym(h, "nrays_multi_get_stats");
        if (!create || !render || !render_rgb8 || !destroy || !last_error || !get_stats || !comm_create_local || !comm_destroy || !set_create || !set_destroy || !render_multi || !multi_stats) {
            std::fprintf(stderr, "libnrays_hip.so lacks an ABI symbol\n"); return 1;
        }
        ms_lib = since(t_);
        t_ = std::chrono::steady_clock::now();
        NraysScene* scene = nullptr; NraysComm* comm = nullptr; NraysSceneSet* set = nullptr;
        if (gpus > 1) { // the frame tiled over `gpus` band owners (scene replicated, RCCL exchange, see include/nrays_abi.h)
            if (comm_create_local((uint32_t)gpus, nullptr, &comm) != NRAYS_OK) { std::fprintf(stderr, "nrays_comm_create_local: %s\n", last_error()); return 1; }
            if (set_create(&sc->desc, comm, &set) != NRAYS_OK) { std::fprintf(stderr, "nrays_scene_set_create: %s\n", last_error()); return 1; }
        } else if (create(&sc->desc, &scene) != NRAYS_OK) { std::fprintf(stderr, "nrays_scene_create: %s\n", last_error()); return 1; }
        ms_create = since(t_); // (a process's first HIP calls — device initialisation, the first hipMemcpy — are inside)
        for (const Camera& c : sc->cameras) {
            NraysRenderParams p; std::memset(&p, 0, sizeof p);
            p.width = (uint32_t)(ow ? ow : (long)c.resolution[0]); p.height = (uint32_t)(oh ? oh : (long)c.resolution[1]);
            p.ray_per_pixel = (uint32_t)(ospp ? ospp : (long)c.aa[0]); p.window_width = owin >= 0 ? owin : c.aa[1];
            p.max_depth = (uint32_t)maxd; p.band_owners = 1;
            for (int a = 0; a < 3; ++a) p.camera_eye[a] = c.eye[a];
            inverse_projection(c, (double)p.width, (double)p.height, p.inv_proj_view);
            std::printf("Casting %u rays per pixels (win. %g).\n", p.ray_per_pixel, p.window_width);
            std::printf("Tracing %llu rays.\n", (unsigned long long)p.width * p.height * p.ray_per_pixel);
            // one GPU: the frame comes back already quantised (Image::to_png's rule, applied on the device: a quarter of the
            // bytes over PCIe); N GPUs: the float frame of nrays_render_multi, quantised here
            std::vector<float> px; std::vector<uint8_t> q8;
            if (set) px.resize((size_t)p.width * p.height * 3); else q8.resize((size_t)p.width * p.height * 3);
            auto t0 = std::chrono::steady_clock::now();
            if ((set ? render_multi(set, &p, px.data()) : render_rgb8(scene, &p, q8.data())) != NRAYS_OK) { std::fprintf(stderr, "nrays_render: %s\n", last_error()); return 1; }
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            ms_render += ms;
            NraysStats st;
            if (set) multi_stats(set, &st); else get_stats(scene, &st);
            ms_gpu += st.kernel_ms_total;
            unsigned long long rays = st.rays_primary + st.rays_reflection + st.rays_refraction + st.rays_shadow;
            std::printf("Rays cast. %llu rays in %.3f ms (%.1f Mrays/s incl. the device-to-host copy; GPU %.3f ms)\n", rays, ms, rays / ms / 1e3, st.kernel_ms_total);
            total_rays += rays;
            std::printf("Saving image to: %s\n", c.output.c_str());
            t_ = std::chrono::steady_clock::now();
            if (set) q8 = quantize_rgb8(px.data(), px.size());
            if (ppm) write_ppm_rgb8(c.output, q8.data(), p.width, p.height);
            else write_png_rgb8(c.output, q8.data(), p.width, p.height);
            ms_save += since(t_);
            std::printf("Image saved.\n");
        }
        if (set) { set_destroy(set); comm_destroy(comm); } else destroy(scene);
        if (times)
            std::printf("{\"loader3d_times_ms\": {\"parse_scene_obj_mtl_textures\": %.3f, \"dlopen_libnrays_hip\": %.3f, \"nrays_scene_create_incl_hip_init\": %.3f, "
                        "\"render_cold_incl_d2h\": %.3f, \"render_gpu_events\": %.3f, \"quantise_encode_write_image\": %.3f, \"total\": %.3f}, \"rays\": %llu, \"cameras\": %zu}\n",
                        ms_parse, ms_lib, ms_create, ms_render, ms_gpu, ms_save, since(t_start), total_rays, sc->cameras.size());
    } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
    return 0;
}
