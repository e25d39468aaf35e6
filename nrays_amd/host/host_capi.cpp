// host_capi.cpp — C entry points of libnrays_host.so: lets Python (tests, bench.py) and other hosts use
// the C++ loader3d front-end.  No pixels are computed here.
#include <cstring>
#include <string>

#include "host.hpp"

using namespace nrays_host;

namespace { thread_local std::string g_err; }

extern "C" {

typedef struct NraysHostCamera { // examples/loader3d.rs:128-135
    double eye[3], at[3], fovy, resolution[2], aa[2];
    char output[256];
} NraysHostCamera;

const char* nrays_host_last_error(void) { return g_err.c_str(); }

int nrays_host_load_scene(const char* path, int allow_standins, void** out) {
    if (!path || !out) { g_err = "null argument"; return NRAYS_ERR_BAD_ARG; }
    try {
        LoadOptions o; o.allow_standins = allow_standins != 0;
        *out = load_scene_file(path, o).release();
        return NRAYS_OK;
    } catch (const std::exception& e) { g_err = e.what(); *out = nullptr; return NRAYS_ERR_BAD_ARG; }
}
const NraysSceneDesc* nrays_host_scene_desc(void* h) { return h ? &((LoadedScene*)h)->desc : nullptr; }
uint32_t nrays_host_num_cameras(void* h) { return h ? (uint32_t)((LoadedScene*)h)->cameras.size() : 0; }
int nrays_host_camera(void* h, uint32_t i, NraysHostCamera* out) {
    if (!h || !out || i >= ((LoadedScene*)h)->cameras.size()) { g_err = "bad camera index"; return NRAYS_ERR_BAD_ARG; }
    const Camera& c = ((LoadedScene*)h)->cameras[i];
    std::memset(out, 0, sizeof *out);
    for (int k = 0; k < 3; ++k) { out->eye[k] = c.eye[k]; out->at[k] = c.at[k]; }
    out->fovy = c.fovy; out->resolution[0] = c.resolution[0]; out->resolution[1] = c.resolution[1]; out->aa[0] = c.aa[0]; out->aa[1] = c.aa[1];
    std::strncpy(out->output, c.output.c_str(), sizeof out->output - 1);
    return NRAYS_OK;
}
int nrays_host_inverse_projection(const NraysHostCamera* c, double width, double height, double out16[16]) {
    if (!c || !out16) { g_err = "null argument"; return NRAYS_ERR_BAD_ARG; }
    try {
        Camera cam; for (int k = 0; k < 3; ++k) { cam.eye[k] = c->eye[k]; cam.at[k] = c->at[k]; } cam.fovy = c->fovy;
        inverse_projection(cam, width, height, out16);
        return NRAYS_OK;
    } catch (const std::exception& e) { g_err = e.what(); return NRAYS_ERR_BAD_ARG; }
}
uint32_t nrays_host_num_warnings(void* h) { return h ? (uint32_t)((LoadedScene*)h)->warnings.size() : 0; }
const char* nrays_host_warning(void* h, uint32_t i) { return (h && i < ((LoadedScene*)h)->warnings.size()) ? ((LoadedScene*)h)->warnings[i].c_str() : ""; }
void nrays_host_free_scene(void* h) { delete (LoadedScene*)h; }

int nrays_host_write_png(const char* path, const float* rgb, uint32_t w, uint32_t h) { // Image::to_png, src/image.rs:60-90
    try { auto q = quantize_rgb8(rgb, (size_t)w * h * 3); write_png_rgb8(path, q.data(), w, h); return NRAYS_OK; }
    catch (const std::exception& e) { g_err = e.what(); return NRAYS_ERR_BAD_ARG; }
}
int nrays_host_write_ppm(const char* path, const float* rgb, uint32_t w, uint32_t h) {
    try { write_ppm(path, rgb, w, h); return NRAYS_OK; } catch (const std::exception& e) { g_err = e.what(); return NRAYS_ERR_BAD_ARG; }
}
// Decodes a PNG into `out` (caller-allocated, capacity bytes); returns channels, sets w/h; < 0 on error.
int nrays_host_read_png(const char* path, uint8_t* out, size_t capacity, uint32_t* w, uint32_t* h) {
    try {
        Image8 im = read_png(path);
        *w = im.width; *h = im.height;
        if (out) { if (im.data.size() > capacity) { g_err = "buffer too small"; return NRAYS_ERR_BAD_ARG; } std::memcpy(out, im.data.data(), im.data.size()); }
        return im.channels;
    } catch (const std::exception& e) { g_err = e.what(); return NRAYS_ERR_BAD_ARG; }
}

// The same for any format stb_image's load recognises by content (PNG, JPEG, BMP, TGA): src/texture2d.rs:95.
int nrays_host_read_image(const char* path, uint8_t* out, size_t capacity, uint32_t* w, uint32_t* h) {
    try {
        Image8 im = read_image(path);
        *w = im.width; *h = im.height;
        if (out) { if (im.data.size() > capacity) { g_err = "buffer too small"; return NRAYS_ERR_BAD_ARG; } std::memcpy(out, im.data.data(), im.data.size()); }
        return im.channels;
    } catch (const std::exception& e) { g_err = e.what(); return NRAYS_ERR_BAD_ARG; }
}

} // extern "C"
