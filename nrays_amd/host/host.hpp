// host.hpp — C++ host-side mirror of the reference's scene model and loader3d front-end.
//
// The reference's host code is Rust (src/*.rs, examples/loader3d.rs); Rust is absent from this image,
// so the host side above the C ABI (include/nrays_abi.h) is written in C++.  Names follow the
// reference: Light (src/light.rs), Texture2d / ImageData (src/texture2d.rs), MtlMaterial
// (src/mtl.rs), obj groups (src/obj.rs), the `.scene` grammar (examples/loader3d.rs:214-906).
// Nothing here computes pixels.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/nrays_abi.h"

namespace nrays_host {

struct Image8 { // what stb_image::load returns: `depth` interleaved u8 channels, top row first
    uint32_t width = 0, height = 0;
    int channels = 0;
    std::vector<uint8_t> data;
};
Image8 read_png(const std::string& path);
Image8 read_image(const std::string& path); // stb_image's `load`: PNG, JPEG, BMP or TGA by content (src/texture2d.rs:95), image_codec.cpp
void write_png_rgb8(const std::string& path, const uint8_t* rgb, uint32_t w, uint32_t h);
std::vector<uint8_t> quantize_rgb8(const float* rgb, size_t n);
void write_ppm(const std::string& path, const float* rgb, uint32_t w, uint32_t h);
void write_ppm_rgb8(const std::string& path, const uint8_t* rgb8, uint32_t w, uint32_t h); // the same file from already quantised bytes

// ImageData (src/texture2d.rs:10-25) after the decode of :99-177: RGBA texels, row 0 = bottom.
struct ImageData {
    uint32_t width = 0, height = 0;
    uint32_t format = NRAYS_TEXEL_RGBA8;
    std::vector<uint8_t> bytes; // RGBA8 or RGBA32F
};
std::shared_ptr<ImageData> decode_texture(const Image8& img, bool opacity); // texture2d.rs:99-177

struct MtlMaterial { // src/mtl.rs:124-161
    std::string name;
    std::string ambiant_texture, diffuse_texture, specular_texture, opacity_map;
    float ambiant[3] = {1, 1, 1}, diffuse[3] = {1, 1, 1}, specular[3] = {1, 1, 1};
    float shininess = 60.0f, alpha = 1.0f;
};
std::vector<MtlMaterial> parse_mtl_file(const std::string& path); // src/mtl.rs:18-89

struct ObjGroup { // one entry of obj::parse's result (src/obj.rs:62-397)
    std::string name;
    std::vector<uint32_t> faces; // 3 per triangle, into the shared vertex arrays
    bool has_mtl = false;
    MtlMaterial mtl;
};
struct ObjFile {
    std::vector<float> coords; // 3 per vertex (f32, obj.rs:197-205)
    std::vector<float> uvs;    // 2 per vertex (zeros when the file has none, obj.rs:377)
    std::vector<ObjGroup> groups; // groups with at least one face, in order of first appearance (D-11)
};
ObjFile parse_obj_file(const std::string& path, const std::string& mtl_base_dir);

struct Camera { // examples/loader3d.rs:128-166
    double eye[3], at[3], fovy;
    double resolution[2];
    double aa[2] = {1.0, 0.0};
    std::string output;
};

// Everything loader3d's parse() produces, flattened for the C ABI; owns all the storage the
// descriptor points to.
struct LoadedScene {
    std::vector<NraysLight> lights;
    std::vector<NraysMaterial> materials;
    std::vector<NraysTexture> textures;
    std::vector<std::shared_ptr<ImageData>> texture_data;
    std::vector<NraysMesh> meshes;
    std::vector<std::shared_ptr<std::vector<double>>> vertex_arrays, uv_arrays;
    std::vector<std::shared_ptr<std::vector<uint32_t>>> index_arrays;
    std::vector<NraysNode> nodes;
    std::vector<Camera> cameras;
    std::vector<std::string> warnings;
    NraysSceneDesc desc;
    void finalize(); // fills `desc` from the vectors (background = white, loader3d.rs:61)
};

struct LoadOptions {
    bool allow_standins = false; // generate the procedural stand-in for a missing media/globe.png
};
std::unique_ptr<LoadedScene> load_scene_file(const std::string& path, const LoadOptions& opt);
std::unique_ptr<LoadedScene> parse_scene(const std::string& text, const std::string& base_dir, const LoadOptions& opt);

// Camera set-up of examples/loader3d.rs:68-79: (P * V)^-1, column-major.
void inverse_projection(const Camera& c, double width, double height, double out16[16]);

} // namespace nrays_host
