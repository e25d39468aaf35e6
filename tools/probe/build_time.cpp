// Host-only timing of build_host_scene on a hair-like mesh (N strands x 8-sided tubes x S segments), no GPU needed:
//   hipcc -O3 -std=c++17 -ffp-contract=off tools/probe/build_time.cpp nrays_amd/csrc/scene_build.cpp nrays_amd/csrc/bvh_build.cpp -o /tmp/build_time && /tmp/build_time 3000 60
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../nrays_amd/csrc/scene_build.h"

int main(int argc, char** argv) {
    int strands = argc > 1 ? atoi(argv[1]) : 3000, segs = argc > 2 ? atoi(argv[2]) : 60;
    std::mt19937_64 rng(0x4A1B);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> verts; std::vector<uint32_t> idx;
    for (int s = 0; s < strands; ++s) {
        double p[3] = {U(rng), U(rng), U(rng)}, d[3] = {U(rng), U(rng), U(rng)};
        for (int k = 0; k <= segs; ++k) {
            for (int a = 0; a < 3; ++a) { d[a] += 0.3 * U(rng); }
            double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); for (int a = 0; a < 3; ++a) d[a] /= n;
            for (int a = 0; a < 3; ++a) p[a] += 0.03 * d[a];
            for (int r = 0; r < 8; ++r) {
                double ang = r * 0.785398163, ox = 0.002 * std::cos(ang), oy = 0.002 * std::sin(ang);
                double v[3] = {p[0] + ox, p[1] + oy, p[2] + 0.5 * (ox - oy)};
                for (int a = 0; a < 3; ++a) verts.push_back((double)(float)v[a]);
            }
            if (k) {
                uint32_t b0 = (uint32_t)(verts.size() / 3) - 16, b1 = b0 + 8;
                for (uint32_t r = 0; r < 8; ++r) {
                    uint32_t r1 = (r + 1) & 7;
                    idx.insert(idx.end(), {b0 + r, b1 + r, b1 + r1, b0 + r, b1 + r1, b0 + r1});
                }
            }
        }
    }
    NraysMesh mesh{}; mesh.num_vertices = (uint32_t)(verts.size() / 3); mesh.num_triangles = (uint32_t)(idx.size() / 3);
    mesh.vertices = verts.data(); mesh.indices = idx.data(); mesh.uvs = nullptr;
    NraysMaterial mat{}; mat.kind = NRAYS_MAT_PHONG; mat.texture_id = -1; mat.alpha_texture_id = -1; mat.shininess = 10.f;
    for (int a = 0; a < 3; ++a) { mat.ambiant[a] = 0.1f; mat.diffuse[a] = 0.8f; mat.specular[a] = 1.f; }
    NraysNode node{}; node.shape_kind = NRAYS_SHAPE_TRIMESH; node.mesh_id = 0; node.material_id = 0; node.alpha = 1.f; node.refr_coeff = 1.0;
    NraysLight light{}; light.pos[1] = 10.0; light.racsample = 1; for (int a = 0; a < 3; ++a) light.color[a] = 1.f;
    NraysSceneDesc d{}; d.num_nodes = 1; d.nodes = &node; d.num_meshes = 1; d.meshes = &mesh; d.num_materials = 1; d.materials = &mat;
    d.num_lights = 1; d.lights = &light; for (int a = 0; a < 3; ++a) d.background[a] = 1.f;
    nrays::HostScene hs; std::string err;
    auto t0 = std::chrono::steady_clock::now();
    int rc = nrays::build_host_scene(&d, hs, err);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("rc %d (%s) triangles %u -> references %zu, nodes %zu, depth %d, %.2f s\n", rc, err.c_str(), mesh.num_triangles, hs.tris.size(), hs.nodes.size(), hs.max_bvh_depth, dt);
    return rc;
}
