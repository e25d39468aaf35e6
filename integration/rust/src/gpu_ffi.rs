//! Raw bindings: the `#[repr(C)]` twins of `include/nrays_abi.h` (ABI version 1) and its entry points.
//! Field order and types must match the header exactly; `tests/test_abi.py` pins the C side's sizes.
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_int, c_void};

pub const NRAYS_ABI_VERSION: u32 = 5; // include/nrays_abi.h; GpuScene::new refuses a library built from another header
pub const NRAYS_OK: c_int = 0;
pub const NRAYS_ERR_BAD_ARG: c_int = -1;
pub const NRAYS_ERR_HIP: c_int = -2;
pub const NRAYS_ERR_OOM: c_int = -3;
pub const NRAYS_ERR_UNSUPPORTED: c_int = -4;
pub const NRAYS_ERR_NO_DEVICE: c_int = -5;
pub const NRAYS_ERR_QUEUE_OVERFLOW: c_int = -6;
pub const NRAYS_ERR_RCCL: c_int = -7;

// NraysShapeKind (examples/loader3d.rs:593-695)
pub const NRAYS_SHAPE_BALL: u32 = 0;
pub const NRAYS_SHAPE_CUBOID: u32 = 1;
pub const NRAYS_SHAPE_CYLINDER: u32 = 2;
pub const NRAYS_SHAPE_CAPSULE: u32 = 3;
pub const NRAYS_SHAPE_CONE: u32 = 4;
pub const NRAYS_SHAPE_PLANE: u32 = 5;
pub const NRAYS_SHAPE_TRIMESH: u32 = 6;
// NraysMaterialKind
pub const NRAYS_MAT_PHONG: u32 = 0;
pub const NRAYS_MAT_NORMAL: u32 = 1;
pub const NRAYS_MAT_UV: u32 = 2;
// NraysTexelFormat / NraysInterpolation / NraysOverflow
pub const NRAYS_TEXEL_RGBA8: u32 = 0;
pub const NRAYS_TEXEL_RGBA32F: u32 = 1;
pub const NRAYS_INTERP_BILINEAR: u32 = 0;
pub const NRAYS_INTERP_NEAREST: u32 = 1;
pub const NRAYS_OVERFLOW_WRAP: u32 = 0;
pub const NRAYS_OVERFLOW_CLAMP: u32 = 1;
pub const NRAYS_UNIQUE_ID_BYTES: usize = 128;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct NraysLight {
    pub pos: [f64; 3],
    pub radius: f64,
    pub racsample: u32,
    pub color: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct NraysTexture {
    pub width: u32,
    pub height: u32,
    pub format: u32,
    pub interp: u32,
    pub overflow: u32,
    pub reserved: u32,
    pub texels: *const c_void,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct NraysMaterial {
    pub kind: u32,
    pub ambiant: [f32; 3],
    pub diffuse: [f32; 3],
    pub specular: [f32; 3],
    pub shininess: f32,
    pub texture_id: i32,
    pub alpha_texture_id: i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct NraysMesh {
    pub num_vertices: u32,
    pub num_triangles: u32,
    pub vertices: *const f64,
    pub uvs: *const f64,
    pub indices: *const u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct NraysNode {
    pub shape_kind: u32,
    pub solid: u32,
    pub params: [f64; 3],
    pub translation: [f64; 3],
    pub axis_angle: [f64; 3],
    pub refl_mix: f32,
    pub refl_atenuation: f32,
    pub alpha: f32,
    pub reserved0: f32,
    pub refr_coeff: f64,
    pub material_id: u32,
    pub mesh_id: i32,
}

#[repr(C)]
pub struct NraysSceneDesc {
    pub background: [f32; 3],
    pub num_lights: u32,
    pub lights: *const NraysLight,
    pub num_materials: u32,
    pub materials: *const NraysMaterial,
    pub num_textures: u32,
    pub textures: *const NraysTexture,
    pub num_meshes: u32,
    pub meshes: *const NraysMesh,
    pub num_nodes: u32,
    pub nodes: *const NraysNode,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct NraysRenderParams {
    pub width: u32,
    pub height: u32,
    pub ray_per_pixel: u32,
    pub max_depth: u32,
    pub window_width: f64,
    pub camera_eye: [f64; 3],
    pub inv_proj_view: [f64; 16],
    pub seed: u64,
    pub band_rows: u32,
    pub band_owner: u32,
    pub band_owners: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct NraysMultiTimings {
    pub render_ms: f64,
    pub exchange_ms: f64,
    pub untile_ms: f64,
    pub frames: u32,
    pub owner: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct NraysStats {
    pub rays_primary: u64,
    pub rays_reflection: u64,
    pub rays_refraction: u64,
    pub rays_shadow: u64,
    pub node_tests: u64,
    pub tri_tests: u64,
    pub prim_tests: u64,
    pub hit_records: u64,
    pub tex_samples: u64,
    pub generations: u32,
    pub instrumented: u32,
    pub kernel_ms_primary: f64,
    pub kernel_ms_total: f64,
    pub frames_timed: u32,
    pub reserved: u32,
    pub rays_primary_traced: u64,
    pub rays_shadow_elided: u64,
    pub node_fetches: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct NraysTileCosts {
    pub tiles: u64,
    pub sum_cycles: u64,
    pub max_cycles: u64,
    pub resident_waves: u64,
    pub shader_clock_hz: f64,
    pub kernel_ms: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct NraysCastResult {
    pub toi: f64,
    pub normal: [f64; 3],
    pub uv: [f64; 2],
    pub node_id: i32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct NraysBlasDump {
    pub num_nodes: u32,
    pub num_refs: u32,
    pub root: i32,
    pub max_depth: i32,
    pub hairy: u32,
    pub node_capacity: u32,
    pub ref_capacity: u32,
    pub pad: u32,
    pub nodes: *mut f32,
    pub tri_ids: *mut u32,
}

pub enum NraysScene {}
pub enum NraysComm {}
pub enum NraysSceneSet {}

extern "C" {
    pub fn nrays_abi_version() -> u32;
    pub fn nrays_last_error() -> *const c_char;

    pub fn nrays_scene_create(desc: *const NraysSceneDesc, out_scene: *mut *mut NraysScene) -> c_int;
    pub fn nrays_scene_destroy(scene: *mut NraysScene);
    pub fn nrays_scene_device_bytes(scene: *const NraysScene) -> u64;
    pub fn nrays_render(scene: *mut NraysScene, params: *const NraysRenderParams, out_rgb: *mut f32) -> c_int;
    pub fn nrays_render_rgb8(scene: *mut NraysScene, params: *const NraysRenderParams, out_rgb8: *mut u8) -> c_int;
    pub fn nrays_render_device(scene: *mut NraysScene, params: *const NraysRenderParams, out_rgb_device: *mut f32, hip_stream: *mut c_void) -> c_int;
    pub fn nrays_render_device_instrumented(scene: *mut NraysScene, params: *const NraysRenderParams, out_rgb_device: *mut f32, hip_stream: *mut c_void) -> c_int;
    pub fn nrays_render_device_counted(scene: *mut NraysScene, params: *const NraysRenderParams, out_rgb_device: *mut f32, hip_stream: *mut c_void, flags: u32) -> c_int;
    pub fn nrays_tile_rows(params: *const NraysRenderParams) -> u32;
    pub fn nrays_untile_device(gathered: *const f32, out_rgb_device: *mut f32, width: u32, height: u32, band_rows: u32, band_owners: u32, hip_stream: *mut c_void) -> c_int;
    pub fn nrays_get_stats(scene: *mut NraysScene, out_stats: *mut NraysStats) -> c_int;
    pub fn nrays_get_primary_kernel_stats(scene: *mut NraysScene, out_stats: *mut NraysStats) -> c_int;
    pub fn nrays_debug_blas_build(mesh: *const NraysMesh, flags: u32, out: *mut NraysBlasDump) -> c_int;
    pub fn nrays_debug_node_aabb(scene: *mut NraysScene, node: u32, out: *mut f64) -> c_int;
    pub fn nrays_debug_scene_flags(scene: *const NraysScene, out: *mut u32) -> c_int;
    pub fn nrays_get_tile_costs(scene: *mut NraysScene, out: *mut NraysTileCosts) -> c_int;
    pub fn nrays_debug_cast_batch(scene: *mut NraysScene, mode: u32, n: u32, origins: *const f64, dirs: *const f64, max_toi: *const f64, out: *mut NraysCastResult) -> c_int;

    pub fn nrays_comm_unique_id(out_id: *mut u8) -> c_int;
    pub fn nrays_comm_create(id: *const u8, num_ranks: u32, rank: u32, out_comm: *mut *mut NraysComm) -> c_int;
    pub fn nrays_comm_create_local(num_owners: u32, devices: *const i32, out_comm: *mut *mut NraysComm) -> c_int;
    pub fn nrays_comm_owners(comm: *const NraysComm) -> u32;
    pub fn nrays_comm_destroy(comm: *mut NraysComm);
    pub fn nrays_scene_set_create(desc: *const NraysSceneDesc, comm: *mut NraysComm, out_set: *mut *mut NraysSceneSet) -> c_int;
    pub fn nrays_scene_set_destroy(set: *mut NraysSceneSet);
    pub fn nrays_scene_set_num_local(set: *const NraysSceneSet) -> u32;
    pub fn nrays_scene_set_local_scene(set: *mut NraysSceneSet, k: u32, out_owner: *mut u32) -> *mut NraysScene;
    pub fn nrays_render_multi(set: *mut NraysSceneSet, params: *const NraysRenderParams, out_rgb: *mut f32) -> c_int;
    pub fn nrays_render_multi_device(set: *mut NraysSceneSet, params: *const NraysRenderParams, out_rgb_device: *mut f32) -> c_int;
    pub fn nrays_multi_sync(set: *mut NraysSceneSet) -> c_int;
    pub fn nrays_multi_get_stats(set: *mut NraysSceneSet, out_stats: *mut NraysStats) -> c_int;
    pub fn nrays_multi_get_timings(set: *mut NraysSceneSet, out_timings: *mut NraysMultiTimings) -> c_int;
}
