//! `gpu` — the MI355X trace loop behind the reference's own types: `GpuScene::new(&Scene)` replaces the BVT builds of
//! `Scene::new` (src/scene.rs:119-133), `gpu::render` has the signature and the meaning of `scene::render`
//! (src/scene.rs:29-36) and returns the same `Image`; `gpu::render_multi` tiles the frame over the GPUs of the node.
//!
//! `SceneNode` holds `Box<RayCast>` / `Arc<Box<Material>>` trait objects (src/scene_node.rs:14-16), which cannot be
//! inspected after the fact, so the plain-data descriptor of a node is captured where the concrete types are still
//! known: in the generic `SceneNode::new<G>` (src/scene_node.rs:22-47, bound `G: FlattenShape` added by the patch)
//! and through `Material::flatten` (a defaulted trait method added by the patch; user-defined materials return `None`
//! and such scenes stay on the CPU path).
//!
//! Dialect: Rust 2015 / nalgebra 0.15 / ncollide3d 0.16, like the rest of the crate.  This file cannot be compiled in
//! the repository that ships it (no Rust toolchain there); shape accessor names follow the ncollide3d 0.16 documentation.

use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::c_void;
use std::ptr;
use std::sync::Arc;

use na::{Matrix4, Point2, Point3, Point4, Vector2, Vector3};
use ncollide3d::shape::{Ball, Capsule, Cone, Cuboid, Cylinder, Plane, TriMesh};

use gpu_ffi::*;
use image::Image;
use light::Light;
use math::{Isometry, Point, Scalar};
use scene::{Scene, Vless};
use scene_node::SceneNode;
use texture2d::{ImageData, Interpolation, Overflow, Texture2d};

// ------------------------------------------------------------------------------------------------ shapes
/// What `SceneNode::new<G>` can still see of its geometry: the arguments the loader passed to the shape constructor
/// (examples/loader3d.rs:601,612,623,634,645,656,695).
#[derive(Clone)]
pub enum ShapeDesc {
    Ball { radius: Scalar },
    Cuboid { half_extents: Vector3<Scalar> },
    Cylinder { half_height: Scalar, radius: Scalar },
    Capsule { half_height: Scalar, radius: Scalar },
    Cone { half_height: Scalar, radius: Scalar },
    Plane { normal: Vector3<Scalar> },
    TriMesh(MeshData),
}

/// The buffers of a `TriMesh` (shared `Arc`s, nothing is copied until the scene is flattened).
#[derive(Clone)]
pub struct MeshData {
    pub vertices: Arc<Vec<Point3<Scalar>>>,
    pub indices: Arc<Vec<Point3<usize>>>,
    pub uvs: Option<Arc<Vec<Point2<Scalar>>>>,
}

/// Implemented for every shape the loader constructs; the bound `G: FlattenShape` on `SceneNode::new` makes any other
/// geometry a compile-time error instead of a silent CPU fallback.
pub trait FlattenShape {
    fn shape_desc(&self) -> ShapeDesc;
}

impl FlattenShape for Ball<Scalar> {
    fn shape_desc(&self) -> ShapeDesc {
        ShapeDesc::Ball { radius: self.radius() }
    }
}
impl FlattenShape for Cuboid<Scalar> {
    fn shape_desc(&self) -> ShapeDesc {
        ShapeDesc::Cuboid { half_extents: *self.half_extents() }
    }
}
impl FlattenShape for Cylinder<Scalar> {
    fn shape_desc(&self) -> ShapeDesc {
        ShapeDesc::Cylinder { half_height: self.half_height(), radius: self.radius() }
    }
}
impl FlattenShape for Capsule<Scalar> {
    fn shape_desc(&self) -> ShapeDesc {
        ShapeDesc::Capsule { half_height: self.half_height(), radius: self.radius() }
    }
}
impl FlattenShape for Cone<Scalar> {
    fn shape_desc(&self) -> ShapeDesc {
        ShapeDesc::Cone { half_height: self.half_height(), radius: self.radius() }
    }
}
impl FlattenShape for Plane<Scalar> {
    fn shape_desc(&self) -> ShapeDesc {
        ShapeDesc::Plane { normal: self.normal().unwrap() }
    }
}
impl FlattenShape for TriMesh<Scalar> {
    fn shape_desc(&self) -> ShapeDesc {
        ShapeDesc::TriMesh(MeshData {
            vertices: self.vertices().clone(),
            indices: self.indices().clone(),
            uvs: self.uvs().clone(),
        })
    }
}

// --------------------------------------------------------------------------------------------- materials
/// Plain-data form of a material; the two textures are still `Texture2d`s, the `TextureTable` turns them into indices.
pub struct MaterialDesc {
    pub kind: u32, // NRAYS_MAT_*
    pub ambiant: [f32; 3],
    pub diffuse: [f32; 3],
    pub specular: [f32; 3],
    pub shininess: f32,
    pub texture: Option<Texture2d>,
    pub alpha: Option<Texture2d>,
}

impl MaterialDesc {
    /// NormalMaterial / UVMaterial: the colour fields are ignored by the library.
    pub fn special(kind: u32) -> MaterialDesc {
        MaterialDesc { kind: kind, ambiant: [0.0; 3], diffuse: [0.0; 3], specular: [0.0; 3], shininess: 0.0, texture: None, alpha: None }
    }
}

/// De-duplicates textures by the address of their shared `ImageData` (the loader's TextureManager hands out one `Arc`
/// per file, src/texture2d.rs:28-48) and keeps the `Arc`s alive until the library has copied the texels.
pub struct TextureTable {
    index: HashMap<(usize, u32, u32), i32>,
    keep: Vec<Arc<ImageData>>,
    pub records: Vec<NraysTexture>,
}

impl TextureTable {
    pub fn new() -> TextureTable {
        TextureTable { index: HashMap::new(), keep: Vec::new(), records: Vec::new() }
    }

    /// Index of `tex` in the texture array of the descriptor (-1 for `None`).
    pub fn id_of(&mut self, tex: &Option<Texture2d>) -> i32 {
        let tex = match *tex {
            Some(ref t) => t,
            None => return -1,
        };
        let data = tex.data(); // &Arc<ImageData>, accessor added by the patch
        let interp = match *tex.interpolation() {
            Interpolation::Bilinear => NRAYS_INTERP_BILINEAR,
            Interpolation::Nearest => NRAYS_INTERP_NEAREST,
        };
        let overflow = match *tex.overflow() {
            Overflow::Wrap => NRAYS_OVERFLOW_WRAP,
            Overflow::ClampToEdges => NRAYS_OVERFLOW_CLAMP,
        };
        let key = (&**data as *const ImageData as usize, interp, overflow);
        if let Some(id) = self.index.get(&key) {
            return *id;
        }
        let dims: Vector2<usize> = data.dims();
        let pixels: &[Point4<f32>] = data.pixels(); // row 0 = bottom row: from_png already flipped Y (texture2d.rs:99-107)
        self.records.push(NraysTexture {
            width: dims.x as u32,
            height: dims.y as u32,
            format: NRAYS_TEXEL_RGBA32F, // Point4<f32> is four packed f32: the reference's own in-memory form
            interp: interp,
            overflow: overflow,
            reserved: 0,
            texels: pixels.as_ptr() as *const c_void,
        });
        self.keep.push(data.clone());
        let id = self.records.len() as i32 - 1;
        self.index.insert(key, id);
        id
    }
}

// ------------------------------------------------------------------------------------------- flat scene
/// Owns every array the `NraysSceneDesc` points into; may be dropped as soon as `nrays_scene_create` has returned
/// (the library copies everything).
pub struct FlatScene {
    lights: Vec<NraysLight>,
    materials: Vec<NraysMaterial>,
    textures: TextureTable,
    mesh_keep: Vec<MeshData>,
    mesh_indices: Vec<Vec<u32>>,
    meshes: Vec<NraysMesh>,
    nodes: Vec<NraysNode>,
    background: [f32; 3],
}

impl FlatScene {
    pub fn desc(&self) -> NraysSceneDesc {
        NraysSceneDesc {
            background: self.background,
            num_lights: self.lights.len() as u32,
            lights: self.lights.as_ptr(),
            num_materials: self.materials.len() as u32,
            materials: self.materials.as_ptr(),
            num_textures: self.textures.records.len() as u32,
            textures: self.textures.records.as_ptr(),
            num_meshes: self.meshes.len() as u32,
            meshes: self.meshes.as_ptr(),
            num_nodes: self.nodes.len() as u32,
            nodes: self.nodes.as_ptr(),
        }
    }
}

fn flatten_light(l: &Light) -> NraysLight {
    NraysLight {
        pos: [l.pos.x, l.pos.y, l.pos.z],
        radius: l.radius,
        racsample: l.racsample as u32, // already floor(sqrt(nsample)), src/light.rs:20
        color: [l.color.x, l.color.y, l.color.z],
    }
}

impl Scene {
    /// The scene as plain data.  `Err` names the first node whose material cannot cross the FFI (a user-defined
    /// `Material` impl): such a scene keeps rendering through `scene::render`.
    pub fn flatten(&self) -> Result<FlatScene, String> {
        let mut flat = FlatScene {
            lights: self.lights().iter().map(flatten_light).collect(),
            materials: Vec::new(),
            textures: TextureTable::new(),
            mesh_keep: Vec::new(),
            mesh_indices: Vec::new(),
            meshes: Vec::new(),
            nodes: Vec::new(),
            background: { let b = self.background(); [b.x, b.y, b.z] }, // accessor added by the patch
        };
        // one NraysMaterial per distinct material object (nodes share them through Arc, loader3d.rs:556)
        let mut material_ids: HashMap<usize, u32> = HashMap::new();
        // one NraysMesh per distinct (vertices, indices) pair: the loader gives every OBJ group its own TriMesh over the
        // shared vertex array (loader3d.rs:690-695)
        let mut mesh_ids: HashMap<(usize, usize), i32> = HashMap::new();

        for (i, node) in self.nodes().iter().enumerate() { // Vec<Arc<SceneNode>> kept by Scene::new (patch)
            let mkey = &**node.material as *const _ as *const u8 as usize;
            let material_id = match material_ids.get(&mkey) {
                Some(id) => *id,
                None => {
                    let d = match node.material.flatten() {
                        Some(d) => d,
                        None => return Err(format!("node {}: this Material implementation has no GPU form", i)),
                    };
                    let rec = NraysMaterial {
                        kind: d.kind,
                        ambiant: d.ambiant,
                        diffuse: d.diffuse,
                        specular: d.specular,
                        shininess: d.shininess,
                        texture_id: flat.textures.id_of(&d.texture),
                        alpha_texture_id: flat.textures.id_of(&d.alpha),
                    };
                    flat.materials.push(rec);
                    let id = flat.materials.len() as u32 - 1;
                    material_ids.insert(mkey, id);
                    id
                }
            };
            let (shape_kind, params, mesh_id) = match node.shape {
                ShapeDesc::Ball { radius } => (NRAYS_SHAPE_BALL, [radius, 0.0, 0.0], -1),
                ShapeDesc::Cuboid { half_extents: h } => (NRAYS_SHAPE_CUBOID, [h.x, h.y, h.z], -1),
                ShapeDesc::Cylinder { half_height, radius } => (NRAYS_SHAPE_CYLINDER, [half_height, radius, 0.0], -1),
                ShapeDesc::Capsule { half_height, radius } => (NRAYS_SHAPE_CAPSULE, [half_height, radius, 0.0], -1),
                ShapeDesc::Cone { half_height, radius } => (NRAYS_SHAPE_CONE, [half_height, radius, 0.0], -1),
                ShapeDesc::Plane { normal: n } => (NRAYS_SHAPE_PLANE, [n.x, n.y, n.z], -1),
                ShapeDesc::TriMesh(ref m) => {
                    let key = (&**m.vertices as *const Vec<Point3<Scalar>> as usize, &**m.indices as *const Vec<Point3<usize>> as usize);
                    let id = match mesh_ids.get(&key) {
                        Some(id) => *id,
                        None => {
                            // Point3<usize> -> 3 x u32 (the ABI's index type); Point3<f64> / Point2<f64> are packed f64
                            let idx: Vec<u32> = m.indices.iter().flat_map(|t| vec![t.x as u32, t.y as u32, t.z as u32]).collect();
                            flat.mesh_indices.push(idx);
                            flat.mesh_keep.push(m.clone());
                            let kept = flat.mesh_keep.last().unwrap();
                            flat.meshes.push(NraysMesh {
                                num_vertices: kept.vertices.len() as u32,
                                num_triangles: kept.indices.len() as u32,
                                vertices: kept.vertices.as_ptr() as *const f64,
                                uvs: match kept.uvs { Some(ref uv) => uv.as_ptr() as *const f64, None => ptr::null() },
                                indices: flat.mesh_indices.last().unwrap().as_ptr(),
                            });
                            let id = flat.meshes.len() as i32 - 1;
                            mesh_ids.insert(key, id);
                            id
                        }
                    };
                    (NRAYS_SHAPE_TRIMESH, [0.0; 3], id)
                }
            };
            let t = node.transform.translation.vector;
            let w = node.transform.rotation.scaled_axis(); // Isometry3::new(t, axisangle) round trip (loader3d.rs:552)
            flat.nodes.push(NraysNode {
                shape_kind: shape_kind,
                solid: node.solid as u32,
                params: params,
                translation: [t.x, t.y, t.z],
                axis_angle: [w.x, w.y, w.z],
                refl_mix: node.refl_mix,
                refl_atenuation: node.refl_atenuation,
                alpha: node.alpha,
                reserved0: 0.0,
                refr_coeff: node.refr_coeff,
                material_id: material_id,
                mesh_id: mesh_id,
            });
        }
        Ok(flat)
    }
}

// ------------------------------------------------------------------------------------------- the drop-in
fn last_error() -> String {
    unsafe { CStr::from_ptr(nrays_last_error()).to_string_lossy().into_owned() }
}

fn params(resolution: &Vless, ray_per_pixel: usize, window_width: Scalar, camera_eye: &Point, projection: &Matrix4<Scalar>) -> NraysRenderParams {
    assert!(ray_per_pixel > 0); // src/scene.rs:37
    let mut m = [0.0f64; 16];
    m.copy_from_slice(projection.as_slice()); // nalgebra stores Matrix4 column-major, which is what the ABI expects
    NraysRenderParams {
        width: resolution.x as u32,
        height: resolution.y as u32,
        ray_per_pixel: ray_per_pixel as u32,
        max_depth: 0, // the energy rule alone, as in the reference (scene.rs:204)
        window_width: window_width,
        camera_eye: [camera_eye.x, camera_eye.y, camera_eye.z],
        inv_proj_view: m,
        seed: 0,
        band_rows: 0,
        band_owner: 0,
        band_owners: 1,
        reserved: 0,
    }
}

/// A scene resident on ONE GPU (the calling thread's current HIP device).
pub struct GpuScene {
    raw: *mut NraysScene,
}
unsafe impl Send for GpuScene {} // one handle must not be used from two threads AT THE SAME TIME (include/nrays_abi.h)

impl GpuScene {
    pub fn new(scene: &Scene) -> Result<GpuScene, String> {
        let v = unsafe { nrays_abi_version() };
        if v != NRAYS_ABI_VERSION { return Err(format!("libnrays_hip.so ABI version {} != {}", v, NRAYS_ABI_VERSION)); }
        let flat = scene.flatten()?;
        let desc = flat.desc();
        let mut raw = ptr::null_mut();
        if unsafe { nrays_scene_create(&desc, &mut raw) } != NRAYS_OK {
            return Err(last_error());
        }
        Ok(GpuScene { raw: raw }) // the library copied everything: `flat` drops here
    }

    pub fn stats(&self) -> NraysStats {
        let mut st = NraysStats::default();
        unsafe { nrays_get_stats(self.raw, &mut st) };
        st
    }
}

impl Drop for GpuScene {
    fn drop(&mut self) {
        unsafe { nrays_scene_destroy(self.raw) }
    }
}

/// Same signature and meaning as `scene::render` (src/scene.rs:29-36), with the scene handle in place of `&Arc<Scene>`.
pub fn render(scene: &GpuScene, resolution: &Vless, ray_per_pixel: usize, window_width: Scalar, camera_eye: Point, projection: Matrix4<Scalar>) -> Image {
    let p = params(resolution, ray_per_pixel, window_width, &camera_eye, &projection);
    println!("Tracing {} rays.", (resolution.y * resolution.x * (ray_per_pixel as f64)) as i32);
    // Vector3<f32> is three packed f32: the frame is written straight into Image's pixel vector,
    // index i + j * resx (src/scene.rs:104)
    let mut px: Vec<Vector3<f32>> = vec![Vector3::new(0.0f32, 0.0, 0.0); (p.width as usize) * (p.height as usize)];
    if unsafe { nrays_render(scene.raw, &p, px.as_mut_ptr() as *mut f32) } != NRAYS_OK {
        panic!("nrays_render: {}", last_error());
    }
    Image::new(resolution.clone(), px)
}

/// The same frame already quantised as `Image::to_png` quantises it (src/image.rs:66-76: c * 255, clamped, truncated),
/// row-major RGB bytes — what loader3d hands to the PNG encoder, a quarter of the bytes over PCIe.
pub fn render_rgb8(scene: &GpuScene, resolution: &Vless, ray_per_pixel: usize, window_width: Scalar, camera_eye: Point, projection: Matrix4<Scalar>) -> Vec<u8> {
    let p = params(resolution, ray_per_pixel, window_width, &camera_eye, &projection);
    let mut px: Vec<u8> = vec![0u8; (p.width as usize) * (p.height as usize) * 3];
    if unsafe { nrays_render_rgb8(scene.raw, &p, px.as_mut_ptr()) } != NRAYS_OK {
        panic!("nrays_render_rgb8: {}", last_error());
    }
    px
}

/// A scene replicated on `num_gpus` GPUs of this node, driven by this one process (framebuffer bands + RCCL exchange
/// inside the library).
pub struct GpuSceneSet {
    comm: *mut NraysComm,
    set: *mut NraysSceneSet,
}
unsafe impl Send for GpuSceneSet {}

impl GpuSceneSet {
    pub fn new(scene: &Scene, num_gpus: u32) -> Result<GpuSceneSet, String> {
        let flat = scene.flatten()?;
        let desc = flat.desc();
        let mut comm = ptr::null_mut();
        if unsafe { nrays_comm_create_local(num_gpus, ptr::null(), &mut comm) } != NRAYS_OK {
            return Err(last_error());
        }
        let mut set = ptr::null_mut();
        if unsafe { nrays_scene_set_create(&desc, comm, &mut set) } != NRAYS_OK {
            let e = last_error();
            unsafe { nrays_comm_destroy(comm) };
            return Err(e);
        }
        Ok(GpuSceneSet { comm: comm, set: set })
    }
}

impl Drop for GpuSceneSet {
    fn drop(&mut self) {
        unsafe {
            nrays_scene_set_destroy(self.set);
            nrays_comm_destroy(self.comm);
        }
    }
}

/// `scene::render` on every GPU of the set; the frame is bit-identical to the single-GPU one.
pub fn render_multi(scene: &GpuSceneSet, resolution: &Vless, ray_per_pixel: usize, window_width: Scalar, camera_eye: Point, projection: Matrix4<Scalar>) -> Image {
    let p = params(resolution, ray_per_pixel, window_width, &camera_eye, &projection);
    let mut px: Vec<Vector3<f32>> = vec![Vector3::new(0.0f32, 0.0, 0.0); (p.width as usize) * (p.height as usize)];
    if unsafe { nrays_render_multi(scene.set, &p, px.as_mut_ptr() as *mut f32) } != NRAYS_OK {
        panic!("nrays_render_multi: {}", last_error());
    }
    Image::new(resolution.clone(), px)
}
