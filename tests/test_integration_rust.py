"""The Rust side of the drop-in (integration/rust/) cannot be compiled here (no Rust toolchain), so
these tests keep it honest mechanically: the `#[repr(C)]` structs of gpu_ffi.rs list exactly the fields of
include/nrays_abi.h in the same order, every entry point of the header is declared, every type gpu.rs relies on is
defined, and the patch applies cleanly to the reference checkout where it is mounted."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST = os.path.join(ROOT, "integration", "rust")
HEADER = open(os.path.join(ROOT, "include", "nrays_abi.h")).read()
FFI = open(os.path.join(RUST, "src", "gpu_ffi.rs")).read()
GPU = open(os.path.join(RUST, "src", "gpu.rs")).read()


def _c_structs():
    out = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \1;", HEADER, re.S):
        body = re.sub(r"/\*.*?\*/", "", m.group(2), flags=re.S)
        out[m.group(1)] = [re.sub(r"\[.*", "", f.strip().split()[-1]).lstrip("*") for f in body.split(";") if f.strip()]
    return out


def _rust_structs():
    out = {}
    for m in re.finditer(r"pub struct (\w+) \{(.*?)\n\}", FFI, re.S):
        out[m.group(1)] = re.findall(r"pub (\w+):", m.group(2))
    return out


def test_repr_c_structs_match_the_header_field_for_field():
    c, r = _c_structs(), _rust_structs()
    assert set(c) == {"NraysLight", "NraysTexture", "NraysMaterial", "NraysMesh", "NraysNode", "NraysSceneDesc", "NraysRenderParams", "NraysStats", "NraysCastResult", "NraysTileCosts", "NraysMultiTimings", "NraysBlasDump"}
    for name, fields in c.items():
        assert r.get(name) == fields, (name, fields, r.get(name))
        assert re.search(r"#\[repr\(C\)\]\s*(#\[derive[^\]]*\]\s*)?pub struct %s " % name, FFI), name


def test_every_entry_point_of_the_header_is_declared():
    c_fns = set(re.findall(r"\b(nrays_\w+)\s*\(", re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)))
    rust_fns = set(re.findall(r"pub fn (nrays_\w+)\(", FFI))
    assert c_fns == rust_fns, (c_fns ^ rust_fns)
    for enum in re.findall(r"(NRAYS_(?:ERR|SHAPE|MAT|TEXEL|INTERP|OVERFLOW)_\w+|NRAYS_OK) = (-?\d+)", HEADER):
        assert re.search(r"pub const %s: \w+ = %s;" % enum, FFI), enum


def test_gpu_rs_defines_what_it_uses():
    for item in ("pub enum ShapeDesc", "pub struct MeshData", "pub trait FlattenShape", "pub struct MaterialDesc", "pub struct TextureTable",
                 "pub struct FlatScene", "pub fn flatten(&self) -> Result<FlatScene, String>", "pub struct GpuScene", "pub fn render(",
                 "pub struct GpuSceneSet", "pub fn render_multi("):
        assert item in GPU, item
    for shape in ("Ball", "Cuboid", "Cylinder", "Capsule", "Cone", "Plane", "TriMesh"):  # loader3d.rs:601-695
        assert "impl FlattenShape for %s<Scalar>" % shape in GPU, shape
    patch = open(os.path.join(RUST, "patches", "0001-gpu-trace-loop.patch")).read()
    for hook in ("+ FlattenShape,", "shape: geometry.shape_desc()", "fn flatten(&self) -> Option<MaterialDesc>", "pub fn nodes(&self)",
                 "pub fn background(&self)", "pub fn data(&self) -> &Arc<ImageData>", "NRAYS_GPUS", 'build   = "build.rs"'):
        assert hook in patch, hook
    # every ffi name gpu.rs calls exists in gpu_ffi.rs
    for name in set(re.findall(r"\b(nrays_\w+)\(", GPU)):
        assert "pub fn %s(" % name in FFI, name


@pytest.mark.skipif(not os.path.isdir("/root/reference/src") or shutil.which("patch") is None, reason="the reference checkout is only mounted in the build container")
def test_patch_applies_to_the_reference_checkout(tmp_path):
    work = tmp_path / "nrays"
    shutil.copytree("/root/reference", str(work), ignore=shutil.ignore_patterns(".git"))
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(RUST, "patches", "0001-gpu-trace-loop.patch")], cwd=str(work),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAILED" not in r.stdout and "fuzz" not in r.stdout
