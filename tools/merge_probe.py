"""One-off probe (GPU box): what would ONE bounding-volume tree over all the meshes of an untransformed scene buy against the reference's two levels (a tree over the
scene nodes, one tree per mesh)?  The sponza stand-in with every group on ONE opaque material (so that both forms trace the same rays), as 276 nodes and as 1 node."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import nrays_amd as nr
from nrays_amd import abi
from tools import scenes_util as su, standins
lib = abi.load_hip_lib()

def scene(merged, n_lights):
    pts, uvs, groups, defs, tex = standins.sponza_geometry(1.0)
    pts = standins._f32(pts * 0.25)
    ka, kd, ks, t, a, ns, d = defs["bricks"]
    mat = nr.PhongMaterial(ka, kd, ks, tex[t] if t else None, None, ns)
    iso = nr.Isometry3((0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
    if merged:
        nodes = [nr.SceneNode(mat, 0.0, 0.0, 1.0, 1.0, iso, nr.TriMesh(pts, np.concatenate([g[1] for g in groups]), uvs))]
    else:
        nodes = [nr.SceneNode(mat, 0.0, 0.0, 1.0, 1.0, iso, nr.TriMesh(pts, idx, uvs)) for _, idx in groups]
    sc0, cam = standins.sponza_scene(n_lights=n_lights)
    return nr.Scene(nodes, sc0.lights(), (1, 1, 1)), cam

for n_lights in (1, 8):
    imgs = []
    for merged in (False, True):
        sc, cam = scene(merged, n_lights)
        p, _ = su.camera_params(cam, 1920, 1080)
        out = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda")
        h = sc.device_handle()
        abi.check(lib.nrays_render_device_instrumented(h, C.byref(p), C.c_void_p(out.data_ptr()), None))
        st = nr.get_stats(sc)
        for _ in range(6): abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(out.data_ptr()), None))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): abi.check(lib.nrays_render_device(h, C.byref(p), C.c_void_p(out.data_ptr()), None))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        imgs.append(out.cpu().numpy().copy())
        print(json.dumps({"lights": n_lights, "merged": merged, "ms": round(dt * 1e3, 4), "rays": st.total_rays(),
                          "node_per_ray": round(st.node_tests / st.total_rays(), 1), "tri_per_ray": round(st.tri_tests / st.total_rays(), 2)}), flush=True)
    print("max |diff| between the two forms:", float(np.abs(imgs[0] - imgs[1]).max()), "pixels differing:", int((imgs[0] != imgs[1]).any(axis=2).sum()))
