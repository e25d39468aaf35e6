"""Helper process of tests/test_multi_gpu.py (not a test): rank `rank` of `world` on HIP device `device` creates the ranked
communicator from the unique id in `idfile` (rank 0 writes it) and prints one line `rc=<status> <last error>`; with
`--render` it also renders a tiled frame through nrays_render_multi and rank 0 prints whether it equals the direct render."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    idfile, world, rank, device = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    import torch
    torch.cuda.set_device(device)
    from nrays_amd import abi, tiling
    lib = abi.load_hip_lib()
    if rank == 0:
        uid = tiling.unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 60:
                print("rc=timeout no id file", flush=True)
                return
            time.sleep(0.05)
        uid = open(idfile, "rb").read()
    comm = C.c_void_p()
    buf = (C.c_uint8 * abi.UNIQUE_ID_BYTES)(*uid)
    rc = lib.nrays_comm_create(buf, world, rank, C.byref(comm))
    print("rc=%d %s" % (rc, lib.nrays_last_error().decode(errors="replace") if rc != 0 else ""), flush=True)
    if rc != 0 or "--render" not in sys.argv:
        if rc == 0:
            lib.nrays_comm_destroy(comm)
        return
    import numpy as np
    from tools import scenes_util as su
    sc, cam = su.mesh_scene()
    p, _ = su.camera_params(cam, 200, 117)
    ss = tiling.SceneSet(sc.descriptor, comm)
    out = np.empty((117, 200, 3), np.float32) if rank == 0 else None
    for _ in range(3):
        abi.check(lib.nrays_render_multi(ss._h, C.byref(p), out.ctypes.data_as(C.POINTER(C.c_float)) if rank == 0 else None))
    if rank == 0:
        ref = np.empty((117, 200, 3), np.float32)
        abi.check(lib.nrays_render(sc.device_handle(), C.byref(p), ref.ctypes.data_as(C.POINTER(C.c_float))))
        print("identical=%d" % int(np.array_equal(out, ref)), flush=True)
    ss.close()
    lib.nrays_comm_destroy(comm)


if __name__ == "__main__":
    main()
