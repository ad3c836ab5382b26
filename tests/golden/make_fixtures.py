"""Regenerates the committed fixtures under tests/golden/.  Run in the BUILD container only
(it reads /root/reference/Models/cornell-box.obj, which does not exist on the GPU box):

    python tests/golden/make_fixtures.py

Writes
  cornell_box.json        the arrays the product's loader produces from the reference's own data
                          file Models/cornell-box.obj (+ .mtl): translated vertices, normals,
                          triangles, materials, lights, vertex_min, camera after load.  DATA only.
  cornell_bvh.json        SBVH + CWBVH of that scene from the product's builders (the BVH2 part is
                          pinned by SURVEY.md §8c known answers, see tests/test_host.py::test_sbvh_cornell_known_answers and ::test_sbvh_reproduces_the_survey_probes_with_spatial_splits).
  oracle_vectors.npz      seeded rays and the oracle's answers (ids, t/u/v bit patterns, visit
                          counters, frame sums) on Cornell and the n=8 tessellation; catches drift
                          of the oracle across compilers/machines.
survey_known_answers.json is NOT generated: it is a transcription of SURVEY.md §8c (values the
survey captured from the reference's own sbvh.h, Rnd.h and data files).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

import caitlynrenderer_amd as cr                                  # noqa: E402
from caitlynrenderer_amd.meshgen import tessellated_cornell       # noqa: E402
from oracle import binding as ob                                  # noqa: E402

REF_OBJ = "/root/reference/Models/cornell-box.obj"
CAM_POS, CAM_LOOK, CAM_FOV = (-2.755610, 2.745992, 7.58545), (-2.755610, 2.745992, 6.58545), 40.0   # Scene.h:468


def seeded_rays(mesh, n, seed):
    """Random rays inside the scene's bounds: origins uniform in the box, directions uniform on the sphere."""
    rng = np.random.default_rng(seed)
    lo, hi = mesh.vertices.min(0), mesh.vertices.max(0)
    rays = np.zeros(n, ob.RAY_DT)
    rays["o"] = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3))
    rays["d"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["tmax"] = np.float32(1e9)
    return rays


def main():
    cam = cr.Camera(CAM_POS, CAM_LOOK, CAM_FOV)
    mesh = cr.Mesh.read_object(REF_OBJ, cam)
    f32 = lambda a: [float(x) for x in np.asarray(a, np.float32).ravel()]
    camera = {"position": f32(cam.position), "right": f32(cam.right), "up": f32(cam.up), "forward": f32(cam.forward), "fov": cam.fov}
    with open(os.path.join(HERE, "cornell_box.json"), "w") as f:
        json.dump({
            "source": "Models/cornell-box.obj + cornell-box.mtl through caitlynrenderer_amd.Mesh.read_object",
            "vertices": f32(mesh.vertices), "normals": f32(mesh.normals),
            "triangles": [int(x) for x in mesh.triangles.ravel()],
            "materials": f32(mesh.materials), "lights": f32(mesh.lights),
            "vertex_min": f32(mesh.vertex_min), "camera": camera,
        }, f)
    data = cr.SceneData.build(mesh, cam)
    with open(os.path.join(HERE, "cornell_bvh.json"), "w") as f:
        json.dump({
            "flat_nodes_bits": [int(x) for x in data.bvh.view(np.uint32).ravel()],
            "triangle_indices": [int(x) for x in data.tri_orig_ids],
            "bvh8_bytes": [int(x) for x in data.bvh8.ravel()],
            "bvh8_tri_slots": [int(x) for x in data.bvh8_tri_slots],
        }, f)

    out = {}
    mesh8 = tessellated_cornell(mesh, 8)
    data8 = cr.SceneData.build(mesh8, cam)
    for tag, m, d in (("cornell", mesh, data), ("tess8", mesh8, data8)):
        o = ob.Oracle(d, 96, 54, 3)
        rays = seeded_rays(m, 4096, 7)
        h, st = o.trace(rays, ob.BVH8, ob.CLOSEST, ob.TIE_LOWEST_ID, stats=True)
        rays_any = rays.copy()
        rays_any["tmax"] = np.float32(2.5)
        ha = o.trace(rays_any, ob.BVH8, ob.ANY)
        out[f"{tag}_rays"] = rays.view(np.uint32).reshape(-1, 8)
        out[f"{tag}_hits"] = h.view(np.uint32).reshape(-1, 4)
        out[f"{tag}_stats"] = st.view(np.uint16).reshape(-1, 2)
        out[f"{tag}_any"] = (ha["tri"] >= 0).astype(np.uint8)
        rnd = cr.Rnd()
        s = np.zeros((54, 96, 3), np.float32)
        for _ in range(2):
            rx, ry = rnd.randf2(), rnd.randf2()
            o.render_frame(rx, ry, s)
        out[f"{tag}_sum2_bits"] = s.view(np.uint32)
    out["rand_seq"] = np.array(ob.rand_sequence(960, 540, 0.6591631, 0.910802, 16), np.float32).view(np.uint32)
    xs = np.float32(np.linspace(-2.0e5, 2.0e5, 4001))
    out["sin_x"] = xs.view(np.uint32)
    out["sin_y"] = np.array([ob.lib().orc_sin(float(x)) for x in xs], np.float32).view(np.uint32)
    out["cos_y"] = np.array([ob.lib().orc_cos(float(x)) for x in xs], np.float32).view(np.uint32)
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
