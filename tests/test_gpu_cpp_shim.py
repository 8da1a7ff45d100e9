"""The header-only C++ mirror (include/szg/pipelines.hpp): a Renderer::recordDraw-style C++
caller renders a frame through the C-ABI; the result must equal the oracle's on the same
scene."""
import os
import subprocess
import sys

import numpy as np
import pytest


pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_cpp_record_draw_matches_oracle(tmp_path):
    import ctypes as C

    from oracle import binding as ob
    from syzygy_amd import abi, lib, scene

    exe = os.path.join(HERE, "cpp", "record_draw")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(HERE, "cpp")], check=True)
    W, H = 200, 120
    out = tmp_path / "frame.bin"
    r = subprocess.run([exe, str(out), str(W), str(H)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, dtype=np.uint16).reshape(H, W, 4)

    # the same scene through the Python host prep + the oracle
    cam = scene.camera_packed(scene.default_camera(), np.float32(W) / np.float32(H))
    a = scene.default_atmosphere()
    a.sunEulerAngles[0] = np.float32(np.float32(3.14159265358979) + np.float32(35.0) * np.float32(3.14159265358979) / np.float32(180.0))
    atm, sun, moon = scene.atmosphere_baked(a, scene.aabb((0.0, -7.0, 39.0), (64.0, 8.0, 46.0)))
    spot = scene.make_spot((1, 0, 0), (-20.0, -28.0, -20.0), scene.eulers_from_forward((20.0, 20.0, 20.0)))
    spots = (abi.SpotLightPacked * 1)(spot)
    boxes = (abi.FillBox * 2)()
    boxes[0].center[:] = [0.0, -8.0, 6.0]
    boxes[1].center[:] = [0.0, -8.0, -6.0]
    for b, metallic in zip(boxes, (0.0, 1.0)):
        b.half_extent[:] = [5.0, 5.0, 5.0]
        b.metallic = metallic
        b.roughness = 60.0 / 255.0
    fill = abi.FillScene(-1.0, 4000.0, 4.0, 60.0 / 255.0, 2, 0, C.cast(boxes, C.POINTER(abi.FillBox)))
    rect = abi.Rect(0, 0, W, H)
    frame = ob.HostFrame(W, H)
    dirs = (abi.DirectionalLightPacked * 2)(sun, moon)
    ob.gbuffer_fill(frame, rect, None, cam, fill, threads=8)
    ob.lights(frame, rect, None, None, cam, dirs, 2, 1, spots, 1, threads=8)
    tl = ob.transmittance_lut(atm, 512, 128, threads=8)
    sl = ob.skyview_lut(atm, cam, tl, 2048, 1024, threads=16)
    ob.composite(frame, rect, None, None, atm, cam, dirs, 0, tl, sl, threads=8)
    assert np.abs(got.astype(np.int32) - frame.color.astype(np.int32)).max() <= 1
    assert (got[..., 3] == 65535).all() and got[..., :3].any()


def test_cpp_record_draw_with_real_meshes_matches_oracle(tmp_path):
    """The same C++ caller with the reference's own argument, a span of mesh instances (the editor's start-up scene built
    in C++): shadow raster into 512^2 maps, G-buffer raster, lights, sky-view pipeline — against the oracle chain."""
    import ctypes as C

    from oracle import binding as ob
    from syzygy_amd import abi, lib, meshes, scene

    exe = os.path.join(HERE, "cpp", "record_draw")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(HERE, "cpp")], check=True)
    W, H, DIM = 200, 120, 512
    out = tmp_path / "frame_meshes.bin"
    r = subprocess.run([exe, str(out), str(W), str(H), "meshes"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, dtype=np.uint16).reshape(H, W, 4)

    cam = scene.camera_packed(scene.default_camera(), np.float32(W) / np.float32(H))
    a = scene.default_atmosphere()
    a.sunEulerAngles[0] = np.float32(np.float32(3.14159265358979) + np.float32(35.0) * np.float32(3.14159265358979) / np.float32(180.0))
    atm, sun, moon = scene.atmosphere_baked(a, scene.aabb((0.0, -7.0, 39.0), (64.0, 8.0, 46.0)))
    spot = scene.make_spot((1, 0, 0), (-20.0, -28.0, -20.0), scene.eulers_from_forward((20.0, 20.0, 20.0)))
    spots = (abi.SpotLightPacked * 1)(spot)
    ms = meshes.reference_default_scene()
    maps = []
    for light in (sun, moon, spot):
        pv = abi.Mat4()
        lib().szg_mat4_mul(C.byref(light.projection), C.byref(light.view), C.byref(pv))
        maps.append(ob.shadow_raster(pv, DIM, ms, threads=8))
    images = (abi.Image * len(maps))(*[ob.host_image(m, abi.SZG_FORMAT_D32_SFLOAT) for m in maps])
    host_maps = abi.ShadowMaps(len(maps), 0, C.cast(images, C.POINTER(abi.Image)))
    rect = abi.Rect(0, 0, W, H)
    frame = ob.HostFrame(W, H)
    dirs = (abi.DirectionalLightPacked * 2)(sun, moon)
    ob.gbuffer_raster(frame, rect, None, cam, ms, threads=8)
    ob.lights(frame, rect, None, host_maps, cam, dirs, 2, 1, spots, 1, threads=8)
    tl = ob.transmittance_lut(atm, 512, 128, threads=8)
    sl = ob.skyview_lut(atm, cam, tl, 2048, 1024, threads=16)
    ob.composite(frame, rect, None, host_maps, atm, cam, dirs, 0, tl, sl, threads=8)
    assert (frame.depth > 0).mean() > 0.1
    assert np.abs(got.astype(np.int32) - frame.color.astype(np.int32)).max() <= 1


def test_cpp_asset_library_gltf_to_frame_matches_oracle(tmp_path):
    """An engine-style C++ caller end to end (include/szg/assets.hpp): AssetLibrary::loadDefaultAssets + loadGLTFFromPath on a
    GLB written to disk, MeshInstanced with two transforms per loaded mesh and the built-in plane as the floor,
    recordDrawCommands + the sky-view pipeline — against the oracle chain fed by the Python mirror of the same loader."""
    import ctypes as C

    from oracle import binding as ob
    from syzygy_amd import abi, assets, lib, meshes, scene
    from tests import gltf_writer as gw

    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "cpp"), "record_draw"], check=True)
    exe = os.path.join(HERE, "cpp", "record_draw")
    rng = np.random.default_rng(5)
    pos, nrm, uv, idx = gw.uv_sphere(10, 20)
    b = gw.GltfBuilder()
    y, x = np.mgrid[0:16, 0:32]
    color = np.stack([(x * 8) & 255, (y * 16) & 255, ((x ^ y) * 16) & 255, np.full_like(x, 255)], -1).astype(np.uint8)
    t_color = b.texture(b.image_uri(gw.data_uri_png(gw.png_rgba8(color))))
    t_mr = b.texture(b.image_view(gw.png_rgba8(rng.integers(0, 256, (4, 4, 4), dtype=np.uint8))))
    b.doc["materials"] = [{"name": "painted", "pbrMetallicRoughness": {"baseColorTexture": {"index": t_color},
                                                                       "metallicRoughnessTexture": {"index": t_mr}}}]
    attrs = {"POSITION": b.accessor(pos), "NORMAL": b.accessor(nrm), "TEXCOORD_0": b.accessor(uv)}
    b.doc["meshes"] = [{"name": "Sphere", "primitives": [{"attributes": attrs, "indices": b.accessor(idx.astype(np.uint16)), "material": 0}]},
                       {"name": "Sphere", "primitives": [{"attributes": attrs, "indices": b.accessor(idx[: len(idx) // 2])}]}]
    path = tmp_path / "two spheres.glb"
    path.write_bytes(b.glb())

    W, H, DIM = 200, 120, 512
    out = tmp_path / "frame_gltf.bin"
    flags = abi.SZG_GLTF_DECODE_BUFFER_VIEW_IMAGES
    r = subprocess.run([exe, str(out), str(W), str(H), "gltf", str(path), str(flags)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    # names are deduplicated like the reference's (assets.cpp:1678-1692)
    assert "mesh_Plane 4 vertices 1 surfaces" in r.stdout and "mesh_Cube 24 vertices" in r.stdout
    assert "mesh_Sphere 231 vertices 1 surfaces" in r.stdout and "mesh_Sphere_2 231 vertices" in r.stdout
    got = np.fromfile(out, dtype=np.uint16).reshape(H, W, 4)

    a = assets.load_gltf(str(path), flags)
    ms = []
    for k in range(2):
        models = [meshes.transform_matrix((-3.0 + 8.0 * k, -6.0, 2.0), (0.3, 0.2, 0.1), (4, 4, 4)),
                  meshes.transform_matrix((6.0, -4.0, 8.0 + 4.0 * k), (0.0, 1.0, 0.0), (3, 5, 3))]
        ms.append(a.instanced(k, models))
    ms.append(meshes.reference_default_scene()[2])

    cam = scene.camera_packed(scene.default_camera(), np.float32(W) / np.float32(H))
    atmosphere = scene.default_atmosphere()
    atmosphere.sunEulerAngles[0] = np.float32(np.float32(3.14159265358979) + np.float32(35.0) * np.float32(3.14159265358979) / np.float32(180.0))
    atm, sun, moon = scene.atmosphere_baked(atmosphere, scene.aabb((0.0, -7.0, 39.0), (64.0, 8.0, 46.0)))
    spot = scene.make_spot((1, 0, 0), (-20.0, -28.0, -20.0), scene.eulers_from_forward((20.0, 20.0, 20.0)))
    spots = (abi.SpotLightPacked * 1)(spot)
    maps = []
    for light in (sun, moon, spot):
        pv = abi.Mat4()
        lib().szg_mat4_mul(C.byref(light.projection), C.byref(light.view), C.byref(pv))
        maps.append(ob.shadow_raster(pv, DIM, ms, threads=8))
    images = (abi.Image * len(maps))(*[ob.host_image(m, abi.SZG_FORMAT_D32_SFLOAT) for m in maps])
    host_maps = abi.ShadowMaps(len(maps), 0, C.cast(images, C.POINTER(abi.Image)))
    rect = abi.Rect(0, 0, W, H)
    frame = ob.HostFrame(W, H)
    dirs = (abi.DirectionalLightPacked * 2)(sun, moon)
    ob.gbuffer_raster(frame, rect, None, cam, ms, threads=8)
    ob.lights(frame, rect, None, host_maps, cam, dirs, 2, 1, spots, 1, threads=8)
    tl = ob.transmittance_lut(atm, 512, 128, threads=8)
    sl = ob.skyview_lut(atm, cam, tl, 2048, 1024, threads=16)
    ob.composite(frame, rect, None, host_maps, atm, cam, dirs, 0, tl, sl, threads=8)
    sphere = (frame.planes()["occlusionRoughnessMetallic"][..., 1] != np.float16(60 / 255)) & (frame.depth > 0)
    assert sphere.mean() > 0.03
    assert np.abs(got.astype(np.int32) - frame.color.astype(np.int32)).max() <= 1


def test_cpp_engine_frame_scene_and_renderer_match_oracle(tmp_path):
    """The engine's own frame in C++ (include/szg/scene.hpp): AssetLibrary::loadDefaultAssets -> Scene::defaultScene (floor that
    casts no shadow, floating cube, two look-at spot lights of strength 30) -> three ticks (sun animation, spinning cube) ->
    calculateShadowBounds -> Renderer::recordDraw. The same scene is rebuilt here from the C-ABI pieces and rendered by the
    oracle chain."""
    import ctypes as C

    from oracle import binding as ob
    from syzygy_amd import abi, assets, lib, meshes, scene

    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "cpp"), "record_draw"], check=True)
    exe = os.path.join(HERE, "cpp", "record_draw")
    W, H, DIM, TICKS = 200, 120, 512, 3
    out = tmp_path / "frame_scene.bin"
    r = subprocess.run([exe, str(out), str(W), str(H), "scene", str(TICKS)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, dtype=np.uint16).reshape(H, W, 4)
    reported = [float(v) for v in r.stdout.split() if v.replace(".", "").replace("-", "").replace("e", "").replace("+", "").isdigit()]

    L = lib()
    cube = assets.default_mesh(abi.SZG_DEFAULT_MESH_CUBE)
    material = meshes.default_material()

    def transform(t, s):
        out_t = abi.Transform()
        out_t.translation[:], out_t.eulerAnglesRadians[:], out_t.scale[:] = list(t), [0.0, 0.0, 0.0], list(s)
        return out_t

    groups = [  # (transform, animation, casts shadow)  scene.cpp:238-279
        (transform((0.0, 0.0, 0.0), (400.0, 1.0, 400.0)), abi.SZG_INSTANCE_ANIMATION_NONE, False),
        (transform((0.0, -4.0, 0.0), (1.0, 1.0, 1.0)), abi.SZG_INSTANCE_ANIMATION_SPIN_ALONG_WORLD_UP, True),
    ]
    atmosphere = scene.default_atmosphere()
    anim = abi.SunAnimation()
    L.szg_sun_animation_default(C.byref(anim))
    anim.time = 0.6
    state = [((abi.Transform * 1)(g[0]), (abi.Transform * 1)(g[0]), (abi.Mat4 * 1)(), (abi.Mat4 * 1)()) for g in groups]
    elapsed, dt = 0.0, 1.0 / 60.0
    for _ in range(TICKS):
        L.szg_scene_tick_sun(C.byref(anim), C.byref(atmosphere), dt)
        for (t0, animation, _), (orig, cur, models, mits) in zip(groups, state):
            L.szg_tick_mesh_instance(animation, orig, cur, 1, elapsed, dt, models, mits)
        elapsed += dt
    bounds_of_cube = abi.AABB()
    bounds_of_cube.center[:], bounds_of_cube.half_extent[:] = [float(v) for v in cube.bounds[0]], [float(v) for v in cube.bounds[1]]
    casters = (abi.ShadowCaster * 2)(*[abi.ShadowCaster(bounds_of_cube, st[1], 1, 1, int(g[2]), 0) for g, st in zip(groups, state)])
    bounds = abi.AABB()
    L.szg_calculate_shadow_bounds(casters, 2, C.byref(bounds))
    # the C++ side reports its sun angle and shadow bounds: the host layers agree before any pixel is compared
    assert np.allclose(reported[-7:], [atmosphere.sunEulerAngles[0]] + list(bounds.center) + list(bounds.half_extent), rtol=1e-6, atol=1e-6)

    ms = [meshes.MeshInstanced(cube.vertices, cube.indices, [(0, len(cube.indices), material)], [st[2][0]], casts_shadow=g[2])
          for g, st in zip(groups, state)]
    spots = (abi.SpotLightPacked * 2)()
    for k, (sign, color) in enumerate([(1.0, (0.0, 1.0, 0.0, 1.0)), (-1.0, (1.0, 0.0, 0.0, 1.0))]):  # scene.cpp:281-331
        look = abi.Transform()
        L.szg_transform_look_at(abi.f3(sign * 8.0, -12.0, sign * 8.0), abi.f3(0.0, -4.0, 0.0), abi.f3(1.0, 1.0, 1.0), C.byref(look))
        p = abi.SpotlightParams()
        p.color[:] = list(color)
        p.strength, p.falloffFactor, p.falloffDistance, p.verticalFOVDegrees, p.horizontalScale = 30.0, 1.0, 1.0, 60.0, 1.0
        p.eulerAngles[:], p.position[:] = list(look.eulerAnglesRadians), list(look.translation)
        p.near_plane, p.far_plane = 0.1, 1000.0
        L.szg_make_spot(C.byref(p), C.byref(spots[k]))

    cam = scene.camera_packed(scene.default_camera(), np.float32(np.float64(W) / np.float64(H)))
    atm, sun, moon = scene.atmosphere_baked(atmosphere, bounds)
    maps = []
    for light in (sun, moon, spots[0], spots[1]):
        pv = abi.Mat4()
        L.szg_mat4_mul(C.byref(light.projection), C.byref(light.view), C.byref(pv))
        maps.append(ob.shadow_raster(pv, DIM, ms, threads=8))
    images = (abi.Image * len(maps))(*[ob.host_image(m, abi.SZG_FORMAT_D32_SFLOAT) for m in maps])
    host_maps = abi.ShadowMaps(len(maps), 0, C.cast(images, C.POINTER(abi.Image)))
    rect = abi.Rect(0, 0, W, H)
    frame = ob.HostFrame(W, H)
    dirs = (abi.DirectionalLightPacked * 2)(sun, moon)
    ob.gbuffer_raster(frame, rect, None, cam, ms, threads=8)
    ob.lights(frame, rect, None, host_maps, cam, dirs, 2, 1, spots, 2, threads=8)
    tl = ob.transmittance_lut(atm, 512, 128, threads=8)
    sl = ob.skyview_lut(atm, cam, tl, 2048, 1024, threads=16)
    ob.composite(frame, rect, None, host_maps, atm, cam, dirs, 0, tl, sl, threads=8)
    assert (frame.depth > 0).mean() > 0.3
    assert np.abs(got.astype(np.int32) - frame.color.astype(np.int32)).max() <= 1


def test_cpp_tiled_frame_through_the_c_abi_collectives(tmp_path):
    """tests/cpp/record_draw.cpp `tiled`: a C++ caller (no Python, no torch) renders the row-tiled frame as the one rank of a
    world of one - szg_rowtile_comm over RCCL, the sky-view LUT row slice + in-place all-gather, the tile gather to the root,
    the compose kernel - and must produce the plain frame bit for bit."""
    exe = os.path.join(HERE, "cpp", "record_draw")
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "cpp"), "record_draw"], check=True)
    W, H = 200, 120
    plain, tiled = tmp_path / "plain.bin", tmp_path / "tiled.bin"
    r = subprocess.run([exe, str(plain), str(W), str(H)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, str(tiled), str(W), str(H), "tiled"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "ranks 1" in r.stdout
    a = np.fromfile(plain, dtype=np.uint16)
    b = np.fromfile(tiled, dtype=np.uint16)
    assert a.shape == b.shape and (a == b).all()


@pytest.mark.parametrize("ranks,no_gather", [(2, False), (3, False), (3, True)])
def test_cpp_n_rank_tiled_frame_through_the_c_abi_collectives(tmp_path, ranks, no_gather):
    """The same C++ caller as N processes (no Python, no torch inside them): rank 0 publishes the communicator id in a file,
    every rank renders its cyclic 8-row blocks, the sky-view LUT slices and their status words are all-gathered, the tiles
    gathered to rank 0 and composed. The processes share the one GPU of the box, where RCCL cannot run two ranks, so
    tests/cpp/mock_rccl.cpp stands in for librccl.so (SZG_RCCL_LIBRARY); everything else is the product's C-ABI. The frame
    must equal the plain single-process frame bit for bit. With 3 ranks the 1024 rows of the sky-view LUT do not divide: every
    rank then computes the whole LUT and the optional second collective is skipped (pipelines.hpp) - same frame.
    `no_gather` (SZG_RCCL_NO_GATHER): szg_rowtile_gather takes its grouped ncclSend / ncclRecv path - the one a library
    without ncclGather gets - with the root's own tile copied outside the group."""
    exe = os.path.join(HERE, "cpp", "record_draw")
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "cpp"), "record_draw", "libmock_rccl.so"], check=True)
    env = dict(os.environ, SZG_RCCL_LIBRARY=os.path.join(HERE, "cpp", "libmock_rccl.so"), SZG_LOG="1")
    if no_gather:
        env["SZG_RCCL_NO_GATHER"] = "1"
    W, H = 200, 240
    plain, tiled, idfile = tmp_path / "plain.bin", tmp_path / "tiled.bin", tmp_path / "comm.id"
    r = subprocess.run([exe, str(plain), str(W), str(H)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    procs = [subprocess.Popen([exe, str(tiled if k == 0 else tmp_path / f"unused{k}.bin"), str(W), str(H), "tiled", str(k), str(ranks), str(idfile)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(ranks)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se + so
        assert f"ranks {ranks}" in so
        assert ("gather: grouped ncclSend/ncclRecv" if no_gather else "gather: ncclGather") in se, se  # SZG_LOG: what was bound
    a = np.fromfile(plain, dtype=np.uint16)
    b = np.fromfile(tiled, dtype=np.uint16)
    assert a.shape == b.shape and (a == b).all()


def test_comm_create_deadline_and_error_paths_of_the_grouped_gather(tmp_path):
    """(a) A rank whose peers never arrive: szg_rowtile_comm_create_deadline returns SZG_ERR_TIMEOUT after its deadline instead
    of hanging in ncclCommInitRank, and the process then exits non-zero (what makes a launcher tear the job down). Run in a
    child process: the blocked initialisation cannot be cancelled, only left behind by exiting.
    (b) An error inside the grouped send / recv path closes the group: a gather to a root outside the communicator is refused
    up front, and after a failing call the next, valid, gather on the same communicator still works (world of one)."""
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "cpp"), "libmock_rccl.so"], check=True)
    env = dict(os.environ, SZG_RCCL_LIBRARY=os.path.join(HERE, "cpp", "libmock_rccl.so"), SZG_RCCL_NO_GATHER="1")
    code = r"""
import ctypes as C, sys, time
import torch
import syzygy_amd
from syzygy_amd import abi
lib = syzygy_amd.lib()
ident = (C.c_ubyte * 256)()
assert lib.szg_rowtile_comm_unique_id(ident) == 0
# (b) world of one, grouped path
comm = C.c_void_p()
assert lib.szg_rowtile_comm_create_deadline(C.byref(comm), 0, 1, ident, 0, 5000) == 0
tile = torch.arange(4096, dtype=torch.uint8, device="cuda")
out = torch.zeros(4096, dtype=torch.uint8, device="cuda")
assert lib.szg_rowtile_gather(comm, None, tile.data_ptr(), 4096, out.data_ptr(), 3) == abi.SZG_ERR_INVALID_ARGUMENT
assert lib.szg_rowtile_gather(comm, None, tile.data_ptr(), 4096, out.data_ptr(), 0) == 0
torch.cuda.synchronize()
assert bool((out == tile).all())
lib.szg_rowtile_comm_destroy(comm)
# (a) rank 0 of 2, nobody else comes
assert lib.szg_rowtile_comm_unique_id(ident) == 0
t0 = time.time()
rc = lib.szg_rowtile_comm_create_deadline(C.byref(comm), 0, 2, ident, 0, 700)
dt = time.time() - t0
print("rc", rc, "dt", round(dt, 2), lib.szg_last_error().decode())
sys.stdout.flush()
import os
os._exit(3 if rc == abi.SZG_ERR_TIMEOUT and 0.6 < dt < 10 else 0)
"""
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=os.path.dirname(HERE))
    assert r.returncode == 3, r.stdout + r.stderr
    assert "did not join within 700 ms" in r.stdout
