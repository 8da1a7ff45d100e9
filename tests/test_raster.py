"""Real-mesh G-buffer / shadow raster (include/szg/raster.h, SURVEY 8 f4 + f3).

PARITY UNPINNED against the reference (fixed-function rasterisation is implementation-defined at the bit level and the
reference holds no fixtures for it). What is checked instead:
  * CPU: the oracle rasteriser against facts that do not come from its own code — the analytic ray-cast fill of the same
    boxes (coverage, depth, position, ORM), watertightness / no double hits of shared edges, top-left ownership, facing and
    culling, closed-form depths, the conventions of offscreen.frag:61-79.
  * GPU (-m gpu): the HIP rasteriser against the oracle, bit for bit, through the C-ABI: analytic-scene meshes, the
    reference's default scene, a triangle soup with primitives crossing the eye plane, row tiles, shadow maps, and the whole
    frame (raster -> lights -> composite).
"""
import ctypes as C

import numpy as np
import pytest

from tests import util
from oracle import binding as ob
from syzygy_amd import abi, meshes


def _planes_equal(got, want):
    """Bit equality; a NaN equals a NaN whatever its payload (x86 and gfx950 differ in the sign of a generated NaN)."""
    for name in ("diffuse", "specular", "normal", "worldPosition", "occlusionRoughnessMetallic"):
        a, b = got[name], want[name]
        bits = np.uint16 if a.dtype == np.float16 else np.uint32
        same = (a.view(bits) == b.view(bits)) | (np.isnan(a) & np.isnan(b))
        assert same.all(), f"{name}: {(~same).sum()} of {same.size} values differ"


def _soup(seed, count, spread=40.0):
    """Random triangles around the default camera, many of them crossing the eye plane or the frustum sides."""
    rng = np.random.default_rng(seed)
    centers = rng.uniform(-spread, spread, (count, 1, 3)).astype(np.float32)
    centers[..., 1] = rng.uniform(-30.0, 5.0, (count, 1))
    pts = (centers + rng.normal(0.0, 6.0, (count, 3, 3))).astype(np.float32)
    v = np.zeros(count * 3, abi.VERTEX_DTYPE)
    v["position"] = pts.reshape(-1, 3)
    n = np.cross(pts[:, 1] - pts[:, 0], pts[:, 2] - pts[:, 0])
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
    v["normal"] = np.repeat(n, 3, axis=0)
    uv = rng.uniform(-2.0, 3.0, (count * 3, 2)).astype(np.float32)
    v["uv_x"], v["uv_y"] = uv[:, 0], uv[:, 1]
    v["color"] = 1.0
    idx = np.arange(count * 3, dtype=np.uint32)
    # both windings of every triangle, so that culling keeps one of them wherever it faces
    idx = np.concatenate([idx, idx.reshape(-1, 3)[:, ::-1].reshape(-1)])
    rngt = np.random.default_rng(seed + 1)
    mat = {"color": (rngt.integers(0, 256, (16, 8, 4), dtype=np.uint8), True),
           "normal": (rngt.integers(64, 192, (8, 8, 4), dtype=np.uint8), False),
           "orm": (rngt.integers(0, 256, (4, 4, 4), dtype=np.uint8), False)}
    models = [meshes.transform_matrix((0, 0, 0), (0, 0, 0), (1, 1, 1)), meshes.transform_matrix((3, -2, 1), (0.3, 0.2, 0.1), (1.5, 0.5, 1))]
    half = len(idx) // 2
    return [meshes.MeshInstanced(v, idx, [(0, half, mat), (half, half, meshes.default_material())], models, name="soup")]


# ---------------------------------------------------------------------------
# CPU: the oracle rasteriser itself
# ---------------------------------------------------------------------------
def test_oracle_raster_matches_the_analytic_fill():
    W, H = 320, 180
    inp = util.Inputs(W, H)
    ms = meshes.meshes_of_fill_scene(inp.synthetic.fill)
    fr, fa = ob.HostFrame(W, H), ob.HostFrame(W, H)
    ob.gbuffer_raster(fr, inp.rect, None, inp.cam, ms, threads=8)
    ob.gbuffer_fill(fa, inp.rect, None, inp.cam, inp.synthetic.fill, threads=8)
    cr, ca = fr.depth > 0, fa.depth > 0
    assert 0.3 < cr.mean() < 0.9
    assert (cr != ca).mean() < 2e-3  # silhouettes may differ by a pixel; on this scene they do not
    both = cr & ca
    assert np.abs(fr.depth - fa.depth)[both].max() < 2e-6
    rel = np.abs(fr.depth - fa.depth)[both] / fa.depth[both]
    assert rel.max() < 5e-4
    dpos = np.abs(fr.position[..., :3] - fa.position[..., :3])[both]
    dist = np.linalg.norm(fa.position[..., :3] - np.array(inp.cam.position[:3], np.float32), axis=-1)[both]
    assert (dpos.max(axis=-1) / dist).max() < 2e-3  # interpolated vs ray-cast positions
    # the flat default normal map tilts the normal by 1/127 (offscreen.frag:47): compare up to that
    dn = np.abs(fr.normal[..., :3].astype(np.float32) - fa.normal[..., :3].astype(np.float32))[both]
    assert dn.max() < 0.012
    assert (fr.orm.view(np.uint16) == fa.orm.view(np.uint16))[both].all()
    # offscreen.frag:61-79 conventions
    assert (fr.diffuse[..., 3][cr] == 1).all() and (fr.diffuse[..., 3][~cr] == 0).all()
    assert (fr.position[..., 3][cr] == 1).all() and (fr.normal[..., 3] == 0).all()
    assert (fr.diffuse.view(np.uint16) == fr.specular.view(np.uint16)).all()
    assert (fr.position[~cr] == 0).all()


def _fullscreen_fan(z=0.5, centre=(0.13, -0.21), n=7):
    """Triangles that tile the whole clip square around an interior point, given directly in clip space
    (identity camera): every pixel must be owned by exactly one of them."""
    ring = [(-1.0, -1.0), (0.2, -1.0), (1.0, -1.0), (1.0, 0.3), (1.0, 1.0), (-0.4, 1.0), (-1.0, 1.0), (-1.0, 0.1)]
    verts = [centre] + ring
    v = np.zeros(len(verts), abi.VERTEX_DTYPE)
    for i, (x, y) in enumerate(verts):
        v[i]["position"] = (x, y, z)
        v[i]["normal"] = (0, 0, -1)
        v[i]["uv_x"], v[i]["uv_y"] = x, y
    tris = []
    for k in range(len(ring)):
        tris += [0, 1 + k, 1 + (k + 1) % len(ring)]  # clockwise in a y-down framebuffer
    return v, np.array(tris, np.uint32)


def _identity_camera():
    cam = abi.CameraPacked()
    eye = np.eye(4, dtype=np.float32)
    for name in ("projection", "inverseProjection", "view", "viewInverseTranspose", "rotation", "projViewInverse"):
        setattr(cam, name, abi.Mat4.from_numpy(eye))
    return cam


@pytest.mark.parametrize("extent", [(64, 48), (97, 53), (256, 144)])
def test_oracle_shared_edges_are_watertight_and_hit_once(extent):
    W, H = extent
    v, idx = _fullscreen_fan()
    cam = _identity_camera()
    rect = abi.Rect(0, 0, W, H)
    ident = meshes.transform_matrix()
    material = meshes.default_material()
    # all triangles together: every pixel covered
    fr = ob.HostFrame(W, H)
    ob.gbuffer_raster(fr, rect, None, cam, [meshes.MeshInstanced(v, idx, [(0, len(idx), material)], [ident])], threads=4)
    assert (fr.depth == 0.5).all()
    # each triangle alone: the coverages partition the frame (no pixel twice, none missing)
    count = np.zeros((H, W), np.int32)
    for t in range(len(idx) // 3):
        f1 = ob.HostFrame(W, H)
        ob.gbuffer_raster(f1, rect, None, cam, [meshes.MeshInstanced(v, idx[3 * t:3 * t + 3], [(0, 3, material)], [ident])], threads=4)
        count += (f1.depth > 0)
    assert (count == 1).all()


def test_oracle_facing_culling_and_depth_order():
    W, H = 64, 64
    cam = _identity_camera()
    rect = abi.Rect(0, 0, W, H)
    ident = meshes.transform_matrix()
    material = meshes.default_material()

    def quad(z, winding):
        v = np.zeros(4, abi.VERTEX_DTYPE)
        for i, (x, y) in enumerate([(-1, -1), (1, -1), (1, 1), (-1, 1)]):
            v[i]["position"] = (x, y, z)
            v[i]["normal"] = (0, 0, -1)
        idx = np.array([0, 1, 2, 0, 2, 3] if winding else [0, 2, 1, 0, 3, 2], np.uint32)
        return meshes.MeshInstanced(v, idx, [(0, 6, material)], [ident])

    f = ob.HostFrame(W, H)
    ob.gbuffer_raster(f, rect, None, cam, [quad(0.25, True)], threads=1)  # clockwise (y down) = front face
    assert (f.depth == 0.25).all()
    ob.gbuffer_raster(f, rect, None, cam, [quad(0.25, False)], threads=1)  # counter-clockwise = back face: culled
    assert (f.depth == 0).all() and (f.diffuse == 0).all()
    # reverse-Z GREATER: the larger depth wins whatever the submission order; equal depth keeps the first
    for order in ([0.25, 0.75], [0.75, 0.25]):
        ob.gbuffer_raster(f, rect, None, cam, [quad(z, True) for z in order], threads=1)
        assert (f.depth == 0.75).all()
    # outside the depth clip volume (z > w or z < 0): no fragments
    for z in (1.5, -0.1):
        ob.gbuffer_raster(f, rect, None, cam, [quad(z, True)], threads=1)
        assert (f.depth == 0).all()
    # shadow pass keeps BACK faces (front-face culling) and writes the largest depth
    ident4 = abi.Mat4.from_numpy(np.eye(4, dtype=np.float32))
    assert (ob.shadow_raster(ident4, 32, [quad(0.5, True)]) == 0).all()
    assert (ob.shadow_raster(ident4, 32, [quad(0.5, False), quad(0.75, False)]) == 0.75).all()
    # depth bias: constant factor in units of the depth's ulp-scale r = 2^(e-23) (Vulkan spec), here e = -1
    biased = ob.shadow_raster(ident4, 32, [quad(0.5, False)], bias_constant=4.0)
    assert (biased == np.float32(0.5) + np.float32(4.0) * np.float32(2.0 ** -24)).all()


def test_oracle_render_flags_and_out_of_range_indices():
    W, H = 48, 32
    inp = util.Inputs(W, H)
    ms = meshes.reference_default_scene()
    full = ob.HostFrame(W, H)
    ob.gbuffer_raster(full, inp.rect, None, inp.cam, ms, threads=4)
    assert (full.depth > 0).any()
    for m in ms:
        m.render = False
    off = ob.HostFrame(W, H)
    off.depth[...] = 7
    ob.gbuffer_raster(off, inp.rect, None, inp.cam, ms, threads=4)
    assert (off.depth == 0).all()  # cleared, nothing drawn
    ms = meshes.reference_default_scene()
    ms[0].indices = ms[0].indices.copy()
    ms[0].indices[:3] = 10 ** 6  # a triangle pointing outside the vertex buffer is dropped, the rest is drawn
    ob.gbuffer_raster(off, inp.rect, None, inp.cam, ms, threads=4)
    assert (off.depth > 0).any()


def test_oracle_row_tiles_equal_the_frame():
    W, H = 96, 72
    inp = util.Inputs(W, H)
    ms = meshes.reference_default_scene() + _soup(5, 40)
    full = ob.HostFrame(W, H)
    ob.gbuffer_raster(full, inp.rect, None, inp.cam, ms, threads=8)
    for rank in range(3):
        tile = util.rowtile(H, 8, rank, 3)
        rows = util.global_rows(H, 8, rank, 3)
        part = ob.HostFrame(W, tile.local_rows)
        ob.gbuffer_raster(part, inp.rect, tile, inp.cam, ms, threads=8)
        assert (part.depth.view(np.uint32) == full.depth[rows].view(np.uint32)).all()
        assert (part.normal.view(np.uint16) == full.normal[rows].view(np.uint16)).all()
        assert (part.position.view(np.uint32) == full.position[rows].view(np.uint32)).all()


# ---------------------------------------------------------------------------
# GPU: HIP rasteriser == oracle, bit for bit
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the product path has no CPU fallback")
    from syzygy_amd import pipelines

    class Ctx:
        pass

    c = Ctx()
    c.pl, c.torch = pipelines, torch
    return c


def _cameras(gpu, cam):
    cameras = gpu.pl.TStagedBuffer(abi.CameraPacked, 1)
    cameras.push(cam)
    cameras.recordCopyToDevice()
    return cameras


def _raster_gpu(gpu, W, H, cam, ms, tile=None):
    rows = H if tile is None else tile.local_rows
    target = gpu.pl.SceneTexture(W, rows)
    deferred = gpu.pl.DeferredShadingPipeline((W, rows), max_spot_lights=1, max_shadow_maps=0)
    deferred.recordGBufferRaster(None, abi.Rect(0, 0, W, H), target, 0, _cameras(gpu, cam), ms, tile=tile)
    gpu.torch.cuda.synchronize()
    planes = deferred.download_gbuffer(W, rows)
    depth = target.depth.cpu().numpy()
    deferred.cleanup()
    return planes, depth


def _scenes():
    inp = util.Inputs(8, 8)
    return {
        "fill_scene": lambda: meshes.meshes_of_fill_scene(inp.synthetic.fill),
        "reference_default": meshes.reference_default_scene,
        "soup": lambda: _soup(11, 300),
        "all": lambda: meshes.reference_default_scene() + _soup(3, 120) + meshes.meshes_of_fill_scene(inp.synthetic.fill),
    }


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fill_scene", "reference_default", "soup", "all"])
@pytest.mark.parametrize("extent", [(240, 136), (97, 61)])
def test_gbuffer_raster_bit_exact(gpu, name, extent):
    W, H = extent
    inp = util.Inputs(W, H)
    ms = _scenes()[name]()
    planes, depth = _raster_gpu(gpu, W, H, inp.cam, ms)
    want = ob.HostFrame(W, H)
    ob.gbuffer_raster(want, inp.rect, None, inp.cam, ms, threads=8)
    assert (want.depth > 0).mean() > 0.05
    assert (depth.view(np.uint32) == want.depth.view(np.uint32)).all()
    _planes_equal(planes, want.planes())


def _hostile_scene(seed, with_default=True):
    """Degenerate and hostile geometry: zero-area and repeated-vertex triangles, constant uvs (singular cotangent frame),
    NaN / inf / huge coordinates, an out-of-range index, a surface reaching past the index buffer, an instance-less and a
    non-rendered mesh, a 1x1 and a missing texture."""
    rng = np.random.default_rng(seed)
    base = _soup(seed, 60)[0]
    v = base.vertices.copy()
    idx = base.indices.copy()
    v["position"][3] = v["position"][4]                      # repeated vertex
    v["position"][6:9] = v["position"][6]                    # zero area
    v["uv_x"][9:12], v["uv_y"][9:12] = 0.25, 0.75            # constant uv: T = B = 0
    v["position"][12] = (np.nan, 0.0, 1.0)
    v["position"][15] = (np.inf, -np.inf, 3.0)
    v["position"][18] = (1.0e30, -1.0e30, 1.0e30)
    v["position"][21] = (1.0e-30, 1.0e-38, -1.0e-30)
    idx[30] = 10 ** 9
    tiny = {"color": (rng.integers(0, 256, (1, 1, 4), dtype=np.uint8), True), "normal": (rng.integers(0, 256, (1, 3, 4), dtype=np.uint8), False)}
    half = len(idx) // 2
    ident = meshes.transform_matrix()
    squash = meshes.transform_matrix((0, -5, 10), (0.5, 1.0, 1.5), (3.0, 0.0, 3.0))  # singular model matrix
    return [
        meshes.MeshInstanced(v, idx, [(0, half, tiny), (half, 10 ** 6, meshes.default_material()), (10 ** 7, 30, tiny)], [ident, squash]),
        meshes.MeshInstanced(v, idx, [(0, half, tiny)], [], name="no instances"),
        meshes.MeshInstanced(v, idx, [(0, half, tiny)], [ident], render=False, name="hidden"),
    ] + (meshes.reference_default_scene() if with_default else [])


@pytest.mark.gpu
@pytest.mark.parametrize("seed,with_default", [(101, True), (202, False), (303, False)])
def test_gbuffer_raster_hostile_geometry_bit_exact(gpu, seed, with_default):
    W, H = 144, 81
    inp = util.Inputs(W, H)
    ms = _hostile_scene(seed, with_default)
    planes, depth = _raster_gpu(gpu, W, H, inp.cam, ms)
    want = ob.HostFrame(W, H)
    ob.gbuffer_raster(want, inp.rect, None, inp.cam, ms, threads=8)
    assert (depth.view(np.uint32) == want.depth.view(np.uint32)).all()
    _planes_equal(planes, want.planes())
    assert (want.depth > 0).mean() > 0.05


def _gltf_scene():
    """A textured sphere written as a GLB (the reference ships assets/sphere.glb, an LFS pointer in this checkout), loaded
    through include/szg/assets.h and instanced twice, next to the editor's floor."""
    from syzygy_amd import assets
    from tests import gltf_writer as gw

    rng = np.random.default_rng(77)
    pos, nrm, uv, idx = gw.uv_sphere(16, 32)
    b = gw.GltfBuilder()
    y, x = np.mgrid[0:32, 0:64]
    color = np.stack([(x * 4) & 255, (y * 8) & 255, ((x ^ y) * 8) & 255, np.full_like(x, 255)], -1).astype(np.uint8)
    normal = np.stack([127 + 60 * np.sin(x / 3.0), 127 + 60 * np.cos(y / 2.0), np.full(x.shape, 230.0), np.zeros(x.shape)], -1).astype(np.uint8)
    mr = rng.integers(0, 256, (8, 8, 4), dtype=np.uint8)
    t_color = b.texture(b.image_uri(gw.data_uri_png(gw.png_rgba8(color))))
    t_normal = b.texture(b.image_view(gw.png_rgba8(normal)))
    t_mr = b.texture(b.image_uri(gw.data_uri_png(gw.png_rgba8(mr))))
    b.doc["materials"] = [{"name": "painted", "pbrMetallicRoughness": {"baseColorTexture": {"index": t_color},
                                                                       "metallicRoughnessTexture": {"index": t_mr}},
                           "normalTexture": {"index": t_normal}}]
    half = len(idx) // 2 // 3 * 3
    b.doc["meshes"] = [{"name": "Sphere", "primitives": [
        {"attributes": {"POSITION": b.accessor(pos), "NORMAL": b.accessor(nrm), "TEXCOORD_0": b.accessor(uv)},
         "indices": b.accessor(idx[:half].astype(np.uint16)), "material": 0},
        {"attributes": {"POSITION": b.accessor(pos), "NORMAL": b.accessor(nrm), "TEXCOORD_0": b.accessor(uv)},
         "indices": b.accessor(idx[half:])}]}]
    a = assets.load_gltf_bytes(b.glb(), is_glb=True, flags=abi.SZG_GLTF_DECODE_BUFFER_VIEW_IMAGES)
    assert a.materials[0]["color"][1] is True and a.materials[0]["normal"] is not None
    models = [meshes.transform_matrix((-3.0, -6.0, 2.0), (0.3, 0.2, 0.1), (4, 4, 4)),
              meshes.transform_matrix((6.0, -4.0, 8.0), (0, 1.0, 0), (3, 5, 3))]
    return [a.instanced(0, models)] + meshes.reference_default_scene()[2:]


@pytest.mark.gpu
@pytest.mark.parametrize("extent", [(320, 180), (131, 77)])
def test_gbuffer_raster_of_a_loaded_gltf_asset_bit_exact(gpu, extent):
    W, H = extent
    inp = util.Inputs(W, H)
    ms = _gltf_scene()
    planes, depth = _raster_gpu(gpu, W, H, inp.cam, ms)
    want = ob.HostFrame(W, H)
    ob.gbuffer_raster(want, inp.rect, None, inp.cam, ms, threads=8)
    assert (want.depth > 0).mean() > 0.2
    assert (depth.view(np.uint32) == want.depth.view(np.uint32)).all()
    _planes_equal(planes, want.planes())
    # the spheres are visible from outside (clockwise-front after the loader's y flip) and carry the sRGB colour map
    sphere_pixels = (want.planes()["occlusionRoughnessMetallic"][..., 1] != np.float16(60 / 255)) & (want.depth > 0)
    assert sphere_pixels.mean() > 0.02


@pytest.mark.gpu
def test_gbuffer_raster_without_geometry_clears_the_targets(gpu):
    """No meshes / nothing rendered: the pass still clears the five planes and the depth (deferred.cpp:560-601)."""
    W, H = 70, 33
    inp = util.Inputs(W, H)
    hidden = meshes.reference_default_scene()
    for m in hidden:
        m.render = False
    for ms in ([], hidden):
        target = gpu.pl.SceneTexture(W, H)
        target.depth.fill_(3.0)
        deferred = gpu.pl.DeferredShadingPipeline((W, H), max_spot_lights=1, max_shadow_maps=0)
        deferred.upload_gbuffer({k: np.full((H, W, 4), 7, np.float16 if k != "worldPosition" else np.float32)
                                 for k in ("diffuse", "specular", "normal", "worldPosition", "occlusionRoughnessMetallic")})
        deferred.recordGBufferRaster(None, inp.rect, target, 0, _cameras(gpu, inp.cam), ms)
        gpu.torch.cuda.synchronize()
        assert (target.depth.cpu().numpy() == 0).all()
        for name, plane in deferred.download_gbuffer(W, H).items():
            assert (plane == 0).all(), name
        deferred.cleanup()


@pytest.mark.gpu
def test_gbuffer_raster_4k_bit_exact(gpu):
    """BASELINE config 3 size: the bench scene as meshes (290 primitives) at 3840x2160, whole frame against the oracle."""
    W, H = 3840, 2160
    inp = util.Inputs(W, H)
    ms = meshes.meshes_of_fill_scene(inp.synthetic.fill)
    planes, depth = _raster_gpu(gpu, W, H, inp.cam, ms)
    want = ob.HostFrame(W, H)
    ob.gbuffer_raster(want, inp.rect, None, inp.cam, ms, threads=16)
    assert 0.55 < (want.depth > 0).mean() < 0.65
    assert (depth.view(np.uint32) == want.depth.view(np.uint32)).all()
    _planes_equal(planes, want.planes())


@pytest.mark.gpu
def test_gbuffer_raster_fullscreen_fan_is_watertight(gpu):
    W, H = 257, 131
    v, idx = _fullscreen_fan()
    ms = [meshes.MeshInstanced(v, idx, [(0, len(idx), meshes.default_material())], [meshes.transform_matrix()])]
    planes, depth = _raster_gpu(gpu, W, H, _identity_camera(), ms)
    assert (depth == 0.5).all() and (planes["diffuse"][..., 3] == 1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("z", [0.0, -0.0, 1.0e-45])
def test_gbuffer_raster_depth_equal_to_the_clear_value_is_not_a_fragment(gpu, z):
    """GREATER against the cleared 0 (deferred.cpp:383-386): a primitive whose depth is exactly 0 (in the wild: the underflow
    of a huge, far primitive; found by tests/sweeps/random_sweep_raster.py seed 21607) covers pixels but produces no fragment,
    in front of or behind a real one; the smallest positive depth does."""
    W, H = 67, 45
    fan = _fullscreen_fan(z=z)
    near = _fullscreen_fan(z=0.25, centre=(-0.3, 0.4))
    material = meshes.default_material()
    for order in ([fan], [fan, near], [near, fan]):
        ms = [meshes.MeshInstanced(v, idx, [(0, len(idx), material)], [meshes.transform_matrix()]) for v, idx in order]
        planes, depth = _raster_gpu(gpu, W, H, _identity_camera(), ms)
        want = ob.HostFrame(W, H)
        ob.gbuffer_raster(want, abi.Rect(0, 0, W, H), None, _identity_camera(), ms, threads=4)
        assert (depth.view(np.uint32) == want.depth.view(np.uint32)).all()
        _planes_equal(planes, want.planes())
        if len(order) == 1:
            written = planes["diffuse"][..., 3] == 1
            assert written.all() if z > 0 else not written.any()


@pytest.mark.gpu
def test_gbuffer_raster_row_tiles_bit_exact(gpu):
    W, H = 200, 120
    inp = util.Inputs(W, H)
    ms = _scenes()["all"]()
    full, full_depth = _raster_gpu(gpu, W, H, inp.cam, ms)
    for nranks, block in ((3, 8), (2, 5)):
        for rank in range(nranks):
            tile = util.rowtile(H, block, rank, nranks)
            rows = util.global_rows(H, block, rank, nranks)
            planes, depth = _raster_gpu(gpu, W, H, inp.cam, ms, tile=tile)
            assert (depth.view(np.uint32) == full_depth[rows].view(np.uint32)).all()
            _planes_equal(planes, {k: a[rows] for k, a in full.items()})


@pytest.mark.gpu
def test_gbuffer_raster_many_primitives_and_buffer_growth(gpu):
    """20k primitives: chunk / box culling path, buffer growth between calls, still bit-identical to the oracle."""
    W, H = 160, 90
    inp = util.Inputs(W, H)
    small = _soup(21, 50)
    big = _soup(22, 5000, spread=120.0)
    deferred = gpu.pl.DeferredShadingPipeline((W, H), max_spot_lights=1, max_shadow_maps=0)
    cameras = _cameras(gpu, inp.cam)
    for ms in (small, big, small):
        target = gpu.pl.SceneTexture(W, H)
        deferred.recordGBufferRaster(None, inp.rect, target, 0, cameras, ms)
        gpu.torch.cuda.synchronize()
        want = ob.HostFrame(W, H)
        ob.gbuffer_raster(want, inp.rect, None, inp.cam, ms, threads=16)
        assert (target.depth.cpu().numpy().view(np.uint32) == want.depth.view(np.uint32)).all()
        _planes_equal(deferred.download_gbuffer(W, H), want.planes())
    deferred.cleanup()


@pytest.mark.gpu
@pytest.mark.parametrize("bias", [(0.0, 0.0), (2.0, 1.5)])
def test_shadow_raster_matches_oracle_and_the_analytic_maps(gpu, bias):
    from syzygy_amd.pipelines import _memcpy2d_from

    W, H, DIM, SPOTS = 64, 36, 256, 2
    inp = util.Inputs(W, H, elevation_degrees=40.0, spots=SPOTS)
    ms = meshes.meshes_of_fill_scene(inp.synthetic.fill)
    lights = gpu.pl.TStagedBuffer(abi.DirectionalLightPacked, 2)
    lights.push([inp.sun, inp.moon])
    lights.recordCopyToDevice()
    deferred = gpu.pl.DeferredShadingPipeline((W, H), max_spot_lights=SPOTS, max_shadow_maps=2 + SPOTS, shadow_map_dim=DIM)
    deferred.setConfiguration(abi.DeferredConfiguration(bias[0], bias[1]))
    deferred.recordShadowRaster(None, lights, inp.spots, ms)
    gpu.torch.cuda.synchronize()
    sm = deferred.shadowMaps()
    packed = [inp.sun, inp.moon] + [inp.spots[i] for i in range(SPOTS)]
    for slot, light in enumerate(packed):
        pv = abi.Mat4()
        from syzygy_amd import lib

        lib().szg_mat4_mul(C.byref(light.projection), C.byref(light.view), C.byref(pv))
        want = ob.shadow_raster(pv, DIM, ms, bias[0], bias[1], threads=8)
        got = _memcpy2d_from(sm.maps[slot], DIM * 4, DIM).cpu().numpy().view(np.float32).reshape(DIM, DIM)
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), f"slot {slot}"
        if bias == (0.0, 0.0):
            analytic = ob.shadow_map(light, DIM, inp.synthetic.fill, threads=8)
            assert ((got > 0) != (analytic > 0)).mean() < 5e-3
            both = (got > 0) & (analytic > 0)
            if both.any():  # silhouette texels may see different boxes; everywhere else the depths agree
                assert ((np.abs(got - analytic)[both] / analytic[both]) > 2e-3).mean() < 2e-3
    assert any((ob.shadow_map(light, DIM, inp.synthetic.fill, threads=8) > 0).any() for light in packed)
    deferred.cleanup()


@pytest.mark.gpu
def test_frame_from_real_meshes_matches_oracle_chain(gpu):
    """recordDrawCommands with scene geometry = meshes (shadow raster, G-buffer raster, lights) + the sky-view pipeline,
    against the oracle running the same chain."""
    torch = gpu.torch
    W, H, DIM, SPOTS = 192, 108, 256, 3
    inp = util.Inputs(W, H, elevation_degrees=40.0, spots=SPOTS)
    ms = meshes.reference_default_scene()
    cameras = _cameras(gpu, inp.cam)
    atmospheres = gpu.pl.TStagedBuffer(abi.AtmospherePacked, 1)
    atmospheres.push(inp.atm)
    atmospheres.recordCopyToDevice()
    lights = gpu.pl.TStagedBuffer(abi.DirectionalLightPacked, 2)
    lights.push([inp.sun, inp.moon])
    lights.recordCopyToDevice()
    target = gpu.pl.SceneTexture(W, H, debug=True)
    deferred = gpu.pl.DeferredShadingPipeline((W, H), max_spot_lights=SPOTS, max_shadow_maps=2 + SPOTS, shadow_map_dim=DIM)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(256, 64), skyview_extent=(256, 128))
    deferred.recordDrawCommandsMeshes(None, inp.rect, target, 1, lights, inp.spots, 0, cameras, ms)
    sky.recordDrawCommands(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    got, got_q = target.debug.cpu().numpy(), target.color_numpy()

    from syzygy_amd import lib

    packed = [inp.sun, inp.moon] + [inp.spots[i] for i in range(SPOTS)]
    maps = []
    for light in packed:
        pv = abi.Mat4()
        lib().szg_mat4_mul(C.byref(light.projection), C.byref(light.view), C.byref(pv))
        maps.append(ob.shadow_raster(pv, DIM, ms, threads=8))
    images = (abi.Image * len(maps))(*[ob.host_image(m, abi.SZG_FORMAT_D32_SFLOAT) for m in maps])
    host_maps = abi.ShadowMaps(len(maps), 0, C.cast(images, C.POINTER(abi.Image)))
    frame = ob.HostFrame(W, H)
    ob.gbuffer_raster(frame, inp.rect, None, inp.cam, ms, threads=8)
    ob.lights(frame, inp.rect, None, host_maps, inp.cam, inp.dirs, 2, 1, inp.spots, SPOTS, threads=8)
    tlut = ob.transmittance_lut(inp.atm, 256, 64, threads=8)
    slut = ob.skyview_lut(inp.atm, inp.cam, tlut, 256, 128, threads=8)
    ob.composite(frame, inp.rect, None, host_maps, inp.atm, inp.cam, inp.dirs, 0, tlut, slut, threads=8)
    assert (frame.depth > 0).mean() > 0.1
    util.assert_close(got, frame.debug, what="frame from real meshes")
    assert np.abs(got_q.astype(np.int32) - frame.color.astype(np.int32)).max() <= 1
    deferred.cleanup()
    sky.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1465, 1484, 1497])
def test_frame_from_hostile_meshes_poisons_the_same_pixels(gpu, seed):
    """The whole chain on degenerate geometry (tests/sweeps/random_sweep_mesh_frames.py found these): a zero-area or constant-uv
    triangle writes a NaN normal into the G-buffer. The sun term survives through its clamps, but the metal reflection's
    environment sample is NaN and `0 * NaN` poisons the pixel in the reference even for metallic == 0, so the composite
    may not skip that term. fp32 frame bit-identical including the NaN pattern."""
    torch = gpu.torch
    from syzygy_amd import lib, scene

    rng = np.random.default_rng(seed)
    W, H, DIM = int(rng.integers(16, 120)), int(rng.integers(9, 70)), int(rng.choice([32, 96]))
    cam = scene.default_camera()
    cam.cameraPosition[:] = [float(rng.uniform(-30, 30)), float(rng.uniform(-30, -1)), float(rng.uniform(-40, 20))]
    cam.eulerAngles[:] = [float(rng.uniform(-1.2, 1.2)), float(rng.uniform(-3.1, 3.1)), 0.0]
    cam.fovDegrees = float(rng.uniform(30.0, 110.0))
    spots_n = int(rng.integers(0, 4))
    inp = util.Inputs(W, H, elevation_degrees=float(rng.uniform(-8.0, 80.0)), spots=max(spots_n, 1), camera=cam)
    assert int(rng.integers(0, 3)) == 2
    ms = _hostile_scene(seed, bool(rng.integers(0, 2)))
    cameras = _cameras(gpu, inp.cam)
    atmospheres = gpu.pl.TStagedBuffer(abi.AtmospherePacked, 1)
    atmospheres.push(inp.atm)
    atmospheres.recordCopyToDevice()
    lights = gpu.pl.TStagedBuffer(abi.DirectionalLightPacked, 2)
    lights.push([inp.sun, inp.moon])
    lights.recordCopyToDevice()
    target = gpu.pl.SceneTexture(W, H, debug=True)
    deferred = gpu.pl.DeferredShadingPipeline((W, H), max_spot_lights=max(spots_n, 1), max_shadow_maps=2 + spots_n, shadow_map_dim=DIM)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(64, 16), skyview_extent=(64, 32))
    deferred.recordDrawCommandsMeshes(None, inp.rect, target, 1, lights, inp.spots if spots_n else None, 0, cameras, ms)
    sky.recordDrawCommands(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    got, got_q = target.debug.cpu().numpy(), target.color_numpy()

    maps = []
    for light in [inp.sun, inp.moon] + [inp.spots[i] for i in range(spots_n)]:
        pv = abi.Mat4()
        lib().szg_mat4_mul(C.byref(light.projection), C.byref(light.view), C.byref(pv))
        maps.append(ob.shadow_raster(pv, DIM, ms, threads=8))
    images = (abi.Image * len(maps))(*[ob.host_image(m, abi.SZG_FORMAT_D32_SFLOAT) for m in maps])
    host_maps = abi.ShadowMaps(len(maps), 0, C.cast(images, C.POINTER(abi.Image)))
    frame = ob.HostFrame(W, H)
    ob.gbuffer_raster(frame, inp.rect, None, inp.cam, ms, threads=8)
    ob.lights(frame, inp.rect, None, host_maps, inp.cam, inp.dirs, 2, 1, inp.spots, spots_n, threads=8)
    tlut = ob.transmittance_lut(inp.atm, 64, 16, threads=8)
    slut = ob.skyview_lut(inp.atm, inp.cam, tlut, 64, 32, threads=8)
    ob.composite(frame, inp.rect, None, host_maps, inp.atm, inp.cam, inp.dirs, 0, tlut, slut, threads=8)
    assert np.isnan(frame.planes()["normal"]).any() and np.isnan(frame.debug).any()
    same = (got.view(np.uint32) == frame.debug.view(np.uint32)) | (np.isnan(got) & np.isnan(frame.debug))
    assert same.all(), f"{(~same).sum()} fp32 values differ; NaNs {np.isnan(got).sum()} vs {np.isnan(frame.debug).sum()}"
    assert np.abs(got_q.astype(np.int32) - frame.color.astype(np.int32)).max() <= 1
    deferred.cleanup()
    sky.destroy()
