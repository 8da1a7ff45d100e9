"""GPU parity tests: every HIP pass, called through the C-ABI, against the CPU oracle on
the same seeded inputs. Tolerance: 1e-4 relative on fp32 values
(north_star), +-1 LSB on the UNORM16 scene colour, bit-exact where the
arithmetic has no transcendental (G-buffer fill, row tiling, compose).
"""
import ctypes as C

import numpy as np
import pytest

from tests import util
from tests.util import RTOL, assert_close

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the product path has no CPU fallback")
    from oracle import binding as ob
    from syzygy_amd import abi, pipelines, scene

    class Ctx:
        pass

    c = Ctx()
    c.ob, c.abi, c.pl, c.scene = ob, abi, pipelines, scene
    return c


def staged(gpu, inp):
    pl, abi = gpu.pl, gpu.abi
    cameras = pl.TStagedBuffer(abi.CameraPacked, 1)
    atmospheres = pl.TStagedBuffer(abi.AtmospherePacked, 1)
    lights = pl.TStagedBuffer(abi.DirectionalLightPacked, 2)
    cameras.push(inp.cam)
    atmospheres.push(inp.atm)
    lights.push([inp.sun, inp.moon])
    for b in (cameras, atmospheres, lights):
        b.recordCopyToDevice()
    return cameras, atmospheres, lights


# ---------------------------------------------------------------------------
# transmittance LUT (BASELINE config 1: 256x64; reference-native 512x128)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("extent", [(256, 64), (512, 128), (64, 16), (33, 7)])
def test_transmittance_lut_matches_oracle(gpu, extent):
    inp = util.Inputs(64, 64)
    cameras, atmospheres, lights = staged(gpu, inp)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=extent, skyview_extent=(64, 32))
    assert sky is not None
    sky.recordTransmittance(None, 0, atmospheres)
    torch.cuda.synchronize()
    got = sky.download_lut(sky.transmittanceLUT())
    want = gpu.ob.transmittance_lut(inp.atm, extent[0], extent[1], threads=8)
    worst = assert_close(got, want, atol=1e-12, what=f"transmittance {extent}")
    exact = float((got == want).mean())
    print(f"transmittance {extent}: worst/tol {worst:.3f}, bit-identical fraction {exact:.3f}")
    sky.destroy()


# ---------------------------------------------------------------------------
# sky-view LUT on an identical transmittance LUT
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("elevation", [70.0, 5.0, -3.0])
def test_skyview_lut_matches_oracle(gpu, elevation):
    inp = util.Inputs(64, 64, elevation_degrees=elevation)
    cameras, atmospheres, lights = staged(gpu, inp)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(512, 128), skyview_extent=(512, 256))
    tlut = gpu.ob.transmittance_lut(inp.atm, 512, 128, threads=8)
    sky.upload_lut(sky.transmittanceLUT(), tlut)
    sky.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
    torch.cuda.synchronize()
    got = sky.download_lut(sky.skyviewLUT())
    want = gpu.ob.skyview_lut(inp.atm, inp.cam, tlut, 512, 256, threads=8)
    worst = assert_close(got[..., :3], want[..., :3], atol=1e-9, what=f"skyview elev {elevation}")
    assert (got[..., 3] == 1.0).all()
    print(f"skyview elev {elevation}: worst/tol {worst:.3f}")
    sky.destroy()


def test_skyview_lut_reference_size_band(gpu):
    """Reference-native 2048x1024 LUT: a band of rows around the horizon against the oracle."""
    inp = util.Inputs(64, 64, elevation_degrees=5.0)
    cameras, atmospheres, lights = staged(gpu, inp)
    sky = gpu.pl.SkyViewComputePipeline.create()
    tlut = gpu.ob.transmittance_lut(inp.atm, 512, 128, threads=8)
    sky.upload_lut(sky.transmittanceLUT(), tlut)
    sky.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
    torch.cuda.synchronize()
    got = sky.download_lut(sky.skyviewLUT())
    want = np.zeros((1024, 2048, 4), np.float32)
    gpu.ob.skyview_lut(inp.atm, inp.cam, tlut, 2048, 1024, row_begin=480, row_end=544, threads=8, out=want)
    assert_close(got[480:544, :, :3], want[480:544, :, :3], atol=1e-9, what="skyview 2048x1024 rows 480..544")
    # mirror symmetry in azimuth is NOT exact (u = .5 + .5 cos), but every texel must be finite and >= 0
    assert np.isfinite(got).all() and (got[..., :3] >= 0).all()
    sky.destroy()


@pytest.mark.parametrize("elevation", [35.0, 0.0, 90.0])
def test_luts_at_structured_camera_altitudes(gpu, elevation):
    """Coincidences a random camera never hits: the camera exactly ON the ground (radius == planet radius: the horizon angle is
    asin(1)), one ulp above and below it, exactly at the top of the atmosphere and just outside it, deep underground - with
    the sun at 35 degrees, exactly on the horizon and exactly at the zenith (the undefined azimuth of SURVEY Q15). Sky-view LUT
    on the GPU's own transmittance LUT against the oracle chain, bit for bit including the NaN pattern."""
    from syzygy_amd import scene

    shell_m = 100000.0  # scene.cpp:52-75: atmosphere radius - planet radius = 0.1 Mm
    for y in (0.0, -np.float32(1e-45), np.float32(1e-45), -0.5, -shell_m, -np.nextafter(np.float32(shell_m), np.float32(2e5)),
              -np.nextafter(np.float32(shell_m), np.float32(0.0)), 250.0, -2.0e5):
        cam = scene.default_camera()
        cam.cameraPosition[:] = [0.0, float(y), 0.0]
        inp = util.Inputs(64, 64, elevation_degrees=elevation, camera=cam)
        cameras, atmospheres, lights = staged(gpu, inp)
        sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(256, 64), skyview_extent=(128, 64))
        sky.recordTransmittance(None, 0, atmospheres)
        sky.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
        torch.cuda.synchronize()
        got_t = sky.download_lut(sky.transmittanceLUT())
        got_s = sky.download_lut(sky.skyviewLUT())
        want_t = gpu.ob.transmittance_lut(inp.atm, 256, 64, threads=8)
        want_s = gpu.ob.skyview_lut(inp.atm, inp.cam, want_t, 128, 64, threads=8)
        assert (got_t.view(np.uint32) == want_t.view(np.uint32)).all(), y
        assert (np.isnan(got_s) == np.isnan(want_s)).all(), y
        ok = ~np.isnan(got_s)
        assert (got_s.view(np.uint32)[ok] == want_s.view(np.uint32)[ok]).all(), y
        sky.destroy()


def test_skyview_lut_row_slices_equal_the_full_lut(gpu):
    """Multi-GPU extension: N row slices (szg_skyview_record_skyview_lut_rows) == the whole LUT, bit for bit;
    the LUT memory aliased as a torch tensor is the same storage."""
    inp = util.Inputs(64, 64, elevation_degrees=20.0)
    cameras, atmospheres, lights = staged(gpu, inp)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(256, 64), skyview_extent=(256, 128))
    sky.recordTransmittance(None, 0, atmospheres)
    sky.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
    torch.cuda.synchronize()
    full = sky.download_lut(sky.skyviewLUT())
    alias = sky.skyviewLUT_tensor()
    assert alias.shape == (128, 256, 4) and (alias.cpu().numpy().view(np.uint32) == full.view(np.uint32)).all()
    alias.zero_()
    from syzygy_amd import rowtile

    for rank in (2, 0, 3, 1):
        b, e = rowtile.lut_rows(128, rank, 4)
        sky.recordSkyViewLUTRows(None, 0, atmospheres, 0, cameras, b, e)
    torch.cuda.synchronize()
    assert (sky.download_lut(sky.skyviewLUT()).view(np.uint32) == full.view(np.uint32)).all()
    from syzygy_amd import SzgError

    with pytest.raises(SzgError):
        sky.recordSkyViewLUTRows(None, 0, atmospheres, 0, cameras, 100, 200)
    sky.destroy()


# ---------------------------------------------------------------------------
# G-buffer fill: arithmetic only -> bit-exact
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("size", [(160, 96), (70, 37), (1, 1)])
def test_gbuffer_fill_bit_exact(gpu, size):
    W, H = size
    inp = util.Inputs(W, H)
    cameras, atmospheres, lights = staged(gpu, inp)
    target = gpu.pl.SceneTexture(W, H)
    deferred = gpu.pl.DeferredShadingPipeline((W, H), max_spot_lights=1, max_shadow_maps=0)
    deferred.recordGBufferFill(None, inp.rect, target, 0, cameras, inp.synthetic.fill)
    torch.cuda.synchronize()
    got = deferred.download_gbuffer(W, H)
    frame = gpu.ob.HostFrame(W, H)
    gpu.ob.gbuffer_fill(frame, inp.rect, None, inp.cam, inp.synthetic.fill, threads=4)
    for name, want in frame.planes().items():
        assert (got[name].view(np.uint8) == want.view(np.uint8)).all(), name
    assert (target.depth.cpu().numpy().view(np.uint32) == frame.depth.view(np.uint32)).all()
    deferred.cleanup()


# ---------------------------------------------------------------------------
# lights pass
# ---------------------------------------------------------------------------
def run_lights_case(gpu, W, H, spots, skip, elevation=70.0, shadow=None, tile=None, poison=None, mutate=None):
    inp = util.Inputs(W, H, elevation_degrees=elevation, spots=spots)
    if mutate is not None:
        mutate(inp)
    cameras, atmospheres, lights = staged(gpu, inp)
    rows = H if tile is None else tile.local_rows
    frame = gpu.ob.HostFrame(W, rows)
    gpu.ob.gbuffer_fill(frame, inp.rect, tile, inp.cam, inp.synthetic.fill, threads=8)
    if poison is not None:
        poison(frame)
    shadow_host = None
    keep = []
    deferred = gpu.pl.DeferredShadingPipeline((W, rows), max_spot_lights=max(spots, 1), max_shadow_maps=spots + 2)
    if shadow is not None:
        images = (gpu.abi.Image * (spots + 2))()
        for slot, depth_map in shadow.items():
            images[slot] = gpu.ob.host_image(depth_map, gpu.abi.SZG_FORMAT_D32_SFLOAT)
            t = torch.from_numpy(depth_map).cuda()
            keep.append(t)
            deferred.setShadowMap(slot, t)
        shadow_host = gpu.abi.ShadowMaps(spots + 2, 0, C.cast(images, C.POINTER(gpu.abi.Image)))
        keep.append(images)
    gpu.ob.lights(frame, inp.rect, tile, shadow_host, inp.cam, inp.dirs, 2, skip, inp.spots, spots, threads=8)

    target = gpu.pl.SceneTexture(W, rows, debug=True)
    deferred.upload_gbuffer(frame.planes())
    deferred.recordLights(None, inp.rect, target, skip, lights, inp.spots if spots else None, 0, cameras, tile=tile)
    torch.cuda.synchronize()
    got = target.debug.cpu().numpy()
    got_q = target.color_numpy()
    deferred.cleanup()
    return got, got_q, frame


@pytest.mark.parametrize("spots,skip", [(0, 1), (0, 0), (1, 1), (16, 1), (64, 1), (256, 0)])
def test_lights_match_oracle(gpu, spots, skip):
    got, got_q, frame = run_lights_case(gpu, 192, 108, spots, skip)
    worst = assert_close(got, frame.debug, what=f"lights spots={spots} skip={skip}")
    assert np.abs(got_q.astype(np.int32) - frame.color.astype(np.int32)).max() <= 1
    assert (got_q[..., 3] == 65535).all()
    print(f"lights spots={spots} skip={skip}: worst/tol {worst:.3f}")


def test_lights_nan_gbuffer_poisons_culled_lights_too(gpu):
    """A pixel outside every spot cone still multiplies each light's term by its BRDF in the reference (lights.comp:141-161):
    0 * NaN = NaN. With non-finite G-buffer values (here NaN / inf normals, specular and ORM in patches of geometry) the kernel
    may not skip culled lights for the affected waves; the NaN pattern must equal the oracle's — with the moon skipped too
    (skip = 2), so that no always-evaluated light hides the difference."""
    def poison(frame):
        geometry = np.argwhere(frame.depth > 0)
        rng = np.random.default_rng(7)
        for plane, value in ((frame.normal, np.nan), (frame.specular, np.inf), (frame.orm, np.nan), (frame.normal, -np.inf)):
            for (y, x) in geometry[rng.choice(len(geometry), 40, replace=False)]:
                plane[y, x, :3] = value

    got, got_q, frame = run_lights_case(gpu, 160, 90, 24, 2, poison=poison)
    assert np.isnan(frame.debug).any() and not np.isnan(frame.debug).all()
    assert (np.isnan(got) == np.isnan(frame.debug)).all()
    ok = ~np.isnan(got)
    assert (got.view(np.uint32)[ok] == frame.debug.view(np.uint32)[ok]).all()
    assert (got_q == frame.color).all()


def test_lights_optimistic_pass_falls_back_when_an_operand_leaves_the_lean_domain(gpu):
    """k_lights runs its light loop optimistically (lean exact operators without per-pair range tests, the operands' lower
    bounds folded into a running minimum) and repeats a wave's loop with the tested form when the minimum says an operand left
    the domain. Force that: pixels whose world position IS a spot light's position (distance 0: falloff 0, colour / 0 = inf,
    inf * 0 = NaN further on) and pixels a hair away from it (d^2 far below 2^-30). The image, including its inf / NaN pattern,
    must still be the oracle's bit for bit."""
    inp = util.Inputs(160, 90, elevation_degrees=70.0, spots=24)
    positions = [tuple(inp.spots[i].position[:3]) for i in range(24)]
    forwards = [np.array(inp.spots[i].forward[:3], np.float64) for i in range(24)]
    camera = np.array(inp.cam.position[:3], np.float64)

    def poison(frame):
        geometry = np.argwhere(frame.depth > 0)
        rng = np.random.default_rng(11)
        picks = geometry[rng.choice(len(geometry), 72, replace=False)]
        for k, (y, x) in enumerate(picks):
            if k < 48:
                p = np.array(positions[k % 24], np.float32)
                if k >= 24:
                    p = p + np.float32(1e-7) * np.array([1.0, -2.0, 0.5], np.float32)
            else:
                # ... and pixels seen exactly against a light's direction: view direction = -light direction, so the half
                # vector light + view has a squared length of ~1e-14 (below the 2^-40 the lean normalisation needs) or 0
                light_dir = -forwards[k % 24] / np.linalg.norm(forwards[k % 24])
                p = (camera + light_dir * (3.0 + 0.37 * (k - 48))).astype(np.float32)
            frame.position[y, x, :3] = p

    got, got_q, frame = run_lights_case(gpu, 160, 90, 24, 2, poison=poison)
    assert not np.isfinite(frame.debug).all()  # the case does produce inf / NaN pixels in the reference's arithmetic
    assert (np.isnan(got) == np.isnan(frame.debug)).all()
    ok = ~np.isnan(got)
    assert (got.view(np.uint32)[ok] == frame.debug.view(np.uint32)[ok]).all()
    assert (got_q == frame.color).all()


def test_lights_with_extreme_falloff_parameters(gpu):
    """Spot lights whose falloff factor and distance are each of 'moderate' magnitude but whose combination is not:
    factor * (d / distance)^2 up to 2^80 and down to 2^-80, colour * strength down to 2^-29 - quotients at the edge of the
    normal range and below it. Such lights must leave the lean exact operators' domain (k_light_prep) and come out
    bit-identical to the oracle's IEEE divisions, denormals included."""
    def mutate(inp):
        settings = [(2.0 ** 25, 2.0 ** -25, 1.0), (2.0 ** -25, 2.0 ** 25, 1.0), (2.0 ** 29, 2.0 ** -20, 2.0 ** -29),
                    (2.0 ** -29, 2.0 ** 29, 2.0 ** 29), (1.0, 2.0 ** -29, 2.0 ** -20), (2.0 ** 20, 1.0, 2.0 ** -29)]
        for i, (factor, distance, strength) in enumerate(settings):
            for k in (i, i + 6):
                inp.spots[k].falloffFactor = factor
                inp.spots[k].falloffDistance = distance
                inp.spots[k].strength = strength

    got, got_q, frame = run_lights_case(gpu, 160, 90, 12, 2, mutate=mutate)
    assert np.isfinite(frame.debug).all()
    assert (got.view(np.uint32) == frame.debug.view(np.uint32)).all()
    assert (got_q == frame.color).all()
    values = np.abs(frame.debug[..., :3])
    print("extreme falloff: smallest non-zero pixel value", values[values > 0].min(), "largest", values.max())
    assert (values > 0).any()


def test_lights_ragged_extent(gpu):
    got, got_q, frame = run_lights_case(gpu, 70, 37, 8, 1)
    assert_close(got, frame.debug, what="lights 70x37")


def test_lights_background_keeps_clear_colour(gpu):
    got, got_q, frame = run_lights_case(gpu, 96, 54, 4, 0)
    background = frame.diffuse[..., 3].astype(np.float32) < 1.0
    assert background.any() and (~background).any()
    assert (got_q[background] == np.array([0, 0, 0, 65535], np.uint16)).all()


def test_lights_with_shadow_maps(gpu):
    """25-tap PCF on nearest / clamp-to-border maps (shadowmap.glinl:32-64): random occluder
    depths for the moon (slot 1) and two spot lights."""
    rng = np.random.default_rng(0x5A2C)
    maps = {1: rng.random((64, 64), dtype=np.float32), 2: rng.random((128, 96), dtype=np.float32),
            5: (rng.random((33, 47), dtype=np.float32) > 0.5).astype(np.float32) * 0.9999}
    got, got_q, frame = run_lights_case(gpu, 160, 90, 8, 1, shadow=maps)
    assert_close(got, frame.debug, what="lights with shadow maps")
    assert np.abs(got_q.astype(np.int32) - frame.color.astype(np.int32)).max() <= 1


def test_lights_capacity_is_an_error(gpu):
    inp = util.Inputs(32, 32, spots=4)
    cameras, atmospheres, lights = staged(gpu, inp)
    target = gpu.pl.SceneTexture(32, 32)
    deferred = gpu.pl.DeferredShadingPipeline((32, 32), max_spot_lights=2, max_shadow_maps=0)
    from syzygy_amd import SzgError

    with pytest.raises(SzgError) as e:
        deferred.recordLights(None, inp.rect, target, 1, lights, inp.spots, 0, cameras)
    assert e.value.code == -5
    deferred.cleanup()


def test_lights_linearity(gpu):
    """all lights at once == sum of single-light runs, before the UNORM clamp."""
    W, H, N = 128, 72, 6
    inp = util.Inputs(W, H, spots=N)
    cameras, atmospheres, lights = staged(gpu, inp)
    target = gpu.pl.SceneTexture(W, H, debug=True)
    deferred = gpu.pl.DeferredShadingPipeline((W, H), max_spot_lights=N, max_shadow_maps=0)
    deferred.recordGBufferFill(None, inp.rect, target, 0, cameras, inp.synthetic.fill)
    deferred.recordLights(None, inp.rect, target, 2, lights, inp.spots, 0, cameras)
    torch.cuda.synchronize()
    total = target.debug.cpu().numpy().astype(np.float64)
    acc = np.zeros_like(total)
    for i in range(N):
        one = (gpu.abi.SpotLightPacked * 1)(inp.spots[i])
        deferred.recordLights(None, inp.rect, target, 2, lights, one, 0, cameras)
        torch.cuda.synchronize()
        acc[..., :3] += target.debug.cpu().numpy()[..., :3]
    assert_close(total[..., :3], acc[..., :3], what="linearity")
    deferred.cleanup()


# ---------------------------------------------------------------------------
# composite on identical LUTs / G-buffer / prior colour
# ---------------------------------------------------------------------------
def run_composite_case(gpu, W, H, elevation, spots=8, sun_shadow=None, tile=None, camera=None, lut=((512, 128), (256, 128)),
                       poison=None):
    inp = util.Inputs(W, H, elevation_degrees=elevation, spots=spots, camera=camera)
    cameras, atmospheres, lights = staged(gpu, inp)
    rows = H if tile is None else tile.local_rows
    frame = gpu.ob.HostFrame(W, rows)
    gpu.ob.gbuffer_fill(frame, inp.rect, tile, inp.cam, inp.synthetic.fill, threads=8)
    if poison is not None:
        poison(frame, inp)
    gpu.ob.lights(frame, inp.rect, tile, None, inp.cam, inp.dirs, 2, 1, inp.spots, spots, threads=8)
    prior = frame.color.copy()
    (tw, th), (sw, sh) = lut
    tlut = gpu.ob.transmittance_lut(inp.atm, tw, th, threads=8)
    slut = gpu.ob.skyview_lut(inp.atm, inp.cam, tlut, sw, sh, threads=8)
    shadow_host, keep = None, []
    images = (gpu.abi.Image * 1)()
    if sun_shadow is not None:
        images[0] = gpu.ob.host_image(sun_shadow, gpu.abi.SZG_FORMAT_D32_SFLOAT)
        shadow_host = gpu.abi.ShadowMaps(1, 0, C.cast(images, C.POINTER(gpu.abi.Image)))
    gpu.ob.composite(frame, inp.rect, tile, shadow_host, inp.atm, inp.cam, inp.dirs, 0, tlut, slut, threads=8)

    target = gpu.pl.SceneTexture(W, rows, debug=True)
    target.color.copy_(torch.from_numpy(prior.view(np.int16)))
    target.depth.copy_(torch.from_numpy(frame.depth))
    deferred = gpu.pl.DeferredShadingPipeline((W, rows), max_spot_lights=1, max_shadow_maps=1)
    deferred.upload_gbuffer(frame.planes())
    if sun_shadow is not None:
        t = torch.from_numpy(sun_shadow).cuda()
        keep.append(t)
        deferred.setShadowMap(0, t)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(tw, th), skyview_extent=(sw, sh))
    sky.upload_lut(sky.transmittanceLUT(), tlut)
    sky.upload_lut(sky.skyviewLUT(), slut)
    sky.recordComposite(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights,
                        tile=tile)
    torch.cuda.synchronize()
    got = target.debug.cpu().numpy()
    got_q = target.color_numpy()
    deferred.cleanup()
    sky.destroy()
    return got, got_q, frame


@pytest.mark.parametrize("elevation", [70.0, 5.0, -3.0])
def test_composite_matches_oracle(gpu, elevation):
    got, got_q, frame = run_composite_case(gpu, 192, 108, elevation)
    worst = assert_close(got, frame.debug, what=f"composite elev {elevation}")
    assert np.abs(got_q.astype(np.int32) - frame.color.astype(np.int32)).max() <= 1
    print(f"composite elev {elevation}: worst/tol {worst:.3f}")


def test_composite_ragged_extent_and_small_luts(gpu):
    got, got_q, frame = run_composite_case(gpu, 70, 37, 20.0, lut=((256, 64), (128, 64)))
    assert_close(got, frame.debug, what="composite 70x37")


def test_composite_with_sun_shadow_map(gpu):
    rng = np.random.default_rng(7)
    shadow = rng.random((96, 96), dtype=np.float32)
    got, got_q, frame = run_composite_case(gpu, 160, 90, 40.0, sun_shadow=shadow)
    assert_close(got, frame.debug, what="composite with sun shadow map")


@pytest.mark.parametrize("elevation", [35.0, 2.0])
def test_composite_structured_coincidences(gpu, elevation):
    """Coincidences a random sweep never draws (round 3 found the lights pass wrong for a pixel AT a light's position this way):
    G-buffer positions exactly on the camera and one ulp beside it (zero-length view segment, NaN direction of
    sampleTransmittanceLUT_Segment), exactly at ground level and at the sub-camera point, far outside the atmosphere, metal
    pixels whose normal is the view direction, its negation, or the zero vector, and normals that make the reflection
    exactly horizontal. The image, with its NaN pattern, must be the oracle's bit for bit."""
    def poison(frame, inp):
        cam = np.array(inp.cam.position[:3], np.float32)
        geometry = np.argwhere(frame.depth > 0)
        rng = np.random.default_rng(23)
        picks = geometry[rng.choice(len(geometry), 120, replace=False)]
        up = np.float32(np.nextafter(np.float32(cam[1]), np.float32(1e9)))
        cases = [cam, np.array([cam[0], up, cam[2]], np.float32), np.array([np.nextafter(cam[0], np.float32(9e9)), cam[1], cam[2]], np.float32),
                 np.array([cam[0], 0.0, cam[2]], np.float32), np.array([3.0, 0.0, -2.0], np.float32), np.array([cam[0], -1e-30, cam[2]], np.float32),
                 np.array([2.0e8, -1.0e8, 5.0e7], np.float32), np.array([0.0, -1.2e5, 0.0], np.float32)]
        for k, (y, x) in enumerate(picks):
            c = k % 12
            if c < len(cases):
                frame.position[y, x, :3] = cases[c]
            else:
                view = cam - frame.position[y, x, :3]
                n = np.linalg.norm(view)
                view = view / n if n > 0 else view
                normal = {8: view, 9: -view, 10: np.zeros(3, np.float32), 11: np.array([0.0, -1.0, 0.0], np.float32)}[c]
                frame.normal[y, x, :3] = normal.astype(np.float16)
                frame.orm[y, x, 2] = np.float16(1.0)  # metallic: the reflection branch is taken
            if k % 3 == 0:
                frame.orm[y, x, 2] = np.float16(1.0)

    got, got_q, frame = run_composite_case(gpu, 160, 90, elevation, poison=poison)
    assert (np.isnan(got) == np.isnan(frame.debug)).all()
    ok = ~np.isnan(got)
    assert (got.view(np.uint32)[ok] == frame.debug.view(np.uint32)[ok]).all()
    assert (got_q == frame.color).all()


def test_composite_camera_looking_down_and_up(gpu):
    """Sky rays that hit the ground (sampleGround path) and a zenith-facing view."""
    from syzygy_amd import scene

    for pitch in (0.9, -0.9):
        cam = scene.default_camera()
        cam.cameraPosition[:] = [0.0, -300.0, -13.0]
        cam.eulerAngles[:] = [pitch, 0.0, 0.3]
        got, got_q, frame = run_composite_case(gpu, 128, 72, 30.0, camera=cam)
        assert_close(got, frame.debug, what=f"composite pitch {pitch}")


# ---------------------------------------------------------------------------
# whole frame, every pass on the GPU (LUTs, fill, lights, composite chained)
# ---------------------------------------------------------------------------
def render_gpu(gpu, inp, tile=None, lut=((512, 128), (512, 256)), debug=True):
    cameras, atmospheres, lights = staged(gpu, inp)
    rows = inp.height if tile is None else tile.local_rows
    target = gpu.pl.SceneTexture(inp.width, rows, debug=debug)
    deferred = gpu.pl.DeferredShadingPipeline((inp.width, rows), max_spot_lights=max(inp.spot_count, 1), max_shadow_maps=0)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=lut[0], skyview_extent=lut[1])
    deferred.recordDrawCommands(None, inp.rect, target, 1, lights, inp.spots if inp.spot_count else None, 0, cameras,
                                inp.synthetic.fill, tile=tile)
    sky.recordDrawCommands(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0,
                           lights, tile=tile)
    torch.cuda.synchronize()
    out = (target.debug.cpu().numpy() if debug else None, target.color_numpy())
    deferred.cleanup()
    sky.destroy()
    return out


def render_oracle(gpu, inp, lut=((512, 128), (512, 256))):
    frame = gpu.ob.HostFrame(inp.width, inp.height)
    gpu.ob.gbuffer_fill(frame, inp.rect, None, inp.cam, inp.synthetic.fill, threads=8)
    gpu.ob.lights(frame, inp.rect, None, None, inp.cam, inp.dirs, 2, 1, inp.spots, inp.spot_count, threads=8)
    tlut = gpu.ob.transmittance_lut(inp.atm, lut[0][0], lut[0][1], threads=8)
    slut = gpu.ob.skyview_lut(inp.atm, inp.cam, tlut, lut[1][0], lut[1][1], threads=8)
    gpu.ob.composite(frame, inp.rect, None, None, inp.atm, inp.cam, inp.dirs, 0, tlut, slut, threads=8)
    return frame


@pytest.mark.parametrize("elevation", [70.0, 5.0, -3.0])
def test_full_frame_chain_matches_oracle(gpu, elevation):
    """End to end: GPU LUTs feed the GPU composite (nothing is uploaded from the oracle).
    The LUT-ratio terms of the march are ill-conditioned (1 - T_a/T_b with T_a ~ T_b), so
    this only holds because the GLSL built-ins are pinned (include/szg/fpmath.h) and every
    other operation is IEEE correctly rounded on both sides."""
    inp = util.Inputs(240, 136, elevation_degrees=elevation, spots=16)
    got, got_q = render_gpu(gpu, inp)
    frame = render_oracle(gpu, inp)
    lsb = np.abs(got_q.astype(np.int32) - frame.color.astype(np.int32))
    rel = util.rel_err(got[..., :3], frame.debug[..., :3], util.ATOL_COLOR)
    print(f"chain elev {elevation}: max LSB diff {lsb.max()}, max rel {rel.max():.3e}, "
          f"frac within 1e-4: {(rel <= RTOL).mean():.5f}")
    assert lsb.max() <= 1
    assert rel.max() <= RTOL


def test_full_frame_chain_with_nonzero_buffer_indices(gpu):
    """The reference addresses its parameter blocks as buffer[index] (atmosphereIndex, cameraIndex, sunLightIndex and the
    directional-light skip count of the push constants: skyview.hpp:64-144, deferred.hpp:82-98). Every other test uses
    index 0; here the blocks in use sit at atmosphere 2 of 3, camera 1 of 3 and lights [poison, sun, moon] with the sun at
    index 1 and a skip count of 2, and every unused slot is filled with 0xFF bytes (NaN): the frame must equal the frame
    rendered from the plain buffers bit for bit, and the oracle's."""
    pl, abi = gpu.pl, gpu.abi
    inp = util.Inputs(200, 120, elevation_degrees=25.0, spots=6)

    def poison(ctype):
        v = ctype()
        C.memset(C.byref(v), 0xFF, C.sizeof(v))
        return v

    cameras = pl.TStagedBuffer(abi.CameraPacked, 3)
    atmospheres = pl.TStagedBuffer(abi.AtmospherePacked, 3)
    lights = pl.TStagedBuffer(abi.DirectionalLightPacked, 3)
    cameras.push([poison(abi.CameraPacked), inp.cam, poison(abi.CameraPacked)])
    atmospheres.push([poison(abi.AtmospherePacked), poison(abi.AtmospherePacked), inp.atm])
    lights.push([poison(abi.DirectionalLightPacked), inp.sun, inp.moon])
    for b in (cameras, atmospheres, lights):
        b.recordCopyToDevice()
    target = pl.SceneTexture(inp.width, inp.height, debug=True)
    deferred = pl.DeferredShadingPipeline((inp.width, inp.height), max_spot_lights=inp.spot_count, max_shadow_maps=0)
    sky = pl.SkyViewComputePipeline.create(transmittance_extent=(512, 128), skyview_extent=(512, 256))
    deferred.recordDrawCommands(None, inp.rect, target, 2, lights, inp.spots, 1, cameras, inp.synthetic.fill)
    sky.recordDrawCommands(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 2, atmospheres, 1, cameras, 1, lights)
    torch.cuda.synchronize()
    got, got_q = target.debug.cpu().numpy(), target.color_numpy()
    deferred.cleanup()
    sky.destroy()
    plain, plain_q = render_gpu(gpu, inp)
    assert np.array_equal(got_q, plain_q)
    assert np.array_equal(got.view(np.uint32), plain.view(np.uint32))
    frame = render_oracle(gpu, inp)
    assert np.abs(got_q.astype(np.int32) - frame.color.astype(np.int32)).max() <= 1
    assert util.rel_err(got[..., :3], frame.debug[..., :3], util.ATOL_COLOR).max() <= RTOL


# ---------------------------------------------------------------------------
# inputs outside the domain of the lean exact ops: the kernels must take their generic code path
# (szg_device.hpp leanAtmosphere / leanRay) and still match the oracle
# ---------------------------------------------------------------------------
def _thin_shell(a):
    a.atmosphereRadiusMegameters = a.planetRadiusMegameters * (1.0 + 1e-4)  # H - Ra + 0.9 Rp < 0


def _tiny_density_scale(a):
    a.altitudeDecayMieMegameters = 2.0 ** -34  # outside [2^-30, 2^30]


def _absorbing_rayleigh(a):  # the Earth defaults have zero Rayleigh absorption and zero ozone scattering; the lean
    a.absorptionRayleighPerMegameter[:] = [0.7, 1.3, 2.9]  # paths drop those exact-zero terms (szg_device.hpp Atm)


def _scattering_ozone(a):
    a.scatteringOzonePerMegameter[:] = [0.3, 0.2, 0.9]


def _absorbing_and_scattering(a):
    _absorbing_rayleigh(a)
    _scattering_ozone(a)


def _one_nonzero_component(a):
    a.absorptionRayleighPerMegameter[:] = [0.0, 0.0, 0.5]


def _negative_zero_coefficient(a):  # -0 clears the sign test: the full sum must be evaluated
    a.scatteringRayleighPerMegameter[0] = -0.0
    a.scatteringMiePerMegameter[2] = -0.0


def _nan_coefficient(a):  # a NaN in the block poisons the frame identically on both sides (generic path)
    a.absorptionOzonePerMegameter[1] = float("nan")


def _infinite_coefficient(a):
    a.scatteringMiePerMegameter[0] = float("inf")


def _dense_atmosphere(a):  # horizontal optical depths of several hundred: LUT texels underflow, the LUT status word
    for k in range(3):     # keeps the transmittance quotients on the generic division
        a.scatteringRayleighPerMegameter[k] *= 40.0


def _large_coefficient(a):  # above 2^10: the extinction bound of the lean in-scatter division does not hold
    a.scatteringMiePerMegameter[:] = [1500.0, 1500.0, 1500.0]


def _camera_deep_underground_lean_floor(a):  # shell so thick that Rp - 80 H is the active floor, not 0.9 Rp
    a.altitudeDecayMieMegameters = 0.0004


@pytest.mark.parametrize("edit", [_thin_shell, _tiny_density_scale, _absorbing_rayleigh, _scattering_ozone, _absorbing_and_scattering,
                                  _one_nonzero_component, _negative_zero_coefficient, _nan_coefficient, _infinite_coefficient,
                                  _dense_atmosphere, _large_coefficient,
                                  _camera_deep_underground_lean_floor])
def test_generic_path_unusual_atmospheres(gpu, edit):
    inp = util.Inputs(128, 72, elevation_degrees=30.0, spots=6, atmosphere_edit=edit)
    got, got_q = render_gpu(gpu, inp, lut=((128, 32), (128, 64)))
    frame = render_oracle(gpu, inp, lut=((128, 32), (128, 64)))
    assert_close(got, frame.debug, what=f"full frame, {edit.__name__}")
    assert np.abs(got_q.astype(np.int32) - frame.color.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_planets_cameras_and_suns(gpu, seed):
    """Randomised frames: planet / shell size, density scales, coefficient magnitudes, sun elevation, camera height and
    orientation drawn at random (some inside, some outside the domain of the lean exact ops). The chained frame must stay
    within the north_star bound of the oracle; today it is bit-identical."""
    from syzygy_amd import scene

    rng = np.random.default_rng(1000 + seed)

    def edit(a):
        a.planetRadiusMegameters = float(rng.uniform(1.0, 12.0))
        a.atmosphereRadiusMegameters = a.planetRadiusMegameters + float(rng.uniform(0.02, 0.4))
        a.altitudeDecayRayleighMegameters = float(rng.uniform(0.002, 0.03))
        a.altitudeDecayMieMegameters = float(rng.uniform(0.0005, 0.005))
        for field in ("scatteringRayleighPerMegameter", "absorptionRayleighPerMegameter", "scatteringMiePerMegameter",
                      "scatteringOzonePerMegameter", "absorptionOzonePerMegameter"):
            scale = float(10.0 ** rng.uniform(-1.0, 1.0))
            vals = getattr(a, field)
            for k in range(3):
                vals[k] = vals[k] * scale
        a.sunAngularRadius = float(rng.uniform(0.002, 0.05))

    cam = scene.default_camera()
    cam.cameraPosition[:] = [float(rng.uniform(-30, 30)), float(-10.0 ** rng.uniform(0.3, 4.7)), float(rng.uniform(-40, 10))]
    cam.eulerAngles[:] = [float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-3.1, 3.1)), 0.0]
    cam.fovDegrees = float(rng.uniform(30.0, 100.0))
    inp = util.Inputs(112, 63, elevation_degrees=float(rng.uniform(-10.0, 90.0)), spots=int(rng.integers(0, 9)), camera=cam,
                      atmosphere_edit=edit)
    got, got_q = render_gpu(gpu, inp, lut=((128, 32), (128, 64)))
    frame = render_oracle(gpu, inp, lut=((128, 32), (128, 64)))
    assert_close(got, frame.debug, what=f"random frame {seed}")
    assert np.abs(got_q.astype(np.int32) - frame.color.astype(np.int32)).max() <= 1
    exact = float((got.view(np.uint32) == frame.debug.view(np.uint32)).mean())
    print(f"random frame {seed}: geometry {float((frame.depth > 0).mean()):.2f}, bit-identical fraction {exact:.5f}")


@pytest.mark.parametrize("seed", [237, 345, 101, 333, 512])
def test_random_frames_including_nan_environments(gpu, seed):
    """A wider random family (sun down to -15 degrees, density scales so small that the extinction underflows to exactly
    0 in the upper shell, coefficient vectors replaced or scaled by 10^+-1.5, cameras from 1 m to 150 km). Seeds 237 and
    345 make the reference math produce NaN in half of the sky-view LUT; 0 * NaN then poisons even non-metal pixels through
    the reflection term (camera.comp:379-386), which the kernel may therefore skip only when the environment is provably
    finite. Bit-identical including the NaN pattern."""
    from syzygy_amd import scene

    rng = np.random.default_rng(seed)

    def edit(a):
        if rng.random() < 0.7:
            a.planetRadiusMegameters = float(rng.uniform(1.0, 12.0))
            a.atmosphereRadiusMegameters = a.planetRadiusMegameters + float(rng.uniform(0.02, 0.4))
        a.altitudeDecayRayleighMegameters = float(rng.uniform(0.002, 0.03))
        a.altitudeDecayMieMegameters = float(rng.uniform(0.0005, 0.005))
        for field in ("scatteringRayleighPerMegameter", "absorptionRayleighPerMegameter", "scatteringMiePerMegameter",
                      "scatteringOzonePerMegameter", "absorptionOzonePerMegameter"):
            vals = getattr(a, field)
            scale = float(10.0 ** rng.uniform(-1.5, 1.5))
            for k in range(3):
                vals[k] = vals[k] * scale
            if rng.random() < 0.25:
                for k in range(3):
                    vals[k] = float(rng.uniform(0, 3))
        a.sunAngularRadius = float(rng.uniform(0.002, 0.05))

    cam = scene.default_camera()
    cam.cameraPosition[:] = [float(rng.uniform(-30, 30)), float(-10.0 ** rng.uniform(0.0, 5.2)), float(rng.uniform(-40, 10))]
    cam.eulerAngles[:] = [float(rng.uniform(-1.2, 1.2)), float(rng.uniform(-3.1, 3.1)), 0.0]
    cam.fovDegrees = float(rng.uniform(30.0, 100.0))
    elevation = float(rng.uniform(-15.0, 90.0))
    inp = util.Inputs(128, 72, elevation_degrees=elevation, spots=int(rng.integers(0, 6)), camera=cam, atmosphere_edit=edit)
    got, got_q = render_gpu(gpu, inp, lut=((128, 32), (128, 64)))
    frame = render_oracle(gpu, inp, lut=((128, 32), (128, 64)))
    assert (np.isnan(got) == np.isnan(frame.debug)).all(), "NaN patterns differ"
    same = (got.view(np.uint32) == frame.debug.view(np.uint32)) | np.isnan(got)
    assert same.all(), f"{(~same).sum()} values differ"
    assert (got_q == frame.color).all()
    print(f"random frame {seed}: elevation {elevation:.1f}, NaN fraction {float(np.isnan(got).mean()):.2f}")


def test_uploaded_lut_with_underflowing_texels(gpu):
    """A transmittance LUT written by the CALLER (here: the oracle's, for an atmosphere so dense that texels underflow to 0)
    is re-scanned before use (k_lut_range): its quotients must take the generic division — a lean division by a zero
    texel would give NaN where IEEE gives inf — and the sky-view LUT still equals the oracle's."""
    inp = util.Inputs(64, 36, elevation_degrees=20.0, atmosphere_edit=_dense_atmosphere)
    cameras, atmospheres, lights = staged(gpu, inp)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(128, 32), skyview_extent=(128, 64))
    tlut = gpu.ob.transmittance_lut(inp.atm, 128, 32, threads=8)
    assert tlut[..., :3].min() < 2.0 ** -50
    sky.upload_lut(sky.transmittanceLUT(), tlut)
    sky.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
    torch.cuda.synchronize()
    got = sky.download_lut(sky.skyviewLUT())
    want = gpu.ob.skyview_lut(inp.atm, inp.cam, tlut, 128, 64, threads=8)
    assert_close(got[..., :3], want[..., :3], atol=1e-12, what="sky-view LUT from an uploaded dense LUT")
    # then a moderate LUT uploaded into the SAME pipeline: the status is recomputed, results again equal
    inp2 = util.Inputs(64, 36, elevation_degrees=20.0)
    cameras2, atmospheres2, _ = staged(gpu, inp2)
    tlut2 = gpu.ob.transmittance_lut(inp2.atm, 128, 32, threads=8)
    sky.upload_lut(sky.transmittanceLUT(), tlut2)
    sky.recordSkyViewLUT(None, 0, atmospheres2, 0, cameras2)
    torch.cuda.synchronize()
    got2 = sky.download_lut(sky.skyviewLUT())
    want2 = gpu.ob.skyview_lut(inp2.atm, inp2.cam, tlut2, 128, 64, threads=8)
    assert (got2.view(np.uint32) == want2.view(np.uint32))[..., :3].all()
    sky.destroy()


def test_generic_path_camera_far_below_ground(gpu):
    """Unphysical on purpose (the reference has a TODO for it, common.glinl:294): rays whose radius drops under
    0.9 R_planet fail leanRay and use the generic operators; results still equal the oracle's, NaNs included."""
    from syzygy_amd import scene

    cam = scene.default_camera()
    cam.cameraPosition[:] = [0.0, 1.2e6, -13.0]  # +y is down: 1200 km below the surface
    inp = util.Inputs(96, 54, elevation_degrees=30.0, spots=2, camera=cam)
    cameras, atmospheres, lights = staged(gpu, inp)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(128, 32), skyview_extent=(128, 64))
    tlut = gpu.ob.transmittance_lut(inp.atm, 128, 32, threads=8)
    sky.upload_lut(sky.transmittanceLUT(), tlut)
    sky.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
    torch.cuda.synchronize()
    got = sky.download_lut(sky.skyviewLUT())
    want = gpu.ob.skyview_lut(inp.atm, inp.cam, tlut, 128, 64, threads=8)
    assert_close(got[..., :3], want[..., :3], atol=1e-9, what="sky-view LUT, camera below ground")
    sky.destroy()


# ---------------------------------------------------------------------------
# row tiling (multi-GPU partition): every rank's tile equals its rows of the whole frame
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("nranks,block_rows", [(2, 8), (3, 4), (8, 16)])
def test_rowtiles_equal_full_frame_bit_exact(gpu, nranks, block_rows):
    W, H = 160, 100
    inp = util.Inputs(W, H, elevation_degrees=25.0, spots=8)
    full_dbg, full_q = render_gpu(gpu, inp, lut=((256, 64), (128, 64)))
    gathered = []
    max_rows = 0
    for rank in range(nranks):
        tile = util.rowtile(H, block_rows, rank, nranks)
        rows = util.global_rows(H, block_rows, rank, nranks)
        assert len(rows) == tile.local_rows
        if tile.local_rows == 0:  # more ranks than row blocks: this rank holds nothing
            gathered.append(np.zeros((0, W, 4), np.uint16))
            continue
        dbg, q = render_gpu(gpu, inp, tile=tile, lut=((256, 64), (128, 64)))
        assert (q == full_q[rows]).all()
        assert (dbg.view(np.uint32) == full_dbg[rows].view(np.uint32)).all()
        gathered.append(q)
        max_rows = max(max_rows, tile.local_rows)
    # compose kernel: the gather buffer (rank-major, padded to the largest tile) -> full image
    from syzygy_amd import lib
    from syzygy_amd._lib import check

    stride_rows = max_rows
    buf = np.zeros((nranks, stride_rows, W, 4), np.uint16)
    for r, q in enumerate(gathered):
        buf[r, : q.shape[0]] = q
    d_buf = torch.from_numpy(buf.view(np.int16)).cuda()
    d_out = torch.zeros((H, W, 4), dtype=torch.int16, device="cuda")
    im = gpu.abi.Image(d_out.data_ptr(), W, H, W * 8, gpu.abi.SZG_FORMAT_RGBA16_UNORM)
    check(lib().szg_compose_rowtiles(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(d_buf.data_ptr()),
                                     stride_rows * W * 8, nranks, block_rows, C.byref(im), W, H))
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy().view(np.uint16) == full_q).all()


# ---------------------------------------------------------------------------
# Aerial-perspective froxel LUT (extension without a reference pass). Texel values are the
# reference's own math at froxel centres -> parity with the oracle. The fast composite that CONSUMES the LUT is
# approximate by design and only has to stay close to the exact composite.
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("elevation,max_distance", [(35.0, 0.032), (5.0, 0.2)])
def test_aerial_lut_matches_oracle(gpu, elevation, max_distance):
    inp = util.Inputs(160, 90, elevation_degrees=elevation)
    cameras, atmospheres, lights = staged(gpu, inp)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(512, 128), skyview_extent=(64, 32))
    tlut = gpu.ob.transmittance_lut(inp.atm, 512, 128, threads=8)
    sky.upload_lut(sky.transmittanceLUT(), tlut)
    sky.recordAerialLUT(None, 0, atmospheres, 0, cameras, max_distance)
    torch.cuda.synchronize()
    lum_im, tr_im = sky.aerialLUT()
    assert (lum_im.width, lum_im.height) == (32, 32 * 32)
    got_lum, got_tr = sky.download_lut(lum_im), sky.download_lut(tr_im)
    want_lum, want_tr = gpu.ob.aerial_lut(inp.atm, inp.cam, tlut, max_distance, threads=8)
    assert_close(got_lum, want_lum, atol=1e-12, what="aerial luminance")
    assert_close(got_tr, want_tr, atol=1e-12, what="aerial transmittance")
    exact = float((got_lum.view(np.uint32) == want_lum.view(np.uint32)).mean())
    print(f"aerial LUT elev {elevation}: bit-identical fraction {exact:.4f}")
    vol = got_lum.reshape(32, 32, 32, 4)
    if np.isfinite(vol).all():  # (froxel rays longer than the distance to the ground are NaN in the reference math too)
        assert (vol[1:, :, :, :3].sum((1, 2, 3)) >= vol[:-1, :, :, :3].sum((1, 2, 3))).all()  # in-scatter grows with depth
    finite = np.isfinite(got_tr[..., :3])
    assert (got_tr[..., :3][finite] <= 1).all() and (got_tr[..., :3][finite] >= 0).all()
    sky.destroy()


def test_fast_composite_is_close_to_the_exact_one(gpu):
    W, H = 320, 180
    inp = util.Inputs(W, H, elevation_degrees=35.0, spots=8)
    cameras, atmospheres, lights = staged(gpu, inp)
    exact_t = gpu.pl.SceneTexture(W, H, debug=True)
    fast_t = gpu.pl.SceneTexture(W, H, debug=True)
    deferred = gpu.pl.DeferredShadingPipeline((W, H), max_spot_lights=8, max_shadow_maps=0)
    sky = gpu.pl.SkyViewComputePipeline.create(skyview_extent=(512, 256))
    for target in (exact_t, fast_t):
        deferred.recordDrawCommands(None, inp.rect, target, 1, lights, inp.spots, 0, cameras, inp.synthetic.fill)
    sky.recordTransmittance(None, 0, atmospheres)
    sky.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
    from syzygy_amd import SzgError

    with pytest.raises(SzgError):  # no aerial LUT yet
        sky.recordCompositeFast(None, fast_t, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    sky.recordAerialLUT(None, 0, atmospheres, 0, cameras, 10.0e-3)  # 10 km
    sky.recordComposite(None, exact_t, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    sky.recordCompositeFast(None, fast_t, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    a, b = exact_t.debug.cpu().numpy(), fast_t.debug.cpu().numpy()
    geo = (exact_t.depth.cpu().numpy() > 0)
    assert (a[~geo].view(np.uint32) == b[~geo].view(np.uint32)).all(), "sky pixels must be untouched by the fast mode"
    rel = util.rel_err(a[geo][:, :3], b[geo][:, :3], util.ATOL_COLOR)
    print(f"fast composite vs exact: max rel {rel.max():.3e}, mean rel {rel.mean():.3e} over {geo.sum()} geometry pixels")
    assert rel.max() < 5e-2 and rel.mean() < 5e-3
    deferred.cleanup()
    sky.destroy()


# ---------------------------------------------------------------------------
# shadow-map generation for the analytic scene and a frame that uses the maps
# ---------------------------------------------------------------------------
def test_shadow_maps_match_oracle_and_shade_the_frame(gpu):
    W, H, DIM, SPOTS = 192, 108, 256, 3
    inp = util.Inputs(W, H, elevation_degrees=40.0, spots=SPOTS)
    cameras, atmospheres, lights = staged(gpu, inp)
    target = gpu.pl.SceneTexture(W, H, debug=True)
    deferred = gpu.pl.DeferredShadingPipeline((W, H), max_spot_lights=SPOTS, max_shadow_maps=2 + SPOTS, shadow_map_dim=DIM)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(256, 64), skyview_extent=(256, 128))
    # recordDrawCommands: shadow maps -> G-buffer -> lights (deferred.cpp:480-787)
    deferred.recordDrawCommands(None, inp.rect, target, 1, lights, inp.spots, 0, cameras, inp.synthetic.fill)
    sky.recordDrawCommands(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    got = target.debug.cpu().numpy()
    got_q = target.color_numpy()

    # oracle: the same maps, slot order sun, moon, spots (lights.comp:138-161)
    packed = [inp.sun, inp.moon] + [inp.spots[i] for i in range(SPOTS)]
    maps = [gpu.ob.shadow_map(light, DIM, inp.synthetic.fill, threads=8) for light in packed]
    sm = deferred.shadowMaps()
    assert sm.count == 2 + SPOTS
    from syzygy_amd.pipelines import _memcpy2d_from

    for slot, want in enumerate(maps):
        im = sm.maps[slot]
        assert (im.width, im.height) == (DIM, DIM)
        dev = _memcpy2d_from(im, DIM * 4, DIM).cpu().numpy().view(np.float32).reshape(DIM, DIM)
        assert (dev.view(np.uint32) == want.view(np.uint32)).all(), f"shadow map slot {slot}"
    assert (maps[0] > 0).any() and (maps[2] > 0).any()

    images = (gpu.abi.Image * len(maps))(*[gpu.ob.host_image(m, gpu.abi.SZG_FORMAT_D32_SFLOAT) for m in maps])
    host_maps = gpu.abi.ShadowMaps(len(maps), 0, C.cast(images, C.POINTER(gpu.abi.Image)))
    frame = gpu.ob.HostFrame(W, H)
    gpu.ob.gbuffer_fill(frame, inp.rect, None, inp.cam, inp.synthetic.fill, threads=8)
    gpu.ob.lights(frame, inp.rect, None, host_maps, inp.cam, inp.dirs, 2, 1, inp.spots, SPOTS, threads=8)
    tlut = gpu.ob.transmittance_lut(inp.atm, 256, 64, threads=8)
    slut = gpu.ob.skyview_lut(inp.atm, inp.cam, tlut, 256, 128, threads=8)
    gpu.ob.composite(frame, inp.rect, None, host_maps, inp.atm, inp.cam, inp.dirs, 0, tlut, slut, threads=8)
    assert_close(got, frame.debug, what="frame with generated shadow maps")
    assert np.abs(got_q.astype(np.int32) - frame.color.astype(np.int32)).max() <= 1

    # and the shadows do something: the same frame without maps is brighter
    deferred2 = gpu.pl.DeferredShadingPipeline((W, H), max_spot_lights=SPOTS, max_shadow_maps=0)
    target2 = gpu.pl.SceneTexture(W, H, debug=True)
    deferred2.recordDrawCommands(None, inp.rect, target2, 1, lights, inp.spots, 0, cameras, inp.synthetic.fill)
    sky.recordComposite(None, target2, inp.rect, deferred2.gbuffer(), deferred2.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    assert target2.debug.cpu().numpy()[..., :3].sum() > got[..., :3].sum()
    deferred.cleanup()
    deferred2.cleanup()
    sky.destroy()


# ---------------------------------------------------------------------------
# Multi-scattering LUT: extension, "parity unpinned" (no reference counterpart) -> own oracle
# ---------------------------------------------------------------------------
def test_multiscatter_lut_matches_its_oracle(gpu):
    inp = util.Inputs(64, 64, elevation_degrees=35.0)
    cameras, atmospheres, lights = staged(gpu, inp)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(512, 128), skyview_extent=(64, 32))
    tlut = gpu.ob.transmittance_lut(inp.atm, 512, 128, threads=8)
    sky.upload_lut(sky.transmittanceLUT(), tlut)
    sky.recordMultiScatterLUT(None, 0, atmospheres)
    torch.cuda.synchronize()
    got = sky.download_lut(sky.multiScatterLUT())
    want, fms = gpu.ob.multiscatter_lut(inp.atm, tlut)
    assert got.shape == (32, 32, 4)
    assert_close(got, want, atol=1e-12, what="multi-scattering LUT")
    print(f"multi-scatter LUT: bit-identical fraction {float((got.view(np.uint32) == want.view(np.uint32)).mean()):.4f}")
    sky.destroy()


# ---------------------------------------------------------------------------
# OETF: integer in, integer out -> bit-exact
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("function", [0, 1])
@pytest.mark.parametrize("size", [(256, 64), (131, 7), (1, 1)])
def test_oetf_bit_exact(gpu, function, size):
    W, H = size
    rng = np.random.default_rng(W * 31 + function)
    img = rng.integers(0, 65536, (H, W, 4), dtype=np.uint16)
    img[0, 0] = [0, 205, 206, 65535]  # around the sRGB cutoff 0.0031308 * 65535 = 205.2
    pad = 5 + (-(W + 5) % 2)  # a target wider than the region, rows still 16-byte multiples
    target = gpu.pl.SceneTexture(W + pad, H + 3)
    target.color.fill_(-12345)
    target.color[:H, :W] = torch.from_numpy(img.view(np.int16)).cuda()
    gpu.pl.recordOETF(None, target, W, H, function)
    torch.cuda.synchronize()
    got = target.color_numpy()
    want = gpu.ob.oetf(img.copy(), function)
    assert (got[:H, :W] == want).all()
    assert (got[:H, :W, 3] == img[..., 3]).all()  # alpha passes through
    untouched = np.int16(-12345).astype(np.int16).view(np.uint16)
    assert (got[H:] == untouched).all() and (got[:, W:] == untouched).all()


@pytest.mark.parametrize("function", [0, 1])
def test_oetf_every_code_value(gpu, function):
    """All 65536 UNORM16 codes in every colour channel (the kernel goes through a 65536-entry table built on the device): the
    whole transfer function, exhaustively, against the oracle's pow() evaluation."""
    codes = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    img = np.stack([codes, codes[::-1, ::-1], codes.T, codes], -1).copy()
    target = gpu.pl.SceneTexture(256, 256)
    target.color[:256, :256] = torch.from_numpy(img.view(np.int16)).cuda()
    gpu.pl.recordOETF(None, target, 256, 256, function)
    torch.cuda.synchronize()
    got = target.color_numpy()[:256, :256]
    want = gpu.ob.oetf(img.copy(), function)
    assert (got == want).all()
    # monotone, fixes 0 and 1, alpha untouched
    curve = got[..., 0].reshape(-1).astype(np.int64)
    assert curve[0] == 0 and curve[-1] == 65535 and (np.diff(curve) >= 0).all() and (got[..., 3] == img[..., 3]).all()


def test_oetf_known_values(gpu):
    img = np.zeros((1, 4, 4), np.uint16)
    img[0, :, 0] = [0, 65535, 32768, 100]
    out = gpu.ob.oetf(img.copy(), 1)
    assert out[0, 0, 0] == 0 and out[0, 1, 0] == 65535
    assert abs(out[0, 2, 0] / 65535 - (1.055 * 0.5000076 ** (1 / 2.4) - 0.055)) < 2e-5
    assert out[0, 3, 0] == round(12.92 * 100)  # linear segment below the cutoff


def test_rowtile_collectives_on_rccl_single_rank(gpu):
    """The two collectives of the multi-GPU path through RCCL itself (a 1-rank group is all one GPU allows): byte
    view gather of int16 tiles, in-place all-gather on the LUT memory the C library owns, async handles."""
    import os

    import torch.distributed as dist

    from syzygy_amd import rowtile

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        tile = torch.randint(-32768, 32767, (16, 24, 4), dtype=torch.int16, device="cuda")
        gathered, work = rowtile.gather_tiles(tile, 0, 1, async_op=True)
        work.wait()
        torch.cuda.synchronize()
        assert gathered.shape == (1, 16, 24, 4) and torch.equal(gathered[0], tile)
        out = rowtile.compose(gathered, 16, 1, 8)
        torch.cuda.synchronize()
        assert torch.equal(out, tile)

        inp = util.Inputs(64, 64)
        cameras, atmospheres, lights = staged(gpu, inp)
        sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(128, 32), skyview_extent=(128, 64))
        sky.recordTransmittance(None, 0, atmospheres)
        sky.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
        torch.cuda.synchronize()
        full = sky.download_lut(sky.skyviewLUT())
        lut = sky.skyviewLUT_tensor()
        lut.zero_()
        b, e = rowtile.lut_rows(64, 0, 1)
        sky.recordSkyViewLUTRows(None, 0, atmospheres, 0, cameras, b, e)
        work = rowtile.allgather_lut(lut, 0, 1, async_op=True, force=True)
        work.wait()
        torch.cuda.synchronize()
        assert (sky.download_lut(sky.skyviewLUT()).view(np.uint32) == full.view(np.uint32)).all()
        sky.destroy()
    finally:
        dist.destroy_process_group()


def test_8k_row_tile_matches_oracle(gpu):
    """7680x4320 (BASELINE config 4): one rank's cyclic row tile (rank 5 of 8, 8-row blocks) rendered on the GPU;
    the oracle renders a 24-row band of the SAME frame through the contiguous-band tiling and the rows both
    hold must agree to +-1 LSB (they are bit-identical today)."""
    W, H = 7680, 4320
    inp = util.Inputs(W, H, elevation_degrees=35.0, spots=64)
    tile = util.rowtile(H, 8, 5, 8)
    assert tile.local_rows == 536  # 540 blocks of 8 rows over 8 ranks: ranks 0-3 hold 68 blocks, ranks 4-7 hold 67
    dbg, q = render_gpu(gpu, inp, tile=tile, lut=((512, 128), (2048, 1024)), debug=False)
    assert q.shape == (536, W, 4) and (q[..., 3] == 65535).all()
    gpu_rows = util.global_rows(H, 8, 5, 8)

    band = 24
    nbands = H // band
    b = int(0.47 * nbands)
    otile = util.rowtile(H, band, b, nbands)
    frame = gpu.ob.HostFrame(W, band)
    gpu.ob.gbuffer_fill(frame, inp.rect, otile, inp.cam, inp.synthetic.fill, threads=16)
    gpu.ob.lights(frame, inp.rect, otile, None, inp.cam, inp.dirs, 2, 1, inp.spots, 64, threads=16)
    tlut = gpu.ob.transmittance_lut(inp.atm, 512, 128, threads=16)
    cameras, atmospheres, lights = staged(gpu, inp)
    skyp = gpu.pl.SkyViewComputePipeline.create()
    skyp.upload_lut(skyp.transmittanceLUT(), tlut)
    skyp.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)  # GPU sky-view LUT for both sides (parity-tested separately)
    torch.cuda.synchronize()
    slut = skyp.download_lut(skyp.skyviewLUT())
    skyp.destroy()
    gpu.ob.composite(frame, inp.rect, otile, None, inp.atm, inp.cam, inp.dirs, 0, tlut, slut, threads=16)
    oracle_rows = util.global_rows(H, band, b, nbands)
    common = np.intersect1d(gpu_rows, oracle_rows)
    assert len(common) >= 8
    gi = np.searchsorted(gpu_rows, common)
    oi = np.searchsorted(oracle_rows, common)
    lsb = np.abs(q[gi].astype(np.int32) - frame.color[oi].astype(np.int32))
    assert lsb.max() <= 1, lsb.max()


def test_empty_draw_rect_is_a_no_op(gpu):
    inp = util.Inputs(32, 32, spots=1)
    cameras, atmospheres, lights = staged(gpu, inp)
    target = gpu.pl.SceneTexture(32, 32)
    target.color.fill_(1234)
    deferred = gpu.pl.DeferredShadingPipeline((32, 32), max_spot_lights=1, max_shadow_maps=0)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(64, 16), skyview_extent=(64, 32))
    empty = gpu.pl.rect(0, 0)
    deferred.recordDrawCommands(None, empty, target, 1, lights, inp.spots, 0, cameras, inp.synthetic.fill)
    sky.recordComposite(None, target, empty, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    assert (target.color.cpu().numpy() == 1234).all()
    deferred.cleanup()
    sky.destroy()


def test_undersized_target_is_rejected(gpu):
    inp = util.Inputs(64, 64, spots=1)
    cameras, atmospheres, lights = staged(gpu, inp)
    target = gpu.pl.SceneTexture(32, 32)
    deferred = gpu.pl.DeferredShadingPipeline((64, 64), max_spot_lights=1, max_shadow_maps=0)
    from syzygy_amd import SzgError

    with pytest.raises(SzgError):
        deferred.recordLights(None, inp.rect, target, 1, lights, inp.spots, 0, cameras)
    deferred.cleanup()


# ---------------------------------------------------------------------------
# BASELINE full sizes through size-independent properties
# ---------------------------------------------------------------------------
def test_4k_frame_properties(gpu):
    """3840x2160, 64 spots (BASELINE config 3): deterministic, finite, alpha opaque, and a
    64-row band equals the oracle's render of the same rows."""
    W, H = 3840, 2160
    inp = util.Inputs(W, H, elevation_degrees=35.0, spots=64)
    dbg1, q1 = render_gpu(gpu, inp, lut=((512, 128), (2048, 1024)))
    dbg2, q2 = render_gpu(gpu, inp, lut=((512, 128), (2048, 1024)))
    assert (q1 == q2).all(), "non-deterministic output"
    assert np.isfinite(dbg1).all()
    assert (q1[..., 3] == 65535).all()
    sky = q1[: H // 4]
    assert sky[..., 2].mean() > sky[..., 0].mean(), "daytime sky should be blue"

    # oracle on a band of rows that crosses the horizon, via the row-tile mechanism
    # (contiguous tiling = one block per rank)
    band_rows = 24
    nranks = H // band_rows
    rank = int(0.42 * nranks)
    tile = util.rowtile(H, band_rows, rank, nranks)
    assert tile.local_rows == band_rows
    frame = gpu.ob.HostFrame(W, band_rows)
    gpu.ob.gbuffer_fill(frame, inp.rect, tile, inp.cam, inp.synthetic.fill, threads=16)
    gpu.ob.lights(frame, inp.rect, tile, None, inp.cam, inp.dirs, 2, 1, inp.spots, 64, threads=16)
    tlut = gpu.ob.transmittance_lut(inp.atm, 512, 128, threads=16)
    # only the GPU's own sky-view LUT is available at full size in reasonable time: use a
    # GPU-produced LUT for BOTH sides so that this compares the 4K composite + lights only
    cameras, atmospheres, lights = staged(gpu, inp)
    skyp = gpu.pl.SkyViewComputePipeline.create()
    skyp.upload_lut(skyp.transmittanceLUT(), tlut)
    skyp.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
    torch.cuda.synchronize()
    slut = skyp.download_lut(skyp.skyviewLUT())
    skyp.destroy()
    gpu.ob.composite(frame, inp.rect, tile, None, inp.atm, inp.cam, inp.dirs, 0, tlut, slut, threads=16)
    rows = util.global_rows(H, band_rows, rank, nranks)
    lsb = np.abs(q1[rows].astype(np.int32) - frame.color.astype(np.int32))
    assert lsb.max() <= 1, lsb.max()


def _oracle_band(gpu, inp, band_rows, fraction, host_maps, tlut, slut, threads=16):
    """The oracle's render of `band_rows` rows of the frame starting at `fraction` of its height (contiguous tiling = one
    block per rank): fill -> lights -> composite on those rows. Returns (global row indices, HostFrame)."""
    H = inp.height
    nranks = H // band_rows
    rank = int(fraction * nranks)
    tile = util.rowtile(H, band_rows, rank, nranks)
    assert tile.local_rows == band_rows
    frame = gpu.ob.HostFrame(inp.width, band_rows)
    gpu.ob.gbuffer_fill(frame, inp.rect, tile, inp.cam, inp.synthetic.fill, threads=threads)
    gpu.ob.lights(frame, inp.rect, tile, host_maps, inp.cam, inp.dirs, 2, 1, inp.spots, inp.spot_count, threads=threads)
    gpu.ob.composite(frame, inp.rect, tile, host_maps, inp.atm, inp.cam, inp.dirs, 0, tlut, slut, threads=threads)
    return util.global_rows(H, band_rows, rank, nranks), frame


def test_c2_full_size_frame_chain(gpu):
    """BASELINE config 2 at its own size: 1920x1080, no spot lights, the sun by the composite (skip count 1) through a
    pipeline-owned 2048^2 sun shadow map generated by recordDrawCommands, the moon by the lights pass, both LUTs at the
    reference's extents (512x128, 2048x1024). The whole chain runs on the GPU; deterministic, finite, opaque; the sun
    shadow map and BOTH LUTs are compared with the oracle's in full, and two 24-row bands (one across the horizon, one in
    the shadowed geometry) with the oracle's chained render of the same rows."""
    W, H, DIM = 1920, 1080, 2048
    inp = util.Inputs(W, H, elevation_degrees=35.0, spots=0)
    cameras, atmospheres, lights = staged(gpu, inp)

    def render():
        target = gpu.pl.SceneTexture(W, H, debug=True)
        deferred = gpu.pl.DeferredShadingPipeline((W, H), max_spot_lights=1, max_shadow_maps=1, shadow_map_dim=DIM)
        sky = gpu.pl.SkyViewComputePipeline.create()
        deferred.recordDrawCommands(None, inp.rect, target, 1, lights, None, 0, cameras, inp.synthetic.fill)
        sky.recordDrawCommands(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
        torch.cuda.synchronize()
        from syzygy_amd.pipelines import _memcpy2d_from

        sm = deferred.shadowMaps()
        assert sm.count == 1 and (sm.maps[0].width, sm.maps[0].height) == (DIM, DIM)
        shadow = _memcpy2d_from(sm.maps[0], DIM * 4, DIM).cpu().numpy().view(np.float32).reshape(DIM, DIM)
        out = (target.debug.cpu().numpy(), target.color_numpy(), shadow, sky.download_lut(sky.transmittanceLUT()),
               sky.download_lut(sky.skyviewLUT()))
        deferred.cleanup()
        sky.destroy()
        return out

    dbg, q, shadow, g_tlut, g_slut = render()
    dbg2, q2, _, _, _ = render()
    assert (q == q2).all() and (dbg.view(np.uint32) == dbg2.view(np.uint32)).all(), "non-deterministic output"
    assert np.isfinite(dbg).all() and (q[..., 3] == 65535).all()
    assert q[: H // 4, :, 2].mean() > q[: H // 4, :, 0].mean(), "daytime sky should be blue"

    want_shadow = gpu.ob.shadow_map(inp.sun, DIM, inp.synthetic.fill, threads=16)
    assert (shadow.view(np.uint32) == want_shadow.view(np.uint32)).all(), "2048^2 sun shadow map"
    assert 0.05 < (want_shadow > 0).mean() < 0.95
    tlut = gpu.ob.transmittance_lut(inp.atm, 512, 128, threads=16)
    slut = gpu.ob.skyview_lut(inp.atm, inp.cam, tlut, 2048, 1024, threads=16)
    assert_close(g_tlut, tlut, atol=1e-12, what="C2 transmittance LUT 512x128")
    assert_close(g_slut[..., :3], slut[..., :3], atol=1e-9, what="C2 sky-view LUT 2048x1024 (whole)")

    images = (gpu.abi.Image * 1)(gpu.ob.host_image(want_shadow, gpu.abi.SZG_FORMAT_D32_SFLOAT))
    host_maps = gpu.abi.ShadowMaps(1, 0, C.cast(images, C.POINTER(gpu.abi.Image)))
    shadowed_px = 0
    for fraction in (0.42, 0.8):
        rows, frame = _oracle_band(gpu, inp, 24, fraction, host_maps, tlut, slut)
        assert_close(dbg[rows], frame.debug, what=f"C2 band at {fraction}")
        lsb = np.abs(q[rows].astype(np.int32) - frame.color.astype(np.int32))
        assert lsb.max() <= 1, (fraction, lsb.max())
        # the sun's shadow map does something on these rows: the same band without it differs
        _, lit = _oracle_band(gpu, inp, 24, fraction, None, tlut, slut)
        shadowed_px += int((lit.color != frame.color).any(axis=-1).sum())
        print(f"C2 band at {fraction}: geometry {float((frame.depth > 0).mean()):.2f}, max UNORM16 diff {lsb.max()} LSB")
    assert shadowed_px > 1000, shadowed_px


def test_c5_full_size_frame_chain(gpu):
    """BASELINE config 5, one view at its own size: 3840x2160 with 256 spot lights, both LUTs at the reference's extents,
    whole chain on the GPU; deterministic, finite, opaque; two 24-row bands (across the horizon and deep in the lit
    geometry) against the oracle's chained render of the same rows. The sky-view LUT is the GPU's on both sides here (the
    oracle's full-size LUT is compared in the C2 test)."""
    W, H, SPOTS = 3840, 2160, 256
    inp = util.Inputs(W, H, elevation_degrees=35.0, spots=SPOTS)
    assert inp.spot_count == SPOTS
    dbg, q = render_gpu(gpu, inp, lut=((512, 128), (2048, 1024)))
    dbg2, q2 = render_gpu(gpu, inp, lut=((512, 128), (2048, 1024)))
    assert (q == q2).all() and (dbg.view(np.uint32) == dbg2.view(np.uint32)).all(), "non-deterministic output"
    assert np.isfinite(dbg).all() and (q[..., 3] == 65535).all()

    tlut = gpu.ob.transmittance_lut(inp.atm, 512, 128, threads=16)
    cameras, atmospheres, lights = staged(gpu, inp)
    skyp = gpu.pl.SkyViewComputePipeline.create()
    skyp.upload_lut(skyp.transmittanceLUT(), tlut)
    skyp.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
    torch.cuda.synchronize()
    slut = skyp.download_lut(skyp.skyviewLUT())
    skyp.destroy()
    for fraction in (0.42, 0.7):
        rows, frame = _oracle_band(gpu, inp, 24, fraction, None, tlut, slut)
        assert_close(dbg[rows], frame.debug, what=f"C5 band at {fraction}")
        lsb = np.abs(q[rows].astype(np.int32) - frame.color.astype(np.int32))
        assert lsb.max() <= 1, (fraction, lsb.max())
        print(f"C5 band at {fraction}: geometry {float((frame.depth > 0).mean()):.2f}, max UNORM16 diff {lsb.max()} LSB")
    # 256 lights light the scene: the lights pass alone (before the composite) is far from black on the geometry rows
    assert q[int(0.7 * H)][..., :3].mean() > 0


# ---------------------------------------------------------------------------
# round 2: LUT reuse, explicit invalidation, draw-rect offsets, staging ring, C-ABI collectives
# ---------------------------------------------------------------------------
def _frame_with(gpu, sky, deferred, target, inp, staged_buffers, lut_images):
    """One frame; returns (colour, transmittance LUT, sky-view LUT). `lut_images` were fetched once up front: every accessor
    call tells the pipeline that the caller may write the texels, which forces a recompute and would hide what reuse does."""
    cameras, atmospheres, lights = staged_buffers
    deferred.recordDrawCommands(None, inp.rect, target, 1, lights, inp.spots if inp.spot_count else None, 0, cameras, inp.synthetic.fill)
    sky.recordDrawCommands(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    return target.color_numpy().copy(), sky.download_lut(lut_images[0]).copy(), sky.download_lut(lut_images[1]).copy()


def test_lut_reuse_gives_identical_frames_and_tracks_every_input(gpu):
    """szg_skyview_set_lut_reuse: frames with reuse == frames without, bit for bit, over a sequence in which
    nothing changes, then the sun moves (atmosphere block), then only the camera moves (sky-view LUT alone), then the caller
    overwrites LUT texels behind the pipeline's back and says so (szg_skyview_invalidate_luts)."""
    W, H = 160, 96
    lut = dict(transmittance_extent=(128, 32), skyview_extent=(256, 128))
    from syzygy_amd import scene

    def inputs(elevation, height):
        cam = scene.default_camera()
        cam.cameraPosition[1] = -height
        return util.Inputs(W, H, elevation_degrees=elevation, spots=2, camera=cam)

    sequence = [inputs(35.0, 10.0), inputs(35.0, 10.0), inputs(35.0, 10.0), inputs(12.0, 10.0), inputs(12.0, 10.0),
                inputs(12.0, 900.0), inputs(12.0, 900.0)]
    results = {}
    for reuse in (False, True):
        sky = gpu.pl.SkyViewComputePipeline.create(**lut)
        alias = sky.skyviewLUT_tensor()  # a pointer the caller keeps for later
        lut_images = (sky.transmittanceLUT(), sky.skyviewLUT())
        sky.setLUTReuse(reuse)
        deferred = gpu.pl.DeferredShadingPipeline((W, H), max_spot_lights=2, max_shadow_maps=0)
        target = gpu.pl.SceneTexture(W, H)
        frames = []
        for inp in sequence:
            frames.append(_frame_with(gpu, sky, deferred, target, inp, staged(gpu, inp), lut_images))
        # the caller scribbles over the sky-view LUT through the pointer it kept, and tells the pipeline: the next frame
        # recomputes it although no parameter block changed
        alias.fill_(float("nan"))
        sky.invalidateLUTs(gpu.abi.SZG_LUT_SKYVIEW)
        frames.append(_frame_with(gpu, sky, deferred, target, sequence[-1], staged(gpu, sequence[-1]), lut_images))
        results[reuse] = frames
        deferred.cleanup()
        sky.destroy()
    for k, (a, b) in enumerate(zip(results[False], results[True])):
        for x, y, what in zip(a, b, ("frame", "transmittance LUT", "sky-view LUT")):
            assert (x.view(np.uint8) == y.view(np.uint8)).all(), f"step {k}: {what} differs with LUT reuse"
    r = results[True]
    assert (r[0][0] == r[2][0]).all() and not (r[2][0] == r[3][0]).all()       # the sun moved: another frame
    assert (r[4][1].view(np.uint32) == r[5][1].view(np.uint32)).all()           # camera moved: same transmittance LUT ...
    assert not (r[4][2].view(np.uint32) == r[5][2].view(np.uint32)).all()       # ... another sky-view LUT
    assert np.isfinite(r[7][2]).all() and (r[7][0] == r[6][0]).all()            # the scribbled LUT was recomputed


def test_lut_reuse_skips_the_lut_passes(gpu):
    """With reuse the second frame of an unchanged scene must not pay for the LUT passes (device time, events)."""
    inp = util.Inputs(64, 64, elevation_degrees=35.0, spots=0)
    cameras, atmospheres, lights = staged(gpu, inp)
    sky = gpu.pl.SkyViewComputePipeline.create()

    def luts_ms():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sky.recordTransmittance(None, 0, atmospheres)
        sky.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    luts_ms()
    plain = min(luts_ms() for _ in range(3))
    sky.setLUTReuse(True)
    first = luts_ms()
    cached = min(luts_ms() for _ in range(3))
    print(f"LUT passes: {plain:.3f} ms recomputed, {first:.3f} ms first frame with reuse, {cached:.3f} ms reused")
    assert first > 0.5 * plain and cached < 0.2 * plain
    sky.destroy()


def test_draw_rect_offset_is_refused(gpu):
    from syzygy_amd import SzgError

    inp = util.Inputs(64, 64, spots=1)
    cameras, atmospheres, lights = staged(gpu, inp)
    target = gpu.pl.SceneTexture(64, 64)
    deferred = gpu.pl.DeferredShadingPipeline((64, 64), max_spot_lights=1, max_shadow_maps=0)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(64, 16), skyview_extent=(64, 32))
    shifted = gpu.abi.Rect(8, 0, 32, 32)
    with pytest.raises(SzgError, match="offset"):
        deferred.recordDrawCommands(None, shifted, target, 1, lights, inp.spots, 0, cameras, inp.synthetic.fill)
    with pytest.raises(SzgError, match="offset"):
        sky.recordDrawCommands(None, target, gpu.abi.Rect(0, -4, 32, 32), deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0,
                               cameras, 0, lights)
    deferred.cleanup()
    sky.destroy()


def test_staged_buffers_survive_frames_in_flight(gpu):
    """TStagedBuffer.recordCopyToDevice is asynchronous: three frames with three different suns are recorded back to back
    WITHOUT a host sync in between (the copies of frame k+1 are staged while frame k's kernels have not started), each into
    its own target; every frame - not only the last - must be its own oracle frame."""
    W, H = 256, 144
    lut = ((128, 32), (256, 128))
    suns = [70.0, 20.0, 5.0]
    inps = [util.Inputs(W, H, elevation_degrees=e, spots=3) for e in suns]
    pl, abi = gpu.pl, gpu.abi
    cameras = pl.TStagedBuffer(abi.CameraPacked, 1)
    atmospheres = pl.TStagedBuffer(abi.AtmospherePacked, 1)
    lights = pl.TStagedBuffer(abi.DirectionalLightPacked, 2)
    targets = [pl.SceneTexture(W, H) for _ in suns]
    deferred = pl.DeferredShadingPipeline((W, H), max_spot_lights=3, max_shadow_maps=0)
    sky = pl.SkyViewComputePipeline.create(transmittance_extent=lut[0], skyview_extent=lut[1])
    # a long-running kernel in front, so that every host-side staging below happens before the first copy has run
    blocker = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    for _ in range(8):
        blocker.mul_(1.0001)
    for inp, target in zip(inps, targets):
        cameras.stage([inp.cam])
        atmospheres.stage([inp.atm])
        lights.stage([inp.sun, inp.moon])
        for b in (cameras, atmospheres, lights):
            b.recordCopyToDevice()
        deferred.recordDrawCommands(None, inp.rect, target, 1, lights, inp.spots, 0, cameras, inp.synthetic.fill)
        sky.recordDrawCommands(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    for inp, target, e in zip(inps, targets, suns):
        want = render_oracle(gpu, inp, lut=lut)
        lsb = np.abs(target.color_numpy().astype(np.int32) - want.color.astype(np.int32)).max()
        assert lsb <= 1, (e, lsb)
    deferred.cleanup()
    sky.destroy()


def test_c_abi_collectives_on_rccl_single_rank(gpu):
    """szg_rowtile_comm (abi.h "Multi-GPU collectives") in a world of one: communicator creation, the in-place all-gather
    of the sky-view LUT slices and the tile gather run on RCCL through the C entry points and leave the data intact."""
    from syzygy_amd import rowtile

    inp = util.Inputs(96, 64, elevation_degrees=25.0, spots=1)
    cameras, atmospheres, lights = staged(gpu, inp)
    comm = rowtile.Comm(0, 1, 0)
    assert comm.size() == 1
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=(128, 32), skyview_extent=(128, 64))
    sky.recordTransmittance(None, 0, atmospheres)
    sky.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
    torch.cuda.synchronize()
    whole = sky.download_lut(sky.skyviewLUT()).copy()
    assert sky.lutRowSlice(0, 1) == (0, 64) and sky.lutRowSlice(3, 4) == (48, 64)
    b, e = sky.lutRowSlice(0, 1)
    sky.recordSkyViewLUTRows(None, 0, atmospheres, 0, cameras, b, e)
    comm.allgather_skyview_lut(sky).wait()
    torch.cuda.synchronize()
    assert (sky.download_lut(sky.skyviewLUT()).view(np.uint32) == whole.view(np.uint32)).all()
    tile = torch.randint(-30000, 30000, (40, 96, 4), dtype=torch.int16, device="cuda")
    gathered, work = comm.gather_tiles(tile)
    work.wait()
    torch.cuda.synchronize()
    assert gathered.shape == (1, 40, 96, 4) and (gathered[0] == tile).all()
    from syzygy_amd import SzgError

    with pytest.raises(SzgError):
        sky.lutRowSlice(0, 3)  # 64 rows do not divide over 3 ranks
    comm.destroy()
    sky.destroy()


def _slut_status_word(gpu, sky):
    """The status dword behind the sky-view LUT's texels (szg_launch.hpp "sky-view LUT block")."""
    im = sky.skyviewLUT()
    word = gpu.abi.Image()
    word.data = im.data + im.width * im.height * 16
    word.width, word.height, word.pitch_bytes, word.format = 1, 1, 16, im.format
    return int(gpu.pl._memcpy2d_from(word, 4, 1).cpu().numpy().view(np.uint32).ravel()[0])


def test_slice_status_travels_with_the_lut_all_gather(gpu):
    """Review of round 1, item 9: the sky-view LUT's status word ("every texel is a finite number", which lets the composite
    leave samples unevaluated) used to be recomputed by a 32 MiB scan after every all-gather of the ranks' slices. Now each
    rank contributes the status of ITS rows and the words are exchanged beside the slices (world of one here, RCCL through the
    C-ABI): a clean slice gives a clean LUT and the plain frame; texels of unknown provenance (scribbled behind the pipeline's
    back, then declared) count as "not known to be finite", and the frame is the oracle's frame for that NaN LUT."""
    from syzygy_amd import rowtile

    W, H = 96, 64
    inp = util.Inputs(W, H, elevation_degrees=30.0, spots=2)
    cameras, atmospheres, lights = staged(gpu, inp)
    lut = ((128, 32), (128, 64))
    comm = rowtile.Comm(0, 1, 0)
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=lut[0], skyview_extent=lut[1])
    alias = sky.skyviewLUT_tensor()
    target = gpu.pl.SceneTexture(W, H, debug=True)
    deferred = gpu.pl.DeferredShadingPipeline((W, H), max_spot_lights=2, max_shadow_maps=0)
    deferred.recordDrawCommands(None, inp.rect, target, 1, lights, inp.spots, 0, cameras, inp.synthetic.fill)
    prior = target.color.clone()
    sky.recordDrawCommands(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    plain = target.debug.cpu().numpy().copy()

    # a slice that is NOT the whole LUT, then the rest: the last launch describes rows [32, 64) only, this rank contributes
    # rows [0, 64) -> "unknown" (1); a launch over exactly the contributed rows -> known and clean (0)
    sky.recordSkyViewLUTRows(None, 0, atmospheres, 0, cameras, 0, 32)
    sky.recordSkyViewLUTRows(None, 0, atmospheres, 0, cameras, 32, 64)
    comm.allgather_skyview_lut(sky).wait()
    torch.cuda.synchronize()
    assert _slut_status_word(gpu, sky) == 1
    sky.recordSkyViewLUTRows(None, 0, atmospheres, 0, cameras, 0, 64)
    comm.allgather_skyview_lut(sky).wait()
    torch.cuda.synchronize()
    assert _slut_status_word(gpu, sky) == 0
    target.color.copy_(prior)
    sky.recordComposite(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    assert (target.debug.cpu().numpy().view(np.uint32) == plain.view(np.uint32)).all()

    # texels of unknown provenance
    tlut = sky.download_lut(sky.transmittanceLUT()).copy()
    alias[: lut[1][1] // 2].fill_(float("nan"))
    torch.cuda.synchronize()
    slut = alias.cpu().numpy().copy()
    sky.invalidateLUTs(gpu.abi.SZG_LUT_SKYVIEW)
    comm.allgather_skyview_lut(sky).wait()
    torch.cuda.synchronize()
    assert _slut_status_word(gpu, sky) == 1
    target.color.copy_(prior)
    sky.recordComposite(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    got = target.debug.cpu().numpy()
    frame = gpu.ob.HostFrame(W, H)
    gpu.ob.gbuffer_fill(frame, inp.rect, None, inp.cam, inp.synthetic.fill, threads=8)
    gpu.ob.lights(frame, inp.rect, None, None, inp.cam, inp.dirs, 2, 1, inp.spots, 2, threads=8)
    gpu.ob.composite(frame, inp.rect, None, None, inp.atm, inp.cam, inp.dirs, 0, tlut, slut, threads=8)
    assert (np.isnan(got) == np.isnan(frame.debug)).all()
    ok = ~np.isnan(got)
    assert np.isnan(got).any() and (got[ok].view(np.uint32) == frame.debug[ok].view(np.uint32)).all()
    comm.destroy()
    deferred.cleanup()
    sky.destroy()


def test_texels_written_through_a_kept_pointer_are_rescanned_after_invalidate(gpu):
    """ADVICE (round 1): the pipeline keeps a status word per LUT (sky-view: "every texel finite") that lets the composite
    leave samples of non-metal pixels unevaluated. A caller that writes texels through a pointer it kept must say so
    (szg_skyview_invalidate_luts); the next composite then re-scans the texels, and a NaN LUT poisons exactly the pixels it
    poisons in the oracle - 0 * NaN in the reflection term of every geometry pixel included."""
    W, H = 96, 64
    inp = util.Inputs(W, H, elevation_degrees=30.0, spots=2)
    cameras, atmospheres, lights = staged(gpu, inp)
    lut = ((128, 32), (128, 64))
    sky = gpu.pl.SkyViewComputePipeline.create(transmittance_extent=lut[0], skyview_extent=lut[1])
    alias = sky.skyviewLUT_tensor()  # kept
    im_t = sky.transmittanceLUT()
    target = gpu.pl.SceneTexture(W, H, debug=True)
    deferred = gpu.pl.DeferredShadingPipeline((W, H), max_spot_lights=2, max_shadow_maps=0)
    deferred.recordDrawCommands(None, inp.rect, target, 1, lights, inp.spots, 0, cameras, inp.synthetic.fill)
    prior = target.color.clone()
    sky.recordDrawCommands(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    clean = target.debug.cpu().numpy().copy()
    assert np.isfinite(clean).all()
    tlut = sky.download_lut(im_t).copy()
    # scribble: half of the LUT becomes NaN, behind the pipeline's back; then the notice, then a composite ALONE
    alias[: lut[1][1] // 2].fill_(float("nan"))
    torch.cuda.synchronize()
    slut = alias.cpu().numpy().copy()
    sky.invalidateLUTs(gpu.abi.SZG_LUT_SKYVIEW)
    target.color.copy_(prior)
    sky.recordComposite(None, target, inp.rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
    torch.cuda.synchronize()
    got = target.debug.cpu().numpy()
    frame = gpu.ob.HostFrame(W, H)
    gpu.ob.gbuffer_fill(frame, inp.rect, None, inp.cam, inp.synthetic.fill, threads=8)
    gpu.ob.lights(frame, inp.rect, None, None, inp.cam, inp.dirs, 2, 1, inp.spots, 2, threads=8)
    gpu.ob.composite(frame, inp.rect, None, None, inp.atm, inp.cam, inp.dirs, 0, tlut, slut, threads=8)
    assert (np.isnan(got) == np.isnan(frame.debug)).all(), "NaN pattern differs from the oracle's"
    assert np.isnan(got).any() and not np.isnan(got).all()
    ok = ~np.isnan(got)
    assert (got[ok].view(np.uint32) == frame.debug[ok].view(np.uint32)).all()
    deferred.cleanup()
    sky.destroy()
