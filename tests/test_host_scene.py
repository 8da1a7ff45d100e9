"""Host input prep (include/szg/host.h) against
  * the reference's own known-answer tests for the euler <-> forward convention:
    geometry/geometrytests.cpp:106-186 (3 + 14 inverse round trips, 22 equality cases,
    tolerance 3 * FLT_EPSILON, geometrytests.cpp:18), and
  * an independent numpy (float64) restatement of the glm 1.0.1 formulas the reference
    calls for projection / view / inverse — those are unpinned by the
    reference's tests.
"""
import ctypes as C
import math

import numpy as np
import pytest

from syzygy_amd import abi, lib, scene

EPS = 3.0 * np.finfo(np.float32).eps  # geometrytests.cpp:18
F = np.float32

# geometrystatics.hpp:7-9
FWD = np.array([0.0, 0.0, 1.0], F)
UP = np.array([0.0, -1.0, 0.0], F)
RIGHT = np.array([1.0, 0.0, 0.0], F)
PI, HALF_PI, QUARTER_PI = F(math.pi), F(math.pi / 2), F(math.pi / 4)


def glm_normalize(v):
    v = np.asarray(v, F)
    d = F(v[0] * v[0]) + F(v[1] * v[1]) + F(v[2] * v[2])
    return (v * (F(1.0) / np.sqrt(F(d)))).astype(F)


def eulers_from_forward(v):
    out = (C.c_float * 3)()
    lib().szg_eulers_from_forward(abi.f3(*[float(x) for x in v]), out)
    return np.array(out, F)


def forward_from_eulers(e):
    out = (C.c_float * 3)()
    lib().szg_forward_from_eulers(abi.f3(*[float(x) for x in e]), out)
    return np.array(out, F)


# geometrytests.cpp:120-186, in order
EQUALITY_CASES = [
    (FWD, (0, 0, 0)),
    (-FWD, (0, 0, -PI)),
    (RIGHT, (0, 0, HALF_PI)),
    (-RIGHT, (0, 0, -HALF_PI)),
    (UP, (HALF_PI, 0, 0)),
    (-UP, (-HALF_PI, 0, 0)),
    (FWD + UP, (QUARTER_PI, 0, 0)),
    (FWD - UP, (-QUARTER_PI, 0, 0)),
    (-FWD - UP, (-QUARTER_PI, 0, PI)),
    (-FWD + UP, (QUARTER_PI, 0, PI)),
    (UP + RIGHT, (QUARTER_PI, 0, HALF_PI)),
    (UP - RIGHT, (QUARTER_PI, 0, -HALF_PI)),
    (-UP - RIGHT, (-QUARTER_PI, 0, -HALF_PI)),
    (-UP + RIGHT, (-QUARTER_PI, 0, HALF_PI)),
    (RIGHT + FWD, (0, 0, QUARTER_PI)),
    (RIGHT - FWD, (0, 0, F(3) * QUARTER_PI)),
    (-RIGHT - FWD, (0, 0, F(-3) * QUARTER_PI)),
    (-RIGHT + FWD, (0, 0, -QUARTER_PI)),
]


@pytest.mark.parametrize("case", range(len(EQUALITY_CASES)))
def test_reference_euler_known_answers(case):
    """eulerAnglesTestEquality, geometrytests.cpp:71-100."""
    unnormalized, expected = EQUALITY_CASES[case]
    got = eulers_from_forward(glm_normalize(unnormalized))
    assert np.all(np.abs(got - np.array(expected, F)) < EPS), (got, expected)


def combos(a, b, c):
    return [a, b, c, a + b, b + c, c + a, a + b + c]


INVERSE_CASES = ([np.array(v, F) for v in ((1, 0, 0), (0, 1, 0), (0, 0, 1))] + combos(FWD, RIGHT, UP) +
                 combos(-FWD, -RIGHT, -UP))


@pytest.mark.parametrize("case", range(len(INVERSE_CASES)))
def test_reference_euler_inverse_round_trip(case):
    """eulerAnglesTestInverse, geometrytests.cpp:19-47 via :106-118."""
    forward = glm_normalize(INVERSE_CASES[case])
    again = forward_from_eulers(eulers_from_forward(forward))
    assert np.all(np.abs(forward - again) < EPS), (forward, again)


def test_zero_forward_gives_zero_eulers():
    assert (eulers_from_forward((0.0, 0.0, 0.0)) == 0).all()  # geometryhelpers.cpp:109-112


# ---- independent float64 restatement of the glm routines ------------------------------
def np_yaw_pitch_roll(yaw, pitch, roll):
    ch, sh, cp, sp, cb, sb = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
    m = np.zeros((4, 4))  # [row, col]
    m[:, 0] = [ch * cb + sh * sp * sb, sb * cp, -sh * cb + ch * sp * sb, 0]
    m[:, 1] = [-ch * sb + sh * sp * cb, cb * cp, sb * sh + ch * sp * cb, 0]
    m[:, 2] = [sh * cp, -sp, ch * cp, 0]
    m[:, 3] = [0, 0, 0, 1]
    return m


def np_orientate4(e):
    return np_yaw_pitch_roll(e[2], e[0], e[1])


def np_translate(p):
    m = np.eye(4)
    m[:3, 3] = p
    return m


def np_perspective_lh_zo(fovy, aspect, near, far):
    t = math.tan(fovy / 2)
    m = np.zeros((4, 4))
    m[0, 0] = 1 / (aspect * t)
    m[1, 1] = 1 / t
    m[2, 2] = far / (far - near)
    m[3, 2] = 1
    m[2, 3] = -(far * near) / (far - near)
    return m


def np_ortho_lh_zo(left, right, b, t, near, far):
    m = np.eye(4)
    m[0, 0] = 2 / (right - left)
    m[1, 1] = 2 / (t - b)
    m[2, 2] = 1 / (far - near)
    m[0, 3] = -(right + left) / (right - left)
    m[1, 3] = -(t + b) / (t - b)
    m[2, 3] = -near / (far - near)
    return m


def close(a, b, tol=2e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.abs(b)))


@pytest.mark.parametrize("eulers", [(0, 0, 0), (0.3, 0.0, -1.2), (-0.7, 0.2, 2.9), (4.3633, 0, 0)])
def test_view_and_transform(eulers):
    pos = (3.0, -10.0, -13.0)
    m = abi.Mat4()
    lib().szg_transform_vk(abi.f3(*pos), abi.f3(*eulers), C.byref(m))
    want = np_translate(pos) @ np_orientate4(eulers)
    assert close(m.to_numpy(), want)
    lib().szg_view_vk(abi.f3(*pos), abi.f3(*eulers), C.byref(m))
    assert close(m.to_numpy(), np.linalg.inv(want), 1e-4)


def test_projection_is_reverse_z():
    """geometryhelpers.cpp:83-95: perspectiveLH_ZO with near/far swapped."""
    m = abi.Mat4()
    lib().szg_projection_vk(70.0, 16 / 9, 0.1, 10000.0, C.byref(m))
    want = np_perspective_lh_zo(math.radians(70.0), 16 / 9, 10000.0, 0.1)
    assert close(m.to_numpy(), want)
    P = m.to_numpy().astype(np.float64)
    near = P @ np.array([0, 0, 0.1, 1.0])
    far = P @ np.array([0, 0, 10000.0, 1.0])
    assert abs(near[2] / near[3] - 1.0) < 1e-5 and abs(far[2] / far[3]) < 1e-5


def test_mat4_inverse_and_mul():
    rng = np.random.default_rng(1)
    for _ in range(20):
        a = rng.normal(size=(4, 4)).astype(np.float32) + np.eye(4, dtype=np.float32) * 3
        b = rng.normal(size=(4, 4)).astype(np.float32)
        ma, mb, out = abi.Mat4.from_numpy(a), abi.Mat4.from_numpy(b), abi.Mat4()
        lib().szg_mat4_mul(C.byref(ma), C.byref(mb), C.byref(out))
        assert close(out.to_numpy(), a.astype(np.float64) @ b.astype(np.float64), 1e-5)
        lib().szg_mat4_inverse(C.byref(ma), C.byref(out))
        assert close(out.to_numpy(), np.linalg.inv(a.astype(np.float64)), 1e-4)
        lib().szg_mat4_inverse_transpose(C.byref(ma), C.byref(out))
        assert close(out.to_numpy(), np.linalg.inv(a.astype(np.float64)).T, 1e-4)


def test_camera_packed_block():
    cam = scene.default_camera()
    assert list(cam.cameraPosition) == [0.0, -10.0, -13.0] and cam.fovDegrees == 70.0  # scene.cpp:77-83
    cam.eulerAngles[:] = [0.2, 0.0, -0.4]
    packed = scene.camera_packed(cam, 16 / 9)
    P = np_perspective_lh_zo(math.radians(70.0), 16 / 9, 10000.0, 0.1)
    T = np_translate(list(cam.cameraPosition)) @ np_orientate4(list(cam.eulerAngles))
    V = np.linalg.inv(T)
    assert close(packed.projection.to_numpy(), P)
    assert close(packed.inverseProjection.to_numpy(), np.linalg.inv(P), 1e-3)
    assert close(packed.view.to_numpy(), V, 1e-4)
    assert close(packed.viewInverseTranspose.to_numpy(), np.linalg.inv(V).T, 1e-4)
    assert close(packed.rotation.to_numpy(), np_orientate4(list(cam.eulerAngles)))
    assert close(packed.projViewInverse.to_numpy(), np.linalg.inv(P @ V), 2e-3)
    assert close(list(packed.forwardWorld), list(np_orientate4(list(cam.eulerAngles)) @ np.array([0, 0, 1, 0])))
    assert list(packed.position) == [0.0, -10.0, -13.0, 1.0]


def test_atmosphere_earth_defaults_and_sun_direction():
    a = scene.default_atmosphere()
    assert a.planetRadiusMegameters == F(6.360) and a.atmosphereRadiusMegameters == F(6.420)  # scene.cpp:56-57
    assert list(a.scatteringRayleighPerMegameter) == [F(5.802), F(13.558), F(33.1)]
    assert a.altitudeDecayRayleighMegameters == F(8.0) / F(1000.0) and a.altitudeDecayMieMegameters == F(1.2) / F(1000.0)
    assert abs(a.sunAngularRadius - math.radians(32 / 60)) < 1e-9
    # for sunEuler = (p, 0, 0) the packed incident direction is (0, sin p, cos p)
    for elevation in (70.0, 5.0, -3.0):
        pitch = math.pi + math.radians(elevation)
        a.sunEulerAngles[:] = [pitch, 0.0, 0.0]
        p = scene.atmosphere_packed(a)
        assert close(list(p.incidentDirectionSun), [0.0, math.sin(pitch), math.cos(pitch)], 1e-6)
        # elevation above the horizon: the direction TO the sun has +y (up) = sin(elevation)
        assert abs(-p.incidentDirectionSun[1] - math.sin(math.radians(elevation))) < 1e-6
    # Mie absorption coefficient is carried in the block even though the shaders never read it
    assert list(p.absorptionMiePerMm) == [F(4.40)] * 3 and list(p.absorptionRayleighPerMm) == [0.0] * 3


def test_sun_animation_noon_and_tick():
    anim = abi.SunAnimation()
    lib().szg_sun_animation_default(C.byref(anim))
    assert anim.time == 0.5 and anim.speed == 100.0
    a = scene.default_atmosphere()
    frozen = abi.SunAnimation(1, 0.5, 100.0, 0)
    lib().szg_scene_tick_sun(C.byref(frozen), C.byref(a), 10.0)
    # time 0.5 -> pitch 3pi/2 -> incident (0, -1, 0): noon (scene.cpp:565-574)
    assert abs(a.sunEulerAngles[0] - 1.5 * math.pi) < 1e-6
    inc = scene.atmosphere_packed(a).incidentDirectionSun
    assert abs(inc[1] + 1.0) < 1e-6 and abs(inc[0]) < 1e-6 and abs(inc[2]) < 1e-6
    lib().szg_scene_tick_sun(C.byref(anim), C.byref(a), 864.0)  # 100 * 864 / 86400 = 1 day fraction -> wraps to 0.5
    assert abs(anim.time - 0.5) < 1e-5
    night = abi.SunAnimation(0, 0.1, 100.0, 1)
    lib().szg_scene_tick_sun(C.byref(night), C.byref(a), 0.0)
    assert abs(night.time - (0.25 - 0.015)) < 1e-7  # skipNight jumps to the sunrise horizon (scene.cpp:546-563)


def test_directional_and_spot_lights():
    bounds = scene.aabb((0, -5, 10), (30, 8, 40))
    a = scene.default_atmosphere(scene.sun_euler_for_elevation(40.0))
    atm, sun, moon = scene.atmosphere_baked(a, bounds)
    assert sun.strength == 4.0 and list(sun.color) == [1.0, 1.0, 1.0, 1.0]  # scene.cpp:584-598
    assert list(moon.color) == [F(0.3), F(0.4), F(0.6), 1.0]
    assert 0.0 <= moon.strength <= F(0.02) + 1e-9
    fwd = scene.forward_from_eulers(tuple(a.sunEulerAngles))
    assert close(list(sun.forward)[:3], fwd)
    # the view is the inverse of a pure rotation; the projection is an orthographic box around the bounds
    V = sun.view.to_numpy().astype(np.float64)
    assert close(V[:3, :3] @ V[:3, :3].T, np.eye(3), 1e-5) and close(V[:3, 3], [0, 0, 0])
    Pm = sun.projection.to_numpy().astype(np.float64)
    assert Pm[3, 3] == 1.0 and Pm[3, 2] == 0.0
    # every corner of the bounds lands inside x,y in [-1,1]... except that projectPointOnPlane
    # (geometryhelpers.cpp:55-61) ADDS the projection, which is restated literally; so only check finiteness
    assert np.isfinite(Pm).all()

    spot = scene.make_spot((1, 0, 0), (-20, -28, -20), scene.eulers_from_forward((1, 1, 1)))
    assert spot.strength == 1000.0 and spot.falloffFactor == 1.0 and spot.falloffDistance == 1.0  # scene.cpp:218-229
    want = np_perspective_lh_zo(math.radians(30.0), 1.0, 1000.0, 0.1)
    assert close(spot.projection.to_numpy(), want)
    assert list(spot.position) == [-20.0, -28.0, -20.0, 1.0]
    f = np.array(list(spot.forward)[:3])
    assert close(f, np.array([1, 1, 1]) / math.sqrt(3), 1e-6)


def test_spot_ring_is_deterministic():
    a = scene.spot_ring(16)
    b = scene.spot_ring(16)
    assert bytes(a) == bytes(b)
    assert len({tuple(l.position) for l in a}) == 16


def test_transform_matrix_and_mesh_instance_tick():
    """Transform::toMatrix (transform.cpp:11-15) and tickMeshInstance (scene.cpp:461-523) against a float64 numpy
    restatement: translate * orientate4 * scale, the diagonal wave y = sin(t + (x + 10 + z + 10) / 3.1415), the spin
    eulers.z += dt, and modelInverseTranspose = transpose(inverse(model))."""
    t, e, s = (3.0, -8.0, 6.0), (0.3, -0.2, 1.1), (5.0, 2.0, 0.5)
    m = abi.Mat4()
    lib().szg_transform_matrix(abi.f3(*t), abi.f3(*e), abi.f3(*s), C.byref(m))
    want = np_translate(t) @ np_orientate4(e) @ np.diag([s[0], s[1], s[2], 1.0])
    assert close(m.to_numpy(), want)

    n = 3
    originals = (abi.Transform * n)()
    for i in range(n):
        originals[i].translation[:] = [float(-4 + 5 * i), -8.0, float(2 - 3 * i)]
        originals[i].eulerAnglesRadians[:] = [0.1 * i, 0.0, 0.2]
        originals[i].scale[:] = [1.0 + i, 1.0, 2.0]
    for animation in (abi.SZG_INSTANCE_ANIMATION_NONE, abi.SZG_INSTANCE_ANIMATION_DIAGONAL_WAVE,
                      abi.SZG_INSTANCE_ANIMATION_SPIN_ALONG_WORLD_UP):
        transforms = (abi.Transform * n)(*originals)
        models, mits = (abi.Mat4 * n)(), (abi.Mat4 * n)()
        elapsed, dt = 12.345, 0.016
        lib().szg_tick_mesh_instance(animation, originals, transforms, n, elapsed, dt, models, mits)
        for i in range(n):
            tr = np.array(originals[i].translation, np.float64)
            eu = np.array(originals[i].eulerAnglesRadians, np.float64)
            if animation == abi.SZG_INSTANCE_ANIMATION_DIAGONAL_WAVE:
                tr[1] += math.sin(elapsed + (tr[0] + 10.0 + tr[2] + 10.0) / 3.1415)
            if animation == abi.SZG_INSTANCE_ANIMATION_SPIN_ALONG_WORLD_UP:
                eu[2] += dt
            assert close(np.array(transforms[i].translation), tr) and close(np.array(transforms[i].eulerAnglesRadians), eu)
            want = np_translate(tr) @ np_orientate4(eu) @ np.diag(list(originals[i].scale) + [1.0])
            assert close(models[i].to_numpy(), want, 1e-5)
            assert close(mits[i].to_numpy(), np.linalg.inv(want).T, 1e-4)


def test_calculate_shadow_bounds_of_the_default_scene():
    """Scene::calculateShadowBounds (scene.cpp:95-148) on the editor's start-up scene (editor.cpp:500-545): two cubes of scale
    5 at (0, -8, +-6) and the 20 x 20 floor at y = -1 -> x in [-20, 20], y in [-13, -1], z in [-20, 20]."""
    cube, plane = abi.AABB(), abi.AABB()
    lib().szg_aabb_create(abi.f3(-1, -1, -1), abi.f3(1, 1, 1), C.byref(cube))
    lib().szg_aabb_create(abi.f3(-1, 0, -1), abi.f3(1, 0, 1), C.byref(plane))

    def transforms(*items):
        arr = (abi.Transform * len(items))()
        for t, (tr, sc) in zip(arr, items):
            t.translation[:], t.eulerAnglesRadians[:], t.scale[:] = list(tr), [0.0, 0.0, 0.0], list(sc)
        return arr

    t1, t2, t3 = transforms(((0, -8, 6), (5, 5, 5))), transforms(((0, -8, -6), (5, 5, 5))), transforms(((0, -1, 0), (20, 1, 20)))
    casters = (abi.ShadowCaster * 3)(abi.ShadowCaster(cube, t1, 1, 1, 1, 0), abi.ShadowCaster(cube, t2, 1, 1, 1, 0),
                                     abi.ShadowCaster(plane, t3, 1, 1, 1, 0))
    out = abi.AABB()
    assert lib().szg_calculate_shadow_bounds(casters, 3, C.byref(out)) == 1
    assert np.allclose(list(out.center), [0.0, -7.0, 0.0]) and np.allclose(list(out.half_extent), [20.0, 6.0, 20.0])
    # a caster that is hidden or casts no shadow does not count; none at all leaves the bounds empty
    casters[2].casts_shadow = 0
    assert lib().szg_calculate_shadow_bounds(casters, 3, C.byref(out)) == 1
    assert np.allclose(list(out.center), [0.0, -8.0, 0.0]) and np.allclose(list(out.half_extent), [5.0, 5.0, 11.0])
    for c in casters:
        c.render = 0
    assert lib().szg_calculate_shadow_bounds(casters, 3, C.byref(out)) == 0
    assert list(out.center) == [0.0, 0.0, 0.0] and list(out.half_extent) == [0.0, 0.0, 0.0]


def test_transform_look_at_is_eulers_from_the_normalised_direction():
    """Transform::lookAt(Ray::create(from, to), scale) (geometry/transform.cpp:17-28)."""
    import ctypes as C

    from syzygy_amd import abi, lib

    rng = np.random.default_rng(12)
    for _ in range(50):
        a, b = rng.normal(0, 10, 3).astype(np.float32), rng.normal(0, 10, 3).astype(np.float32)
        scale = rng.uniform(0.5, 2.0, 3).astype(np.float32)
        t = abi.Transform()
        lib().szg_transform_look_at(abi.f3(*a), abi.f3(*b), abi.f3(*scale), C.byref(t))
        assert (np.array(t.translation, np.float32) == a).all() and (np.array(t.scale, np.float32) == scale).all()
        d = (b - a).astype(np.float32)
        forward = d * (np.float32(1.0) / np.sqrt(np.dot(d, d).astype(np.float32)))
        want = (C.c_float * 3)()
        lib().szg_eulers_from_forward(abi.f3(*forward), want)
        assert np.allclose(np.array(t.eulerAnglesRadians), np.array(want), rtol=0, atol=2e-7)
        # and the forward vector of those eulers is the direction again
        back = (C.c_float * 3)()
        lib().szg_forward_from_eulers(t.eulerAnglesRadians, back)
        assert np.allclose(np.array(back), d / np.linalg.norm(d), atol=2e-6)
