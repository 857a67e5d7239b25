"""Cameras.cpp restated on the host (product) and in the oracle, pinned by the reference's own
camera tests: unit_tests/Cameras.test.cpp:22-44 (values), Cameras.cpp:164-189 and
Technique.cpp:118-152 (inline unittest blocks)."""
import math

import numpy as np
import pytest

import master_amd as ma
import oracle

IMPLS = [("product", ma.camera_setup, ma.ray_direction, ma.pixel_position),
         ("oracle", oracle.camera_setup, oracle.ray_direction, oracle.pixel_position)]


def cam(pos, direction, up, fovx):
    return ma.Camera((ma.C.c_float * 3)(*pos), (ma.C.c_float * 3)(*direction), (ma.C.c_float * 3)(*up), fovx)


@pytest.mark.parametrize("name,setup,raydir,pixpos", IMPLS)
def test_fov_values_of_reference_gtest(name, setup, raydir, pixpos):
    c = cam((0, 0, 0), (0, 0, -1), (0, 1, 0), math.pi / 2)
    # Cameras.test.cpp:24-25: fovy == pi/2 at aspect 1, 1.2870022 at aspect 4/3
    assert setup(c, 1.0).fovy == pytest.approx(math.pi / 2, rel=4e-7)
    assert setup(c, 4.0 / 3.0).fovy == pytest.approx(1.2870022, rel=4e-7)


@pytest.mark.parametrize("name,setup,raydir,pixpos", IMPLS)
def test_corner_ray_directions_of_reference_gtest(name, setup, raydir, pixpos):
    # Cameras.test.cpp:32-47 (800x600, fovx 90 deg): centre ray -> (0,0,-1).  The 800x600 corner values of
    # that (stale, excluded-from-build) gtest, (-+0.685994, -+0.514496, -0.514496), belong to the removed
    # Cameras::shoot API that treated the angle as VERTICAL; today's ray_direction (Cameras.cpp:120-127) with
    # fovy derived from fovx (Cameras.cpp:81-88) gives x/z = tan(fovx/2) = 1 at the left/right edge instead.
    c = cam((0, 0, 0), (0, 0, -1), (0, 1, 0), math.pi / 2)
    f = setup(c, 800.0 / 600.0)
    m = np.array(list(f.view_to_world), np.float32).reshape(3, 3).T  # column-major -> matrix
    np.testing.assert_allclose(m @ raydir(400.0, 300.0, 800.0, 600.0, f.focal_length_y), [0, 0, -1], atol=1e-5)
    corner = m @ raydir(0.0, 0.0, 800.0, 600.0, f.focal_length_y)
    np.testing.assert_allclose(corner, np.array([-4 / 3, -1, -4 / 3]) / np.linalg.norm([4 / 3, 1, 4 / 3]), atol=1e-6)
    np.testing.assert_allclose(m @ raydir(800.0, 600.0, 800.0, 600.0, f.focal_length_y), -corner * [1, 1, -1], atol=1e-6)
    f1 = setup(c, 1.0)
    np.testing.assert_allclose(m @ raydir(0.0, 0.0, 100.0, 100.0, f1.focal_length_y), [-0.577, -0.577, -0.577], atol=1e-3)
    # Cameras.test.cpp:50-64: camera looking along +x
    c2 = cam((0, 0, 0), (1, 0, 0), (0, 1, 0), math.pi / 2)
    f2 = setup(c2, 1.0)
    m2 = np.array(list(f2.view_to_world), np.float32).reshape(3, 3).T
    np.testing.assert_allclose(m2 @ raydir(0.0, 0.0, 100.0, 100.0, f2.focal_length_y), [0.577, -0.577, -0.577], atol=1e-3)


@pytest.mark.parametrize("name,setup,raydir,pixpos", IMPLS)
def test_ray_direction_pixel_position_round_trip(name, setup, raydir, pixpos):
    # Cameras.cpp:164-173: resolution 800x600, fov_y = pi/2, position (123.4, 345.0)
    fl = 1.0 / math.tan(math.pi / 4)
    d = raydir(123.4, 345.0, 800.0, 600.0, fl)
    p = pixpos(d, 800.0, 600.0, fl)
    np.testing.assert_allclose(p, [123.4, 345.0], rtol=0, atol=64 * 345.0 * 1.2e-7)  # almost_eq: 64 ULP (unittest.cpp:173-175)


@pytest.mark.parametrize("name,setup,raydir,pixpos", IMPLS)
def test_look_at_frames_of_inline_unittests(name, setup, raydir, pixpos):
    # Technique.cpp:118-136: position (1,3,2), direction (1,0,10), up (0,1,0)
    d = np.array([1.0, 0.0, 10.0]); dn = d / np.linalg.norm(d)
    f = setup(cam((1, 3, 2), dn, (0, 1, 0), 1.0), 1.0)
    v2w = np.array(list(f.view_to_world), np.float32).reshape(3, 3)  # rows = columns of the mat3
    np.testing.assert_allclose(v2w[1], [0, 1, 0], atol=2e-7)                       # tangent[0] = view_to_world[1]
    np.testing.assert_allclose(v2w[2], -dn, atol=2e-7)                             # tangent[1] = view_to_world[2] = -direction
    np.testing.assert_allclose(v2w[0], np.cross(d, [0, 1, 0]) / np.linalg.norm(np.cross(d, [0, 1, 0])), atol=2e-7)
    # Technique.cpp:142-152
    d = np.array([1.0, 2.0, 3.0]); dn = d / np.linalg.norm(d)
    f = setup(cam((1, 1, 1), dn, (0, 1, 0), 1.0), 1.0)
    np.testing.assert_allclose(np.array(list(f.view_to_world), np.float32).reshape(3, 3)[2], -dn, atol=2e-7)
    # Cameras.cpp:175-189: looking down -z: world_to_view columns are the axes
    f = setup(cam((1, 3, 2), (0, 0, -1), (0, 1, 0), 1.0), 1.0)
    w2v = np.array(list(f.world_to_view), np.float32).reshape(3, 3)
    np.testing.assert_allclose(w2v, np.eye(3), atol=1e-7)
    np.testing.assert_allclose(list(f.position), [1, 3, 2])


def test_product_and_oracle_camera_agree_bitwise():
    """Host code of the product and the oracle state the same arithmetic (fma placement included):
    axis-aligned cameras hide a mismatch, rotated ones (TestCaseFurnace, random) do not."""
    import glob
    import os

    from conftest import ROOT

    cams = []
    for f in glob.glob(os.path.join(ROOT, "scenes", "*.miscene")):
        cams += ma.Scene.load(f).cameras
    rng = np.random.default_rng(0)
    for _ in range(50):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        cams.append(cam(rng.normal(size=3), d, (0, 0, 1), float(rng.uniform(0.2, 2.5))))
    for c in cams:
        for aspect in (1.0, 4.0 / 3.0, 16.0 / 9.0, 0.5):
            assert bytes(ma.camera_setup(c, aspect)) == bytes(oracle.camera_setup(c, aspect))


def test_cornell_camera_fov(cornell):
    # lens 45 mm / sensor 30 mm => fovx = 2 atan(15/45) = 36.87 deg
    assert math.degrees(cornell.cameras[0].fovx) == pytest.approx(36.87, abs=0.01)
