"""The progressive display contract of the reference app (ContentView.swift:172-235,
Shaders.metal:151-158): oracle known answers on the CPU, HIP kernel against the oracle
bit for bit on the GPU, and the view's pass-counter rules.  No reference test covers this
step: the expected values below are worked by hand from the kernel's expressions."""
import numpy as np
import pytest

import oracle_bridge as ob
from raytracer_amd import scenes
from raytracer_amd.scene import CameraConfig


def test_oracle_present_hand_worked_values():
    # pass 1: the old content has weight 0.  0.5 -> 512; sqrt(0.5)*255 = 180.3 -> 180
    acc, bgra = ob.progressive_present([[0.5, 2.0, 0.0]], 1, 1, [0x3FFFFFFF])
    r, g, b = acc[0] & 1023, (acc[0] >> 10) & 1023, (acc[0] >> 20) & 1023
    assert (r, g, b) == (512, 1023, 0)                   # 2.0 is clamped to 1 -> 1024 -> min 1023
    assert list(bgra[0]) == [0, 255, 180, 255]           # b, g, r, a
    # pass 2 with colour 1.0 on r: old = (512 + 0.49999997)/1024, t = 1/2 -> 0.75024414 -> 768
    acc2, bgra2 = ob.progressive_present([[1.0, 1.0, 0.0]], 1, 2, acc)
    assert acc2[0] & 1023 == 768 and bgra2[0][2] == 221  # sqrt(0.75024414)*255 = 220.87
    # samples_per_pixel divides the sums
    acc3, _ = ob.progressive_present([[2.0, 4.0, 0.0]], 4, 1, [0])
    assert (acc3[0] & 1023, (acc3[0] >> 10) & 1023) == (512, 1023)


def test_oracle_present_running_mean_stays_put_and_converges():
    rs = np.random.RandomState(3)
    colour = rs.uniform(0, 1, size=(500, 3)).astype(np.float32)
    acc = np.zeros(500, dtype=np.uint32)
    for k in range(1, 200):
        acc, bgra = ob.progressive_present(colour, 1, k, acc)
    got = np.stack([(acc >> (10 * c)) & 1023 for c in range(3)], axis=-1)
    # the same colour every pass: the stored mean never drifts by more than one 10-bit step
    assert np.abs(got - np.floor(colour * 1024)).max() <= 1
    assert (bgra[:, 3] == 255).all()
    # alternating 0 / 1 converges to one half
    acc = np.zeros(4, dtype=np.uint32)
    for k in range(1, 200):
        acc, _ = ob.progressive_present(np.full((4, 3), float(k & 1), dtype=np.float32), 1, k, acc)
    assert np.abs(((acc & 1023).astype(int)) - 512).max() <= 8


def test_camera_orbit_accessors_round_trip():
    """Config.swift:43-59."""
    cam = CameraConfig(look_from=(13, 2, 3), look_at=(0, 0, 0))
    yaw, pitch, dist = cam.relative_position
    assert abs(dist - np.sqrt(13 * 13 + 4 + 9)) < 1e-5
    assert abs(yaw - np.degrees(np.arctan2(3, 13))) < 1e-4 and abs(pitch - np.degrees(np.arctan2(2, np.hypot(13, 3)))) < 1e-4
    other = cam.copy()
    other.relative_position = (yaw, pitch, dist)
    assert np.allclose(other.look_from, [13, 2, 3], atol=1e-5)
    other.relative_position = (90.0, 0.0, 5.0)
    assert np.allclose(other.look_from, [0, 0, 5], atol=1e-5)
    assert other != cam and cam == cam.copy()


@pytest.mark.gpu
def test_present_kernel_matches_oracle_bit_for_bit():
    import ctypes as C
    import torch
    from raytracer_amd._native import check, lib
    from raytracer_amd.renderer import Context
    ctx = Context(0)
    rs = np.random.RandomState(11)
    w, h = 173, 61                                   # ragged: not a multiple of the workgroup
    n = w * h
    acc_ref = rs.randint(0, 1 << 30, size=n).astype(np.uint32)
    acc_ref[::3] = 0x3FFFFFFF                        # saturated pixels
    acc_dev = torch.from_numpy(acc_ref.astype(np.int32)).cuda()
    out_dev = torch.zeros((n, 4), dtype=torch.uint8, device="cuda")
    for k, spp in [(1, 1), (2, 1), (3, 4), (7, 1), (199, 2), (200, 1)]:
        ps = (rs.uniform(0, 1.3, size=(n, 3)) * spp).astype(np.float32)
        ps[rs.randint(0, n, 50)] = 0.0
        ps[rs.randint(0, n, 50)] *= 1e-6
        ps[rs.randint(0, n, 20)] = 1e9               # fireflies
        ps[rs.randint(0, n, 300)] = 0.5 * spp        # round values: products that land on a 10-bit boundary
        ps[rs.randint(0, n, 300)] = 0.25 * spp
        ps_dev = torch.from_numpy(ps).cuda()
        check(lib.rt_progressive_present(ctx.h, ps_dev.data_ptr(), spp, k, acc_dev.data_ptr(), out_dev.data_ptr(), w, h,
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        acc_ref, bgra_ref = ob.progressive_present(ps, spp, k, acc_ref)
        assert np.array_equal(acc_dev.cpu().numpy().astype(np.uint32), acc_ref), "pass %d" % k
        assert np.array_equal(out_dev.cpu().numpy(), bgra_ref), "pass %d" % k
    assert lib.rt_progressive_present(ctx.h, ps_dev.data_ptr(), 1, 0, acc_dev.data_ptr(), out_dev.data_ptr(), w, h, None) < 0


@pytest.mark.gpu
def test_progressive_view_follows_the_app_loop():
    import torch
    from raytracer_amd.renderer import ProgressiveView
    scene = scenes.simple_balls()
    view = ProgressiveView(scene, 96, 54, seed=5)
    acc = np.zeros(96 * 54, dtype=np.uint32)
    for k in range(1, 6):
        assert view.draw()
        torch.cuda.synchronize()
        assert view.pass_counter == k
        ps = view.renderer.accum.cpu().numpy().reshape(-1, 3)        # this pass alone
        acc, bgra = ob.progressive_present(ps, 1, k, acc)
        assert np.array_equal(view.acc.cpu().numpy().astype(np.uint32).ravel(), acc)
        assert np.array_equal(view.drawable.cpu().numpy().reshape(-1, 4), bgra)
    mean5 = view.running_mean()
    assert 0.2 < mean5.mean() < 1.0
    # a camera change restarts the mean (ContentView.swift:107-113), the same camera does not
    same = scene.camera.copy()
    view.set_camera(same)
    assert view.pass_counter == 5
    moved = scene.camera.copy()
    moved.relative_position = (30.0, 20.0, 6.0)
    view.set_camera(moved)
    assert view.pass_counter == 0
    view.draw()
    torch.cuda.synchronize()
    assert view.pass_counter == 1
    assert np.abs(view.running_mean() - mean5).mean() > 1e-3
    # the view pauses once 200 passes are in (ContentView.swift:187-190)
    view.pass_counter = 198
    assert view.draw() and not view.paused
    assert view.draw() and view.paused and view.pass_counter == 200
    assert not view.draw() and view.pass_counter == 200
    view.resize(64, 36)
    assert view.pass_counter == 0 and not view.paused and view.draw()


@pytest.mark.gpu
@pytest.mark.timeout(200)
def test_overlapped_passes_give_the_words_and_pixels_of_the_sequential_loop():
    """ProgressiveView(overlap=True) alternates passes between two streams; run in a child process under its own time
    limit (tests/progressive_child.py), so that trouble between streams would end there and not in this process."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        p = subprocess.run([sys.executable, os.path.join(here, "progressive_child.py")], capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired:
        pytest.skip("the child with two streams in flight did not finish in 150 s")
    assert p.returncode == 0, p.stderr[-2000:]
    assert "overlapped passes equal the sequential loop" in p.stdout
