"""Frames per second of the interactive loop (ProgressiveView: one sample per pixel, depth 10, a fresh seed and a present
step per frame, ContentView.swift:182-224) with and without overlapping successive passes on two streams.
usage (GPU box): python tools/interactive_fps.py [workload]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.renderer import ProgressiveView  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_random_spheres_1080p"
scene, w, h, _, _, _ = scenes.make_workload(name)
for overlap in (False, True, False, True):
    view = ProgressiveView(scene, w, h, seed=1, overlap=overlap)
    for _ in range(10):
        view.draw()
    torch.cuda.synchronize()
    view.set_needs_redraw()
    t0 = time.perf_counter()
    frames = 0
    while view.draw():
        frames += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s %dx%d, overlap %s: %d frames in %.1f ms = %.0f frames/s (%.3f ms per frame)" % (name, w, h, overlap, frames, dt * 1e3, frames / dt, dt / frames * 1e3),
          flush=True)
