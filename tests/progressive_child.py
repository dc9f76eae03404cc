"""Child process of tests/test_progressive.py: the interactive loop with successive passes overlapped on two streams
must produce, pass for pass, the 10-bit words and the bgra8 pixels of the loop that runs one pass at a time."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.renderer import ProgressiveView  # noqa: E402

scene = scenes.simple_balls()
plain = ProgressiveView(scene, 160, 90, seed=9)
both = ProgressiveView(scene, 160, 90, seed=9, overlap=True)
for k in range(1, 13):
    assert plain.draw() and both.draw()
    if k in (1, 2, 5, 12):
        torch.cuda.synchronize()
        assert np.array_equal(plain.acc.cpu().numpy(), both.acc.cpu().numpy()), "running mean differs at pass %d" % k
        assert np.array_equal(plain.drawable.cpu().numpy(), both.drawable.cpu().numpy()), "drawable differs at pass %d" % k
both.wait()
torch.cuda.current_stream().synchronize()
assert both.pass_counter == 12 and int(both.acc.abs().sum().item()) > 0
print("overlapped passes equal the sequential loop")
