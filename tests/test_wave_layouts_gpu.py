"""Round 4: the wide gradient kernels run eight waves per block (two per SIMD, 32 columns each) where round 3 ran four of 64, and small
minibatches run as 16-row half groups.  Same operands, same products; what differs is the order of a few f32 sums (bias gradients per lane
instead of per MFMA tile, eight head partial sums instead of four, another split of the row groups over the 256 blocks) -- so the two layouts must agree to f32 rounding, far inside the
tolerances the kernels are checked against torch with (tests/test_bf16_gpu.py, tests/test_ppo_gpu.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _grad(tmp_path, name, task, dtype, batch, **env):
    out = str(tmp_path / f"{name}.npy")
    e = dict(os.environ, **env)
    subprocess.run([sys.executable, os.path.join(HERE, "_grad_dump.py"), task, dtype, str(batch), out], check=True, env=e, timeout=300)
    return np.load(out)


@pytest.mark.parametrize("task,dtype,batch,switch", [("ball3d", "bf16", 16384, "TMA_BF_NW4"), ("crawler", "bf16", 16384, "TMA_BF_NW4"),
                                                      ("gridworld", "f32", 16384, "TMA_WIDE_NW4"), ("basic", "f32", 256, "TMA_WIDE_NW4"),
                                                      ("basic", "f32", 256, "TMA_NO_HALF_GROUPS")])
def test_eight_wave_and_half_group_layouts_agree_with_the_round_3_layouts(tmp_path, task, dtype, batch, switch):
    new = _grad(tmp_path, "new", task, dtype, batch)
    old = _grad(tmp_path, "old", task, dtype, batch, **{switch: "1"})
    assert np.isfinite(new).all() and float(np.abs(new).max()) > 0
    scale = float(np.abs(old).max())
    assert float(np.abs(new - old).max()) <= 2e-5 * scale, (float(np.abs(new - old).max()), scale)
