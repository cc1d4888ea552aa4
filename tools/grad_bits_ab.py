#!/usr/bin/env python3
"""Bit-level A/B of one minibatch gradient between two builds of the library (tests/_grad_dump.py in a subprocess per library):
`python tools/grad_bits_ab.py A.so B.so [task dtype batch]...` prints, per policy-parameter block, how many entries differ and by how much."""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a, b, rest = sys.argv[1], sys.argv[2], sys.argv[3:] or ["ball3d", "bf16", "16384"]
for i in range(0, len(rest), 3):
    task, dt, batch = rest[i:i + 3]
    outs = []
    for lib in (a, b):
        f = tempfile.mktemp(suffix=".npy")
        subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_grad_dump.py"), task, dt, batch, f], check=True,
                       env=dict(os.environ, TMA_LIB_PATH=os.path.abspath(lib)), stderr=subprocess.DEVNULL)
        outs.append(np.load(f))
        os.unlink(f)
    ga, gb = outs
    diff = ga != gb
    print(f"{task} {dt} batch {batch}: {int(diff.sum())} of {ga.size} entries differ, max |a - b| = {float(np.abs(ga - gb).max()):.3e}, max |a| = {float(np.abs(ga).max()):.3e}")
    if diff.any():
        idx = np.nonzero(diff)[0]
        # contiguous runs of differing indices (parameter blocks: W1, b1, W2, b2, W3, b3 per net)
        runs, s = [], idx[0]
        for p, q in zip(idx[:-1], idx[1:]):
            if q - p > 64:
                runs.append((s, p))
                s = q
        runs.append((s, idx[-1]))
        print("   differing index ranges:", [(int(x), int(y)) for x, y in runs][:12])
