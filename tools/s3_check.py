#!/usr/bin/env python3
"""mfma_dtype = 'bf16x3' against the exact-f32 update on the same rollout: gradient difference and kernel time.  usage: s3_check.py [task batch]..."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:] or ["gridworld", "16384", "ball3d", "16384", "basic", "16384"]
for i in range(0, len(args), 2):
    task, batch = args[i], args[i + 1]
    g = {}
    for dt in ("f32", "bf16x3"):
        f = tempfile.mktemp(suffix=".npy")
        subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_grad_dump.py"), task, dt, batch, f], check=True, stderr=subprocess.DEVNULL)
        g[dt] = np.load(f)
        os.unlink(f)
    a, b = g["f32"], g["bf16x3"]
    n = min(len(a), len(b))
    d = np.abs(a[:n] - b[:n])
    print(f"{task} batch {batch}: max |g_f32 - g_split| = {d.max():.3e} = {d.max() / np.abs(a).max():.2e} of max |g| ({np.abs(a).max():.3e}); finite {np.isfinite(b).all()}; "
          f"worst index {int(d.argmax())} of {n}", flush=True)
