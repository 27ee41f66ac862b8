#!/usr/bin/env python3
"""Writes the reference's six images (tests/golden/images.npz: data, not code) as raw 8-bit files + a manifest for pin_opencv."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1] if len(sys.argv) > 1 else "pin_fixtures"
os.makedirs(out, exist_ok=True)
z = np.load(os.path.join(ROOT, "tests", "golden", "images.npz"))
with open(os.path.join(out, "manifest.txt"), "w") as m:
    for k in z.files:
        a = np.ascontiguousarray(z[k], np.uint8)
        a.tofile(os.path.join(out, k + ".raw"))
        m.write("%s %d %d\n" % (k, a.shape[1], a.shape[0]))
print("wrote %d images to %s" % (len(z.files), out))
