#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build container only: it reads the reference's
image files; nothing at test time reads /root/reference).

  images.npz   the reference's own image data: Thirdparty/DBoW2/demo/images/image{0..3}.png (640x480 gray) and
               demo/initImages/*.png (752x480 RGB) converted to gray with the integer formula
               Y = (R*4899 + G*9617 + B*1868 + 8192) >> 14   
  golden.npz   outputs of the CPU oracle (oracle/orbx_oracle.cpp) on those images: keypoints, descriptors and
               SearchForInitialization results for the `canonical` (1000,1.2,8,20,7) and `as_shipped` (2000,1.2,8,0,0)
               presets.  The reference itself cannot run here (no OpenCV), so these pin the ORACLE against regressions;
               parity with an OpenCV-linked build of the reference stays unpinned.
"""
import glob
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

REF = "/root/reference"
PRESETS = {"canonical": (1000, 1.2, 8, 20, 7), "as_shipped": (2000, 1.2, 8, 0, 0)}


def rgb2gray(a):
    a = a.astype(np.int64)
    return ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def main():
    imgs = {}
    for i in range(4):
        imgs["dbow%d" % i] = np.array(Image.open("%s/Thirdparty/DBoW2/demo/images/image%d.png" % (REF, i)))
    for i, f in enumerate(sorted(glob.glob(REF + "/demo/initImages/*.png"))):
        imgs["init%d" % i] = rgb2gray(np.array(Image.open(f)))
    for k, v in imgs.items():
        assert v.dtype == np.uint8 and v.ndim == 2, (k, v.shape, v.dtype)
    np.savez_compressed(os.path.join(HERE, "images.npz"), **imgs)

    gold = {}
    for pname, p in PRESETS.items():
        ex = O.Extractor(*p)
        res = {}
        for name, im in imgs.items():
            if pname == "as_shipped" and not name.startswith("init"):
                continue
            r, k, d = ex(im, cap=p[0] + 64)
            res[name] = (k, d)
            gold["%s/%s/ret" % (pname, name)] = np.int32(r)
            gold["%s/%s/kps" % (pname, name)] = k
            gold["%s/%s/desc" % (pname, name)] = d
            gold["%s/%s/ncand" % (pname, name)] = np.array([len(ex.level_candidates(l)) for l in range(p[2])], np.int32)
        pairs = [("init0", "init1")] + ([("dbow0", "dbow1"), ("dbow2", "dbow3")] if pname == "canonical" else [])
        for a, b in pairs:
            h, w = imgs[a].shape
            nm, m12, st = O.match_init(res[a][0], res[a][1], res[b][0], res[b][1], (0, w, 0, h), 100, 0.9, True)
            gold["%s/%s-%s/nmatches" % (pname, a, b)] = np.int32(nm)
            gold["%s/%s-%s/matches12" % (pname, a, b)] = m12
            gold["%s/%s-%s/stats" % (pname, a, b)] = st
            print(pname, a, b, "N", len(res[a][0]), len(res[b][0]), "nmatches", nm, "stats", st)
    np.savez_compressed(os.path.join(HERE, "golden.npz"), **gold)
    print("wrote", len(imgs), "images and", len(gold), "golden arrays")


if __name__ == "__main__":
    main()
