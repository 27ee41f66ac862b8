"""CPU-only: pins the oracle (oracle/orbx_oracle.cpp) against every known-answer the reference's source holds for this
path (K1-K4, constants) and against the committed golden vectors."""
import hashlib
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

QUOTAS = {1000: [217, 181, 151, 126, 105, 87, 73, 60], 1250: [271, 226, 189, 157, 131, 109, 91, 76],
          2000: [434, 362, 302, 251, 209, 175, 145, 122], 4000: [869, 724, 603, 503, 419, 349, 291, 242],
          8000: [1737, 1448, 1207, 1005, 838, 698, 582, 485]}
UMAX = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


def test_quotas_and_umax(oracle):  # K1, K2 (Features/ORBextractor.cpp:529-548, 562-594)
    for nf, q in QUOTAS.items():
        t = oracle.Extractor(nf, 1.2, 8, 20, 7).tables()
        assert t["quota"].tolist() == q and int(t["quota"].sum()) == nf  # "Sum of features", cpp:549
        assert t["umax"].tolist() == UMAX
    t = oracle.Extractor(1000, 1.2, 8, 20, 7).tables()
    s = np.float32(1.0)
    for l in range(8):
        assert t["scale"][l] == s
        assert t["inv_scale"][l] == np.float32(1.0) / s
        s = np.float32(np.float64(s) * np.float64(np.float32(1.2)))


@pytest.mark.parametrize("rel", ["oracle/orbx_pattern_data.inc", "orb_slam_tracking_amd/csrc/orbx_pattern_data.inc"])
def test_pattern_table_sha(rel):  # K3 (cpp:233-490)
    txt = open(os.path.join(ROOT, rel)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    vals = [int(v) for v in re.findall(r"-?\d+", txt)]
    assert len(vals) == 1024 and min(vals) == -13 and max(vals) == 12
    assert vals[:4] == [8, -3, 9, 5] and vals[-4:] == [-1, -6, 0, -11]
    assert hashlib.sha256(struct.pack("<1024i", *vals)).hexdigest() == \
        "7e645581387b82784797e8adddb9b6f0c12611859fda09ca8a9bec96d767a05f"


def test_hamming_kats(oracle):  # K4 (Thirdparty/DBoW2/src/FORB.cpp:77-101)
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, 32, dtype=np.uint8)
    assert oracle.hamming(a, a) == 0
    assert oracle.hamming(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256
    b = a.copy()
    b[7] ^= 0x10
    assert oracle.hamming(a, b) == 1
    for _ in range(50):
        x, y = rng.integers(0, 256, (2, 32), dtype=np.uint8)
        assert oracle.hamming(x, y) == int(np.unpackbits(x ^ y).sum())


def test_fast_atan2(oracle):
    assert oracle.fast_atan2(0.0, 0.0) == 0.0
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = (float(v) for v in rng.integers(-1200000, 1200000, 2))
        a = oracle.fast_atan2(y, x)
        ref = np.degrees(np.arctan2(y, x)) % 360.0
        d = abs(a - ref)
        assert min(d, 360 - d) < 0.02  # cv::fastAtan2's documented accuracy is ~0.3 deg; the polynomial is far better
        assert 0.0 <= a <= 360.0


def test_resize_properties(oracle):
    const = np.full((48, 64), 137, np.uint8)
    assert (oracle.resize_linear(const, 53, 40) == 137).all()  # Q11 weights sum to 2048
    ramp = np.tile(np.arange(0, 240, 2, dtype=np.uint8), (30, 1))
    out = oracle.resize_linear(ramp, 100, 25)
    assert (np.diff(out.astype(int), axis=1) >= 0).all()
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (40, 60), dtype=np.uint8)
    assert np.array_equal(oracle.resize_linear(img, 60, 40), img)  # identity scale


def test_gaussian_properties(oracle):
    const = np.full((20, 30), 201, np.uint8)
    assert (oracle.gaussian7(const) == 201).all()  # taps [18,34,48,56,48,34,18] sum to 256
    imp = np.zeros((21, 21), np.uint8)
    imp[10, 10] = 255
    out = oracle.gaussian7(imp).astype(int)
    k = np.array([18, 34, 48, 56, 48, 34, 18])
    exp = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(out[7:14, 7:14], exp) and out.sum() == exp.sum()
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (17, 23), dtype=np.uint8)
    pad = np.pad(img, 3, mode="reflect").astype(np.int64)  # numpy 'reflect' == BORDER_REFLECT_101
    hz = sum(k[i] * pad[:, i:i + 23] for i in range(7))
    vt = sum(k[i] * hz[i:i + 17, :] for i in range(7))
    assert np.array_equal(oracle.gaussian7(img), ((vt + 32768) >> 16).astype(np.uint8))


def test_fast_definition(oracle):
    """cv::FAST semantics against a literal numpy evaluation on a small random image."""
    rng = np.random.default_rng(4)
    img = (rng.integers(0, 256, (24, 28)) // 32 * 32).astype(np.uint8)
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
            (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    h, w = img.shape
    s = np.zeros((h, w), int)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            d = [int(img[y, x]) - int(img[y + dy, x + dx]) for dx, dy in ring]
            best = -999
            for st in range(16):
                arc = [d[(st + j) % 16] for j in range(9)]
                best = max(best, min(arc), -max(arc))
            s[y, x] = best
            assert oracle.fast_strength(img, x, y) == best
    for t in (0, 7, 20, 40):
        score = np.where(s > t, s - 1, 0)
        exp = []
        for y in range(3, h - 3):
            for x in range(3, w - 3):
                if s[y, x] > t:
                    nb = score[y - 1:y + 2, x - 1:x + 2].copy()
                    nb[1, 1] = -1
                    if (score[y, x] > nb).all():
                        exp.append((x, y, score[y, x]))
        got = oracle.fast(img, t, True)
        assert [tuple(int(v) for v in r) for r in got] == exp
        nonms = oracle.fast(img, t, False)
        assert len(nonms) == int((s[3:h - 3, 3:w - 3] > t).sum())


def test_golden_vectors(oracle, images, golden):
    """Oracle reproduces the committed golden outputs bit for bit (regression pin; see tests/golden/make_fixtures.py)."""
    presets = {"canonical": (1000, 1.2, 8, 20, 7), "as_shipped": (2000, 1.2, 8, 0, 0)}
    for pname, p in presets.items():
        ex = oracle.Extractor(*p)
        res = {}
        for name, im in images.items():
            key = "%s/%s/kps" % (pname, name)
            if key not in golden:
                continue
            r, k, d = ex(im, cap=p[0] + 64)
            assert r == int(golden["%s/%s/ret" % (pname, name)])
            assert k.tobytes() == golden[key].tobytes()
            assert np.array_equal(d, golden["%s/%s/desc" % (pname, name)])
            res[name] = (k, d)
        for key in [k for k in golden if k.startswith(pname) and k.endswith("/matches12")]:
            a, b = key.split("/")[1].split("-")
            h, w = images[a].shape
            nm, m12, st = oracle.match_init(res[a][0], res[a][1], res[b][0], res[b][1], (0, w, 0, h), 100, 0.9, True)
            assert nm == int(golden[key.replace("matches12", "nmatches")])
            assert np.array_equal(m12, golden[key]) and np.array_equal(st, golden[key.replace("matches12", "stats")])


def test_behavioural_pins(golden):
    """The only behavioural pins the reference holds: `Sum of features == nfeatures` and
    `nmatches >= 100` on the two init images with the shipped settings (demo_initialization.cpp:110)."""
    assert int(golden["as_shipped/init0-init1/nmatches"]) >= 100
    assert len(golden["as_shipped/init0/kps"]) == 2000 and len(golden["as_shipped/init1/kps"]) == 2000
    assert golden["canonical/dbow0/desc"].shape == (1000, 32) and golden["canonical/dbow0/desc"].dtype == np.uint8
    k = golden["canonical/dbow0/kps"]
    assert (np.diff(k["octave"]) >= 0).all()  # level-major output order (cpp:1587-1650)
    assert (k["class_id"] == -1).all() and (k["x"] >= 19).all() and (k["y"] >= 19).all()


def test_matcher_quirks(oracle):
    """Q14/Q15/Q17/Q19 of SURVEY appendix B on hand-built inputs."""
    KP = oracle.KP
    k1 = np.zeros(3, KP)
    k2 = np.zeros(2, KP)
    k1["x"], k1["y"] = [100, 102, 104], [100, 100, 100]
    k2["x"], k2["y"] = [101, 300], [100, 300]
    k1["angle"] = [10, 10, 10]
    k2["angle"] = [10, 10]
    d2 = np.zeros((2, 32), np.uint8)
    d1 = np.zeros((3, 32), np.uint8)
    d1[0, 0] = 0b00000111  # dist 3 to train 0
    d1[1, 0] = 0b00000001  # dist 1 -> steals train 0
    d1[2, 0] = 0b00000011  # dist 2 -> train 0 hidden by vMatchedDistance (1 <= 2): bestDist stays INT_MAX
    nm, m12, st = oracle.match_init(k1, d1, k2, d2, (0, 640, 0, 480), 100, 0.9, True)
    assert m12.tolist() == [-1, 0, -1]
    assert st.tolist() == [1, 0, 0]  # Q19: query 2 counted as invalid-by-distance
    assert nm == 1                   # +1 (q0) -1 +1 (q1 steals); both share the kept histogram bin
    # second-best absent => ratio test passes with INT_MAX (Q17); octave>0 queries/trains ignored (Q13)
    k1["octave"] = [0, 1, 0]
    nm, m12, st = oracle.match_init(k1, d1, k2, d2, (0, 640, 0, 480), 100, 0.9, False)
    assert m12.tolist() == [-1, -1, 0] and nm == 1  # q2 (dist 2) steals from q0 (dist 3)


def _undistort_np(cam, x, y):
    """Second, independent restatement of cv::undistortPoints(R = I, P = K) in numpy f64: the same
    operation order without OpenCV's zero-coefficient terms (adding/multiplying exact zeros does not change a result)."""
    fx, fy, cx, cy, k1, k2, p1, p2 = [np.float64(np.float32(v)) for v in cam]
    ifx, ify = np.float64(1.0) / fx, np.float64(1.0) / fy
    x = (np.float64(np.float32(x)) - cx) * ifx
    y = (np.float64(np.float32(y)) - cy) * ify
    x0, y0 = x, y
    for _ in range(5):
        r2 = x * x + y * y
        icdist = np.float64(1.0) / (np.float64(1.0) + (k2 * r2 + k1) * r2)
        dx = np.float64(2.0) * p1 * x * y + p2 * (r2 + np.float64(2.0) * x * x)
        dy = p1 * (r2 + np.float64(2.0) * y * y) + np.float64(2.0) * p2 * x * y
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    return np.float32(fx * x + cx), np.float32(fy * y + cy)


def test_undistort_keypoints_and_bounds(oracle, golden):
    """Frame::UndistortKeyPoints / ComputeImageBounds (Frame.cpp:101-161) with the camera of the
    reference's Settings.yaml."""
    cam = oracle.SETTINGS_CAMERA
    k = golden["as_shipped/init0/kps"]
    u = oracle.undistort_keypoints(k, cam)
    for f in ("size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(u[f], k[f])  # Frame.cpp:156-159: only pt changes
    for i in range(0, len(k), 7):
        ex, ey = _undistort_np(cam, k["x"][i], k["y"][i])
        assert u["x"][i] == ex and u["y"][i] == ey
    # forward model check: distorting the result again lands near the measured pixel (5 fixed-point iterations leave
    # up to ~0.1 px at the image edge with this k1)
    fx, fy, cx, cy, k1, k2, _, _ = cam
    xn, yn = (u["x"].astype(np.float64) - cx) / fx, (u["y"].astype(np.float64) - cy) / fy
    r2 = xn * xn + yn * yn
    cd = 1 + k1 * r2 + k2 * r2 * r2
    assert np.abs(xn * cd * fx + cx - k["x"]).max() < 0.25 and np.abs(yn * cd * fy + cy - k["y"]).max() < 0.25
    # the principal point is a fixed point; no distortion = copy (Frame.cpp:137-140)
    pp = np.zeros(1, oracle.KP)
    pp["x"], pp["y"] = np.float32(cam[2]), np.float32(cam[3])
    o = oracle.undistort_keypoints(pp, cam)
    assert o["x"][0] == pp["x"][0] and o["y"][0] == pp["y"][0]
    nod = cam[:4] + (0.0, 0.1, 0.0, 0.0)
    assert oracle.undistort_keypoints(k, nod).tobytes() == k.tobytes()
    # image bounds: barrel distortion pushes the corners outwards; ints truncated from floats (Frame.cpp:123-126)
    b = oracle.image_bounds(cam, 640, 480)
    corners = [(0, 0), (640, 0), (0, 480), (640, 480)]
    un = [_undistort_np(cam, *c) for c in corners]
    assert b == (int(min(un[0][0], un[2][0])), int(max(un[1][0], un[3][0])), int(min(un[0][1], un[1][1])),
                 int(max(un[2][1], un[3][1])))
    assert b[0] < 0 and b[2] < 0 and b[1] > 640 and b[3] > 480
    assert oracle.image_bounds(nod, 640, 480) == (0, 640, 0, 480)
    # tangential terms exercised too (p1, p2 != 0)
    camt = cam[:6] + (0.0011, -0.0007)
    ut = oracle.undistort_keypoints(k[:50], camt)
    for i in range(50):
        ex, ey = _undistort_np(camt, k["x"][i], k["y"][i])
        assert ut["x"][i] == ex and ut["y"][i] == ey


def test_to_gray(oracle, images):
    """Converter::toGray (Utils/Converter.cpp:5-19) with the 14-bit cvtColor coefficients."""
    rng = np.random.default_rng(11)
    for (h, w) in ((480, 752), (33, 641), (3, 5), (1, 1)):
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        a = im.astype(np.int64)
        assert np.array_equal(oracle.to_gray(im, True), ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + 8192) >> 14))
        assert np.array_equal(oracle.to_gray(im, False), ((a[..., 2] * 4899 + a[..., 1] * 9617 + a[..., 0] * 1868 + 8192) >> 14))
        assert np.array_equal(oracle.to_gray(im[..., 0].copy()), im[..., 0])                    # one channel: copyTo
    assert oracle.to_gray(np.zeros((4, 4, 4), np.uint8)) is None                                # "Wrong image format"
    assert oracle.to_gray(np.full((2, 2, 3), 255, np.uint8), True).tolist() == [[255, 255], [255, 255]]  # coefficients sum to 2^14
    # the committed gray fixtures of the two init images are this conversion of the reference's RGB PNGs
    pngs = sorted(__import__("glob").glob("/root/reference/demo/initImages/*.png"))
    if len(pngs) == 2:
        try:
            from PIL import Image
        except ImportError:
            return
        for i, f in enumerate(pngs):
            assert np.array_equal(oracle.to_gray(np.array(Image.open(f).convert("RGB")), True), images["init%d" % i])


def _check_h_np(H21, H12, k1, k2, first, second, sigma):
    """Independent numpy-f32 restatement of CheckHomography (same operation order, one IEEE f32 operation per step)."""
    f = np.float32
    h, hi = H21.reshape(9).astype(f), H12.reshape(9).astype(f)
    th, inv = f(5.991), f(1.0 / np.float64(f(sigma) * f(sigma)))
    score, inl = f(0), []
    for a, b in zip(first, second):
        u1, v1, u2, v2 = f(k1["x"][a]), f(k1["y"][a]), f(k2["x"][b]), f(k2["y"][b])
        ok = True
        w = f(1.0 / np.float64(f(f(hi[6] * u2) + f(hi[7] * v2)) + hi[8]))
        uu, vv = f(f(f(f(hi[0] * u2) + f(hi[1] * v2)) + hi[2]) * w), f(f(f(f(hi[3] * u2) + f(hi[4] * v2)) + hi[5]) * w)
        chi = f(f(f(f(u1 - uu) * f(u1 - uu)) + f(f(v1 - vv) * f(v1 - vv))) * inv)
        if chi > th: ok = False
        else: score = f(score + f(th - chi))
        w = f(1.0 / np.float64(f(f(h[6] * u1) + f(h[7] * v1)) + h[8]))
        uu, vv = f(f(f(f(h[0] * u1) + f(h[1] * v1)) + h[2]) * w), f(f(f(f(h[3] * u1) + f(h[4] * v1)) + h[5]) * w)
        chi = f(f(f(f(u2 - uu) * f(u2 - uu)) + f(f(v2 - vv) * f(v2 - vv))) * inv)
        if chi > th: ok = False
        else: score = f(score + f(th - chi))
        inl.append(ok)
    return score, np.array(inl, bool)


def _check_f_np(F21, k1, k2, first, second, sigma):
    f = np.float32
    m = F21.reshape(9).astype(f)
    th, ths, inv = f(3.841), f(5.991), f(1.0 / np.float64(f(sigma) * f(sigma)))
    score, inl = f(0), []
    for a, b in zip(first, second):
        u1, v1, u2, v2 = f(k1["x"][a]), f(k1["y"][a]), f(k2["x"][b]), f(k2["y"][b])
        ok = True
        a2, b2, c2 = f(f(f(m[0] * u1) + f(m[1] * v1)) + m[2]), f(f(f(m[3] * u1) + f(m[4] * v1)) + m[5]), f(f(f(m[6] * u1) + f(m[7] * v1)) + m[8])
        num = f(f(f(a2 * u2) + f(b2 * v2)) + c2)
        chi = f(f(f(num * num) / f(f(a2 * a2) + f(b2 * b2))) * inv)
        if chi > th: ok = False
        else: score = f(score + f(ths - chi))
        a1, b1, c1 = f(f(f(m[0] * u2) + f(m[3] * v2)) + m[6]), f(f(f(m[1] * u2) + f(m[4] * v2)) + m[7]), f(f(f(m[2] * u2) + f(m[5] * v2)) + m[8])
        num = f(f(f(a1 * u1) + f(b1 * v1)) + c1)
        chi = f(f(f(num * num) / f(f(a1 * a1) + f(b1 * b1))) * inv)
        if chi > th: ok = False
        else: score = f(score + f(ths - chi))
        inl.append(ok)
    return score, np.array(inl, bool)


def test_initializer_scoring_loops(oracle):
    """CheckHomography / CheckFundamental (Initializer.cpp:268-438): the C restatement equals an
    independent numpy-f32 evaluation bit for bit, and behaves as a score should (true model wins, outliers rejected)."""
    k1, k2, m12, H21, H12, F21 = oracle.scoring_case(3, n=200, n_models=6)
    first = np.nonzero(m12 >= 0)[0]
    second = m12[first]
    with np.errstate(all="ignore"):
        for sigma in (1.0, 1.7):
            scores = []
            for i in range(len(H21)):
                sc, inl = oracle.check_homography(H21[i], H12[i], k1, k2, m12, sigma)
                esc, einl = _check_h_np(H21[i], H12[i], k1, k2, first, second, sigma)
                assert sc == esc and np.array_equal(inl, einl)
                scores.append(sc)
                sc, inl = oracle.check_fundamental(F21[i], k1, k2, m12, sigma)
                esc, einl = _check_f_np(F21[i], k1, k2, first, second, sigma)
                assert sc == esc and np.array_equal(inl, einl)
            assert int(np.argmax(scores)) == 0            # hypothesis 0 is the generating homography
    sc, inl = oracle.check_homography(H21[0], H12[0], k1, k2, m12, 1.0)
    assert 0.6 < inl.mean() < 0.95 and len(inl) == len(first)  # ~20 % of the pairs are gross outliers
    sc0, inl0 = oracle.check_homography(H21[0], H12[0], k1, k2, np.full(len(k1), -1, np.int32), 1.0)
    assert sc0 == 0 and len(inl0) == 0


def _check_rt_np(R, t, K, k1, k2, first, second, inliers, th2):
    """Independent float64 evaluation of CheckRT's geometry with numpy's LAPACK SVD: which inlier matches triangulate in front
    of camera 1 with a small reprojection error in both views (cosine < 0.99998), booked like the reference books them."""
    P1 = np.hstack([K, np.zeros((3, 1))])
    P2 = K @ np.hstack([R, t.reshape(3, 1)])
    O2 = -R.T @ t
    sel = np.nonzero(inliers)[0]
    good, pts, cos_all = {}, {}, []
    for i, m in enumerate(sel):
        u1, v1, u2, v2 = float(k1["x"][first[m]]), float(k1["y"][first[m]]), float(k2["x"][second[m]]), float(k2["y"][second[m]])
        A = np.stack([u1 * P1[2] - P1[0], v1 * P1[2] - P1[1], u2 * P2[2] - P2[0], v2 * P2[2] - P2[1]])
        X = np.linalg.svd(A)[2][3]
        x = X[:3] / X[3]
        oc2 = x - O2
        cosp = x @ oc2 / (np.linalg.norm(x) * np.linalg.norm(oc2))
        xc2 = R @ x + t
        if x[2] <= 0 and cosp < 0.99998:
            continue
        p1 = K @ x; p1 = p1[:2] / p1[2]
        p2 = K @ xc2; p2 = p2[:2] / p2[2]
        e1, e2 = (p1[0] - u1) ** 2 + (p1[1] - v1) ** 2, (p2[0] - u2) ** 2 + (p2[1] - v2) ** 2
        if e1 > th2 or e2 > th2:
            continue
        cos_all.append(cosp)
        pts[first[i]] = x
        good[first[i]] = cosp < 0.99998
    return good, pts, sorted(cos_all)


def test_check_rt(oracle):
    """Initializer::CheckRT (Initializer.cpp:569-713): the restatement (fixed one-sided Jacobi SVD for the DLT, OpenCV's
    float / double mixing) against a float64 LAPACK evaluation of the same geometry.  Tolerances: the restatement rounds the
    homogeneous point and most intermediate results to float32 like the reference (relative 2e-4 on points, 5e-3 degrees on
    the parallax); decisions may only differ for points within 1e-3 of a threshold (none in these cases)."""
    for seed in range(4):
        K, R, t, k1, k2, m12, cands = oracle.two_view_case(seed, n=300 + 50 * seed)
        first = np.nonzero(m12 >= 0)[0].astype(np.int32)
        second = m12[first]
        rng = np.random.default_rng(seed)
        inliers = (rng.random(len(first)) < 0.85).astype(np.uint8)
        for (Rc, tc) in cands:
            R32, t32, K32 = Rc.astype(np.float32), tc.astype(np.float32), K.astype(np.float32)
            n, good, p3d, par = oracle.check_rt(R32, t32, K32, k1, k2, m12, inliers, 4.0)
            g, pts, cosv = _check_rt_np(R32.astype(np.float64), t32.astype(np.float64), K32.astype(np.float64), k1, k2, first, second,
                                        inliers, 4.0)
            assert n == len(cosv)
            assert set(np.nonzero(good)[0].tolist()) == {k for k, v in g.items() if v}
            for k, x in pts.items():
                # (a point near infinity -- cosine >= 0.99998, not flagged good -- divides by a float32 w close to 0: looser)
                assert np.allclose(p3d[k], x, rtol=2e-4 if g[k] else 5e-2, atol=1e-5), (k, p3d[k], x)
            booked = set(pts)
            assert all(not p3d[k].any() for k in range(len(k1)) if k not in booked)  # zeros where nothing was booked
            if n:
                exp = np.degrees(np.arccos(cosv[min(50, n - 1)]))
                assert abs(float(par) - exp) < 5e-3 + 1e-4 * exp, (par, exp)
            else:
                assert par == 0
    # the true pose reconstructs most inliers, a wrong translation sign none (camera-1 depth test)
    K, R, t, k1, k2, m12, cands = oracle.two_view_case(9)
    first = np.nonzero(m12 >= 0)[0]
    ones = np.ones(len(first), np.uint8)
    n_true = oracle.check_rt(cands[0][0], cands[0][1], K, k1, k2, m12, ones)[0]
    n_neg = oracle.check_rt(cands[1][0], cands[1][1], K, k1, k2, m12, ones)[0]
    assert n_true > 0.6 * len(first) and n_neg == 0
    # degenerate inputs: no matches, no inliers
    n, good, p3d, par = oracle.check_rt(R, t, K, k1, k2, np.full(len(k1), -1, np.int32), np.zeros(0, np.uint8))
    assert n == 0 and not good.any() and par == 0
    n, good, p3d, par = oracle.check_rt(R, t, K, k1, k2, m12, np.zeros(len(first), np.uint8))
    assert n == 0 and not good.any() and not p3d.any() and par == 0


def _resize_np(src, dw, dh):
    """Independent numpy evaluation of cv::resize(INTER_LINEAR) on 8UC1: per axis
    f = (d + 0.5) * scale - 0.5 in float32, clamp, Q11 weights by round-half-even, horizontal pass in int32, vertical pass
    ((b0 * (T0 >> 4)) >> 16) + ((b1 * (T1 >> 4)) >> 16) + 2 >> 2 -- vectorised over the whole image, no oracle code."""
    sh, sw = src.shape

    def taps(dn, sn, fix):
        scale = 1.0 / (float(dn) / sn)
        f = ((np.arange(dn) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if fix:  # columns: the offset is pulled inside and the fraction dropped; rows are only clipped when they are read
            lo = s < 0
            s[lo], f[lo] = 0, 0
            hi = s >= sn - 1
            s[hi], f[hi] = sn - 1, 0
        c1 = np.rint(f * np.float32(2048)).astype(np.int64)        # np.rint = round half to even = cvRound
        c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        return np.clip(s, 0, sn - 1), np.clip(s + 1, 0, sn - 1), c0, c1
    sx0, sx1, a0, a1 = taps(dw, sw, True)
    sy0, sy1, b0, b1 = taps(dh, sh, False)
    S = src.astype(np.int64)
    T = S[:, sx0] * a0 + S[:, sx1] * a1                           # horizontal pass, no shift
    out = (((b0[:, None] * (T[sy0] >> 4)) >> 16) + ((b1[:, None] * (T[sy1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def test_resize_against_numpy(oracle):
    """A2 pinned by an independent evaluation (the property test above only checks constants, monotony and identity): random
    images, the pyramid's own chain of sizes (640x480 -> 533x400 -> ... and 752x480 -> 627x400), odd sizes, up-scaling."""
    rng = np.random.default_rng(12)
    for (sw, sh, dw, dh) in ((640, 480, 533, 400), (533, 400, 444, 333), (214, 161, 179, 134), (752, 480, 627, 400), (61, 47, 53, 40),
                             (33, 29, 32, 28), (40, 30, 57, 41), (7, 5, 3, 2), (100, 100, 100, 100)):
        img = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        assert np.array_equal(oracle.resize_linear(img, dw, dh), _resize_np(img, dw, dh)), (sw, sh, dw, dh)
    ramp = (np.add.outer(np.arange(120), np.arange(160)) % 256).astype(np.uint8)
    assert np.array_equal(oracle.resize_linear(ramp, 133, 100), _resize_np(ramp, 133, 100))


def _fast_atan2_np(y, x):
    """Independent float32 evaluation of cv::fastAtan2: every operation rounded to float32 by numpy."""
    f = np.float32
    p1, p3 = f(0.9997878412794807) * f(180 / np.pi), f(-0.3258083974640975) * f(180 / np.pi)
    p5, p7 = f(0.1555786518463281) * f(180 / np.pi), f(-0.04432655554792128) * f(180 / np.pi)
    eps = f(2.2204460492503131e-16)
    y, x = f(y), f(x)
    ax, ay = np.abs(x), np.abs(y)
    if ax >= ay:
        c = ay / (ax + eps)
        c2 = c * c
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    else:
        c = ax / (ay + eps)
        c2 = c * c
        a = f(90.0) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    if x < 0:
        a = f(180.0) - a
    if y < 0:
        a = f(360.0) - a
    return f(a)


def test_fast_atan2_against_numpy_f32(oracle):
    """A5 bit for bit against a numpy float32 evaluation of the same polynomial (the property test above only bounds the error
    against atan2): integer moments of the size IC_Angle produces, axes, diagonals, tiny and huge values."""
    rng = np.random.default_rng(13)
    cases = [(0, 0), (0, 5), (5, 0), (0, -5), (-5, 0), (7, 7), (-7, 7), (7, -7), (-7, -7), (1, 1200000), (1200000, 1), (1e-30, 1e-30), (3e38, 1e38)]
    cases += [tuple(float(v) for v in rng.integers(-1500000, 1500000, 2)) for _ in range(5000)]
    with np.errstate(all="ignore"):
        for (y, x) in cases:
            a, b = np.float32(oracle.fast_atan2(y, x)), _fast_atan2_np(y, x)
            assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b)), (y, x, a, b)


def test_opencv_variants(oracle):
    """The two OpenCV-release dependent constants, selectable (orbo_set_opencv_variant / orbx_set_opencv_variant): Gaussian Q8
    taps with every tap rounded (sum 257, saturating) and the 15-bit BGR2GRAY coefficients, each against numpy; the defaults
    are restored and still give the committed fixtures (test_golden_vectors)."""
    rng = np.random.default_rng(14)
    try:
        oracle.set_opencv_variant(1, 1)
        k = np.array([18, 34, 49, 55, 49, 34, 18])
        img = rng.integers(0, 256, (19, 27), dtype=np.uint8)
        img[:6, :9] = 255  # saturation: 255 * 257 * 257 + 32768 >= 2^24
        pad = np.pad(img, 3, mode="reflect").astype(np.int64)
        hz = sum(k[i] * pad[:, i:i + 27] for i in range(7))
        vt = sum(k[i] * hz[i:i + 19, :] for i in range(7))
        exp = np.minimum((vt + 32768) >> 16, 255).astype(np.uint8)
        assert np.array_equal(oracle.gaussian7(img), exp) and exp.max() == 255 and ((vt + 32768) >> 16).max() > 255
        im = rng.integers(0, 256, (31, 45, 3), dtype=np.uint8)
        a = im.astype(np.int64)
        assert np.array_equal(oracle.to_gray(im, True), (a[..., 0] * 9798 + a[..., 1] * 19235 + a[..., 2] * 3735 + 16384) >> 15)
        assert np.array_equal(oracle.to_gray(im, False), (a[..., 2] * 9798 + a[..., 1] * 19235 + a[..., 0] * 3735 + 16384) >> 15)
        assert oracle.to_gray(np.full((2, 2, 3), 255, np.uint8), True).tolist() == [[255, 255], [255, 255]]  # 9798 + 19235 + 3735 = 2^15
        # the variants change descriptors (blur) but not keypoints
        from orb_slam_tracking_amd import synth
        fr = synth.synth(320, 240, 21)
        oe = oracle.Extractor(400, 1.2, 5, 20, 7)
        _, k1, d1 = oe(fr)
        oracle.set_opencv_variant(0, 0)
        _, k0, d0 = oe(fr)
        assert k0.tobytes() == k1.tobytes() and not np.array_equal(d0, d1)
    finally:
        oracle.set_opencv_variant(0, 0)


# one of the 45 (scale factor, nlevels <= 16, nfeatures in {500 .. 8000}) with a scale factor in [1.01, 2] for which powf and
# (float)pow(double) give different per-level quotas (cpp:536): 1.01168489, nine levels, 500 features
EXOTIC_SCALE = float(np.uint32(0x3F817EE4).view(np.float32))


def test_libm_variants(oracle):
    """The libm reading of the reference's unqualified cos / sin / pow on floats (cpp:174, cpp:536; orbo_set_libm_variant /
    orbx_set_libm_variant).  ORBX_LIBM_FLOAT = glibc >= 2.28's sinf / cosf, restated in the oracle: EVERY f32 angle of [0, 360]
    (1,135,869,953 of them) against this host's own cosf / sinf -- 0 differing, in the FMA form and in the form without; the two
    readings themselves differ (483,807 cosines / 1,001,902 sines with glibc 2.35), and at 96 angles a rotated sample point
    lands on another pixel: one of them is pinned here through the descriptor."""
    hi = int(np.float32(360.0).view(np.uint32))
    out = oracle.libm_sweep(0, hi, max(1, min(os.cpu_count() or 1, 16)))
    assert out[:4] == [0, 0, 0, 0], out
    assert out[4] > 0 and out[5] > 0, out  # (the readings are different functions)
    rng = np.random.default_rng(5)
    blurred = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    moved = float(np.uint32(0x40E8408C).view(np.float32))   # 7.25787926 degrees: six of the 512 points move
    same = 7.25
    try:
        d = {}
        for v in (0, 1):
            oracle.set_libm_variant(v)
            d[v] = (oracle.descriptor(blurred, 32.0, 32.0, moved), oracle.descriptor(blurred, 32.0, 32.0, same))
            c, s = oracle.sincos_deg_batch(np.array([moved], np.float32))
            d[v] += (c[0], s[0])
        assert not np.array_equal(d[0][0], d[1][0]) and np.array_equal(d[0][1], d[1][1])
        assert (d[0][2], d[0][3]) != (d[1][2], d[1][3])
        # the constructor's pow: equal quotas for the usual parameters, different ones for an exotic scale factor
        q = {}
        for v in (0, 1):
            oracle.set_libm_variant(v)
            q[v] = (oracle.Extractor(1000, 1.2, 8, 20, 7).tables()["quota"], oracle.Extractor(500, EXOTIC_SCALE, 9, 20, 7).tables()["quota"])
        assert np.array_equal(q[0][0], q[1][0]) and q[0][0].sum() == 1000
        assert not np.array_equal(q[0][1], q[1][1]) and q[0][1].sum() == q[1][1].sum() == 500
    finally:
        oracle.set_libm_variant(oracle.LIBM_DEFAULT)
