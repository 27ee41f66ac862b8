"""GPU parity tests (run on the MI355X box with -m gpu): the HIP path, called through the C ABI, against the CPU oracle
and the committed golden vectors.  Bit-exact for keypoints (x, y, size, angle, response, octave), descriptor bytes,
matches12 and nmatches.  (Angles are f32 values compared bitwise: tolerance 0.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CANON = (1000, 1.2, 8, 20, 7)
SHIPPED = (2000, 1.2, 8, 0, 0)


@pytest.fixture(autouse=True, params=["fast_default", "fast_wave_per_cell"])
def fast_kernel_choice(request, orbx):
    """Launches of up to 5000 FAST cells (eight 640x480 frames) take k_fast, a workgroup per cell; the diagnostic knob
    fast_wg_max_cells = 0 (orbx_debug_set) sends them through k_fast_wave, one wave per cell, like the large batches.  Every test of
    this module runs both ways."""
    orbx.debug_set("fast_wg_max_cells", 0 if request.param == "fast_wave_per_cell" else None)
    yield
    orbx.debug_set("fast_wg_max_cells", None)


@pytest.fixture(scope="module")
def ext640(orbx):
    e = orbx.ORBextractor(*CANON, max_width=752, max_height=480, max_batch=8)
    yield e
    e.close()


def _same(kg, dg, ko, do):
    assert len(kg) == len(ko), (len(kg), len(ko))
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        bad = np.nonzero(kg[f] != ko[f])[0]
        assert len(bad) == 0, (f, bad[:5], kg[f][bad[:5]], ko[f][bad[:5]])
    assert kg.tobytes() == ko.tobytes()
    bad = np.nonzero((dg != do).any(1))[0]
    assert len(bad) == 0, ("descriptors", bad[:5])


def test_golden_fixtures_canonical(orbx, ext640, images, golden):
    for name in ("dbow0", "dbow1", "dbow2", "dbow3", "init0", "init1"):
        r, k, d = ext640(images[name])
        assert r == int(golden["canonical/%s/ret" % name])
        _same(k, d, golden["canonical/%s/kps" % name], golden["canonical/%s/desc" % name])


def test_golden_fixtures_as_shipped(orbx, images, golden):
    """The preset the shipped Settings.yaml really yields (thresholds 0/0, 2000 features)."""
    e = orbx.ORBextractor(*SHIPPED, max_width=752, max_height=480, max_batch=2)
    frames = []
    for name in ("init0", "init1"):
        r, k, d = e(images[name])
        _same(k, d, golden["as_shipped/%s/kps" % name], golden["as_shipped/%s/desc" % name])
        frames.append(orbx.Frame.from_arrays(k, d, (0, 752, 0, 480)))
    m = orbx.ORBmatcher(0.9, True, extractor=e)
    nm, m12 = m.SearchForInitialization(frames[0], frames[1], 100)
    assert nm == int(golden["as_shipped/init0-init1/nmatches"]) and nm >= 100  # demo_initialization.cpp:110
    assert np.array_equal(m12, golden["as_shipped/init0-init1/matches12"])
    assert list(m.last_stats) == golden["as_shipped/init0-init1/stats"].tolist()
    e.close()


def test_golden_matches_canonical(orbx, ext640, images, golden):
    for a, b in (("init0", "init1"), ("dbow0", "dbow1"), ("dbow2", "dbow3")):
        fa = orbx.Frame.from_arrays(golden["canonical/%s/kps" % a], golden["canonical/%s/desc" % a], (0, images[a].shape[1], 0, 480))
        fb = orbx.Frame.from_arrays(golden["canonical/%s/kps" % b], golden["canonical/%s/desc" % b], (0, images[b].shape[1], 0, 480))
        m = orbx.ORBmatcher(0.9, True, extractor=ext640)
        nm, m12 = m.SearchForInitialization(fa, fb, 100)
        assert nm == int(golden["canonical/%s-%s/nmatches" % (a, b)])
        assert np.array_equal(m12, golden["canonical/%s-%s/matches12" % (a, b)])
        assert list(m.last_stats) == golden["canonical/%s-%s/stats" % (a, b)].tolist()


def test_getters_match_oracle(orbx, ext640, oracle):
    t = oracle.Extractor(*CANON).tables()
    assert ext640.GetLevels() == 8 and ext640.GetScaleFactor() == np.float32(1.2)
    assert np.array_equal(ext640.GetScaleFactors(), t["scale"]) and np.array_equal(ext640.GetInverseScaleFactors(), t["inv_scale"])
    assert np.array_equal(ext640.GetScaleSigmaSquares(), t["sigma2"]) and np.array_equal(ext640.GetInverseScaleSigmaSquares(), t["inv_sigma2"])
    assert np.array_equal(ext640.GetNumFeaturesPerLevel(), t["quota"]) and np.array_equal(ext640.umax(), t["umax"])


def test_pyramid_and_candidates_per_level(orbx, ext640, oracle, images):
    """K1 (resize chain) and K2 (per-cell FAST + NMS + fallback, in the reference's candidate order) stage by stage."""
    oe = oracle.Extractor(*CANON)
    for name in ("dbow1", "init0"):
        im = images[name]
        ext640(im)
        oe(im)
        for l in range(8):
            assert ext640.level_size(l) == oe.level_size(l)
            assert np.array_equal(ext640.image_pyramid(l), oe.level_image(l)), (name, l)
            assert np.array_equal(ext640.debug_candidates(0, l), oe.level_candidates(l)), (name, l)
        ring = ext640.image_pyramid(2, border=19)
        assert np.array_equal(ring, np.pad(oe.level_image(2), 19, mode="reflect"))  # mvImagePyramid's REFLECT_101 ring


def test_synthetic_frames_and_batch(orbx, ext640, oracle):
    from orb_slam_tracking_amd import synth
    frames = synth.synth_frames(8, 640, 480, 1000)
    oe = oracle.Extractor(*CANON)
    single = [ext640(f) for f in frames]
    batch = ext640.extract_batch(frames)
    for f in range(8):
        r, k, d = oe(frames[f])
        assert single[f][0] == r == batch[f][0]
        _same(single[f][1], single[f][2], k, d)
        _same(batch[f][1], batch[f][2], k, d)
    # matching consecutive pairs
    for p in range(4):
        fa = orbx.Frame.from_arrays(batch[2 * p][1], batch[2 * p][2], (0, 640, 0, 480))
        fb = orbx.Frame.from_arrays(batch[2 * p + 1][1], batch[2 * p + 1][2], (0, 640, 0, 480))
        nm, m12 = orbx.ORBmatcher(0.9, True, extractor=ext640).SearchForInitialization(fa, fb, 100)
        onm, om12, _ = oracle.match_init(fa.mvKeysUn, fa.mDescriptors, fb.mvKeysUn, fb.mDescriptors, (0, 640, 0, 480), 100, 0.9, True)
        assert nm == onm and np.array_equal(m12, om12)
        assert nm > 20  # the two frames show the same scene


def test_lapping_area_order(orbx, ext640, oracle, images):
    """Stereo keypoints (x in [lap0, lap1]) are written back to front (cpp:1637-1646)."""
    oe = oracle.Extractor(*CANON)
    r, k, d = ext640(images["dbow0"], None, (100, 300))
    ro, ko, do = oe(images["dbow0"], lap=(100, 300))
    assert r == ro and 0 < r < len(k)
    _same(k, d, ko, do)


def test_edge_cases(orbx, ext640):
    assert ext640(np.zeros((0, 0), np.uint8))[0] == -1  # cpp:1536
    flat = np.full((480, 640), 128, np.uint8)
    r, k, d = ext640(flat)
    assert r == 0 and len(k) == 0 and d.shape == (0, 32)  # cpp:1567-1571
    with pytest.raises(orbx.OrbxError) as e:
        ext640(np.zeros((120, 160), np.uint8))  # level 7 would be 45x33: no FAST cell fits (UB upstream)
    assert e.value.code == orbx.E_TOOSMALL
    r, k, d = ext640(np.full((600, 800), 7, np.uint8))  # larger than the context was created for: it grows (cpp:1531-1545)
    assert r == 0 and len(k) == 0
    with pytest.raises(orbx.OrbxError):
        orbx.ORBextractor(1000, 1.0, 8, 20, 7)  # exit(1) upstream (cpp:502-505)


def test_frame_size_limit(orbx, oracle):
    """ORBX_MAX_FRAME_DIM (documented deviation, include/orbx.h): a 4096-pixel-wide frame is taken and equals the oracle (nine
    quadtree roots, 12-bit candidate coordinates used to the last column); 4100 pixels in either direction return ORBX_E_BADARG with a
    message that names the limit -- not a crash, not a silent truncation."""
    from orb_slam_tracking_amd import synth
    params = (600, 1.2, 4, 20, 7)
    e = orbx.ORBextractor(*params, max_width=640, max_height=480, max_batch=1)
    oe = oracle.Extractor(*params)
    f = synth.synth(4096, 300, 41)
    (r, k, d), = e.extract_batch(f[None])
    ro, ko, do = oe(f)
    assert r == ro and len(k) > 300
    _same(k, d, ko, do)
    for (w, h) in ((4100, 300), (300, 4100)):
        with pytest.raises(orbx.OrbxError) as ex:
            e.extract_batch(synth.synth(w, h, 42)[None])
        assert ex.value.code == orbx.E_BADARG and "4096" in str(ex.value)
    (r, k, d), = e.extract_batch(f[None])  # the context is still usable
    _same(k, d, ko, do)
    e.close()


def test_odd_sizes_and_fallback_cells(orbx, oracle):
    """Ragged geometry: odd widths, nIni == 2, dark low-contrast cells that need the minThFAST retry."""
    from orb_slam_tracking_amd import synth
    for (w, h, seed) in ((701, 397, 3), (333, 333, 4), (1000, 420, 5)):
        e = orbx.ORBextractor(500, 1.2, 6, 25, 5, max_width=w, max_height=h, max_batch=1)
        oe = oracle.Extractor(500, 1.2, 6, 25, 5)
        img = synth.synth(w, h, seed)
        img[:, : w // 2] = (img[:, : w // 2].astype(np.int32) // 6 + 60).astype(np.uint8)  # low contrast half
        r, k, d = e(img)
        ro, ko, do = oe(img)
        assert r == ro
        _same(k, d, ko, do)
        for l in range(6):
            assert np.array_equal(e.debug_candidates(0, l), oe.level_candidates(l))
        e.close()


def test_fast_retry_threshold_boundaries(orbx, oracle):
    """k_fast_wave sweeps a cell again at minThFAST only if a pixel can pass the quick reject there: patches whose contrast sits
    exactly at, one below and one above each threshold (a bright / dark 3 x 3 block differs from the ground by d: d > th passes),
    noise of every amplitude around minThFAST, flat ground, and all of it beside strong corners -- candidates per level and the
    final result against the oracle, for two threshold pairs and through both FAST kernels."""
    rng = np.random.default_rng(77)
    w, h = 640, 480
    for (ini, mn) in ((20, 7), (12, 3), (9, 8)):
        img = np.full((h, w), 100, np.uint8)
        tile = 40
        amps = [0, mn - 1, mn, mn + 1, (ini + mn) // 2, ini - 1, ini, ini + 1, 60]
        for ty in range(h // tile):
            for tx in range(w // tile):
                y0, x0 = ty * tile, tx * tile
                kind = (ty * (w // tile) + tx) % 27
                a = amps[kind % 9]
                if kind < 9:  # isolated blocks of contrast +-a
                    for _ in range(3):
                        y, x = y0 + int(rng.integers(4, tile - 8)), x0 + int(rng.integers(4, tile - 8))
                        img[y:y + 3, x:x + 3] = 100 + (a if rng.integers(0, 2) else -a)
                elif kind < 18:  # uniform noise of amplitude a / 2 (largest difference a)
                    lo = a // 2
                    img[y0:y0 + tile, x0:x0 + tile] = 100 - lo + rng.integers(0, a + 1, (tile, tile))
                else:  # a single pixel of contrast a (a ring of 16 darker pixels around it)
                    img[y0 + tile // 2, x0 + tile // 2] = 100 + a
        e = orbx.ORBextractor(600, 1.2, 5, ini, mn, max_width=w, max_height=h, max_batch=1)
        oe = oracle.Extractor(600, 1.2, 5, ini, mn)
        r, k, d = e(img)
        ro, ko, do = oe(img)
        assert r == ro
        _same(k, d, ko, do)
        for l in range(5):
            assert np.array_equal(e.debug_candidates(0, l), oe.level_candidates(l)), (ini, mn, l)
        e.close()


def test_fast_retry_cells_on_the_cell_grid(orbx, oracle):
    """The retry rule cell by cell (cpp:1109-1123): on the FAST cell grid of level 0 (36 x 38 pixel cells from
    (16, 16)) a chessboard of cells that are EMPTY at iniThFAST and NON-EMPTY at minThFAST (one 3 x 3 block of contrast between the
    thresholds: swept twice, the second sweep lists it), cells empty at BOTH (flat: k_fast_wave's flat-cell rule skips their second
    sweep; noise whose largest difference stays below minThFAST: the same; noise that passes the quick reject at minThFAST but
    holds no arc of 9: swept twice for nothing) and cells with a strong corner (one sweep) -- every kind beside every other,
    peaks also right at a cell's first and last detected pixel.  Candidates per level and the final result against the oracle,
    as one frame (k_fast, or k_fast_wave under the module's knob) and as a batch of 12 copies with different noise (k_fast_wave)."""
    import torch
    rng = np.random.default_rng(2026)
    w, h = 640, 480
    wc, hc = 36, 38  # ceil((640 - 32) / 17), ceil((480 - 32) / 12)
    for (ini, mn) in ((20, 7), (15, 5)):
        frames = []
        for rep in range(12):
            img = np.full((h, w), 120, np.uint8)
            for ci in range(12):
                for cj in range(17):
                    y0, x0 = 16 + ci * hc, 16 + cj * wc
                    kind = (ci * 5 + cj * 3 + rep) % 6
                    # where the block sits inside the cell: anywhere, or in a corner of the cell's own detection area
                    py, px = (3, 3) if (ci + cj) % 5 == 0 else (hc - 1, wc - 1) if (ci + cj) % 5 == 1 else (int(rng.integers(6, hc - 8)), int(rng.integers(6, wc - 8)))
                    if kind == 0:      # flat: empty at both
                        pass
                    elif kind == 1:    # a peak between the thresholds: empty at iniTh, one corner at minTh (a lone pixel: its ring is the
                        a = int(rng.integers(mn + 1, ini + 1))  # ground, its strength the contrast; a flat 3 x 3 block ties with itself in the NMS)
                        img[y0 + py + 1, x0 + px + 1] = 120 + (a if rng.integers(0, 2) else -a)
                    elif kind == 2:    # a strong peak: listed in the first sweep
                        img[y0 + py + 1, x0 + px + 1] = 120 + (3 * ini if rng.integers(0, 2) else -3 * ini)
                    elif kind == 3:    # noise below minTh: no pixel passes the quick reject there
                        img[y0 + 3:y0 + hc, x0 + 3:x0 + wc] = 120 + rng.integers(0, mn, (hc - 3, wc - 3))
                    elif kind == 4:    # sparse single pixels of contrast just above minTh: they pass the quick reject of their neighbours' rings
                        for _ in range(6):
                            img[y0 + int(rng.integers(4, hc - 4)), x0 + int(rng.integers(4, wc - 4))] = 120 + mn + 2
                    else:              # a peak exactly AT minTh (strength == minTh: no corner; the cell is swept twice for nothing)
                        img[y0 + py + 1, x0 + px + 1] = 120 + mn
            frames.append(img)
        frames = np.stack(frames)
        oe = oracle.Extractor(700, 1.2, 4, ini, mn)
        e1 = orbx.ORBextractor(700, 1.2, 4, ini, mn, max_width=w, max_height=h, max_batch=1)
        r, k, d = e1(frames[0])
        ro, ko, do = oe(frames[0])
        assert r == ro
        _same(k, d, ko, do)
        cands = [oe.level_candidates(l) for l in range(4)]
        for l in range(4):
            assert np.array_equal(e1.debug_candidates(0, l), cands[l]), (ini, mn, l)
        # the chessboard really holds cells of every kind at level 0: some corners only below iniTh, none in the flat cells
        c0 = cands[0]
        assert len(c0) > 30 and (c0[:, 2] < ini).any() and (c0[:, 2] >= ini).any(), (len(c0), c0[:, 2].min(), c0[:, 2].max())
        e1.close()
        B, cap = len(frames), 700
        eb = orbx.ORBextractor(700, 1.2, 4, ini, mn, max_width=w, max_height=h, max_batch=B)
        d_img = torch.from_numpy(frames).cuda()
        d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
        d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
        d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
        eb.extract_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, cap)
        assert eb.debug_last_launch()["fast_wave"] == 1  # (12 x 381 cells: beyond k_fast's launch size)
        n = d_n.cpu().numpy()
        kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
        dd = d_d.cpu().numpy().reshape(B, cap, 32)
        for f in range(B):
            _, ko, do = oe(frames[f], cap=cap)
            assert n[f] == len(ko), (ini, mn, f, n[f], len(ko))
            _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
            for l in (0, 3):
                assert np.array_equal(eb.debug_candidates(f, l), oe.level_candidates(l)), (ini, mn, f, l)
        eb.close()


def test_other_pyramid_parameters(orbx, oracle, images):
    for p in ((1250, 1.2, 8, 20, 7), (300, 1.5, 4, 30, 10), (700, 1.1, 10, 12, 12), (200, 1.2, 1, 20, 7)):
        e = orbx.ORBextractor(*p, max_width=640, max_height=480, max_batch=1)
        oe = oracle.Extractor(*p)
        r, k, d = e(images["dbow3"])
        ro, ko, do = oe(images["dbow3"], cap=p[0] + 64)
        assert r == ro
        _same(k, d, ko, do)
        e.close()


def test_brute_force_match_configs(orbx, ext640, oracle):
    """BASELINE configs 3/5 style: window covers the frame, so every octave-0 pair is a candidate."""
    from orb_slam_tracking_amd import synth
    for n, seed, ratio, ori in ((600, 5, 0.9, True), (2000, 6, 0.9, True), (900, 7, 0.6, False)):
        kA, dA, kB, dB = synth.synth_desc(n, seed)
        fa = orbx.Frame.from_arrays(kA, dA, (0, 3840, 0, 2160))
        fb = orbx.Frame.from_arrays(kB, dB, (0, 3840, 0, 2160))
        m = orbx.ORBmatcher(ratio, ori, extractor=ext640)
        nm, m12 = m.SearchForInitialization(fa, fb, 4096)
        onm, om12, ost = oracle.match_init(kA, dA, kB, dB, (0, 3840, 0, 2160), 4096, ratio, ori)
        assert nm == onm and np.array_equal(m12, om12) and list(m.last_stats) == ost.tolist()
        assert (m12 >= 0).sum() > n // 4


def test_brute_force_on_the_matrix_cores(orbx, oracle):
    """k_match_bf_mfma takes the blocks of 256 queries whose windows cover every train, in launches with at least 256 such blocks
    (fewer stay on k_match_wide_lists): 64 pairs of two to five blocks in one call, every pair different -- set sizes that are no multiple of 256, 64
    or 32, descriptors from few prototypes (many candidates per query, lists that overflow and send the pair to the reference's
    loop), other octaves mixed in -- against the oracle pair by pair, and the same call on the vector form (knob match_no_mfma).
    Then a window that only the queries in the middle of the frame pass, with the queries sorted by x: the blocks in the middle
    go to the matrix cores, the ones at the sides to k_match_wide_lists, in one pair."""
    import torch
    rng = np.random.default_rng(123)
    dev = torch.device("cuda", 0)
    KP = orbx.KEYPOINT_DTYPE
    P, cap = 64, 1100
    W, H = 3840, 2160
    ext = orbx.ORBextractor(cap, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1)

    def run(sets, window, ratio, ori, reps=1, min_mfma_blocks=1):
        k_all = np.zeros((2 * P, cap), KP)
        d_all = np.zeros((2 * P, cap, 32), np.uint8)
        n_all = np.zeros(2 * P, np.int32)
        for i, (k1, d1, k2, d2) in enumerate(sets):
            for f, (k, d) in ((2 * i, (k1, d1)), (2 * i + 1, (k2, d2))):
                k_all[f, :len(k)] = k
                d_all[f, :len(k)] = d
                n_all[f] = len(k)
        d_k = torch.from_numpy(k_all.view(np.uint8).reshape(-1).copy()).to(dev)
        d_d = torch.from_numpy(d_all.reshape(-1)).to(dev)
        d_n = torch.from_numpy(n_all).to(dev)
        first = np.tile(np.arange(0, 2 * P, 2, dtype=np.int32), reps)  # (every pair `reps` times in the launch: more blocks)
        NP = len(first)
        out = []
        for knob in (None, 1):
            orbx.debug_set("match_no_mfma", knob)
            d_m = torch.full((NP * cap,), -7, dtype=torch.int32, device=dev)
            d_nm = torch.zeros(NP, dtype=torch.int32, device=dev)
            d_st = torch.zeros(3 * NP, dtype=torch.int32, device=dev)
            before = ext.debug_match_counters()["bf_mfma_blocks"]
            ext.match_pairs_device(first, first + 1, d_k, d_d, d_n, (0, W, 0, H), d_m, d_nm, d_st, window, ratio, ori, cap)
            torch.cuda.synchronize()
            took = ext.debug_match_counters()["bf_mfma_blocks"] - before
            # the matrix kernel really listed blocks of this launch (a changed launch rule must not leave the
            # test green on the vector form alone) -- and under the knob it listed none
            if knob is None:
                assert took >= min_mfma_blocks, (took, min_mfma_blocks)
            else:
                assert took == 0, took
            out.append((d_m.cpu().numpy().reshape(NP, cap), d_nm.cpu().numpy(), d_st.cpu().numpy().reshape(NP, 3)))
        orbx.debug_set("match_no_mfma", None)
        for i, (k1, d1, k2, d2) in enumerate(sets):
            onm, om12, ost = oracle.match_init(k1, d1, k2, d2, (0, W, 0, H), window, ratio, ori)
            for (m, nm, st) in out:
                for j in range(i, NP, P):
                    assert nm[j] == onm and np.array_equal(m[j, :len(k1)], om12) and st[j].tolist() == ost.tolist(), (j, len(k1), len(k2))
        return out

    try:
        sets = []
        for i in range(P):
            n = int(rng.integers(530, cap - 8))  # more than 512: the wide path, three to five blocks of 256 queries
            if i % 7 == 3:  # few prototypes: tens of candidates per query; the smallest prototype sets overflow the lists
                protos = rng.integers(0, 256, (int(rng.choice([3, 40, 400])), 32), dtype=np.uint8)
                sets.append(_clustered_desc_pair(orbx, rng, n, protos, 10, W, H, 0.85, 6))
            else:
                protos = rng.integers(0, 256, (n, 32), dtype=np.uint8)
                sets.append(_clustered_desc_pair(orbx, rng, n, protos, 40, W, H, 1.0 if i % 3 else 0.8, 4))
        # (128 pairs of two to five blocks: the 256 blocks the kernel wants -- every block passes the brute-force test, all are its)
        o1 = run(sets, 8192, 0.9, True, reps=2, min_mfma_blocks=256)
        assert sum(int((m >= 0).sum()) for m in o1[0][0]) > P * 100
        # a window the middle of the frame passes: |x - bbx0| < r and |x - bbx1| < r need x in (W - r, r)
        sets2 = []
        for (k1, d1, k2, d2) in sets:
            o = np.argsort(k1["x"], kind="stable")
            sets2.append((k1[o], d1[o], k2, d2))
        run(sets2, 2600, 0.8, False, reps=2)
        # one block of fewer than 256 queries per pair (the last waves without a query), trains that are no multiple of 32, every
        # pair five times in the launch so that it has its 256 blocks
        sets3 = [(k1[:int(rng.integers(70, 250))], d1, k2, d2) for (k1, d1, k2, d2) in sets[:48]]
        sets3 = [(k1, d1[:len(k1)], k2, d2) for (k1, d1, k2, d2) in sets3] + sets[48:]
        run(sets3, 8192, 0.9, True, reps=5)
    finally:
        orbx.debug_set("match_no_mfma", None)
        ext.close()


def _flip_bits(rng, d, max_flips):
    for i in range(len(d)):
        for bit in rng.integers(0, 256, int(rng.integers(0, max_flips + 1))):
            d[i, bit >> 3] ^= np.uint8(1 << (bit & 7))
    return d


def _clustered_desc_pair(orbx, rng, n, protos, flips, w, h, level0_share, jitter):
    """Frame A: descriptors = prototypes with a few flipped bits (few prototypes -> many near-duplicates); frame B: a
    permutation of most of A, moved by up to `jitter` px, with more flipped bits, other octaves and angles mixed in."""
    k1 = np.zeros(n, orbx.KEYPOINT_DTYPE)
    k1["x"] = rng.uniform(0, w - 1, n).astype(np.float32)
    k1["y"] = rng.uniform(0, h - 1, n).astype(np.float32)
    k1["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    k1["octave"] = np.where(rng.uniform(0, 1, n) < level0_share, 0, rng.integers(1, 8, n))
    d1 = _flip_bits(rng, protos[rng.integers(0, len(protos), n)].copy(), flips)
    perm = rng.permutation(n)[:n - 7]
    k2 = k1[perm].copy()
    k2["x"] = np.clip(k2["x"] + rng.uniform(-jitter, jitter, len(perm)), 0, w - 1).astype(np.float32)
    k2["y"] = np.clip(k2["y"] + rng.uniform(-jitter, jitter, len(perm)), 0, h - 1).astype(np.float32)
    k2["angle"] = ((k2["angle"] + 20 + rng.choice([0, 0, 0, 0, 90, 200], len(perm)) + rng.uniform(-3, 3, len(perm))) % 360)
    k2["octave"] = np.where(rng.uniform(0, 1, len(perm)) < 0.9, k2["octave"], rng.integers(0, 3, len(perm)))
    d2 = _flip_bits(rng, d1[perm].copy(), flips)
    return k1, d1, k2, d2


def test_wide_matcher_contention(orbx, ext640, oracle):
    """Pairs beyond the LDS instances (more than 512 octave-0 queries / eligible trains): descriptors drawn from few
    prototypes, so trains are claimed by several queries, stolen and blocked, and the fixpoint needs several sweeps.
    Every case must equal the oracle whichever kernel ends up taking it; the first cases must be taken by the wide path
    itself (checked by switching the general kernel off)."""
    rng = np.random.default_rng(17)
    cases = (  # n, prototypes, max flipped bits, window, w, h, octave-0 share, ratio, checkOri, wide path must take it
        (1500, 1500, 20, 300, 1920, 1080, 0.7, 0.9, True, True),
        (3000, 900, 12, 150, 3840, 2160, 0.8, 0.9, True, True),      # ~3 near-duplicates per prototype
        (4000, 4000, 25, 4096, 3840, 2160, 1.0, 0.9, True, True),    # brute force, 4000 x 4000
        (2500, 2500, 25, 4096, 3840, 2160, 0.9, 0.6, False, False),  # ratio 0.6: longer lists
        (1800, 150, 8, 4096, 1920, 1080, 0.9, 0.9, True, False),     # heavy contention: may overflow claims / lists
        (2400, 200, 6, 400, 1280, 720, 0.9, 0.9, True, False),       # near-duplicates inside the windows
        (4600, 4600, 40, 200, 3840, 2160, 1.0, 0.9, True, False),    # more than 4096 queries: the general kernel
        # either side of the point where matchWidePrep stops storing the trains by grid column (a window that spans half the grid)
        (2000, 2000, 20, 230, 1280, 720, 0.8, 0.9, True, True),
        (2000, 2000, 20, 300, 1280, 720, 0.8, 0.9, True, True),
        (1200, 400, 10, 60, 752, 480, 0.9, 0.9, True, False),        # narrow windows, near-duplicates, columns of a small frame
        # octave-0 trains thinly spread over more than 1024 keypoints in no level order: k_match_jacobi's staging loop crosses its
        # 256-train limit in its SECOND chunk of 1024 keypoints (the early exit must be decided uniformly)
        (2600, 2600, 20, 300, 1920, 1080, 0.2, 0.9, True, True),
        (1501, 1501, 20, 100, 1920, 1080, 0.8, 0.9, True, True),     # odd capacity: the staged train records stay 16-byte aligned
    )
    for (n, npro, flips, win, w, h, share, ratio, ori, must) in cases:
        protos = rng.integers(0, 256, (npro, 32), dtype=np.uint8)
        k1, d1, k2, d2 = _clustered_desc_pair(orbx, rng, n, protos, flips, w, h, share, min(win, 300) / 3)
        fa, fb = orbx.Frame.from_arrays(k1, d1, (0, w, 0, h)), orbx.Frame.from_arrays(k2, d2, (0, w, 0, h))
        m = orbx.ORBmatcher(ratio, ori, extractor=ext640)
        onm, om12, ost = oracle.match_init(k1, d1, k2, d2, (0, w, 0, h), win, ratio, ori)
        nm, m12 = m.SearchForInitialization(fa, fb, win)
        assert nm == onm and np.array_equal(m12, om12) and list(m.last_stats) == ost.tolist(), (n, npro, win)
        if must:
            assert onm > 50
            orbx.debug_set("match_no_general", 1)
            try:
                nm2, m12b = m.SearchForInitialization(fa, fb, win)
            finally:
                orbx.debug_set("match_no_general", None)
            assert nm2 == onm and np.array_equal(m12b, om12), "the wide path handed the pair on: %r" % ((n, npro, win),)


def test_match_edge_cases(orbx, ext640, oracle):
    KP = orbx.KEYPOINT_DTYPE
    empty = orbx.Frame.from_arrays(np.zeros(0, KP), np.zeros((0, 32), np.uint8), (0, 640, 0, 480))
    k = np.zeros(5, KP)
    k["x"], k["y"] = [10, 630, 320, 639.6, 0.2], [10, 470, 240, 479.7, 0.1]  # incl. grid cells 64/48 -> dropped (Q12)
    d = np.arange(5 * 32, dtype=np.uint8).reshape(5, 32)
    f5 = orbx.Frame.from_arrays(k, d, (0, 640, 0, 480))
    m = orbx.ORBmatcher(0.9, True, extractor=ext640)
    assert m.SearchForInitialization(empty, f5, 100)[0] == 0
    nm, m12 = m.SearchForInitialization(f5, empty, 100)
    assert nm == 0 and (m12 == -1).all()
    nm, m12 = m.SearchForInitialization(f5, f5, 100)
    onm, om12, _ = oracle.match_init(k, d, k, d, (0, 640, 0, 480), 100, 0.9, True)
    assert nm == onm and np.array_equal(m12, om12)
    # quirk: stolen match in a pruned histogram bin is decremented twice (Q15)
    k1 = np.zeros(40, KP)
    k2 = np.zeros(40, KP)
    rng = np.random.default_rng(11)
    k1["x"], k1["y"] = rng.integers(50, 590, 40), rng.integers(50, 430, 40)
    k2["x"], k2["y"] = k1["x"], k1["y"]
    k1["angle"] = rng.integers(0, 360, 40)
    k2["angle"] = (k1["angle"] + rng.choice([0, 0, 0, 90, 200], 40)) % 360
    d1 = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    d2 = d1.copy()
    d2[5] = d2[6]  # collisions
    d1[7] = d1[8]
    fa, fb = orbx.Frame.from_arrays(k1, d1, (0, 640, 0, 480)), orbx.Frame.from_arrays(k2, d2, (0, 640, 0, 480))
    for win in (30, 100, 700):
        nm, m12 = m.SearchForInitialization(fa, fb, win)
        onm, om12, ost = oracle.match_init(k1, d1, k2, d2, (0, 640, 0, 480), win, 0.9, True)
        assert nm == onm and np.array_equal(m12, om12) and list(m.last_stats) == ost.tolist()


def test_device_resident_api(orbx, oracle):
    """Frames in HBM in, results in HBM out (torch only provides the device memory), incl. a non-4-aligned stride."""
    import torch
    from orb_slam_tracking_amd import synth
    assert torch.cuda.is_available()
    B, cap = 6, 1000
    oe = oracle.Extractor(*CANON)
    for (w, h) in ((640, 480), (642, 481)):
        frames = synth.synth_frames(B, w, h, 2000)
        e = orbx.ORBextractor(*CANON, max_width=w, max_height=h, max_batch=B)
        d_img = torch.from_numpy(frames).cuda()
        d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
        d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
        d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
        e.extract_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, cap)
        n = d_n.cpu().numpy()
        kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
        dd = d_d.cpu().numpy().reshape(B, cap, 32)
        ora = [oe(f) for f in frames]
        for f in range(B):
            assert n[f] == len(ora[f][1])
            _same(kk[f, :n[f]], dd[f, :n[f]], ora[f][1], ora[f][2])
        first, second = np.arange(0, B, 2, dtype=np.int32), np.arange(1, B, 2, dtype=np.int32)
        d_m = torch.zeros((B // 2) * cap, dtype=torch.int32, device="cuda")
        d_nm = torch.zeros(B // 2, dtype=torch.int32, device="cuda")
        d_st = torch.zeros(B // 2 * 3, dtype=torch.int32, device="cuda")
        e.match_pairs_device(first, second, d_k, d_d, d_n, (0, w, 0, h), d_m, d_nm, d_st, 100, 0.9, True, cap)
        mm, nm, st = d_m.cpu().numpy().reshape(B // 2, cap), d_nm.cpu().numpy(), d_st.cpu().numpy().reshape(-1, 3)
        for p in range(B // 2):
            a, b = ora[2 * p], ora[2 * p + 1]
            onm, om12, ost = oracle.match_init(a[1], a[2], b[1], b[2], (0, w, 0, h), 100, 0.9, True)
            assert nm[p] == onm and np.array_equal(mm[p, :len(om12)], om12) and st[p].tolist() == ost.tolist()
        e.close()


@pytest.mark.parametrize("cfg", [(1920, 1080, 4000), (3840, 2160, 8000)])
def test_full_size_configs(orbx, oracle, cfg):
    """BASELINE configs 3 and 5 at full size: extraction equals the oracle; extract -> match of a shifted frame is
    idempotent and finds the translation."""
    from orb_slam_tracking_amd import synth
    w, h, nf = cfg
    a, b = synth.synth_pair(w, h, 2)
    e = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
    oe = oracle.Extractor(nf, 1.2, 8, 20, 7)
    res = e.extract_batch(np.stack([a, b]))
    for img, (r, k, d) in zip((a, b), res):
        ro, ko, do = oe(img, cap=nf + 64)
        assert r == ro
        _same(k, d, ko, do)
    again = e.extract_batch(np.stack([a, b]))
    assert again[0][1].tobytes() == res[0][1].tobytes() and np.array_equal(again[1][2], res[1][2])  # idempotent
    fa = orbx.Frame.from_arrays(res[0][1], res[0][2], (0, w, 0, h))
    fb = orbx.Frame.from_arrays(res[1][1], res[1][2], (0, w, 0, h))
    nm, m12 = orbx.ORBmatcher(0.9, True, extractor=e).SearchForInitialization(fa, fb, 4096)
    onm, om12, _ = oracle.match_init(res[0][1], res[0][2], res[1][1], res[1][2], (0, w, 0, h), 4096, 0.9, True)
    assert nm == onm and np.array_equal(m12, om12)
    ok = m12 >= 0
    dx = res[1][1]["x"][m12[ok]] - res[0][1]["x"][ok]
    dy = res[1][1]["y"][m12[ok]] - res[0][1]["y"][ok]
    assert ok.sum() > 100 and abs(np.median(dx) - 7) <= 1 and abs(np.median(dy) - 4) <= 1  # B is A shifted by (+7, +4)
    e.close()


def _rowmajor_cands(rng, W, H, n):
    pos = np.sort(rng.choice(W * H, size=n, replace=False))
    r = rng.integers(1, int(rng.integers(2, 200)), size=n)
    return np.stack([pos % W, pos // W, r], 1).astype(np.float32)


def test_device_std_sort_replay(orbx, ext640, oracle):
    """The one-lane replay of libstdc++'s introsort gives the library's exact permutation, ties included."""
    rng = np.random.default_rng(21)
    cases = []
    for n in (1, 2, 3, 15, 16, 17, 18, 33, 100, 257, 1000, 2500):
        for hi in (2, 5, 60):
            t = np.stack([rng.integers(2, 2 + hi, n), rng.integers(0, 40, n) * 16, np.arange(n)], 1)
            cases.append(t)
    cases.append(np.stack([np.arange(3000), np.zeros(3000, int), np.arange(3000)], 1))         # sorted
    cases.append(np.stack([np.arange(3000)[::-1], np.zeros(3000, int), np.arange(3000)], 1))   # reversed
    cases.append(np.stack([np.full(3000, 7), np.full(3000, 3), np.arange(3000)], 1))           # all equal
    # median-of-3 killer (drives introsort into its heapsort fallback)
    n = 4096
    k = n // 2
    killer = np.zeros(n, int)
    for i in range(1, k + 1):
        if i & 1:
            killer[i - 1] = i
            killer[i] = k + i
        killer[k + i - 1] = 2 * i
    cases.append(np.stack([killer, np.zeros(n, int), np.arange(n)], 1))
    # n <= 512 runs the workgroup-parallel replay (closed-form partitions, breadth first): sizes around its limits, sorted /
    # reversed / constant inputs, few distinct keys, and the median-of-3 killer (depth budget -> per-range heapsort)
    # (sizes with bit 1 set take the form k_octree_big uses: the whole sort from the partition phase's own ranges, wave tasks included)
    for n in (19, 20, 31, 34, 64, 66, 128, 130, 131, 200, 202, 254, 255, 256, 300, 303, 400, 449, 450, 510, 511, 512):
        for hi in (1, 2, 3, 9, 1000):
            cases.append(np.stack([rng.integers(2, 2 + hi, n), rng.integers(0, 6, n) * 16, np.arange(n)], 1))
        cases.append(np.stack([np.arange(n), np.zeros(n, int), np.arange(n)], 1))
        cases.append(np.stack([np.arange(n)[::-1], np.zeros(n, int), np.arange(n)], 1))
        cases.append(np.stack([np.full(n, 7), np.full(n, 3), np.arange(n)], 1))
    for n in (64, 130, 254, 256, 510, 512):
        k = n // 2
        killer = np.zeros(n, int)
        for i in range(1, k + 1):
            if i & 1:
                killer[i - 1] = i
                killer[i] = k + i
            killer[k + i - 1] = 2 * i
        cases.append(np.stack([killer, np.zeros(n, int), np.arange(n)], 1))
    for t in cases:
        got = ext640.debug_std_sort(t)
        exp = oracle.std_sort_sized(t)
        assert np.array_equal(got, exp), len(t)


def test_device_octree_random(orbx, ext640, oracle):
    """Device DistributeOctTree (both kernels) vs the oracle on random candidate sets: same keys, same list order."""
    rng = np.random.default_rng(5)
    done = 0
    for it in range(160):
        W, H = int(rng.integers(40, 900)), int(rng.integers(40, 700))
        if round(float(np.float32(W) / np.float32(H))) < 1:
            continue
        n = int(rng.integers(0, min(W * H // 4, 1800 if it % 3 else 6000)))
        xyr = _rowmajor_cands(rng, W, H, n)
        N = int(rng.integers(0, max(2, 2 * n // 3 + 2)))
        exp = oracle.distribute(xyr, 16, 16 + W, 16, 16 + H, N)[:N]  # the pipeline truncates to the quota (cpp:1159-1161)
        for variant in (0, 1, 2, 3, 4, 5):
            got = ext640.debug_distribute_device(xyr, 16, 16 + W, 16, 16 + H, N, variant)
            assert got.shape == exp.shape and np.array_equal(got, exp), (it, variant, W, H, n, N)
        done += 1
    assert done > 100


def test_device_octree_small_quotas(orbx, ext640, oracle):
    """Quotas of a few keypoints against many candidates (nfeatures = 30 leaves 11, 7, 5, 3, 2, 2 per level): the last,
    partial pass then creates more multi-key children than the quota (up to N + 4), and with several roots (nIni > 1) the
    first pass alone exceeds 4 N nodes.  Found by tools/fuzz_parity.py (714 x 634, (30, 1.5, 6, 4, 4))."""
    rng = np.random.default_rng(23)
    for it in range(90):
        H = int(rng.integers(60, 500))
        W = int(H * float(rng.choice([1.0, 1.2, 1.6, 2.4, 3.3, 4.4])))
        n = int(rng.integers(20, min(W * H // 4, 700 if it % 3 else 9000)))
        xyr = _rowmajor_cands(rng, W, H, n)
        for N in (1, 2, 3, 5, 11, int(rng.integers(4, 40))):
            exp = oracle.distribute(xyr, 16, 16 + W, 16, 16 + H, N)[:N]
            for variant in (0, 1, 2, 3, 4, 5):
                got = ext640.debug_distribute_device(xyr, 16, 16 + W, 16, 16 + H, N, variant)
                assert got.shape == exp.shape and np.array_equal(got, exp), (it, variant, W, H, n, N)
    from orb_slam_tracking_amd import synth
    params = (30, 1.5, 6, 4, 4)
    fr = synth.synth_frames(2, 714, 634, 5133)
    e = orbx.ORBextractor(*params, max_width=714, max_height=634, max_batch=2)
    oe = oracle.Extractor(*params)
    for f, (r, k, d) in zip(fr, e.extract_batch(fr)):
        ro, ko, do = oe(f)
        assert r == ro
        _same(k, d, ko, do)
    e.close()


@pytest.mark.parametrize("W,H", [(595, 368), (501, 150), (333, 100), (596, 368), (513, 256), (1025, 512), (769, 256), (767, 256), (1535, 512)])
def test_device_octree_root_boundary_columns(orbx, ext640, oracle, W, H):
    """Rectangles with two or three roots whose boundary is not integral (odd width: hX = 297.5; three roots: hX = 167.0 / 111.0
    ...): the reference's root = x / hX (cpp:747) puts the column x = (int)hX into root 0, outside that root's rectangle
    [0, (int)hX), where DivideNode routes it right for ever -- only the full 16-level path code tells it from the rectangle's
    last column.  The LDS kernel's 32-bit sort keys drop the digits beyond the one-pixel depth D, so D must cover the split at
    which that column parts from the rectangle's last one (octDepthBits: ceil(hX) + 1; rectangles of power-of-two width -- 513 / 2,
    1025 / 2, 769 / 3 -- are the tight case); candidates are packed onto the boundary columns to make a wrong D visible."""
    rng = np.random.default_rng(W)
    nIni = round(float(np.float32(W) / np.float32(H)))
    assert nIni >= 2
    hX = np.float32(W) / np.float32(nIni)
    cols = sorted({int(c) for i in range(1, nIni) for c in (int(hX * i) - 1, int(hX * i), int(hX * i) + 1) if 0 <= c < W})
    pts = {(x, y) for x in cols for y in range(0, H, 1 + int(rng.integers(0, 2)))}
    while len(pts) < 1500:
        pts.add((int(rng.integers(0, W)), int(rng.integers(0, H))))
    xyr = np.array(sorted([(x, y, int(rng.integers(6, 40))) for x, y in pts], key=lambda t: (t[1], t[0])), np.float32)
    for N in (60, 217, 250, 400):
        exp = oracle.distribute(xyr, 16, 16 + W, 16, 16 + H, N)[:N]
        for variant in (0, 1, 2, 3, 4, 5):
            got = ext640.debug_distribute_device(xyr, 16, 16 + W, 16, 16 + H, N, variant)
            assert got.shape == exp.shape and np.array_equal(got, exp), (N, variant)
    # the deep case: a quota that forces the tree down to single pixels right at a boundary -- the column pair ((int)hX - 1,
    # (int)hX) over 140 rows, and hardly anything else
    for i in range(1, nIni):
        c = int(hX * i)
        h0 = min(140, H - 2)
        pts2 = {(x, y) for x in (c - 1, c) for y in range(1, 1 + h0)}
        while len(pts2) < 2 * h0 + 40:
            pts2.add((int(rng.integers(0, W)), int(rng.integers(0, H))))
        xyr2 = np.array(sorted([(x, y, int(rng.integers(6, 40))) for x, y in pts2], key=lambda t: (t[1], t[0])), np.float32)
        for N in (200, 250, 256):
            exp = oracle.distribute(xyr2, 16, 16 + W, 16, 16 + H, N)[:N]
            for variant in (0, 1, 2, 3, 4, 5):
                got = ext640.debug_distribute_device(xyr2, 16, 16 + W, 16, 16 + H, N, variant)
                assert got.shape == exp.shape and np.array_equal(got, exp), ("deep", i, N, variant)


@pytest.mark.parametrize("shape", [(608, 448, 217), (720, 448, 434), (1888, 1048, 869), (3808, 2128, 1737), (147, 102, 60)])
def test_device_octree_level_geometries(orbx, ext640, oracle, shape):
    W, H, N = shape
    rng = np.random.default_rng(W)
    for dens in (0.002, 0.01, 0.05):
        n = int(W * H * dens)
        xyr = _rowmajor_cands(rng, W, H, n)
        xyr[:, 2] = rng.integers(6, 12, n)  # heavy response ties
        half = n // 2  # cluster half of the points
        xyr[:half, 0] = np.clip(np.round(W * 0.3 + rng.normal(0, W * 0.05, half)), 0, W - 1)
        xyr[:half, 1] = np.clip(np.round(H * 0.6 + rng.normal(0, H * 0.05, half)), 0, H - 1)
        key = xyr[:, 1] * 4096 + xyr[:, 0]
        _, uniq = np.unique(key, return_index=True)
        xyr = xyr[uniq]  # unique, and sorted row-major by construction of np.unique
        exp = oracle.distribute(xyr, 16, 16 + W, 16, 16 + H, N)[:N]
        for variant in (0, 1, 2, 3, 4, 5):
            got = ext640.debug_distribute_device(xyr, 16, 16 + W, 16, 16 + H, N, variant)
            assert np.array_equal(got, exp), (dens, variant)


@pytest.mark.parametrize("shape", [(720, 448, 434), (1888, 1048, 869), (3808, 2128, 1737), (2640, 1480, 1207), (1500, 300, 500)])
def test_device_octree_many_workgroups(orbx, ext640, oracle, shape):
    """The kernels of large units (k_octree_buckets: a workgroup per bucket of keys; k_octree_big: the tree arithmetic) WITHOUT the
    one-workgroup kernel behind them (variant 6): evenly spread candidates at the densities FAST leaves on real frames must be
    taken by them and equal the oracle; a scene that overfills a bucket's slot (everything in one blob) must be handed on --
    ORBX_E_CAPACITY here, the global-scratch kernel in the pipeline (variant 5, which must equal the oracle always)."""
    W, H, N = shape
    rng = np.random.default_rng(W + 1)
    taken = 0
    for dens in (0.0015, 0.006, 0.02):
        n = int(W * H * dens)
        xyr = _rowmajor_cands(rng, W, H, n)
        xyr[:, 2] = rng.integers(6, 14, len(xyr))  # heavy response ties: the first of the highest responses decides
        for q in (N, max(N // 3, 1), min(3 * N, 2000)):
            exp = oracle.distribute(xyr, 16, 16 + W, 16, 16 + H, q)[:q]
            try:
                got = ext640.debug_distribute_device(xyr, 16, 16 + W, 16, 16 + H, q, 6)
            except orbx.OrbxError as e:  # (a small quota allows only shallow buckets: at the highest density they may overflow)
                assert e.code == orbx.E_CAPACITY and dens > 0.01 and q < N, (dens, q)
                continue
            assert got.shape == exp.shape and np.array_equal(got, exp), (dens, q)
            taken += 1
    assert taken >= 8
    # one dense blob: more keys than a bucket's slot holds
    n = 30000
    pts = {(int(x), int(y)) for x, y in zip(np.clip(rng.normal(W * 0.4, 25, n), 0, W - 1), np.clip(rng.normal(H * 0.5, 25, n), 0, H - 1))}
    xyr = np.array(sorted([(x, y, int(rng.integers(6, 40))) for x, y in pts], key=lambda t: (t[1], t[0])), np.float32)
    exp = oracle.distribute(xyr, 16, 16 + W, 16, 16 + H, N)[:N]
    assert np.array_equal(ext640.debug_distribute_device(xyr, 16, 16 + W, 16, 16 + H, N, 5), exp)
    try:
        got = ext640.debug_distribute_device(xyr, 16, 16 + W, 16, 16 + H, N, 6)
        assert np.array_equal(got, exp)  # (a geometry whose buckets hold the blob after all)
    except orbx.OrbxError as e:
        assert e.code == orbx.E_CAPACITY


def test_dense_cluster_does_not_alternate_redos(orbx, oracle):
    """a scene with one dense cluster of corners at 1920x1080 / 4000 features.  The many-workgroup selection picks its
    bucket depth from the previous batch; from the candidate COUNT alone the level-0 unit would get a depth whose bucket under the
    cluster overflows, be redone in place, report the largest count, get the deepest depth, report its true count from there and
    return to the depth that overflowed -- every second batch through the one-workgroup code.  With the fullest bucket's fill and
    its depth in the feedback no batch after the second redoes a unit, and every batch equals the oracle."""
    import torch
    w, h, B, cap = 1920, 1080, 4, 4000
    params = (4000, 1.2, 8, 20, 7)
    rng = np.random.default_rng(77)
    frames = np.full((B, h, w), 90, np.uint8)
    for f in range(B):
        frames[f, 300:640, 500:840] = rng.integers(0, 256, (340, 340), dtype=np.uint8)  # the cluster
        for _ in range(60):  # a few corners elsewhere
            y, x = int(rng.integers(30, h - 40)), int(rng.integers(30, w - 40))
            frames[f, y:y + 9, x:x + 9] = int(rng.integers(150, 256))
    oe = oracle.Extractor(*params)
    _, ko, do = oe(frames[0], cap=cap)
    c0 = oe.level_candidates(0)
    assert len(c0) > 1500, len(c0)  # (what makes a depth-1 bucket of 1024 slots overflow)
    e = orbx.ORBextractor(*params, max_width=w, max_height=h, max_batch=B)
    d_img = torch.from_numpy(frames).cuda()
    d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
    d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    redone = []
    for it in range(8):
        e.extract_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, cap)
        assert e.debug_last_launch()["octree_instance"] == 0  # (large units: the many-workgroup kernels)
        redone.append(int(sum(e.debug_selection_units(f)[1].sum() for f in range(B))))
        n = d_n.cpu().numpy()
        kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
        dd = d_d.cpu().numpy().reshape(B, cap, 32)
        assert n[0] == len(ko)
        _same(kk[0, :n[0]], dd[0, :n[0]], ko, do)
    print("units redone per batch:", redone)
    assert sum(redone[2:]) == 0, redone
    e.close()


@pytest.mark.parametrize("B", [4, 40])
def test_redone_and_failed_units_through_both_bookkeepers(orbx, oracle, B):
    """the per-frame bookkeeping of the selection stage exists once (orbx_device.h) and runs in two places -- in
    k_sel_compact, and for launches of up to 256 units inside k_describe_patch, which then reads the staging lists itself (B = 4:
    32 units; B = 40: 320 units, k_sel_compact).  Through both: a unit the many-workgroup kernels REDO in place (bucket depth 0
    forced: the cluster overfills its bucket) gives the oracle's result and carries the tag bit; a unit that FAILS (the redo switched
    off) raises ORBX_E_CAPACITY at the batch's wait."""
    import torch
    w, h, cap = 1920, 1080, 4000
    params = (4000, 1.2, 8, 20, 7)
    rng = np.random.default_rng(78)
    frames = np.full((B, h, w), 90, np.uint8)
    for f in range(B):
        frames[f, 300:640, 500:840] = rng.integers(0, 256, (340, 340), dtype=np.uint8)
    oe = oracle.Extractor(*params)
    ora = [oe(frames[f], cap=cap) for f in (0, B - 1)]
    e = orbx.ORBextractor(*params, max_width=w, max_height=h, max_batch=B)
    d_img = torch.from_numpy(frames).cuda()
    d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
    d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    try:
        orbx.debug_set("no_split", 1)  # (one launch of B frames: B * 8 units)
        orbx.debug_set("oct_big_depth", 0)
        for it in range(2):
            e.extract_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, cap)
            info = e.debug_last_launch()
            assert info["staged_lists"] == (1 if B * 8 <= 256 else 0) and info["octree_instance"] == 0, info
            assert e.debug_selection_units(0)[1][0] == 1 and e.debug_selection_units(B - 1)[1][0] == 1  # level 0 was redone
            n = d_n.cpu().numpy()
            kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
            dd = d_d.cpu().numpy().reshape(B, cap, 32)
            for f, (_, ko, do) in zip((0, B - 1), ora):
                assert n[f] == len(ko)
                _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
        orbx.debug_set("oct_big_no_fallback", 1)
        with pytest.raises(orbx.OrbxError) as ex:
            e.extract_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, cap)
        assert ex.value.code == orbx.E_CAPACITY
        assert e.debug_selection_units(0)[0][0] < 0
        orbx.debug_set("oct_big_no_fallback", None)
        orbx.debug_set("oct_big_depth", None)
        e.extract_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, cap)  # the context works on
        n = d_n.cpu().numpy()
        assert n[0] == len(ora[0][1])
    finally:
        for k in ("no_split", "oct_big_depth", "oct_big_no_fallback"):
            orbx.debug_set(k, None)
        e.close()


def test_fused_two_stream_batch(orbx, oracle):
    """orbx_extract_match_batch_device (half-batches on two streams, matching fused behind extraction) gives exactly
    the results of the separate calls and of the oracle, incl. pairs that straddle the halves or are out of order."""
    import torch
    from orb_slam_tracking_amd import synth
    B, cap, w, h = 40, 1000, 640, 480
    frames = synth.synth_frames(B, w, h, 3000)
    oe = oracle.Extractor(*CANON)
    ora = [oe(f) for f in frames]
    e = orbx.ORBextractor(*CANON, max_width=w, max_height=h, max_batch=B)
    d_img = torch.from_numpy(frames).cuda()
    for first, second in ((np.arange(0, B, 2), np.arange(1, B, 2)),                       # consecutive pairs
                          (np.array([0, 2, 19, 21, 38, 5]), np.array([1, 3, 20, 22, 39, 30]))):  # straddling / unordered
        d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
        d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
        d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
        P = len(first)
        d_m = torch.zeros(P * cap, dtype=torch.int32, device="cuda")
        d_nm = torch.zeros(P, dtype=torch.int32, device="cuda")
        d_st = torch.zeros(P * 3, dtype=torch.int32, device="cuda")
        e.extract_match_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, first, second, (0, w, 0, h), d_m, d_nm, d_st, 100,
                                     0.9, True, cap)
        n = d_n.cpu().numpy()
        kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
        dd = d_d.cpu().numpy().reshape(B, cap, 32)
        for f in range(B):
            assert n[f] == len(ora[f][1])
            _same(kk[f, :n[f]], dd[f, :n[f]], ora[f][1], ora[f][2])
        mm, nm, st = d_m.cpu().numpy().reshape(P, cap), d_nm.cpu().numpy(), d_st.cpu().numpy().reshape(-1, 3)
        for p in range(P):
            a, b = ora[int(first[p])], ora[int(second[p])]
            onm, om12, ost = oracle.match_init(a[1], a[2], b[1], b[2], (0, w, 0, h), 100, 0.9, True)
            assert nm[p] == onm and np.array_equal(mm[p, :len(om12)], om12) and st[p].tolist() == ost.tolist()
    e.close()


def test_async_batches_equal_sync(orbx):
    """orbx_extract_match_batch_device_async: three batches issued back to back (two in flight, the third call waits for the
    oldest) into alternating output sets give what the synchronous call gives; stage profiling survives the overlap."""
    import torch
    from orb_slam_tracking_amd import synth
    B, cap, w, h = 32, 1000, 640, 480
    e = orbx.ORBextractor(*CANON, max_width=w, max_height=h, max_batch=B)
    first = np.arange(0, B, 2, dtype=np.int32)
    imgs = [torch.from_numpy(synth.synth_frames(B, w, h, seed0=300 + 50 * i)).cuda() for i in range(3)]

    def outs():
        return dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
                    n=torch.zeros(B, dtype=torch.int32, device="cuda"), m=torch.full(((B // 2) * cap,), -5, dtype=torch.int32, device="cuda"),
                    nm=torch.zeros(B // 2, dtype=torch.int32, device="cuda"))
    ref = []
    for im in imgs:
        o = outs()
        e.extract_match_batch_device(im, B, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h), o["m"], o["nm"],
                                     None, 100, 0.9, True, cap)
        ref.append({k: v.cpu().numpy().copy() for k, v in o.items()})
    e.profile_enable(True)
    e.profile_reset()
    sets = [outs(), outs(), outs()]
    for i, im in enumerate(imgs):
        o = sets[i]
        e.extract_match_batch_device_async(im, B, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h), o["m"],
                                           o["nm"], None, 100, 0.9, True, cap)
    e.wait_one()
    e.wait()
    e.wait()  # nothing in flight: a no-op
    prof = e.profile_get()
    e.profile_enable(False)
    assert all(prof[s][1] > 0 and prof[s][0] > 0 for s in ("pyramid", "fast", "select", "describe", "match"))
    for i in range(3):
        got = {k: v.cpu().numpy() for k, v in sets[i].items()}
        n = got["n"]
        assert np.array_equal(n, ref[i]["n"]) and np.array_equal(got["nm"], ref[i]["nm"])
        kk, rk = got["k"].reshape(B, cap * 28), ref[i]["k"].reshape(B, cap * 28)
        dd, rd = got["d"].reshape(B, cap * 32), ref[i]["d"].reshape(B, cap * 32)
        mm, rm = got["m"].reshape(B // 2, cap), ref[i]["m"].reshape(B // 2, cap)
        for f in range(B):
            assert np.array_equal(kk[f, :n[f] * 28], rk[f, :n[f] * 28]) and np.array_equal(dd[f, :n[f] * 32], rd[f, :n[f] * 32])
        for p_ in range(B // 2):
            assert np.array_equal(mm[p_, :n[2 * p_]], rm[p_, :n[2 * p_]])
    # a synchronous call right behind asynchronous ones
    o = outs()
    e.extract_match_batch_device_async(imgs[0], B, w, h, w, w * h, sets[0]["k"], sets[0]["d"], sets[0]["n"], first, first + 1,
                                       (0, w, 0, h), sets[0]["m"], sets[0]["nm"], None, 100, 0.9, True, cap)
    e.extract_match_batch_device(imgs[1], B, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h), o["m"], o["nm"],
                                 None, 100, 0.9, True, cap)
    assert np.array_equal(o["n"].cpu().numpy(), ref[1]["n"]) and np.array_equal(o["nm"].cpu().numpy(), ref[1]["nm"])
    assert np.array_equal(sets[0]["nm"].cpu().numpy(), ref[0]["nm"])
    # another batch size and another pair list behind a batch in flight (both make the context drain first)
    B2 = 20
    f2 = np.arange(1, B2, 2, dtype=np.int32)  # pairs (2k + 1, 2k): B matched against A
    o1, o2 = outs(), outs()
    e.extract_match_batch_device(imgs[2], B2, w, h, w, w * h, o1["k"], o1["d"], o1["n"], f2, f2 - 1, (0, w, 0, h), o1["m"], o1["nm"],
                                 None, 100, 0.9, True, cap)
    e.extract_match_batch_device_async(imgs[0], B, w, h, w, w * h, sets[1]["k"], sets[1]["d"], sets[1]["n"], first, first + 1,
                                       (0, w, 0, h), sets[1]["m"], sets[1]["nm"], None, 100, 0.9, True, cap)
    e.extract_match_batch_device_async(imgs[2], B2, w, h, w, w * h, o2["k"], o2["d"], o2["n"], f2, f2 - 1, (0, w, 0, h), o2["m"],
                                       o2["nm"], None, 100, 0.9, True, cap)
    e.wait()
    assert np.array_equal(sets[1]["nm"].cpu().numpy(), ref[0]["nm"]) and np.array_equal(sets[1]["n"].cpu().numpy(), ref[0]["n"])
    assert np.array_equal(o2["n"].cpu().numpy()[:B2], o1["n"].cpu().numpy()[:B2])
    assert np.array_equal(o2["nm"].cpu().numpy()[:B2 // 2], o1["nm"].cpu().numpy()[:B2 // 2])
    assert np.array_equal(o2["n"].cpu().numpy()[:B2], ref[2]["n"][:B2]) and o1["nm"].cpu().numpy()[:B2 // 2].sum() > 100
    e.close()


def test_host_async_batches_equal_oracle(orbx, oracle):
    """orbx_extract_match_batch_host_async: frames from host memory (page-locked and pageable, a row stride that differs from the
    width, frames with gaps between them), results into host arrays, on three lanes with three batches in flight, with depth 0, and
    with a frame size that makes the lanes grow -- every frame and every pair against the oracle.  A wide window sends pairs
    through the wide matcher kernels, which must travel with the batch (the copies back run behind the kernels)."""
    import torch
    from orb_slam_tracking_amd import synth
    KP = orbx.KEYPOINT_DTYPE
    B, cap = 18, 1000
    oe = oracle.Extractor(*CANON)
    e = orbx.ORBextractor(*CANON, max_width=400, max_height=300, max_batch=4)  # (smaller than what comes: the lanes grow)
    first = np.arange(0, B - 1, 2, dtype=np.int32)

    def outs(pin):
        mk = (lambda n, dt: torch.zeros(n, dtype=dt).pin_memory()) if pin else (lambda n, dt: torch.zeros(n, dtype=dt))
        return dict(k=mk(B * cap * 28, torch.uint8), d=mk(B * cap * 32, torch.uint8), n=mk(B, torch.int32), m=mk((B // 2) * cap, torch.int32),
                    nm=mk(B // 2, torch.int32), st=mk((B // 2) * 3, torch.int32))

    def check(o, frames, w, h, win):
        n = o["n"].numpy()
        kk = o["k"].numpy().view(KP).reshape(B, cap)
        dd = o["d"].numpy().reshape(B, cap, 32)
        mm = o["m"].numpy().reshape(B // 2, cap)
        ora = [oe(f) for f in frames]
        for f in range(B):
            assert n[f] == len(ora[f][1])
            _same(kk[f, :n[f]], dd[f, :n[f]], ora[f][1], ora[f][2])
        for p_ in range(B // 2):
            a, b = ora[2 * p_], ora[2 * p_ + 1]
            onm, om12, ost = oracle.match_init(a[1], a[2], b[1], b[2], (0, w, 0, h), win, 0.9, True)
            assert o["nm"].numpy()[p_] == onm and np.array_equal(mm[p_, :len(om12)], om12)
            assert o["st"].numpy().reshape(-1, 3)[p_].tolist() == ost.tolist()

    cases = []  # (host array handed over, frames, w, h, stride, frame stride, window, pinned outputs)
    for i, (w, h, pad, gap, win, pin) in enumerate([(640, 480, 0, 0, 100, True), (640, 480, 0, 0, 4096, True), (324, 243, 12, 1000, 100, True),
                                                    (640, 480, 0, 0, 100, False), (500, 375, 0, 0, 100, True)]):
        fr = synth.synth_frames(B, w, h, seed0=900 + 50 * i)
        stride, fs = w + pad, (w + pad) * h + gap
        host = torch.zeros(B * fs, dtype=torch.uint8)
        if pin:
            host = host.pin_memory()
        hv = host.numpy()
        for f in range(B):
            hv[f * fs:f * fs + stride * h].reshape(h, stride)[:, :w] = fr[f]
        cases.append((host, fr, w, h, stride, fs, win, pin))
    for depth in (3, 0):
        e.set_pipeline_depth(depth)
        sets = [outs(c[7]) for c in cases]
        for j, (host, fr, w, h, stride, fs, win, pin) in enumerate(cases):
            o = sets[j]
            e.extract_match_batch_host_async(host, B, w, h, stride, fs, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h), o["m"],
                                             o["nm"], o["st"], win, 0.9, True, cap)
        e.wait()
        for j, (host, fr, w, h, stride, fs, win, pin) in enumerate(cases):
            check(sets[j], fr, w, h, win)
    with pytest.raises(orbx.OrbxError):
        e.extract_match_batch_host_async(None, B, 640, 480, 640, 640 * 480, sets[0]["k"], sets[0]["d"], sets[0]["n"], first, first + 1,
                                         (0, 640, 0, 480), sets[0]["m"], sets[0]["nm"], None, 100, 0.9, True, cap)
    # no pairs at all (extraction only), no stats array; a capacity below the extractor's quota sum and a stride below the width
    host, fr, w, h, stride, fs, win, pin = cases[0]
    o = outs(True)
    none = np.zeros(0, np.int32)
    e.set_pipeline_depth(2)
    e.extract_match_batch_host_async(host, B, w, h, stride, fs, o["k"], o["d"], o["n"], none, none, (0, w, 0, h), None, None, None, win, 0.9, True, cap)
    e.wait()
    n = o["n"].numpy()
    kk = o["k"].numpy().view(KP).reshape(B, cap)
    dd = o["d"].numpy().reshape(B, cap, 32)
    for f in (0, B - 1):
        _, ko, do = oe(fr[f])
        assert n[f] == len(ko)
        _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
    with pytest.raises(orbx.OrbxError) as ex:
        e.extract_match_batch_host_async(host, B, w, h, stride, fs, o["k"], o["d"], o["n"], none, none, (0, w, 0, h), None, None, None, win, 0.9, True, 500)
    assert ex.value.code == orbx.E_CAPACITY
    with pytest.raises(orbx.OrbxError) as ex:
        e.extract_match_batch_host_async(host, B, w, h, w - 4, fs, o["k"], o["d"], o["n"], none, none, (0, w, 0, h), None, None, None, win, 0.9, True, cap)
    assert ex.value.code == orbx.E_BADARG
    e.wait()
    e.close()


def test_pipeline_lanes_equal_sync(orbx):
    """orbx_set_pipeline_depth(3): seven stream-ordered batches go, whole, to three lanes (three in flight, a fourth call waits for
    the oldest) and give what the synchronous call gives; batch size, pair list and frame size change on the way; synchronous
    calls in between; depth back to 0; stage profiling sums over the lanes."""
    import torch
    from orb_slam_tracking_amd import synth
    B, cap, w, h = 34, 1000, 640, 480
    e = orbx.ORBextractor(*CANON, max_width=w, max_height=h, max_batch=B)
    first = np.arange(0, B - 1, 2, dtype=np.int32)
    imgs = [torch.from_numpy(synth.synth_frames(B, w, h, seed0=700 + 40 * i)).cuda() for i in range(4)]

    def outs():
        return dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
                    n=torch.zeros(B, dtype=torch.int32, device="cuda"), m=torch.full(((B // 2) * cap,), -5, dtype=torch.int32, device="cuda"),
                    nm=torch.zeros(B // 2, dtype=torch.int32, device="cuda"))

    def same(got, ref, nb=B):
        g = {k: v.cpu().numpy() for k, v in got.items()}
        n = g["n"][:nb]
        assert np.array_equal(n, ref["n"][:nb]) and np.array_equal(g["nm"][:nb // 2], ref["nm"][:nb // 2])
        kk, rk = g["k"].reshape(B, cap * 28), ref["k"].reshape(B, cap * 28)
        dd, rd = g["d"].reshape(B, cap * 32), ref["d"].reshape(B, cap * 32)
        mm, rm = g["m"].reshape(B // 2, cap), ref["m"].reshape(B // 2, cap)
        for f in range(nb):
            assert np.array_equal(kk[f, :n[f] * 28], rk[f, :n[f] * 28]) and np.array_equal(dd[f, :n[f] * 32], rd[f, :n[f] * 32])
        for p_ in range(nb // 2):
            assert np.array_equal(mm[p_, :n[2 * p_]], rm[p_, :n[2 * p_]])

    ref = []
    for im in imgs:
        o = outs()
        e.extract_match_batch_device(im, B, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h), o["m"], o["nm"],
                                     None, 100, 0.9, True, cap)
        ref.append({k: v.cpu().numpy().copy() for k, v in o.items()})
    e.set_pipeline_depth(3)
    e.profile_enable(True)
    e.profile_reset()
    sets = [outs() for _ in range(3)]
    order = [0, 1, 2, 3, 1, 0, 2]
    done = []
    for j, i in enumerate(order):
        o = sets[j % 3]
        if j >= 3:  # the set is about to be reused: its batch (j - 3) is the oldest in flight
            e.wait_one()
            same(o, ref[order[j - 3]])
            done.append(j - 3)
        e.extract_match_batch_device_async(imgs[i], B, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h), o["m"],
                                           o["nm"], None, 100, 0.9, True, cap)
    e.wait()
    for j in range(len(order) - 3, len(order)):
        same(sets[j % 3], ref[order[j]])
    prof = e.profile_get()
    e.profile_enable(False)
    assert all(prof[s][1] > 0 and prof[s][0] > 0 for s in ("pyramid", "fast", "select", "describe", "match"))
    # a synchronous call between stream-ordered ones; then a smaller batch with another pair list on the lanes
    o = outs()
    e.extract_match_batch_device_async(imgs[3], B, w, h, w, w * h, sets[0]["k"], sets[0]["d"], sets[0]["n"], first, first + 1,
                                       (0, w, 0, h), sets[0]["m"], sets[0]["nm"], None, 100, 0.9, True, cap)
    e.extract_match_batch_device(imgs[1], B, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h), o["m"], o["nm"],
                                 None, 100, 0.9, True, cap)
    same(o, ref[1])
    same(sets[0], ref[3])
    B2 = 20
    f2 = np.arange(0, B2, 2, dtype=np.int32)
    e.extract_match_batch_device_async(imgs[2], B2, w, h, w, w * h, sets[1]["k"], sets[1]["d"], sets[1]["n"], f2, f2 + 1, (0, w, 0, h),
                                       sets[1]["m"], sets[1]["nm"], None, 100, 0.9, True, cap)
    e.extract_match_batch_device_async(imgs[0], B, w, h, w, w * h, sets[2]["k"], sets[2]["d"], sets[2]["n"], first, first + 1,
                                       (0, w, 0, h), sets[2]["m"], sets[2]["nm"], None, 100, 0.9, True, cap)
    e.wait()
    same(sets[1], ref[2], nb=B2)
    same(sets[2], ref[0])
    # back to the two-halves mode
    e.set_pipeline_depth(0)
    e.extract_match_batch_device_async(imgs[1], B, w, h, w, w * h, sets[0]["k"], sets[0]["d"], sets[0]["n"], first, first + 1,
                                       (0, w, 0, h), sets[0]["m"], sets[0]["nm"], None, 100, 0.9, True, cap)
    e.wait()
    same(sets[0], ref[1])
    e.close()


def test_wide_matcher_issued_late(orbx, oracle):
    """The wide matcher kernels travel with a batch only while batches need them.  After a run of batches that did not (few
    keypoints), batches that do get them at their wait - also stream-ordered, two in flight - and then with the batch again."""
    import torch
    from orb_slam_tracking_amd import synth
    params = (2000, 1.2, 2, 20, 7)  # two levels: ~1100 octave-0 keypoints per frame -> beyond k_match_jacobi's tables
    B, cap, w, h = 32, 2000, 640, 480
    e = orbx.ORBextractor(*params, max_width=w, max_height=h, max_batch=B)
    oe = oracle.Extractor(*params)
    rich = synth.synth_frames(B, w, h, seed0=4200)
    flat = np.full((B, h, w), 90, np.uint8)
    flat[:, 100:140, 200:260] = 200  # one rectangle: a handful of corners
    d_rich, d_flat = torch.from_numpy(rich).cuda(), torch.from_numpy(flat).cuda()
    first = np.arange(0, B, 2, dtype=np.int32)

    def outs():
        return dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
                    n=torch.zeros(B, dtype=torch.int32, device="cuda"), m=torch.zeros((B // 2) * cap, dtype=torch.int32, device="cuda"),
                    nm=torch.zeros(B // 2, dtype=torch.int32, device="cuda"), st=torch.zeros(B // 2 * 3, dtype=torch.int32, device="cuda"))

    def run(img, o, async_):
        f = e.extract_match_batch_device_async if async_ else e.extract_match_batch_device
        f(img, B, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h), o["m"], o["nm"], o["st"], 100, 0.9, True, cap)
    # expected results of the rich batch from the oracle (pairs 0 and B/2 - 1 and one in the second half)
    exp = {}
    for p_ in (0, B // 4, B // 2 - 1):
        a, b = oe(rich[2 * p_]), oe(rich[2 * p_ + 1])
        exp[p_] = oracle.match_init(a[1], a[2], b[1], b[2], (0, w, 0, h), 100, 0.9, True)
        assert (a[1]["octave"] == 0).sum() > 600

    def check(o):
        nm, st = o["nm"].cpu().numpy(), o["st"].cpu().numpy().reshape(-1, 3)
        mm = o["m"].cpu().numpy().reshape(B // 2, cap)
        for p_, (onm, om12, ost) in exp.items():
            assert nm[p_] == onm and np.array_equal(mm[p_, :len(om12)], om12) and st[p_].tolist() == ost.tolist(), p_
    o = outs()
    run(d_rich, o, False)   # a fresh context issues the wide kernels with the batch
    check(o)
    scratch = outs()
    for _ in range(10):     # ten batches that need nothing: the context stops issuing them
        run(d_flat, scratch, False)
    o1, o2 = outs(), outs()
    run(d_rich, o1, True)   # two rich batches in flight without the wide kernels: both completed at their waits
    run(d_rich, o2, True)
    e.wait()
    check(o1)
    check(o2)
    o3 = outs()
    run(d_rich, o3, False)  # and with the batch again
    check(o3)
    e.close()


def test_cpp_shim_equals_oracle(orbx, oracle, tmp_path):
    """The reference's demo call sequence through the C++ drop-in classes (include/orbx_shim.hpp)."""
    import subprocess
    from orb_slam_tracking_amd import synth
    from test_host import build_shim_demo
    exe = build_shim_demo(orbx, str(tmp_path))
    a, b = synth.synth_pair(640, 480, 77)
    fa, fb = tmp_path / "a.raw", tmp_path / "b.raw"
    fa.write_bytes(a.tobytes())
    fb.write_bytes(b.tobytes())
    p = subprocess.run([exe, "640", "480", str(fa), str(fb), "1000", "20", "7"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True)
    assert p.returncode == 0, p.stdout
    assert "Sum of features = 1000" in p.stdout and "invalidMatchByDistance:" in p.stdout  # cpp:549, ORBmatcher.cpp:144-147
    res = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][0].split()[1:]
    oe = oracle.Extractor(*CANON)
    _, ka, da = oe(a)
    _, kb, db = oe(b)
    nm, m12, _ = oracle.match_init(ka, da, kb, db, (0, 640, 0, 480), 100, 0.9, True)

    def fnv(buf):
        h = 1469598103934665603
        for byte in buf:
            h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    assert [int(v) for v in res[:3]] == [len(ka), len(kb), nm]
    assert int(res[3]) == fnv(ka.tobytes()) and int(res[4]) == fnv(da.tobytes()) and int(res[5]) == fnv(m12.astype(np.int32).tobytes())
    # the same with the Frame constructor's undistortion and image bounds (Settings.yaml camera)
    p = subprocess.run([exe, "640", "480", str(fa), str(fb), "1000", "20", "7", "1"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    res = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][0].split()[1:]
    cam = oracle.SETTINGS_CAMERA
    nm, m12, _ = oracle.match_init(oracle.undistort_keypoints(ka, cam), da, oracle.undistort_keypoints(kb, cam), db,
                                   oracle.image_bounds(cam, 640, 480), 100, 0.9, True)
    assert [int(v) for v in res[:3]] == [len(ka), len(kb), nm] and int(res[5]) == fnv(m12.astype(np.int32).tobytes())


def test_undistort_and_bounds(orbx, ext640, oracle, images, golden):
    """Frame::UndistortKeyPoints / ComputeImageBounds (Frame.cpp:101-161) on the device, bit-exact
    against the oracle, with the camera of the reference's Settings.yaml and with tangential terms."""
    rng = np.random.default_rng(5)
    pts = np.zeros(5000, orbx.KEYPOINT_DTYPE)
    pts["x"] = rng.uniform(-50, 700, len(pts)).astype(np.float32)
    pts["y"] = rng.uniform(-50, 530, len(pts)).astype(np.float32)
    pts["octave"] = rng.integers(0, 8, len(pts))
    pts["angle"] = rng.uniform(0, 360, len(pts)).astype(np.float32)
    k = golden["as_shipped/init0/kps"]
    cams = [oracle.SETTINGS_CAMERA, oracle.SETTINGS_CAMERA[:6] + (0.0011, -0.0007), (500.0, 510.0, 320.0, 240.0, 0.21, -0.4, 0.0, 0.0),
            oracle.SETTINGS_CAMERA[:4] + (0.0, 0.3, 0.01, 0.01),      # k1 == 0: copy, whatever the other coefficients
            (100.0, 100.0, 320.0, 240.0, -0.9, 0.0, 0.0, 0.0)]        # strong: icdist < 0 branch for far points
    for cam in cams:
        for src in (k, pts, pts[:1], pts[:0]):
            assert ext640.undistort_keypoints(src, cam).tobytes() == oracle.undistort_keypoints(src, cam).tobytes()
        for (w, h) in ((640, 480), (1920, 1080), (752, 480)):
            assert ext640.image_bounds(cam, w, h) == oracle.image_bounds(cam, w, h)


def test_frames_with_distortion_match(orbx, oracle, images):
    """Config C1 as the demo runs it: Frame(im, K, distCoef) -> mvKeysUn + undistorted bounds -> SearchForInitialization
    (grid over the undistorted bounds, Frame.cpp:44-46, 70-99), host API and device-resident API."""
    import torch
    cam = oracle.SETTINGS_CAMERA
    K = np.array([[cam[0], 0, cam[2]], [0, cam[1], cam[3]], [0, 0, 1]], np.float32)
    dist = np.array(cam[4:], np.float32)
    a, b = images["init0"], images["init1"]
    h, w = a.shape
    e = orbx.ORBextractor(*SHIPPED, max_width=w, max_height=h, max_batch=2)
    oe = oracle.Extractor(*SHIPPED)
    fa, fb = orbx.Frame(a, 0.0, e, K, dist), orbx.Frame(b, 1.0, e, K, dist)
    oa, ob = oe(a), oe(b)
    ua, ub = oracle.undistort_keypoints(oa[1], cam), oracle.undistort_keypoints(ob[1], cam)
    ob_bounds = oracle.image_bounds(cam, w, h)
    assert fa.bounds == ob_bounds and fb.bounds == ob_bounds
    assert fa.mvKeysUn.tobytes() == ua.tobytes() and fb.mvKeysUn.tobytes() == ub.tobytes()
    assert fa.mvKeys.tobytes() == oa[1].tobytes()
    m = orbx.ORBmatcher(0.9, True)
    nm, m12 = m.SearchForInitialization(fa, fb, 100)
    onm, om12, ost = oracle.match_init(ua, oa[2], ub, ob[2], ob_bounds, 100, 0.9, True)
    assert nm == onm and np.array_equal(m12, om12) and list(m.last_stats) == ost.tolist()
    assert nm >= 100  # demo_initialization.cpp:110
    # device-resident chain: extract -> undistort (in place) -> match
    cap = 2000
    d_img = torch.from_numpy(np.stack([a, b])).cuda()
    d_k = torch.zeros(2 * cap * 28, dtype=torch.uint8, device="cuda")
    d_d = torch.zeros(2 * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(2, dtype=torch.int32, device="cuda")
    e.extract_batch_device(d_img, 2, w, h, w, w * h, d_k, d_d, d_n, cap)
    e.undistort_batch_device(2, d_k, d_n, cam, d_k, cap)
    kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(2, cap)
    assert kk[0, :len(ua)].tobytes() == ua.tobytes() and kk[1, :len(ub)].tobytes() == ub.tobytes()
    d_m = torch.zeros(cap, dtype=torch.int32, device="cuda")
    d_nm = torch.zeros(1, dtype=torch.int32, device="cuda")
    e.match_pairs_device(np.array([0]), np.array([1]), d_k, d_d, d_n, ob_bounds, d_m, d_nm, None, 100, 0.9, True, cap)
    assert int(d_nm.item()) == onm and np.array_equal(d_m.cpu().numpy()[:len(om12)], om12)
    e.close()


def test_to_gray_and_colour_chain(orbx, ext640, oracle):
    """Converter::toGray on the device (host and device-resident API), then colour frames -> gray ->
    extraction equal to the oracle's chain."""
    import torch
    rng = np.random.default_rng(12)
    for (h, w) in ((480, 752), (33, 641), (3, 5), (1, 1), (17, 1024)):
        im = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for rgb in (False, True):
            assert np.array_equal(ext640.to_gray(im, rgb), oracle.to_gray(im, rgb))
        assert np.array_equal(ext640.to_gray(im[..., 1].copy()), im[..., 1])
        assert np.array_equal(ext640.to_gray(im[:, :, :1].copy()), im[..., 0])
        padded = np.zeros((h, w + 3, 3), np.uint8)   # a row stride that is not a multiple of 4 for odd w
        padded[:, :w] = im
        assert np.array_equal(ext640.to_gray(padded[:, :w], True), oracle.to_gray(im, True))
    with pytest.raises(orbx.OrbxError) as ei:
        ext640.to_gray(np.zeros((4, 4, 4), np.uint8))
    assert ei.value.code == orbx.E_BADARG
    with pytest.raises(orbx.OrbxError) as ei:
        ext640.to_gray(np.zeros((0, 0, 3), np.uint8))
    assert ei.value.code == orbx.E_EMPTY
    # device-resident: B colour frames (RGB order) built from synthetic gray scenes with a colour cast
    from orb_slam_tracking_amd import synth
    B, W, H, cap = 4, 640, 480, 1000
    g = synth.synth_frames(B, W, H, 4100).astype(np.int32)
    col = np.stack([np.clip(g + 9, 0, 255), g, np.clip(g * 3 // 4 + 20, 0, 255)], axis=-1).astype(np.uint8)
    d_col = torch.from_numpy(col).cuda()
    for (gs, off) in ((W, 0), (W + 5, 1)):  # aligned and unaligned gray destinations
        d_buf = torch.zeros(B * gs * H + 8, dtype=torch.uint8, device="cuda")
        d_gray = d_buf[off:]
        ext640.to_gray_batch_device(d_col, B, W, H, W * 3, W * H * 3, 3, True, d_gray, gs, gs * H)
        got = d_gray[:B * gs * H].cpu().numpy().reshape(B, H, gs)[:, :, :W]
        want = np.stack([oracle.to_gray(c, True) for c in col])
        assert np.array_equal(got, want)
    d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
    d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    e = orbx.ORBextractor(*CANON, max_width=W, max_height=H, max_batch=B)
    e.extract_batch_device(d_gray, B, W, H, gs, gs * H, d_k, d_d, d_n, cap)
    n = d_n.cpu().numpy()
    kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
    dd = d_d.cpu().numpy().reshape(B, cap, 32)
    oe = oracle.Extractor(*CANON)
    for f in range(B):
        _, ko, do = oe(want[f])
        assert n[f] == len(ko)
        _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
    e.close()


@pytest.mark.parametrize("nosplit", [False, True])
def test_banded_pyramid_large_batch(orbx, oracle, nosplit, request):
    """Batches of >= 32 frames per stream build the pyramid with k_pyramid_bands (one launch, row bands with halos; three
    bands per 640x480 frame from 86 frames per stream, more and thinner ones below: ceil(256 / frames)): pyramid levels and extraction results equal the oracle, for an
    even-sized and an odd-height frame size."""
    import torch
    from orb_slam_tracking_amd import synth
    orbx.debug_set("no_split", 1 if nosplit else None)
    request.addfinalizer(lambda: orbx.debug_set("no_split", None))
    cap = 1000
    for (w, h, B) in ((640, 480, 72), (324, 243, 66), (322, 243, 40)):  # the last: rows not 4-byte aligned (unaligned 8-byte taps)
        frames = synth.synth_frames(B, w, h, 5200)
        oe = oracle.Extractor(*CANON)
        e = orbx.ORBextractor(*CANON, max_width=w, max_height=h, max_batch=B)
        d_img = torch.from_numpy(frames).cuda()
        d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
        d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
        d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
        e.extract_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, cap)
        n = d_n.cpu().numpy()
        kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
        dd = d_d.cpu().numpy().reshape(B, cap, 32)
        for f in list(range(0, B, 5)) + [B // 2 - 1, B // 2, B - 1]:
            _, ko, do = oe(frames[f])
            assert n[f] == len(ko)
            _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
            if f in (0, B // 2, B - 1):
                for l in range(1, 8):
                    assert np.array_equal(e.image_pyramid(l, f), oe.level_image(l)), (w, h, f, l)
        e.close()


def _same_f32(a, b):
    """bitwise equal except that any NaN equals any NaN (x86 and the GPU produce different default-NaN signs)"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    nan = np.isnan(a)
    return np.array_equal(nan, np.isnan(b)) and a[~nan].tobytes() == b[~nan].tobytes()


def test_initializer_scoring_loops(orbx, ext640, oracle):
    """CheckHomography / CheckFundamental for stacks of RANSAC hypotheses on the device equal the
    oracle bit for bit (scores are sequential f32 sums), incl. degenerate models, N not a multiple of 64 and N = 0."""
    for (seed, n, nm) in ((1, 400, 200), (2, 63, 5), (3, 129, 9)):
        k1, k2, m12, H21, H12, F21 = oracle.scoring_case(seed, n=n, n_models=nm)
        H21[-1] = 0; H12[-1] = 0; F21[-1] = 0           # degenerate: 1/0 and 0/0 paths
        for sigma in (1.0, 2.5):
            with np.errstate(all="ignore"):
                sc, inl, best = ext640.check_homography(H21, H12, k1, k2, m12, sigma)
                ref = [oracle.check_homography(H21[i], H12[i], k1, k2, m12, sigma) for i in range(nm)]
                assert _same_f32(sc, [r[0] for r in ref])
                assert all(np.array_equal(inl[i], ref[i][1]) for i in range(nm))
                esc, ebest = np.float32(0), -1
                for i in range(nm):
                    if ref[i][0] > esc:
                        esc, ebest = ref[i][0], i
                assert best == ebest == 0
                sc, inl, best = ext640.check_fundamental(F21, k1, k2, m12, sigma)
                ref = [oracle.check_fundamental(F21[i], k1, k2, m12, sigma) for i in range(nm)]
                assert _same_f32(sc, [r[0] for r in ref])
                assert all(np.array_equal(inl[i], ref[i][1]) for i in range(nm))
    none = np.full(len(k1), -1, np.int32)
    sc, inl, best = ext640.check_homography(H21, H12, k1, k2, none, 1.0)
    assert (sc == 0).all() and inl.shape == (len(H21), 0) and best == -1


def test_random_geometries_batched(orbx, oracle):
    """Seeded random frame sizes through the batched device path (banded pyramid from 32 frames per stream, geometry-sized
    FAST tiles, LDS quadtree): every level image and the extraction of sampled frames equal the oracle."""
    import torch
    from orb_slam_tracking_amd import synth
    rng = np.random.default_rng(77)
    cap, compared = 500, 0
    for trial in range(7):
        w = int(rng.integers(60, 120)) * 4           # 4-aligned rows: the dword / banded paths
        h = int(rng.integers(200, 400))
        B = int(rng.choice([33, 40, 64, 70]))
        params = (500, float(rng.choice([1.2, 1.1, 1.3])), int(rng.choice([4, 6, 8])), 20, 7)
        if trial == 6:  # a level wider than 2048 pixels: the batch takes the per-level pyramid launches, not k_pyramid_bands
            w, h, B, params = 2560, 160, 33, (500, 1.2, 3, 20, 7)
        oe = oracle.Extractor(*params)
        try:
            e = orbx.ORBextractor(*params, max_width=w, max_height=h, max_batch=B)
        except orbx.OrbxError as err:
            assert err.code == orbx.E_TOOSMALL, (w, h, params)   # a level narrower than one FAST cell: UB upstream
            continue
        frames = synth.synth_frames(B, w, h, 9000 + trial)
        d_img = torch.from_numpy(frames).cuda()
        d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
        d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
        d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
        try:
            e.extract_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, cap)
        except orbx.OrbxError as err:
            assert err.code == orbx.E_TOOSMALL, (w, h, params)   # a level narrower than one FAST cell: UB upstream
            e.close()
            continue
        n = d_n.cpu().numpy()
        kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
        dd = d_d.cpu().numpy().reshape(B, cap, 32)
        for f in (0, B // 2 - 1, B // 2, B - 1):
            _, ko, do = oe(frames[f])
            assert n[f] == len(ko), (w, h, B, params, f)
            _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
            for l in range(1, params[2]):
                assert np.array_equal(e.image_pyramid(l, f), oe.level_image(l)), (w, h, B, params, f, l)
        compared += 1
        e.close()
    assert compared >= 5


@pytest.mark.parametrize("sf,nlev", [(2.0, 3), (1.5, 4), (1.05, 8), (1.01, 3)])
def test_banded_pyramid_scale_factors(orbx, oracle, sf, nlev):
    """k_pyramid_bands at the ends of its range: scale 2 (the taps of 4 outputs fill all 8 loaded bytes), a scale close to 1,
    two scales close to 1; a width that is not a multiple of 4, so the last group of a row is partial."""
    import torch
    from orb_slam_tracking_amd import synth
    w, h, B, cap = 486, 363, 33, 300
    params = (300, sf, nlev, 20, 7)
    oe = oracle.Extractor(*params)
    e = orbx.ORBextractor(*params, max_width=w, max_height=h, max_batch=B)
    frames = synth.synth_frames(B, w, h, 4100)
    d_img = torch.from_numpy(frames).cuda()
    d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
    d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    e.extract_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, cap)
    n = d_n.cpu().numpy()
    kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
    dd = d_d.cpu().numpy().reshape(B, cap, 32)
    for f in (0, 16, B - 1):
        _, ko, do = oe(frames[f], cap=cap)
        assert n[f] == len(ko)
        _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
        for l in range(1, nlev):
            assert np.array_equal(e.image_pyramid(l, f), oe.level_image(l)), (sf, f, l)
    e.close()


def test_banded_pyramid_frames_that_overlap(orbx, oracle):
    """A frame stride smaller than a frame (sliding windows over one tall image): nothing is known to follow any frame's last row,
    so every frame takes level 1 of k_pyramid_bands through the clamped single-dword loads (PyrBands.safeFrom = 0)."""
    import torch
    from orb_slam_tracking_amd import synth
    w, h, B, cap = 400, 300, 34, 300
    params = (300, 1.2, 6, 20, 7)
    step = (h // 2) * w                                     # frame f starts half a frame after frame f - 1
    tall = synth.synth_frames(1, w, (B - 1) * (h // 2) + h, 4300)[0]
    d_img = torch.from_numpy(tall).cuda()
    oe = oracle.Extractor(*params)
    e = orbx.ORBextractor(*params, max_width=w, max_height=h, max_batch=B)
    d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
    d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    e.extract_batch_device(d_img, B, w, h, w, step, d_k, d_d, d_n, cap)
    n = d_n.cpu().numpy()
    kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
    dd = d_d.cpu().numpy().reshape(B, cap, 32)
    for f in (0, 1, 17, B - 1):
        frame = tall[f * (h // 2):f * (h // 2) + h]
        _, ko, do = oe(frame, cap=cap)
        assert n[f] == len(ko), f
        _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
    e.close()


def test_sincos_matches_libm(orbx, ext640, oracle):
    """The descriptor's cos / sin (f64 evaluation of the f32 angle, rounded to f32, cpp:173-174): the device's f64 sincos and
    the host libm agree after the rounding for every angle fastAtan2 can produce that is tried here -- all 360 * 2^7
    multiples of 1/128 degree, the neighbourhoods of the multiples of 45 degrees float by float, and 3 million random
    floats in [0, 360]."""
    rng = np.random.default_rng(2)
    parts = [np.arange(0, 360 * 128 + 1, dtype=np.float32) / np.float32(128)]
    for m in range(0, 361, 45):
        c = np.float32(m)
        lo = c
        vals = [c]
        up = c
        for _ in range(2000):
            lo = np.nextafter(lo, np.float32(-1))
            up = np.nextafter(up, np.float32(400))
            vals += [lo, up]
        parts.append(np.array([v for v in vals if 0 <= v <= 360], np.float32))
    parts.append(rng.uniform(0, 360, 3_000_000).astype(np.float32))
    a = np.concatenate(parts)
    c, s = ext640.debug_sincos(a)
    ec, es = oracle.sincos_deg_batch(a)
    bad = np.nonzero((c != ec) | (s != es))[0]
    assert len(bad) == 0, (len(bad), a[bad[:5]], c[bad[:5]], ec[bad[:5]], s[bad[:5]], es[bad[:5]])


def test_libm_float_variant(orbx, oracle):
    """orbx_set_libm_variant: ORBX_LIBM_FLOAT (the default since round 5) -- cos / sin of cpp:174 read as glibc's cosf / sinf
    (restated on the device) and the constructor's pow as powf (cpp:536) -- and ORBX_LIBM_DOUBLE, each against the oracle with the
    same reading: the (cos, sin) pairs for all multiples of
    1/128 degree, 3 million random angles and the known angles at which the two readings put a sample point on different pixels
    (bitwise; tools/sincos_exhaustive.py 1 sweeps every f32 angle); whole extractions; a scale factor whose quotas the reading
    changes (the context re-plans its buffers)."""
    from orb_slam_tracking_amd import synth
    rng = np.random.default_rng(3)
    moved = np.array([0x40506FD5, 0x4062B34D, 0x40E8408C, 0x40E8408D, 0x40F35541, 0x40F35542, 0x410B3FE6, 0x4126BE39, 0x4152F830,
                      0x416EF42B, 0x416EF42C, 0x41790BB9], np.uint32).view(np.float32)
    a = np.concatenate([np.arange(0, 360 * 128 + 1, dtype=np.float32) / np.float32(128), moved,
                        np.float32(2.0) ** np.arange(-30, 0, dtype=np.float32),  # tiny angles: sinf returns its argument below 2^-12
                        rng.uniform(0, 360, 3_000_000).astype(np.float32), rng.uniform(0, 50, 500_000).astype(np.float32)])
    exotic = float(np.uint32(0x3F817EE4).view(np.float32))  # tests/test_oracle.py EXOTIC_SCALE
    frames = [synth.synth(640, 480, 31), synth.synth(640, 480, 32)]
    e = orbx.ORBextractor(*CANON, max_width=640, max_height=480, max_batch=2)
    e2 = orbx.ORBextractor(500, exotic, 9, 20, 7, max_width=320, max_height=240, max_batch=2)
    try:
        oracle.set_libm_variant(0)
        e.set_libm_variant(0)
        e2.set_libm_variant(0)
        c0, s0 = e.debug_sincos(a)
        ec0, es0 = oracle.sincos_deg_batch(a)
        assert np.array_equal(c0.view(np.uint32), ec0.view(np.uint32)) and np.array_equal(s0.view(np.uint32), es0.view(np.uint32))
        oracle.set_libm_variant(1)
        e.set_libm_variant(1)
        c, s = e.debug_sincos(a)
        ec, es = oracle.sincos_deg_batch(a)
        bad = np.nonzero((c.view(np.uint32) != ec.view(np.uint32)) | (s.view(np.uint32) != es.view(np.uint32)))[0]
        assert len(bad) == 0, (len(bad), a[bad[:5]], c[bad[:5]], ec[bad[:5]], s[bad[:5]], es[bad[:5]])
        assert ((c != c0) | (s != s0)).sum() > 1000  # (the readings differ for ~0.13 % of the angles)
        oe = oracle.Extractor(*CANON)
        for fr in frames:
            r, k, d = e(fr)
            orr, ok, od = oe(fr)
            assert r == orr
            _same(k, d, ok, od)
        q0 = np.array(e2.GetNumFeaturesPerLevel())
        e2.set_libm_variant(1)
        oe2 = oracle.Extractor(500, exotic, 9, 20, 7)
        q1 = np.array(e2.GetNumFeaturesPerLevel())
        assert np.array_equal(q1, oe2.tables()["quota"]) and not np.array_equal(q0, q1)
        fr = synth.synth(320, 240, 33)
        r, k, d = e2(fr)
        orr, ok, od = oe2(fr)
        assert r == orr
        _same(k, d, ok, od)
        oracle.set_libm_variant(0)
        e2.set_libm_variant(0)
        assert np.array_equal(np.array(e2.GetNumFeaturesPerLevel()), q0)
        r, k, d = e2(fr)
        oe0 = oracle.Extractor(500, exotic, 9, 20, 7)
        orr, ok, od = oe0(fr)
        _same(k, d, ok, od)
        # lanes created AFTER the switch take the parent's reading (their own constructor ran with the default one): a stream-ordered
        # batch on two lanes with the DOUBLE quotas
        import torch
        e2.set_pipeline_depth(2)
        B2, cap2 = 2, 500
        fr2 = synth.synth_frames(B2, 320, 240, 41)
        d_img = torch.from_numpy(fr2).cuda()
        outs2 = [dict(k=torch.zeros(B2 * cap2 * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(B2 * cap2 * 32, dtype=torch.uint8, device="cuda"),
                      n=torch.zeros(B2, dtype=torch.int32, device="cuda"), m=torch.zeros(cap2, dtype=torch.int32, device="cuda"),
                      nm=torch.zeros(1, dtype=torch.int32, device="cuda")) for _ in range(2)]
        for o in outs2:
            e2.extract_match_batch_device_async(d_img, B2, 320, 240, 320, 320 * 240, o["k"], o["d"], o["n"], np.array([0], np.int32), np.array([1], np.int32),
                                                (0, 320, 0, 240), o["m"], o["nm"], None, 100, 0.9, True, cap2)
        e2.wait()
        for o in outs2:
            n = o["n"].cpu().numpy()
            kk = o["k"].cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B2, cap2)
            dd = o["d"].cpu().numpy().reshape(B2, cap2, 32)
            for f in range(B2):
                _, ko, do = oe0(fr2[f])
                assert n[f] == len(ko)
                _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
        e2.set_pipeline_depth(0)
        with pytest.raises(Exception):
            e.set_libm_variant(2)
    finally:
        oracle.set_libm_variant(oracle.LIBM_DEFAULT)
        e.close()
        e2.close()


def test_check_rt(orbx, ext640, oracle):
    """Initializer::CheckRT (Initializer.cpp:569-713) on the device for the four (R, t) candidates of a decomposition at once,
    against the CPU restatement: nGood and vbTriGood equal; vP3D and the parallax within 1e-4 relative (floating point:
    both run the same fixed Jacobi SVD in uncontracted f64, so they normally agree bitwise -- counted, not required)."""
    exact = total = 0
    for seed in range(5):
        K, R, t, k1, k2, m12, cands = oracle.two_view_case(100 + seed, n=200 + 150 * seed, outliers=0.25, noise=0.6)
        first = np.nonzero(m12 >= 0)[0]
        rng = np.random.default_rng(seed)
        inl = (rng.random(len(first)) < 0.8).astype(np.uint8)
        Rs = np.stack([c[0] for c in cands]).astype(np.float32)
        ts = np.stack([c[1] for c in cands]).astype(np.float32)
        ng, good, p3d, par = ext640.check_rt(Rs, ts, K, k1, k2, m12, inl, 4.0)
        for m in range(len(cands)):
            on, ogood, op3d, opar = oracle.check_rt(Rs[m], ts[m], K, k1, k2, m12, inl, 4.0)
            assert ng[m] == on and np.array_equal(good[m], ogood), (seed, m, ng[m], on)
            assert np.allclose(p3d[m], op3d, rtol=1e-4, atol=1e-6) and abs(float(par[m]) - float(opar)) <= 1e-4 * max(1.0, float(opar))
            exact += int(p3d[m].tobytes() == op3d.tobytes() and par[m].tobytes() == np.float32(opar).tobytes())
            total += 1
    assert exact >= total - 2, (exact, total)  # bitwise equality is the rule
    # edge cases: no hypothesis, no inlier, no match
    K, R, t, k1, k2, m12, cands = oracle.two_view_case(7, n=120)
    first = np.nonzero(m12 >= 0)[0]
    ng, good, p3d, par = ext640.check_rt(np.zeros((0, 3, 3), np.float32), np.zeros((0, 3), np.float32), K, k1, k2, m12, np.ones(len(first), np.uint8))
    assert len(ng) == 0
    ng, good, p3d, par = ext640.check_rt(R, t, K, k1, k2, m12, np.zeros(len(first), np.uint8))
    assert ng[0] == 0 and not good.any() and not p3d.any() and par[0] == 0
    ng, good, p3d, par = ext640.check_rt(R, t, K, k1, k2, np.full(len(k1), -1, np.int32), np.zeros(0, np.uint8))
    assert ng[0] == 0 and par[0] == 0


def test_opencv_variant_constants(orbx, oracle):
    """orbx_set_opencv_variant: the rounded Gaussian taps [18,34,49,55,..] (sum 257, saturating) and the 15-bit BGR2GRAY
    coefficients in the kernels == the oracle with the same variant, on frames with saturated regions; the default variant
    afterwards == the default oracle again."""
    import torch
    from orb_slam_tracking_amd import synth
    w, h, B = 640, 480, 4
    frames = synth.synth_frames(B, w, h, 3300)
    frames[0, 100:260, 200:420] = 255   # bright plateau next to texture: the saturating case of the sum-257 taps
    frames[0, 120:250:9, 210:410:7] = 0
    e = orbx.ORBextractor(*CANON, max_width=w, max_height=h, max_batch=B)
    oe = oracle.Extractor(*CANON)
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (97, 203, 3), dtype=np.uint8)
    try:
        for gv, cv_ in ((1, 1), (1, 0), (0, 1), (0, 0)):
            e.set_opencv_variant(gv, cv_)
            oracle.set_opencv_variant(gv, cv_)
            res = e.extract_batch(frames)
            for f in range(B):
                ro, ko, do = oe(frames[f])
                assert res[f][0] == ro
                _same(res[f][1], res[f][2], ko, do)
            for b in (True, False):
                assert np.array_equal(e.to_gray(rgb, b), oracle.to_gray(rgb, b))
            d_src = torch.from_numpy(rgb).cuda()
            d_g = torch.zeros((97, 203), dtype=torch.uint8, device="cuda")
            e.to_gray_batch_device(d_src, 1, 203, 97, 203 * 3, 0, 3, True, d_g, 203, 0)
            assert np.array_equal(d_g.cpu().numpy(), oracle.to_gray(rgb, True))
        with pytest.raises(orbx.OrbxError):
            e.set_opencv_variant(2, 0)
    finally:
        oracle.set_opencv_variant(0, 0)
        e.close()


@pytest.mark.parametrize("params", [(500, 1.2, 8, 20, 7), (300, 2.0, 4, 20, 7), (300, 1.05, 6, 20, 7), (400, 1.37, 5, 20, 7)])
def test_tiled_pyramid_small_batches(orbx, oracle, params):
    """Batches of up to 8 frames build the pyramid with k_pyramid_tiles (one launch, a workgroup per tile of a frame, the level chain
    through LDS, every tile with a halo of the pixels its higher levels read): every level image equals the oracle's cv::resize chain
    (Features/ORBextractor.cpp:1660-1713) for frame sizes with ragged tile grids, a scale of 2 and scales close to 1, a strided
    caller image; 9 frames take k_pyramid_bands where the geometry allows it (round 6; one launch per level before) and, under the
    knob no_bands, one launch per level."""
    import torch
    from orb_slam_tracking_amd import synth
    nlev, cap = params[2], params[0]
    for (w, h, B, pad) in ((640, 480, 1, 0), (641, 479, 3, 0), (322, 243, 8, 0), (173, 131, 2, 0), (752, 480, 5, 16), (1280, 720, 1, 0), (640, 480, 9, 0),
                           (640, 480, 9, -1)):
        if min(w, h) / params[1] ** (nlev - 1) < 70:  # the smallest level must hold a FAST cell grid (ORBX_E_TOOSMALL otherwise)
            continue
        level_by_level = pad < 0  # (the last case: the per-level launches, which no default batch size takes any more)
        pad = max(pad, 0)
        orbx.debug_set("no_bands", 1 if level_by_level else None)
        frames = synth.synth_frames(B, w, h, 6100 + w)
        stride = w + pad
        buf = np.zeros((B, h, stride), np.uint8)
        buf[:, :, :w] = frames
        oe = oracle.Extractor(*params)
        e = orbx.ORBextractor(*params, max_width=w, max_height=h, max_batch=B)
        d_img = torch.from_numpy(buf).cuda()
        d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
        d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
        d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
        e.extract_batch_device(d_img, B, w, h, stride, stride * h, d_k, d_d, d_n, cap)
        info = e.debug_last_launch()
        orbx.debug_set("no_bands", None)
        assert (info["pyramid_banded"] == 2) if B <= 8 else (info["pyramid_banded"] == 0 if level_by_level else info["pyramid_banded"] in (0, 1)), (w, h, B, info)
        if B > 8 and not level_by_level and params[1] == 1.2:
            assert info["pyramid_banded"] == 1, info  # (scale 1.2 on aligned 640x480 frames: nothing keeps the banded kernel away)
        # launches of up to 256 (frame, level) units: k_describe_patch indexes the selection's staging lists itself (no k_sel_compact)
        assert info["staged_lists"] == (1 if B * nlev <= 256 else 0), (w, h, B, info)
        n = d_n.cpu().numpy()
        kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
        dd = d_d.cpu().numpy().reshape(B, cap, 32)
        for f in sorted({0, B // 2, B - 1}):
            _, ko, do = oe(frames[f], cap=cap)
            assert n[f] == len(ko), (w, h, f)
            _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
            for l in range(1, nlev):
                assert np.array_equal(e.image_pyramid(l, f), oe.level_image(l)), (params, w, h, f, l)
        e.close()


def test_banded_pyramid_column_strips(orbx, oracle):
    """k_pyramid_bands with column strips (round 5: few large frames; a workgroup = a row band x a column strip, each strip with the
    groups its share of the next level reads): every pyramid level of every frame byte for byte against the oracle, for 2 .. 8
    strips forced through the diagnostic knobs on frame sizes whose levels end in partial groups, at two scale factors; then the
    library's own choice for the four-frame halves of a 3840x2160 batch (24 bands x 8 strips), whole extraction against the oracle."""
    import torch
    from orb_slam_tracking_amd import synth
    try:
        orbx.debug_set("bands_min_frames", 1)
        for (w, h, B, params, K, S) in ((1280, 720, 3, (1500, 1.2, 8, 20, 7), 8, 2), (1284, 723, 2, (1500, 1.2, 8, 20, 7), 6, 4),
                                        (2044, 600, 2, (1500, 1.2, 6, 20, 7), 4, 8), (1600, 900, 2, (1000, 1.5, 5, 20, 7), 5, 3),
                                        (900, 500, 2, (1000, 1.2, 8, 20, 7), 3, 8)):  # (the last: fewer strips than asked, a level is too narrow)
            orbx.debug_set("pyr_bands", K)
            orbx.debug_set("pyr_strips", S)
            frames = synth.synth_frames(B, w, h, 8100 + w)
            oe = oracle.Extractor(*params)
            e = orbx.ORBextractor(*params, max_width=w, max_height=h, max_batch=B)
            cap = params[0]
            d_img = torch.from_numpy(frames).cuda()
            d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
            d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
            d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
            e.extract_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, cap)
            info = e.debug_last_launch()
            assert info["pyramid_banded"] == 1 and info["pyramid_bands"] % K == 0 and (info["pyramid_bands"] == K * S or w == 900), info
            n = d_n.cpu().numpy()
            kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
            dd = d_d.cpu().numpy().reshape(B, cap, 32)
            for f in range(B):
                _, ko, do = oe(frames[f], cap=cap)
                for l in range(1, params[2]):
                    assert np.array_equal(e.image_pyramid(l, f), oe.level_image(l)), (w, h, K, S, f, l)
                assert n[f] == len(ko)
                _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
            e.close()
    finally:
        for k in ("bands_min_frames", "pyr_bands", "pyr_strips"):
            orbx.debug_set(k, None)
    w, h, B, cap = 3840, 2160, 8, 8000
    params = (8000, 1.2, 8, 20, 7)
    frames = synth.synth_frames(B, w, h, 7400)
    oe = oracle.Extractor(*params)
    e = orbx.ORBextractor(*params, max_width=w, max_height=h, max_batch=B)
    d_img = torch.from_numpy(frames).cuda()
    d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
    d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    e.extract_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, cap)  # synchronous: two halves of four frames
    info = e.debug_last_launch()
    assert info["pyramid_banded"] == 1 and info["pyramid_bands"] == 24 * 8 and info["split"] == 1 and info["frames_per_launch"] == 4, info
    n = d_n.cpu().numpy()
    kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
    dd = d_d.cpu().numpy().reshape(B, cap, 32)
    for f in (0, 3, 4, B - 1):
        _, ko, do = oe(frames[f], cap=cap)
        for l in (1, 4, 7):
            assert np.array_equal(e.image_pyramid(l, f), oe.level_image(l)), (f, l)
        assert n[f] == len(ko)
        _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
    e.close()


def test_banded_pyramid_wide_levels(orbx, oracle):
    """k_pyramid_bands on 3840x2160 frames (BASELINE config 5's size; levels 1 .. 7 up to 3200 pixels wide: 400 thread-columns of two
    4-pixel groups, one row per pass): eight frames on a lane take it in 16 bands x 4 column strips (the four-frame halves of a
    synchronous call in 24 x 8: test_banded_pyramid_column_strips); the extraction results (keypoints of all eight levels, descriptors) equal the oracle."""
    import torch
    from orb_slam_tracking_amd import synth
    w, h, B, cap = 3840, 2160, 8, 8000
    params = (8000, 1.2, 8, 20, 7)
    frames = synth.synth_frames(B, w, h, 7300)
    oe = oracle.Extractor(*params)
    e = orbx.ORBextractor(*params, max_width=w, max_height=h, max_batch=B)
    e.set_pipeline_depth(1)  # whole batches on one lane: a single launch of eight frames
    d_img = torch.from_numpy(frames).cuda()
    d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
    d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    d_m = torch.zeros((B // 2) * cap, dtype=torch.int32, device="cuda")
    d_nm = torch.zeros(B // 2, dtype=torch.int32, device="cuda")
    first = np.arange(0, B, 2, dtype=np.int32)
    e.extract_match_batch_device_async(d_img, B, w, h, w, w * h, d_k, d_d, d_n, first, first + 1, (0, w, 0, h), d_m, d_nm, None, 100, 0.9, True, cap)
    info = e.debug_last_launch()
    e.wait()
    assert info["pyramid_banded"] == 1 and info["frames_per_launch"] == B, info
    n = d_n.cpu().numpy()
    kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
    dd = d_d.cpu().numpy().reshape(B, cap, 32)
    for f in (0, B - 1):
        _, ko, do = oe(frames[f], cap=cap)
        assert n[f] == len(ko)
        _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
    e.close()
