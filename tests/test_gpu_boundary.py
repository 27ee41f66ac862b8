"""GPU tests of the drop-in boundary's host logic (run with -m gpu): a context that grows with the frames it is given
(ORBextractor::operator() takes any image, Features/ORBextractor.cpp:1531-1545), synchronous calls behind stream-ordered
batches, ordering against a caller's stream.  Everything is compared with the CPU oracle bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CANON = (1000, 1.2, 8, 20, 7)


def _same(kg, dg, ko, do):
    assert len(kg) == len(ko), (len(kg), len(ko))
    assert kg.tobytes() == ko.tobytes()
    assert np.array_equal(dg, do)


def test_context_grows_with_frame_and_batch(orbx, oracle):
    """One context created for 640x480 / one frame: 640x480, then 3840x2160, then 640x480 again, then a batch of 5 frames
    through the device API (a drop-in must reallocate, not return ORBX_E_BADARG)."""
    import torch
    from orb_slam_tracking_amd import synth
    e = orbx.ORBextractor(*CANON, max_width=640, max_height=480, max_batch=1)
    oe = oracle.Extractor(*CANON)
    small, big = synth.synth(640, 480, 11), synth.synth(3840, 2160, 4)
    for im in (small, big, small):
        r, k, d = e(im)
        ro, ko, do = oe(im)
        assert r == ro
        _same(k, d, ko, do)
        for l in (0, 3, 7):  # mvImagePyramid of the grown context
            assert np.array_equal(e.image_pyramid(l), oe.level_image(l))
    B, cap, w, h = 5, 1000, 752, 480
    frames = synth.synth_frames(B, w, h, 77)
    d_img = torch.from_numpy(frames).cuda()
    d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
    d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    e.extract_batch_device(d_img, B, w, h, w, w * h, d_k, d_d, d_n, cap)
    n = d_n.cpu().numpy()
    kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
    dd = d_d.cpu().numpy().reshape(B, cap, 32)
    for f in range(B):
        _, ko, do = oe(frames[f])
        assert n[f] == len(ko)
        _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
    # a frame the path cannot take (a level narrower than one FAST cell) is refused before any growth
    with pytest.raises(orbx.OrbxError) as ex:
        e(np.zeros((100, 2000), np.uint8))
    assert ex.value.code == orbx.E_TOOSMALL
    e.close()


def test_sync_host_call_behind_async_batch(orbx, oracle):
    """a stream-ordered batch is in flight; orbx_extract_batch then copies page-locked host frames into the
    context's staging buffer on the first stream and extracts a batch of the same size (>= 16 frames: two half-batch
    streams).  The second stream must wait for that copy."""
    import torch
    from orb_slam_tracking_amd import synth
    B, cap, w, h = 32, 1000, 640, 480
    e = orbx.ORBextractor(*CANON, max_width=w, max_height=h, max_batch=B)
    oe = oracle.Extractor(*CANON)
    dev_np = synth.synth_frames(B, w, h, seed0=900)
    dev_frames = torch.from_numpy(dev_np).cuda()
    host_frames = torch.empty((B, h, w), dtype=torch.uint8).pin_memory()
    first = np.arange(0, B, 2, dtype=np.int32)
    o = dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
             n=torch.zeros(B, dtype=torch.int32, device="cuda"), m=torch.zeros((B // 2) * cap, dtype=torch.int32, device="cuda"),
             nm=torch.zeros(B // 2, dtype=torch.int32, device="cuda"))
    for rep in range(3):
        frames = synth.synth_frames(B, w, h, seed0=1200 + 40 * rep)
        # prime the staging buffer with other content, so that a stale read cannot pass by accident
        e.extract_batch(synth.synth_frames(B, w, h, seed0=5000 + rep))
        host_frames.numpy()[:] = frames
        e.extract_match_batch_device_async(dev_frames, B, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h),
                                           o["m"], o["nm"], None, 100, 0.9, True, cap)
        res = e.extract_batch(host_frames.numpy())
        e.wait()
        for f in (0, 1, B // 2 - 1, B // 2, B // 2 + 1, B - 1):
            _, ko, do = oe(frames[f])
            _same(res[f][1], res[f][2], ko, do)
        # the pyramid reader drains both streams too
        e.extract_match_batch_device_async(dev_frames, B, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h),
                                           o["m"], o["nm"], None, 100, 0.9, True, cap)
        oe(dev_np[B - 1])
        assert np.array_equal(e.image_pyramid(5, frame=B - 1), oe.level_image(5))
    e.close()


def test_ordering_against_torch_stream(orbx, oracle):
    """The frames are produced by torch work queued on torch's current stream right before the call: the binding orders the
    context's private streams behind it (orbx_order_after).  A long torch kernel queue in front makes a missing ordering
    visible; order_before lets a torch consumer read the results of an _async batch without a host wait."""
    import torch
    from orb_slam_tracking_amd import synth
    B, cap, w, h = 16, 1000, 640, 480
    e = orbx.ORBextractor(*CANON, max_width=w, max_height=h, max_batch=B)
    oe = oracle.Extractor(*CANON)
    frames = synth.synth_frames(B, w, h, seed0=4100)
    src = torch.from_numpy(frames).cuda()
    d_img = torch.zeros_like(src)
    big = torch.zeros(64 << 20, dtype=torch.float32, device="cuda")
    d_k = torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda")
    d_d = torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    first = np.arange(0, B, 2, dtype=np.int32)
    d_m = torch.zeros((B // 2) * cap, dtype=torch.int32, device="cuda")
    d_nm = torch.zeros(B // 2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(20):
        big.add_(1.0)  # ~ms of queued work in front of the producer
    d_img.copy_(src)   # the producer of the frames, on torch's stream
    e.extract_match_batch_device_async(d_img, B, w, h, w, w * h, d_k, d_d, d_n, first, first + 1, (0, w, 0, h), d_m, d_nm, None,
                                       100, 0.9, True, cap)
    e.order_before(torch.cuda.current_stream().cuda_stream)
    n_copy = d_n.clone()  # torch consumer, no host-side wait in between
    torch.cuda.synchronize()
    e.wait()
    n = d_n.cpu().numpy()
    assert np.array_equal(n_copy.cpu().numpy(), n)
    kk = d_k.cpu().numpy().view(orbx.KEYPOINT_DTYPE).reshape(B, cap)
    dd = d_d.cpu().numpy().reshape(B, cap, 32)
    for f in range(B):
        _, ko, do = oe(frames[f])
        assert n[f] == len(ko)
        _same(kk[f, :n[f]], dd[f, :n[f]], ko, do)
    e.close()


@pytest.mark.parametrize("lanes", [0, 2])
def test_order_before_with_pairs_beyond_the_fast_matcher(orbx, oracle, lanes):
    """the event-only contract of orbx_order_before must also hold for batches whose pairs overflow k_match_jacobi
    (~1100 octave-0 keypoints per frame here) at a moment when the context has stopped issuing the wide matcher kernels with
    its batches: (a) a batch already in flight without them is completed by the first orbx_order_before; (b) every later batch
    carries them, so a torch consumer ordered by the event alone reads final nmatches / matches12 -- no host-side wait."""
    import torch
    from orb_slam_tracking_amd import synth
    params = (2000, 1.2, 2, 20, 7)
    B, cap, w, h = 8, 2000, 640, 480
    e = orbx.ORBextractor(*params, max_width=w, max_height=h, max_batch=B)
    if lanes:
        e.set_pipeline_depth(lanes)
    oe = oracle.Extractor(*params)
    rich = synth.synth_frames(B, w, h, seed0=4300)
    d_rich = torch.from_numpy(rich).cuda()
    first = np.arange(0, B, 2, dtype=np.int32)
    exp = []
    for p_ in range(B // 2):
        a, b = oe(rich[2 * p_]), oe(rich[2 * p_ + 1])
        assert (a[1]["octave"] == 0).sum() > 600
        exp.append(oracle.match_init(a[1], a[2], b[1], b[2], (0, w, 0, h), 100, 0.9, True))

    def outs():
        return dict(k=torch.zeros(B * cap * 28, dtype=torch.uint8, device="cuda"), d=torch.zeros(B * cap * 32, dtype=torch.uint8, device="cuda"),
                    n=torch.zeros(B, dtype=torch.int32, device="cuda"), m=torch.full(((B // 2) * cap,), -9, dtype=torch.int32, device="cuda"),
                    nm=torch.full((B // 2,), -9, dtype=torch.int32, device="cuda"))

    def issue(o):
        e.extract_match_batch_device_async(d_rich, B, w, h, w, w * h, o["k"], o["d"], o["n"], first, first + 1, (0, w, 0, h), o["m"], o["nm"],
                                           None, 100, 0.9, True, cap)

    def consume(o):  # a torch consumer behind the event only
        e.order_before(torch.cuda.current_stream().cuda_stream)
        nm, m = o["nm"].clone(), o["m"].clone()
        torch.cuda.current_stream().synchronize()  # (of torch's stream: the context's own waits are not called)
        nm, m = nm.cpu().numpy(), m.cpu().numpy().reshape(B // 2, cap)
        for p_, (onm, om12, _) in enumerate(exp):
            assert nm[p_] == onm and np.array_equal(m[p_, :len(om12)], om12), p_
    o1, o2, o3 = outs(), outs(), outs()
    issue(o1)                                        # a fresh context does not issue the wide kernels with its batches
    assert e.debug_last_launch()["wide_with_batch"] == 0
    consume(o1)                                      # (a) completed by the first orbx_order_before
    issue(o2)
    assert e.debug_last_launch()["wide_with_batch"] == 1
    issue(o3)
    consume(o2)                                      # (b) the wide kernels travelled with the batches
    consume(o3)
    e.wait()
    e.close()


def _fnv(b):
    h = 1469598103934665603
    for x in bytes(b):
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_cpp_opencv_shim_reference_frame_unchanged_call_sites(orbx, oracle, tmp_path, images):
    """The reference-shaped Frame class of tests/cpp/shim_opencv_frame.cpp runs the reference's call sites verbatim through
    the -DORBX_WITH_OPENCV branch of the shim (operator() with cv:: types; ORBmatcher(0.9, true) without setExtractor --
    the device context comes from Frame::mpORBextractor); one extractor object takes 3840x2160, then two 752x480 frames,
    then 3840x2160 again (the context grows).  The quiet flag is on: no stdout line of the reference may appear."""
    import subprocess
    from test_host import build_shim_opencv_frame
    from orb_slam_tracking_amd import synth
    exe = build_shim_opencv_frame(orbx, str(tmp_path))
    a, b = images["init0"], images["init1"]
    big = synth.synth(3840, 2160, 4)
    pa, pb, pg = tmp_path / "a.raw", tmp_path / "b.raw", tmp_path / "big.raw"
    pa.write_bytes(a.tobytes()); pb.write_bytes(b.tobytes()); pg.write_bytes(big.tobytes())
    h, w = a.shape
    p = subprocess.run([exe, str(w), str(h), str(pa), str(pb), "2000", "20", "7", "3840", "2160", str(pg)], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.strip().splitlines()
    assert [l.split()[0] for l in lines] == ["GROW", "RESULT", "GROW"], p.stdout  # and nothing else on stdout (quiet flag)
    oe = oracle.Extractor(2000, 1.2, 8, 20, 7)
    _, kg, dg = oe(big, cap=4096)
    lvl1_w = oe.level_size(1)[0]
    for gl in (lines[0], lines[2]):
        t = gl.split()
        assert int(t[1]) == len(kg) and int(t[2]) == _fnv(kg.tobytes()) and int(t[3]) == _fnv(dg.tobytes()) and int(t[4]) == lvl1_w
    _, ka, da = oe(a, cap=4096)
    _, kb, db = oe(b, cap=4096)
    nm, m12, _ = oracle.match_init(ka, da, kb, db, (0, w, 0, h), 100, 0.9, True)
    t = lines[1].split()
    assert (int(t[1]), int(t[2]), int(t[3])) == (len(ka), len(kb), nm) and nm > 20
    assert int(t[4]) == _fnv(ka.tobytes()) and int(t[5]) == _fnv(da.tobytes()) and int(t[6]) == _fnv(m12.astype(np.int32).tobytes())


def test_regression_r01_fault_tiny_units_on_global_scratch(orbx, oracle):
    """Round-1 fault (gpurun_out repro.log .. repro5.log: `oracle 300` / `oracle 10` -> memory access
    fault): a (frame, level) unit with a handful of candidates on the global-scratch selection kernel.  The register bitonic
    sort works on max(pow2ceil(n), workgroup size) padded keys, so the unit's key / node arrays must be allocated for at
    least the workgroup size however small n is (octScratchBytes pads both to >= 1024 entries).  Every candidate count
    around the padding steps, every selection kernel variant, quotas below and above the count."""
    e = orbx.ORBextractor(*CANON, max_width=640, max_height=480, max_batch=1)
    rng = np.random.default_rng(77)
    W, H = 608, 448
    for n in (1, 2, 3, 10, 15, 16, 17, 63, 64, 65, 255, 256, 257, 300, 434, 1023, 1024, 1025, 2047, 2048, 2049):
        pos = np.sort(rng.choice(W * H, size=n, replace=False))
        xyr = np.stack([pos % W, pos // W, rng.integers(1, 200, n)], 1).astype(np.float32)
        for N in (1, 10, 217, 300, 434):
            exp = oracle.distribute(xyr, 16, 16 + W, 16, 16 + H, N)[:N]
            for variant in (1, 0, 2, 3, 4):
                got = e.debug_distribute_device(xyr, 16, 16 + W, 16, 16 + H, N, variant)
                assert got.shape == exp.shape and np.array_equal(got, exp), (n, N, variant)
    e.close()


def test_pipeline_depth_needs_an_own_stream(orbx):
    """orbx_set_pipeline_depth on a context created on a caller's stream is refused (the lanes run on streams of their own)."""
    import torch
    st = torch.cuda.Stream()
    e = orbx.ORBextractor(500, 1.2, 4, 20, 7, max_width=320, max_height=240, max_batch=4, stream=int(st.cuda_stream))
    with pytest.raises(orbx.OrbxError) as ei:
        e.set_pipeline_depth(2)
    assert ei.value.code == orbx.E_BADARG
    e.set_pipeline_depth(0)
    e.close()
    e = orbx.ORBextractor(500, 1.2, 4, 20, 7, max_width=320, max_height=240, max_batch=4)
    e.set_pipeline_depth(2)
    e.set_pipeline_depth(0)
    e.close()
