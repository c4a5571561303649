"""Input-side row: JPEG decode parity.

Chain of evidence: Pillow (= what the reference runs: Image.open(path).convert('RGB')) -> golden fixtures
(tests/golden/jpeg_cases.npz, written by oracle/make_golden_jpeg.py) -> C oracle (oracle/jpeg_oracle.c) -> HIP
decoder through the C ABI (cald_jpeg_decode_batch).  Bit-exact at every link.
"""
import io

import numpy as np
import pytest

try:
    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = 1 << 24
except Exception:  # pragma: no cover
    Image = None


def _golden_cases(golden):
    g = golden("jpeg_cases")
    return [(g["file_%d" % i].tobytes(), g["rgb_%d" % i]) for i in range(int(g["n"]))]


def _encode(a, gray=False, **kw):
    im = Image.fromarray(a)
    if gray:
        im = im.convert("L")
    bio = io.BytesIO()
    im.save(bio, "JPEG", **kw)
    return bio.getvalue()


def _pil_decode(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def _sweep_of_flavours(seed, sizes):
    from cald_amd import synth
    rng = np.random.default_rng(seed)
    blobs = []
    for k, (H, W) in enumerate(sizes):
        a = np.ascontiguousarray(synth.synth_image(3000 + seed + k, max(H, 33), max(W, 33))[:H, :W])
        if k % 3 == 0:
            a = (a.astype(np.int32) + rng.integers(-20, 21, a.shape)).clip(0, 255).astype(np.uint8)
        kw = dict(quality=[35, 75, 90, 97][k % 4], subsampling=[2, 2, 1, 0][k % 4], optimize=bool(k & 1))
        if k % 5 == 4:
            kw["restart_marker_blocks"] = 1 + k % 7
        gray = k % 6 == 5
        if gray:
            kw.pop("subsampling")
        blobs.append(_encode(a, gray=gray, **kw))
    return blobs


# ------------------------------------------------------------------ CPU: oracle pinned to Pillow
def test_jpeg_oracle_matches_pillow_golden(oracle, golden):
    for data, ref in _golden_cases(golden):
        H, W, _ = oracle.jpeg_info(data)
        assert (H, W) == ref.shape[:2]
        assert np.array_equal(oracle.jpeg_decode(data), ref)


@pytest.mark.skipif(Image is None, reason="Pillow not importable")
def test_jpeg_oracle_matches_live_pillow(oracle):
    sizes = [(8, 8), (16, 16), (17, 23), (1, 1), (2, 3), (3, 5), (5, 40), (33, 31), (64, 48), (100, 75), (3, 200), (200, 3),
             (4, 4), (6, 5), (120, 160), (375, 500)]
    for data in _sweep_of_flavours(1, sizes):
        assert np.array_equal(oracle.jpeg_decode(data), _pil_decode(data))


@pytest.mark.skipif(Image is None, reason="Pillow not importable")
def test_jpeg_unsupported_flavours_are_rejected_not_guessed(oracle):
    from cald_amd import pool
    a = np.zeros((24, 24, 3), np.uint8)
    progressive = _encode(a, progressive=True)
    cmyk = io.BytesIO()
    Image.fromarray(a).convert("CMYK").save(cmyk, "JPEG")
    for blob in (progressive, cmyk.getvalue()):
        with pytest.raises(NotImplementedError):
            oracle.jpeg_info(blob)
        with pytest.raises(NotImplementedError):        # C ABI: CALD_ERR_UNSUPPORTED, host-only call
            pool.jpeg_info(blob)
    with pytest.raises(ValueError):
        oracle.jpeg_info(b"\x89PNG not a jpeg")
    with pytest.raises(RuntimeError):
        pool.jpeg_info(b"\x89PNG not a jpeg")
    assert pool.jpeg_info(_encode(a, quality=80)) == (24, 24, 3)
    assert pool.jpeg_info(_encode(a, gray=True)) == (24, 24, 1)


def test_device_pool_layout_is_aligned_and_ordered():
    from cald_amd.pool import DevicePool
    shapes = [(5, 7), (375, 500), (1, 1), (333, 500)]
    off, total = DevicePool._layout(shapes)
    assert off[0] == 0 and np.all(off % 256 == 0) and np.all(np.diff(off) > 0)
    for (H, W), o, nxt in zip(shapes, off, list(off[1:]) + [total]):
        assert o + H * W * 3 <= nxt


# ------------------------------------------------------------------ GPU: HIP decoder through the C ABI
@pytest.mark.gpu
def test_gpu_jpeg_decode_matches_golden_and_oracle(oracle, golden):
    import torch
    from cald_amd import pool
    cases = _golden_cases(golden)
    outs = pool.decode_jpeg_batch([d for d, _ in cases])
    torch.cuda.synchronize()
    for (data, ref), o in zip(cases, outs):
        got = o.cpu().numpy()
        assert np.array_equal(got, ref)
        assert np.array_equal(got, oracle.jpeg_decode(data))


@pytest.mark.gpu
@pytest.mark.skipif(Image is None, reason="Pillow not importable")
def test_gpu_jpeg_decode_ragged_batch_matches_pillow(oracle):
    from cald_amd import pool
    sizes = [(375, 500), (500, 333), (8, 8), (17, 23), (1, 1), (2, 3), (281, 500), (64, 48), (100, 75), (3, 200), (200, 3),
             (442, 500), (5, 40), (33, 31), (480, 640), (600, 800), (16, 16), (120, 160), (7, 9), (250, 250)]
    blobs = _sweep_of_flavours(2, sizes)
    outs = pool.decode_jpeg_batch(blobs)
    for data, o in zip(blobs, outs):
        got = o.cpu().numpy()
        assert np.array_equal(got, _pil_decode(data))
        assert np.array_equal(got, oracle.jpeg_decode(data))


@pytest.mark.gpu
@pytest.mark.skipif(Image is None, reason="Pillow not importable")
def test_gpu_jpeg_unsupported_raises_and_decodes_nothing_on_cpu():
    from cald_amd import pool
    a = np.zeros((24, 24, 3), np.uint8)
    with pytest.raises(NotImplementedError):
        pool.decode_jpeg_batch([_encode(a, quality=80), _encode(a, progressive=True)])


@pytest.mark.gpu
@pytest.mark.skipif(Image is None, reason="Pillow not importable")
def test_gpu_device_pool_sweep_equals_loader_sweep():
    """JPEG files -> DevicePool (GPU decode, HBM resident) -> get_uncertainty == the same sweep fed by a
    reference-style loader of PIL-decoded float CHW tensors (to_tensor), position for position."""
    import torch
    from cald_amd import detector, pool, synth, sweep
    model = detector.fasterrcnn_resnet50_fpn_feature(num_classes=21, min_size=300, max_size=500).to("cuda")
    model.load_state_dict(synth.pseudo_trained_frcnn(21, 50, seed=0))
    model.eval()
    imgs = [np.ascontiguousarray(synth.synth_image(40 + i, 200 + 16 * i, 300 - 8 * i)) for i in range(5)]
    blobs = [_encode(a, quality=88, subsampling=2) for a in imgs]
    dp = pool.DevicePool.from_jpeg_bytes(blobs, chunk=2)
    assert len(dp) == 5 and dp.nbytes >= sum(a.size for a in imgs)
    for i, b in enumerate(blobs):
        assert np.array_equal(dp[i].cpu().numpy(), _pil_decode(b))
    augs = ["flip", "cut_out", "smaller_resize"]
    subset = [3, 0, 4, 1]
    u_pool, c_pool = sweep.get_uncertainty(model, dp.loader(subset), augs, 21, base_seed=9)
    ref_loader = [([torch.from_numpy(_pil_decode(blobs[i]).copy()).permute(2, 0, 1).float().div(255)], [None]) for i in subset]
    u_ref, c_ref = sweep.get_uncertainty(model, ref_loader, augs, 21, base_seed=9)
    assert u_pool == u_ref
    assert all(np.array_equal(a, b) for a, b in zip(c_pool, c_ref))
    # a pool built from decoded arrays is the same pool
    dp2 = pool.DevicePool.from_arrays([_pil_decode(b) for b in blobs])
    assert all(torch.equal(dp[i], dp2[i]) for i in range(5))
