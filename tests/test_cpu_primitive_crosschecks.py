"""Independent cross-checks of the third-party primitives the oracles RESTATE (DESIGN section 2 has the
table).  TensorFlow, OpenCV and pycocotools are absent from the image, so oracle and TF / cv2 stand-ins share one author; where
the image holds a second implementation of the same mathematics it is used here:

  * bilinear sampling of ``crop_and_resize`` (RoIAlign), ``cv2.remap`` and ``grid_sample``  <-  scipy.ndimage.map_coordinates(order=1)
  * TF1-legacy ``resize_bilinear`` (both align_corners forms) and nearest                   <-  float64 loops straight from the
                                                                                               published kernel definition
  * cv2.resize INTER_LINEAR on uint8 (11-bit fixed point)                                    <-  float64 half-pixel bilinear (<= 1 LSB)
                                                                                               and PIL's own resampler (loose bound)
  * COCO RLE strings                                                                         <-  the format's published LEB128-like
                                                                                               coding, decoded by an independent reader
These pin the MATHEMATICS (sample positions, weights, rounding direction); the libraries' own last-bit behaviour stays unpinned."""
import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle import cv_resize_oracle as CR
from oracle import merge_oracle as MO
from oracle import proposal_oracle as PO
from oracle import pwc_oracle as O
from oracle import refinement_oracle as RO


def test_roi_align_sampling_against_scipy_map_coordinates():
    """model.py:300-374 -> tf.image.crop_and_resize: output sample i of a box maps to y1*(H-1) + i*(y2-y1)*(H-1)/(crop-1) in
    feature-map pixels (the reference's fpcoor remap chooses y1, y2 so that the samples sit at the centres of crop equal bins of
    the box), bilinear between the four neighbours, then a 2x2 average."""
    rng = np.random.default_rng(0)
    H, W, C, out = 23, 31, 5, 7
    fm = rng.standard_normal((1, C, H, W)).astype(np.float32)
    boxes = np.array([[2.3, 3.1, 17.8, 15.2], [0.6, 0.7, 29.0, 21.5], [10.0, 4.0, 12.5, 20.0]], np.float32)     # x1 y1 x2 y2, inside
    got = PO.roi_align(torch.from_numpy(fm), boxes, out).numpy()
    crop = 2 * out
    for r, (x0, y0, x1, y1) in enumerate(boxes.astype(np.float64)):
        # bin centres of crop equal bins: x0 + (i + 0.5) * (x1 - x0) / crop - 0.5   (pixel centres at integer + 0.5 -> index space)
        ys = y0 + (np.arange(crop) + 0.5) * (y1 - y0) / crop - 0.5
        xs = x0 + (np.arange(crop) + 0.5) * (x1 - x0) / crop - 0.5
        yy, xx = np.meshgrid(ys, xs, indexing="ij")
        ref = np.stack([ndimage.map_coordinates(fm[0, c].astype(np.float64), [yy, xx], order=1, mode="nearest") for c in range(C)])
        ref = ref.reshape(C, out, 2, out, 2).mean(axis=(2, 4))
        assert np.abs(got[r] - ref).max() < 2e-5, r


def test_crop_and_resize_extrapolation_rule_is_zero_outside_the_map():
    """TF's kernel writes extrapolation_value (0) for a sample whose coordinate lies outside [0, H-1] -- not a blend with an
    imaginary zero border (scipy's 'constant' mode would blend)."""
    fm = torch.ones((1, 1, 6, 6))
    got = PO.roi_align(fm, np.array([[-4.0, -4.0, 4.0, 4.0]], np.float32), 2).numpy()[0, 0]
    # 4x4 samples at -3.5, -1.5, 0.5, 2.5 (index space: minus 0.5 more): the first two rows / columns are outside -> 0
    assert np.allclose(got, [[0.0, 0.0], [0.0, 1.0]])


def test_remap_fixed_point_against_scipy_on_the_1_32_lattice():
    """cv2.remap INTER_LINEAR quantises the map to 1/32 pixel and weighs with products of 5-bit fractions; on coordinates that
    ARE multiples of 1/32 the float64 bilinear value (scipy) rounded half-up must be reproduced exactly."""
    rng = np.random.default_rng(1)
    h, w = 19, 27
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    mp = np.zeros((h, w, 2), np.float32)
    mp[..., 0] = np.arange(w)[None, :] + rng.integers(-64, 64, (h, w)) / 32.0
    mp[..., 1] = np.arange(h)[:, None] + rng.integers(-64, 64, (h, w)) / 32.0
    got = MO.remap_linear_u8(img, mp).astype(np.int64)
    ref = ndimage.map_coordinates(np.pad(img.astype(np.float64), 2), [mp[..., 1].astype(np.float64) + 2, mp[..., 0].astype(np.float64) + 2],
                                  order=1, mode="constant", cval=0.0)
    inside = (mp[..., 0] >= -1) & (mp[..., 0] <= w) & (mp[..., 1] >= -1) & (mp[..., 1] <= h)      # BORDER_CONSTANT 0 one pixel out
    assert np.array_equal(got[inside], np.floor(ref[inside] + 0.5).astype(np.int64))
    # the quantisation itself: an arbitrary map differs from the exact bilinear value by at most the 1/32-pixel snap
    mp2 = mp + rng.uniform(-0.5, 0.5, mp.shape).astype(np.float32) / 32.0
    got2 = MO.remap_linear_u8(img, mp2).astype(np.float64)
    snap = np.rint(mp2.astype(np.float64) * 32) / 32
    ref2 = ndimage.map_coordinates(np.pad(img.astype(np.float64), 2), [snap[..., 1] + 2, snap[..., 0] + 2], order=1, mode="constant")
    assert np.abs(got2 - np.floor(ref2 + 0.5))[inside].max() <= 1.0


def test_pwc_warp_against_scipy_map_coordinates():
    """PWCNet.py:140-176: sample image 2 at (x + u, y + v) bilinearly, zero outside, and zero where the sampled ones-mask < 0.9999."""
    rng = np.random.default_rng(2)
    h, w = 12, 17
    x = rng.standard_normal((1, 3, h, w)).astype(np.float32)
    flo = (rng.uniform(-3, 3, (1, 2, h, w))).astype(np.float32)
    got = O.warp(torch.from_numpy(x), torch.from_numpy(flo)).numpy()[0]
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    sy, sx = yy + flo[0, 1], xx + flo[0, 0]
    ones = ndimage.map_coordinates(np.pad(np.ones((h, w)), 1), [sy + 1, sx + 1], order=1, mode="constant")
    for c in range(3):
        ref = ndimage.map_coordinates(np.pad(x[0, c].astype(np.float64), 1), [sy + 1, sx + 1], order=1, mode="constant")
        ref = ref * (ones >= 0.9999)
        sure = np.abs(ones - 0.9999) > 1e-5                      # (the threshold itself is evaluated in fp32 by the reference)
        assert np.abs(got[c] - ref)[sure].max() < 1e-5


def _resize_bilinear_tf_bruteforce(x, oh, ow, align):
    """tensorflow/core/kernels/resize_bilinear_op.cc (TF 1.x, half_pixel_centers absent): in = out * scale, scale = in/out or
    (in-1)/(out-1); lower = floor, upper = min(lower + 1, in - 1), lerp = in - lower.  float64 loops."""
    h, w = x.shape
    sy = (h - 1) / (oh - 1) if (align and oh > 1) else h / oh
    sx = (w - 1) / (ow - 1) if (align and ow > 1) else w / ow
    out = np.zeros((oh, ow))
    for i in range(oh):
        fy = i * sy
        y0 = int(np.floor(fy))
        y1 = min(y0 + 1, h - 1)
        for j in range(ow):
            fx = j * sx
            x0 = int(np.floor(fx))
            x1 = min(x0 + 1, w - 1)
            top = x[y0, x0] + (x[y0, x1] - x[y0, x0]) * (fx - x0)
            bot = x[y1, x0] + (x[y1, x1] - x[y1, x0]) * (fx - x0)
            out[i, j] = top + (bot - top) * (fy - y0)
    return out


@pytest.mark.parametrize("shape,out,align", [((7, 5), (13, 11), False), ((7, 5), (13, 11), True), ((25, 25), (97, 97), True),
                                             ((97, 97), (385, 385), False), ((9, 9), (4, 3), False)])
def test_tf_legacy_resize_bilinear_against_a_float64_brute_force(shape, out, align):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape)
    got = RO.resize_bilinear_tf(torch.from_numpy(x.astype(np.float32))[None, None], out[0], out[1], align)[0, 0].numpy()
    # (TF evaluates out_index * scale in float32: on a 385-wide axis the sample position itself carries ~1e-5 of a pixel)
    assert np.abs(got - _resize_bilinear_tf_bruteforce(x, out[0], out[1], align)).max() < 1e-4
    # nearest (legacy): index = min(floor(out_index * in / out), in - 1)
    nn = RO.resize_nearest_tf(torch.from_numpy(x.astype(np.float32))[None, None], out[0], out[1])[0, 0].numpy()
    ref = x[np.minimum((np.arange(out[0]) * shape[0] / out[0]).astype(int), shape[0] - 1)][:, np.minimum(
        (np.arange(out[1]) * shape[1] / out[1]).astype(int), shape[1] - 1)]
    assert np.array_equal(nn, ref.astype(np.float32))


@pytest.mark.parametrize("src,dst", [((480, 854), (749, 1333)), ((1080, 1920), (750, 1333)), ((480, 854), (512, 896))])
def test_cv2_resize_restatement_against_float64_bilinear_and_pil(src, dst):
    """cv2.resize(INTER_LINEAR) on uint8 = half-pixel-centre bilinear evaluated in 11-bit fixed point: within ONE grey level of the
    float64 evaluation of the same formula everywhere (proposal_net/common.py:35-62, script_pwc_multi.py:38-45 shapes).  PIL's
    BILINEAR is the same kernel when up-sampling (its support grows only when shrinking) but rounds differently: a second,
    loose witness of the sample positions (<= 2 grey levels on > 99.9 % of a smooth image, never more than 3)."""
    from PIL import Image
    rng = np.random.default_rng(4)
    (h, w), (dh, dw) = src, dst
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 100 * np.sin(xx / 37.0 + yy / 23.0), 127 + 90 * np.cos(xx / 19.0 - yy / 41.0), 40 + 0.2 * xx + 0.1 * yy], -1)
    img = np.clip(img + rng.normal(0, 4, img.shape), 0, 255).astype(np.uint8)
    got = CR.resize_linear_u8(img, dw, dh).astype(np.float64)
    fy = np.clip((np.arange(dh) + 0.5) * h / dh - 0.5, 0, h - 1)
    fx = np.clip((np.arange(dw) + 0.5) * w / dw - 0.5, 0, w - 1)
    gy, gx = np.meshgrid(fy, fx, indexing="ij")
    ref = np.stack([ndimage.map_coordinates(img[..., c].astype(np.float64), [gy, gx], order=1, mode="nearest") for c in range(3)], -1)
    assert np.abs(got - ref).max() <= 1.0                                   # the <= 1 LSB bound of the fixed-point evaluation
    assert (np.abs(got - np.rint(ref)) > 0).mean() < 0.25                   # and mostly the correctly rounded value
    if dh >= h and dw >= w:
        pil = np.asarray(Image.fromarray(img).resize((dw, dh), Image.BILINEAR)).astype(np.float64)
        d = np.abs(got - pil)
        assert d.max() <= 3 and (d > 2).mean() < 1e-3
    # the float path (flow post-resize): plain fp32 evaluation of the same weights
    f = rng.standard_normal((h // 4, w // 4)).astype(np.float32)
    gotf = CR.resize_linear_f32(f, w, h)
    fy = np.clip((np.arange(h) + 0.5) * f.shape[0] / h - 0.5, 0, f.shape[0] - 1)
    fx = np.clip((np.arange(w) + 0.5) * f.shape[1] / w - 0.5, 0, f.shape[1] - 1)
    gy, gx = np.meshgrid(fy, fx, indexing="ij")
    assert np.abs(gotf - ndimage.map_coordinates(f.astype(np.float64), [gy, gx], order=1, mode="nearest")).max() < 1e-4      # fp32 sample positions


def _coco_counts_from_string(s: str):
    """Independent reader of the COCO RLE string (cocoapi common/maskApi.c rleFrString, published format): 5 data bits + a
    continuation bit per character (offset 48), sign extension from the last group, values from the third on are deltas to the
    value two places back."""
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return cnts


def test_coco_rle_strings_against_an_independent_reader_of_the_published_format():
    from premvos_amd import rle
    rng = np.random.default_rng(5)
    for h, w in ((7, 5), (48, 85), (480, 854)):
        m = np.zeros((h, w), np.uint8)
        for _ in range(6):
            y, x = rng.integers(0, h), rng.integers(0, w)
            m[y:y + rng.integers(1, h // 2 + 2), x:x + rng.integers(1, w // 2 + 2)] ^= 1
        for enc in (RO.rle_encode(m), rle.encode(m)):
            cnts = _coco_counts_from_string(enc["counts"])
            assert sum(cnts) == h * w and all(c >= 0 for c in cnts)
            flat = np.concatenate([np.full(c, i & 1, np.uint8) for i, c in enumerate(cnts)])
            assert np.array_equal(flat.reshape(w, h).T, m)              # column-major runs, starting with a run of zeros
