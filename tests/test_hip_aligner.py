"""GPU: the clip aligner - af_warp_affine_clip_u8 behind the FasterCropAlignXRay mirror against the CPU
restatement oracle/aligner_oracle.py.  Integer work: BIT-EXACT.  The oracle's similarity fit is pinned by the reference's
own numpy code (tests/golden/f8_aligner.npz); its warp restates OpenCV's fixed-point algorithm and is PARITY UNPINNED
against cv2 itself (absent from the build image; the reference holds no aligned frame)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_npz

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import aligner_oracle as ao  # noqa: E402
from af_mi355x import aligner  # noqa: E402

pytestmark = pytest.mark.gpu


def _clip(rng, g, tag, crop_jitter=True):
    """landmarks of a golden case + random crops whose sizes follow the boxes (some crops smaller than their box)"""
    infos, images = [], []
    for l5, l68, box in zip(g[tag + "_ldm5"], g[tag + "_ldm68"], g[tag + "_boxes"]):
        bw, bh = int(box[2] - box[0]), int(box[3] - box[1])
        ih = bh - (int(rng.integers(0, 9)) if crop_jitter else 0)
        iw = bw - (int(rng.integers(0, 9)) if crop_jitter else 0)
        images.append(rng.integers(0, 256, size=(ih, iw, 3), dtype=np.uint8))
        infos.append((None, l5.copy(), l68.copy(), box.copy()))
    return infos, images


@pytest.mark.parametrize("tag,size", [("t32_224", 224), ("t1_256", 256), ("t8_mirrored", 224)])
def test_aligned_clip_bit_exact_vs_oracle(tag, size):
    g = load_npz("f8_aligner.npz")
    rng = np.random.default_rng(77)
    infos, images = _clip(rng, g, tag)
    want68, want = ao.crop_align([(a, b.copy(), c.copy(), d.copy()) for a, b, c, d in infos], images, size=size)
    got68, got = aligner.FasterCropAlignXRay(size)(infos, images)
    assert isinstance(got, np.ndarray) and got.dtype == np.uint8 and got.shape == want.shape == (len(images), size, size, 3)
    np.testing.assert_allclose(got68, want68, rtol=1e-10, atol=1e-9)
    assert (got == want).all(), "aligned frames differ in %d bytes" % int((got != want).sum())
    assert got.any()                                           # the faces really land inside the output
    dev = aligner.FasterCropAlignXRay(size, return_ldm5=True)(infos, images, device_output=True)
    assert len(dev) == 3 and dev[2].is_cuda and dev[2].dtype == torch.uint8 and (dev[2].cpu().numpy() == want).all()


@pytest.mark.parametrize("m", [
    [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]],                       # identity: exact copy + zero border
    [[1.0, 0.0, 7.0], [0.0, 1.0, 3.0]],                       # integer shift
    [[1.0, 0.0, 0.5], [0.0, 1.0, 0.25]],                      # sub-pixel shift
    [[0.83, -0.21, 5.3], [0.21, 0.83, -2.7]],                 # rotation + scale
    [[-0.57, 0.065, 206.7], [0.065, 0.57, -93.7]],            # reflection
    [[3.1, 0.4, -300.0], [-0.4, 3.1, -250.0]],                # magnification, mostly outside
    [[1e-3, 0.0, 10.0], [0.0, 1e-3, 10.0]],                   # everything collapses onto one source pixel far away
    [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]],                       # singular: OpenCV inverts with D = 0 -> all zeros map
])
def test_warp_kernel_bit_exact_for_hand_picked_transforms(m):
    rng = np.random.default_rng(3)
    al = aligner.FasterCropAlignXRay(96)
    h, w = 120, 150
    images = [rng.integers(0, 256, size=(100, 130, 3), dtype=np.uint8), rng.integers(0, 256, size=(120, 150, 3), dtype=np.uint8),
              rng.integers(0, 256, size=(1, 1, 3), dtype=np.uint8)]
    diff = np.array([[20, 20], [0, 0], [149, 119]])
    got = al.warp_clip(images, diff, h, w, np.array(m)).cpu().numpy()
    for i, (im, d) in enumerate(zip(images, diff)):
        canvas = np.zeros((h, w, 3), dtype=np.uint8)
        canvas[d[1]:d[1] + im.shape[0], d[0]:d[0] + im.shape[1]] = im
        want = ao.warp_affine_u8(canvas, np.array(m), 96)
        assert (got[i] == want).all(), (i, int((got[i] != want).sum()))


def test_more_frames_than_one_launch_and_errors():
    rng = np.random.default_rng(9)
    al = aligner.FasterCropAlignXRay(32)
    n = 70                                                    # > AF_ALIGN_MAX_FRAMES: two launches
    images = [rng.integers(0, 256, size=(40, 40, 3), dtype=np.uint8) for _ in range(n)]
    diff = np.zeros((n, 2), dtype=np.int64)
    m = np.array([[0.7, 0.1, 1.0], [-0.1, 0.7, 2.0]])
    got = al.warp_clip(images, diff, 40, 40, m).cpu().numpy()
    for i in (0, 63, 64, 69):
        assert (got[i] == ao.warp_affine_u8(images[i], m, 32)).all()
    assert al.warp_clip([], diff[:0], 40, 40, m).shape == (0, 32, 32, 3)
    with pytest.raises(ValueError, match="does not fit"):     # numpy's slice assignment refuses this in the reference
        al.warp_clip(images[:1], np.array([[5, 0]]), 40, 40, m)
    with pytest.raises(AssertionError):
        al.warp_clip([images[0].astype(np.float32)], diff[:1], 40, 40, m)


def test_aligner_feeds_classifier_without_host_round_trip(weights0):
    """aligned clip stays in HBM: FasterCropAlignXRay(device_output=True) -> I3D8x8.forward_clips_u8; same logit as the
    caller's as_tensor / permute / normalise path on the host copy of the same clip."""
    from af_mi355x import synth
    from af_mi355x.classifier import Classifier
    g = load_npz("f8_aligner.npz")
    infos, images = _clip(np.random.default_rng(1), g, "t32_224")
    _, clip_dev = aligner.FasterCropAlignXRay(224)(infos, images, device_output=True)
    clf = Classifier(precision="f32")
    clf.network.load_state_dict(weights0)
    clf = clf.to("cuda").eval()
    with torch.inference_mode():
        y_dev = clf.network.forward_clips_u8(clip_dev[None])["final_output"]
        y_host = clf(synth.normalize_like_callers(clip_dev.cpu()[None]).cuda())["final_output"]
    assert y_dev.shape == (1, 1) and torch.isfinite(y_dev).all()
    assert abs(float(y_dev) - float(y_host)) <= 2e-4


def test_streaming_aligner_equals_the_batch_call():
    """StreamingCropAligner: crops pushed frame by frame (incl. more frames than a window and a wrap of its slot ring),
    align_last(n) == FasterCropAlignXRay on the last n frames - same fit, same kernel: bit-exact, landmarks identical."""
    g = load_npz("f8_aligner.npz")
    rng = np.random.default_rng(5)
    infos, images = _clip(rng, g, "t32_224")
    sal = aligner.StreamingCropAligner(224, capacity=12, max_crop_pixels=max(im.shape[0] * im.shape[1] for im in images))
    ref = aligner.FasterCropAlignXRay(224)
    for i, (info, im) in enumerate(zip(infos, images)):
        sal.push(info, im)
        if i in (7, 20, 31):                                     # before and after the 12-slot ring wrapped
            n = 8
            got68, got = sal.align_last(n)
            want68, want = ref(infos[i + 1 - n:i + 1], images[i + 1 - n:i + 1])
            np.testing.assert_allclose(got68, want68, rtol=0, atol=0)
            assert (got.cpu().numpy() == want).all()
    with pytest.raises(ValueError):
        sal.align_last(12)                                       # only capacity - 1 frames stay resident


def test_non_contiguous_crops_and_other_stream_push():
    """Crops handed over as VIEWS of the camera frame (frame[y0:y1, x0:x1]: rows not contiguous) go through contiguous temporaries
    inside the staging call - which must keep them alive until its copy threads have returned (round-3 advisor finding) - and give
    the same clip as their contiguous copies, bit for bit; frames pushed to the streaming aligner from a side stream are ordered
    in front of the warp by the slot events."""
    g = load_npz("f8_aligner.npz")
    rng = np.random.default_rng(11)
    infos, images = _clip(rng, g, "t32_224")
    views = []
    for im in images:                                            # every crop as an interior view of a larger frame
        frame = rng.integers(0, 256, size=(im.shape[0] + 7, im.shape[1] + 9, 3), dtype=np.uint8)
        frame[3:3 + im.shape[0], 4:4 + im.shape[1]] = im
        v = frame[3:3 + im.shape[0], 4:4 + im.shape[1]]
        assert not v.flags.c_contiguous
        views.append(v)
    al = aligner.FasterCropAlignXRay(224)
    want68, want = al(infos, images)
    for _ in range(3):                                           # repeated: freed temporaries would be recycled by the allocator
        got68, got = al(infos, views)
        np.testing.assert_allclose(got68, want68, rtol=0, atol=0)
        assert (got == want).all()
    sal = aligner.StreamingCropAligner(224, capacity=40, max_crop_pixels=max(im.shape[0] * im.shape[1] for im in images))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for info, v in zip(infos, views):
            sal.push(info, v)
    got68, got = sal.align_last(32)                              # current stream != the stream the crops were uploaded on
    torch.cuda.synchronize()
    assert (got.cpu().numpy() == want).all()
