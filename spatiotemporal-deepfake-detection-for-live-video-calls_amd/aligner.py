"""Clip aligner on the MI355X: drop-in for the reference's ``FasterCropAlignXRay``
(altfreezing/test_tools/faster_crop_align_xray.py:11-88), the step right before the classifier in every caller
(demo.py:315, demo2.py:301, TEST2.py:401, feature.py:84).

One similarity transform is fitted over the 5-point landmarks of ALL frames of a clip (test_tools/warp_for_xray.py:556-560:
least squares for a non-reflective similarity and for its mirror image, the smaller residual wins), the 68-point landmarks
are mapped through it, and every frame's crop is pasted on a common canvas and warped to ``size`` x ``size``.  The fit is
a 4-unknown least-squares problem on T*5 points and stays numpy on the host, like the reference; the T warps
(``cv2.warpAffine`` on the CPU there, inside the timed region of demo.py) are one HIP launch from the uploaded crops:
``af_warp_affine_clip_u8`` (csrc/af_align.hip).  With ``device_output=True`` the aligned clip stays in HBM as the uint8
(T, size, size, 3) tensor ``I3D8x8.forward_clips_u8`` consumes - no host round trip between aligner and classifier.

The warp's arithmetic is OpenCV's fixed-point bilinear one, bit-exact against the CPU restatement in oracle/; against
cv2 itself its parity is unpinned (cv2 is absent from the build image, the reference pins no version and holds no
aligned frame).  There is no CPU fallback: without the HIP library the call fails.
"""
import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

STD_POINTS_317 = np.array([[85.82991, 115.7792], [169.0532, 114.3381], [127.574, 167.0006],
                           [90.6964, 204.7014], [167.3069, 203.3733]]) + 30.0            # warp_for_xray.py:532-545
STD_POINTS_256 = STD_POINTS_317 - np.array([30.0, 60.0])                                 # :547-549


def _nonreflective(src: np.ndarray, dst: np.ndarray):
    """least squares for dst ~ [x y 1] . [[sc -ss] [ss sc] [tx ty]] read backwards (cp2tform's convention,
    warp_for_xray.py:224-334): solves for the map dst -> src and returns its inverse, the forward 3x3 (row vectors)."""
    x, y = dst[:, 0:1], dst[:, 1:2]
    one, zero = np.ones_like(x), np.zeros_like(x)
    A = np.vstack((np.hstack((x, y, one, zero)), np.hstack((y, -x, zero, one))))
    b = np.vstack((src[:, 0:1], src[:, 1:2]))
    if np.linalg.matrix_rank(A) < 4:
        raise Exception("cp2tform:twoUniquePointsReq")
    sc, ss, tx, ty = np.squeeze(np.linalg.lstsq(A, b, rcond=-1)[0])
    fwd = np.linalg.inv(np.array([[sc, -ss, 0.0], [ss, sc, 0.0], [tx, ty, 1.0]]))
    fwd[:, 2] = (0.0, 0.0, 1.0)
    return fwd


def _apply(trans: np.ndarray, pts: np.ndarray) -> np.ndarray:
    return (np.hstack((pts, np.ones((pts.shape[0], 1)))) @ trans)[:, :2]


def estimate_batch_transform(all_src_pts, tgt_pts: np.ndarray):
    """(tfm 2x3 for the warp, trans 3x3) of ``estimiate_batch_transform`` (warp_for_xray.py:556-560).  Like the reference's
    findSimilarity (:337-425) the mirrored fit reflects the target array in place, so BOTH residuals are measured against
    the reflected targets - kept, it decides which solution wins."""
    src = np.asarray(all_src_pts, dtype=np.float64).reshape(-1, 2)
    tgt = np.repeat(np.asarray(tgt_pts, dtype=np.float64)[None], len(all_src_pts), 0).reshape(-1, 2)
    plain = _nonreflective(src, tgt)
    tgt[:, 0] *= -1.0
    mirrored = _nonreflective(src, tgt) @ np.diag([-1.0, 1.0, 1.0])
    trans = plain if np.linalg.norm(_apply(plain, src) - tgt) <= np.linalg.norm(_apply(mirrored, src) - tgt) else mirrored
    return trans[:, 0:2].T, trans


class FasterCropAlignXRay:
    """``FasterCropAlignXRay(size)(landmarks, images)`` -> ``(landmarks68, images)`` like the reference (same argument
    meaning: per frame ``(_, ldm5 (5,2), ldm68 (68,2), box (x0,y0,x1,y1))`` relative to the frame's crop, and the crop as an
    HxWx3 uint8 array).  ``images`` come back as a (T, size, size, 3) uint8 numpy array, or - ``device_output=True`` - as a
    CUDA tensor of that shape."""

    def __init__(self, size: int = 256, return_ldm5: bool = False, device: Optional[torch.device] = None):
        self.image_size = int(size)
        self.std_points = STD_POINTS_256 * size / 256.0
        self.return_ldm5 = return_ldm5
        self.device = device

    def __call__(self, landmarks, images: Optional[Sequence[np.ndarray]] = None, jitter: bool = False, device_output: bool = False):
        landmarks = [lm[:4] for lm in landmarks]
        boxes = np.array([box for _, _, _, box in landmarks])
        five = np.array([l5 for _, l5, _, _ in landmarks])
        l68 = np.array([l for _, _, l, _ in landmarks])
        left_top = boxes[:, :2].min(0)
        w, h = boxes[:, 2:].max(0) - left_top                      # the canvas all crops are pasted on
        diff = boxes[:, :2] - left_top[None]
        five_c, l68_c = five + diff[:, None, :], l68 + diff[:, None, :]
        fit_pts = five_c.copy()
        if jitter:
            fit_pts += np.random.uniform(-4, 4, fit_pts.shape)
        tfm, trans = estimate_batch_transform(fit_pts, self.std_points)
        t68 = np.array([_apply(trans, l) for l in l68_c])
        t5 = np.array([_apply(trans, l) for l in five_c])
        if images is None:
            return (t5, t68) if self.return_ldm5 else t68
        aligned = self.warp_clip(images, diff, int(h), int(w), tfm)
        if not device_output:
            aligned = aligned.cpu().numpy()
        return (t5, t68, aligned) if self.return_ldm5 else (t68, aligned)

    def warp_clip(self, images: Sequence[np.ndarray], diff: np.ndarray, h: int, w: int, tfm: np.ndarray) -> torch.Tensor:
        """the ``process_single`` loop (:75-88) for the whole clip: upload the crops, one launch per <= 64 frames"""
        dev = self.device or torch.device("cuda", torch.cuda.current_device())
        out = torch.empty((len(images), self.image_size, self.image_size, 3), dtype=torch.uint8, device=dev)
        if len(images) == 0:
            return out
        with torch.cuda.device(dev):
            crops, offs, host = self.stage_crops(images, dev)
            self.launch_warps(crops, offs, [im.shape for im in images], diff, h, w, tfm, out)
            torch.cuda.current_stream(dev).synchronize()      # `crops` / `host` must outlive the asynchronous copy and launch
        return out

    @staticmethod
    def stage_crops(images: Sequence[np.ndarray], dev):
        """crops -> one pinned host buffer -> one H2D copy; returns (device bytes, per-frame byte offsets, host buffer)"""
        offs, total = [], 0
        for im in images:
            if not (isinstance(im, np.ndarray) and im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3):
                raise AssertionError("aligner: images must be HxWx3 uint8 numpy arrays")
            offs.append(total)
            total += (im.size + 15) // 16 * 16
        host = torch.empty(total, dtype=torch.uint8, pin_memory=True)
        hv = host.numpy()
        for im, o in zip(images, offs):
            hv[o:o + im.size] = np.ascontiguousarray(im).reshape(-1)
        return host.to(dev, non_blocking=True), offs, host

    def launch_warps(self, crops: torch.Tensor, offs, shapes, diff, h: int, w: int, tfm: np.ndarray, out: torch.Tensor):
        """enqueue af_warp_affine_clip_u8 for the frames of one clip (device-resident crops) on the current stream"""
        from . import _lib                                   # fails loudly when libafhip.so is missing
        dev = crops.device
        n, size = len(offs), self.image_size
        m = (C.c_double * 6)(*np.asarray(tfm, dtype=np.float64).reshape(6).tolist())
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for lo in range(0, n, _lib.ALIGN_MAX_FRAMES):
            hi = min(n, lo + _lib.ALIGN_MAX_FRAMES)
            frames = (_lib.AlignFrame * (hi - lo))()
            for k, i in enumerate(range(lo, hi)):
                ih, iw = int(shapes[i][0]), int(shapes[i][1])
                x, y = int(diff[i][0]), int(diff[i][1])
                if x < 0 or y < 0 or x + iw > w or y + ih > h:
                    # numpy refuses new_image[y:y+ih, x:x+iw] = image for a crop that sticks out of the canvas
                    raise ValueError("aligner: frame %d (%dx%d at %d,%d) does not fit the %dx%d canvas" % (i, iw, ih, x, y, w, h))
                frames[k] = _lib.AlignFrame(offs[i], ih, iw, x, y)
            _lib.check(_lib.lib.af_warp_affine_clip_u8(C.c_void_p(crops.data_ptr()), C.cast(frames, C.c_void_p), hi - lo, h, w, m,
                                                       size, C.c_void_p(out[lo:hi].data_ptr()), stream), "warp_affine_clip_u8")


def synthetic_clip(frames: int = 32, seed: int = 0, mirrored: bool = False):
    """(infos, crops) of a synthetic tracked face for tests / the bench: 5 points = the standard points under a random
    similarity + per-frame jitter, tracker boxes of slightly different origin and size per frame, random-noise crops."""
    rng = np.random.default_rng(seed)
    std = STD_POINTS_317 - 30.0
    ang, sc = rng.uniform(-0.5, 0.5), rng.uniform(0.6, 1.6)
    rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]) * sc
    infos, crops = [], []
    for _ in range(frames):
        p5 = std @ rot.T + rng.uniform(40, 60, size=(1, 2)) + rng.normal(0, 1.5, size=(5, 2))
        if mirrored:
            p5[:, 0] = 400 - p5[:, 0]
        p68 = p5.mean(0, keepdims=True) + rng.normal(0, 40 * sc, size=(68, 2))
        x0, y0 = rng.integers(100, 140, size=2)
        bw, bh = rng.integers(380, 460, size=2)
        infos.append((None, p5, p68, np.array([x0, y0, x0 + bw, y0 + bh], dtype=np.int64)))
        crops.append(rng.integers(0, 256, size=(int(bh), int(bw), 3), dtype=np.uint8))
    return infos, crops
