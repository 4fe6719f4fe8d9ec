"""Clip aligner on the MI355X: drop-in for the reference's ``FasterCropAlignXRay``
(altfreezing/test_tools/faster_crop_align_xray.py:11-88), the step right before the classifier in every caller
(demo.py:315, demo2.py:301, TEST2.py:401, feature.py:84).

One similarity transform is fitted over the 5-point landmarks of ALL frames of a clip (test_tools/warp_for_xray.py:556-560:
least squares for a non-reflective similarity and for its mirror image, the smaller residual wins), the 68-point landmarks
are mapped through it, and every frame's crop is pasted on a common canvas and warped to ``size`` x ``size``.  The fit is
a 4-unknown least-squares problem on T*5 points and stays on the host, like the reference - in closed form (the normal
equations of a similarity are diagonal in centred coordinates: ~30 us instead of two SVD-based ``lstsq`` + ``matrix_rank``
calls; same answer to 1e-12); the T warps (``cv2.warpAffine`` on the CPU there, inside the timed region of demo.py) are one
HIP launch from the uploaded crops: ``af_warp_affine_clip_u8`` (csrc/af_align.hip).  The crops go through a persistent ring
of pinned staging buffers (filled by a few copy threads, one asynchronous H2D copy per clip, no allocation and no
synchronisation per call: round 3).  With ``device_output=True`` the aligned clip stays in HBM as the uint8
(T, size, size, 3) tensor ``I3D8x8.forward_clips_u8`` consumes - no host round trip between aligner and classifier.

The warp's arithmetic is OpenCV's fixed-point bilinear one, bit-exact against the CPU restatement in oracle/; against
cv2 itself its parity is unpinned (cv2 is absent from the build image, the reference pins no version and holds no
aligned frame).  There is no CPU fallback: without the HIP library the call fails.
"""
import ctypes as C
import os
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Optional, Sequence

import numpy as np
import torch

STD_POINTS_317 = np.array([[85.82991, 115.7792], [169.0532, 114.3381], [127.574, 167.0006],
                           [90.6964, 204.7014], [167.3069, 203.3733]]) + 30.0            # warp_for_xray.py:532-545
STD_POINTS_256 = STD_POINTS_317 - np.array([30.0, 60.0])                                 # :547-549


def _nonreflective(src: np.ndarray, dst: np.ndarray):
    """least squares for dst ~ [x y 1] . [[sc -ss] [ss sc] [tx ty]] read backwards (cp2tform's convention,
    warp_for_xray.py:224-334): solves for the map dst -> src and returns its inverse, the forward 3x3 (row vectors).
    The reference calls numpy.linalg.lstsq on the 2N x 4 system  u = sc x + ss y + tx,  v = sc y - ss x + ty  (x, y = dst,
    u, v = src); in centred coordinates its normal equations are diagonal, so the minimiser is written down directly."""
    x, y, u, v = dst[:, 0], dst[:, 1], src[:, 0], src[:, 1]
    xm, ym, um, vm = x.mean(), y.mean(), u.mean(), v.mean()
    xc, yc, uc, vc = x - xm, y - ym, u - um, v - vm
    den = float(xc @ xc + yc @ yc)
    # rank(A) < 4 <=> all target points coincide (numpy.linalg.matrix_rank's verdict in the reference, :262-263)
    if not den > 1e-20 * (1.0 + float(x @ x + y @ y)):
        raise Exception("cp2tform:twoUniquePointsReq")
    sc = float(xc @ uc + yc @ vc) / den
    ss = float(yc @ uc - xc @ vc) / den
    tx, ty = um - sc * xm - ss * ym, vm - sc * ym + ss * xm
    # inverse of [[sc, -ss, 0], [ss, sc, 0], [tx, ty, 1]]
    det = sc * sc + ss * ss
    a, b = sc / det, ss / det
    return np.array([[a, b, 0.0], [-b, a, 0.0], [-(tx * a - ty * b), -(tx * b + ty * a), 1.0]])


def _apply(trans: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """row-vector points (..., 2) through a 3x3 forward matrix"""
    return pts @ trans[:2, :2] + trans[2, :2]


def estimate_batch_transform(all_src_pts, tgt_pts: np.ndarray):
    """(tfm 2x3 for the warp, trans 3x3) of ``estimiate_batch_transform`` (warp_for_xray.py:556-560).  Like the reference's
    findSimilarity (:337-425) the mirrored fit reflects the target array in place, so BOTH residuals are measured against
    the reflected targets - kept, it decides which solution wins."""
    src = np.asarray(all_src_pts, dtype=np.float64).reshape(-1, 2)
    tgt = np.tile(np.asarray(tgt_pts, dtype=np.float64), (len(all_src_pts), 1))
    plain = _nonreflective(src, tgt)
    tgt[:, 0] *= -1.0
    mirrored = _nonreflective(src, tgt) @ np.diag([-1.0, 1.0, 1.0])
    trans = plain if np.linalg.norm(_apply(plain, src) - tgt) <= np.linalg.norm(_apply(mirrored, src) - tgt) else mirrored
    return trans[:, 0:2].T, trans


class _StagingRing:
    """persistent pinned host buffers + their device twins, used round-robin: a slot is rewritten only after the event
    recorded behind its last consumer (the warp launch) has completed, so neither the asynchronous copy nor the kernel can
    see a buffer change under them; `slots` clips may be in flight."""

    def __init__(self, slots: int = 3):
        self.host = [None] * slots
        self.dev = [None] * slots
        self.done = [None] * slots
        self.next = 0
        self.lock = threading.Lock()

    def acquire(self, nbytes: int, dev) -> int:
        with self.lock:
            k = self.next
            self.next = (k + 1) % len(self.host)
        if self.done[k] is not None:
            self.done[k].synchronize()
        if self.host[k] is None or self.host[k].numel() < nbytes or self.dev[k].device != dev:
            cap = max(nbytes + nbytes // 4, 1 << 20)
            self.host[k] = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
            self.dev[k] = torch.empty(cap, dtype=torch.uint8, device=dev)
        return k


# also cut the COLUMNS the warp cannot sample (rows are always cut)?  Measured on one box with the C staging copy: rows only 1 448
# clips/s host-inclusive, rows + columns 1 020-1 100 - 22 % fewer bytes, but 8 000 short strided memcpys per clip run far below the
# rate of 32 long ones.  Off; AF_ALIGN_COLS=1 switches it on for A/B runs.
_CLIP_COLUMNS = os.environ.get("AF_ALIGN_COLS", "0") == "1"
_PLAN_IN_C = os.environ.get("AF_ALIGN_PLAN", "1") == "1"        # 0: the per-frame planning in Python (round 3), for A/B runs
_COPY_THREADS = 2          # measured on the MI355X host: 1 thread 45 GB/s, 2 threads 72 GB/s, 4+ slower (memory-bound copies)
_copy_pool = None


def _pool():
    global _copy_pool
    if _copy_pool is None:
        _copy_pool = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix="af-align-stage")
    return _copy_pool


def _copy_group(pairs):
    for dst, im in pairs:
        if im.flags.c_contiguous:
            np.copyto(dst, im.reshape(-1))
        else:                                    # a column-clipped view: one strided copy, row by row, no temporary
            np.copyto(dst.reshape(im.shape), im)


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
        self._ring = _StagingRing()

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
        t68, t5 = _apply(trans, l68_c), _apply(trans, five_c)
        if images is None:
            return (t5, t68) if self.return_ldm5 else t68
        aligned = self.warp_clip(images, diff, int(h), int(w), tfm)
        if not device_output:
            aligned = aligned.cpu().numpy()
        return (t5, t68, aligned) if self.return_ldm5 else (t68, aligned)

    def warp_clip(self, images: Sequence[np.ndarray], diff: np.ndarray, h: int, w: int, tfm: np.ndarray) -> torch.Tensor:
        """the ``process_single`` loop (:75-88) for the whole clip: upload the crops, one launch per <= 64 frames.  Everything
        is enqueued on the current stream and the call returns without synchronising (the staging slot is protected by an
        event, see _StagingRing); `out` is valid for stream-ordered consumers, `.cpu()` waits for it."""
        dev = self.device or torch.device("cuda", torch.cuda.current_device())
        dev = torch.device(dev)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        out = torch.empty((len(images), self.image_size, self.image_size, 3), dtype=torch.uint8, device=dev)
        if len(images) == 0:
            return out
        if not _CLIP_COLUMNS and _PLAN_IN_C:
            return self._warp_clip_planned(images, diff, h, w, tfm, out, dev)
        # only the part of a crop the warp can touch is uploaded: the destination square maps to a parallelogram of the canvas;
        # a crop cut to rows [r0, r1) x columns [c0, c1) is the same picture as that smaller crop pasted (c0, r0) further in
        # (everything else it would have covered is never sampled), so the kernel's frame table takes the moved origin and the
        # smaller size and nothing else changes
        for i, im in enumerate(images):
            x, y = int(diff[i][0]), int(diff[i][1])
            if im.ndim == 3 and (x < 0 or y < 0 or x + im.shape[1] > w or y + im.shape[0] > h):
                # numpy refuses new_image[y:y+ih, x:x+iw] = image for a crop that sticks out of the canvas
                raise ValueError("aligner: frame %d (%dx%d at %d,%d) does not fit the %dx%d canvas" % (i, im.shape[1], im.shape[0], x, y, w, h))
        shapes = []
        for im in images:
            if not (isinstance(im, np.ndarray) and im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3):
                raise AssertionError("aligner: images must be HxWx3 uint8 numpy arrays")
        images, shapes, diff = self._clip_rect(images, diff, tfm)
        with torch.cuda.device(dev):
            crops, offs, slot = self.stage_crops_ring(images, dev)
            self.launch_warps(crops, offs, shapes, diff, h, w, tfm, out)
            ev = torch.cuda.Event()
            ev.record()
            self._ring.done[slot] = ev
        return out

    def _warp_clip_planned(self, images, diff, h: int, w: int, tfm, out: torch.Tensor, dev) -> torch.Tensor:
        """warp_clip with the per-frame work in C (round 4): ONE Python pass over the crops collects (address, pitch, shape, paste
        offset); af_align_plan_u8 checks the canvas fit, cuts every crop to the rows the warp can sample and fills the staging
        table and the kernel's frame table; then the staging copy (two threads), one H2D copy, one warp launch per <= 64 frames.
        Same arithmetic as the Python path it replaces (_clip_rect + stage_crops_ring + launch_warps, still used by the column-cut
        experiment and the streaming aligner): the clip is bit for bit the same."""
        from . import _lib
        n = len(images)
        crops = (_lib.AlignCrop * n)()
        keep = []                                      # every array whose address goes into `crops` lives until the copies return
        for i, im in enumerate(images):
            if not (isinstance(im, np.ndarray) and im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3):
                raise AssertionError("aligner: images must be HxWx3 uint8 numpy arrays")
            st = im.strides
            if st[2] != 1 or st[1] != 3 or st[0] < im.shape[1] * 3:
                im = np.ascontiguousarray(im)
                st = im.strides
            keep.append(im)
            crops[i] = _lib.AlignCrop(im.__array_interface__["data"][0], st[0], im.shape[0], im.shape[1], int(diff[i][0]), int(diff[i][1]))
        rects = (_lib.StageRect * n)()
        frames = (_lib.AlignFrame * n)()
        total, bad = C.c_int64(0), C.c_int32(-1)
        m = (C.c_double * 6)(*np.asarray(tfm, dtype=np.float64).reshape(6).tolist())
        rc = _lib.lib.af_align_plan_u8(crops, n, int(h), int(w), m, self.image_size, rects, frames, C.byref(total), C.byref(bad))
        if rc != 0 and bad.value >= 0:
            i, im = bad.value, keep[bad.value]
            # numpy refuses new_image[y:y+ih, x:x+iw] = image for a crop that sticks out of the canvas
            raise ValueError("aligner: frame %d (%dx%d at %d,%d) does not fit the %dx%d canvas"
                             % (i, im.shape[1], im.shape[0], int(diff[i][0]), int(diff[i][1]), w, h))
        _lib.check(rc, "align_plan_u8")
        with torch.cuda.device(dev):
            k = self._ring.acquire(total.value, dev)
            base = self._ring.host[k].data_ptr()
            nt = min(_COPY_THREADS, n)
            if nt > 1 and total.value >= (1 << 20):
                cuts = [n * t // nt for t in range(nt + 1)]
                def part(t):
                    _lib.check(_lib.lib.af_stage_rows_u8(C.c_void_p(base), C.byref(rects, cuts[t] * C.sizeof(_lib.StageRect)), cuts[t + 1] - cuts[t]),
                               "stage_rows_u8")
                list(_pool().map(part, range(nt)))
            else:
                _lib.check(_lib.lib.af_stage_rows_u8(C.c_void_p(base), rects, n), "stage_rows_u8")
            del keep
            d = self._ring.dev[k]
            d[:total.value].copy_(self._ring.host[k][:total.value], non_blocking=True)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for lo in range(0, n, _lib.ALIGN_MAX_FRAMES):
                hi = min(n, lo + _lib.ALIGN_MAX_FRAMES)
                _lib.check(_lib.lib.af_warp_affine_clip_u8(C.c_void_p(d.data_ptr()), C.byref(frames, lo * C.sizeof(_lib.AlignFrame)), hi - lo,
                                                           int(h), int(w), m, self.image_size, C.c_void_p(out[lo:hi].data_ptr()), stream),
                           "warp_affine_clip_u8")
            ev = torch.cuda.Event()
            ev.record()
            self._ring.done[k] = ev
        return out

    def _clip_rect(self, images: Sequence[np.ndarray], diff: np.ndarray, tfm: np.ndarray):
        """the rectangle of the canvas the size x size destination can sample (bilinear taps, fixed-point rounding: 3 pixels of
        margin) -> per frame the part of the crop inside it (a rows x bytes view, no copy), its (h, w, 3) shape and the paste offset
        moved accordingly.  Rows are always cut (a contiguous view); columns only with AF_ALIGN_COLS=1 (see _CLIP_COLUMNS)."""
        m = np.asarray(tfm, dtype=np.float64).reshape(2, 3)
        det = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
        if not np.isfinite(det) or abs(det) < 1e-12:
            return images, [im.shape for im in images], diff # singular map: OpenCV's D = 0 path samples around one point; keep everything
        s = float(self.image_size - 1)
        corners = np.array([[0.0, 0.0], [s, 0.0], [0.0, s], [s, s]])
        # dst = M [x y 1]^T  ->  src = M^-1 (dst - t)
        dx, dy = corners[:, 0] - m[0, 2], corners[:, 1] - m[1, 2]
        xs = (m[1, 1] * dx - m[0, 1] * dy) / det
        ys = (-m[1, 0] * dx + m[0, 0] * dy) / det
        if not (np.isfinite(xs).all() and np.isfinite(ys).all()):
            return images, [im.shape for im in images], diff
        ylo, yhi = int(np.floor(ys.min())) - 3, int(np.ceil(ys.max())) + 4
        xlo, xhi = int(np.floor(xs.min())) - 3, int(np.ceil(xs.max())) + 4
        out_images, shapes, out_diff = [], [], np.array(diff, dtype=np.int64, copy=True)
        for i, im in enumerate(images):
            x0, y0 = int(out_diff[i][0]), int(out_diff[i][1])
            r0, r1 = max(0, ylo - y0), min(im.shape[0], yhi - y0)
            c0, c1 = max(0, xlo - x0), min(im.shape[1], xhi - x0)
            if not _CLIP_COLUMNS:
                c0, c1 = 0, im.shape[1]
            if r1 <= r0 or c1 <= c0:                         # the warp never reaches this crop: one pixel keeps the frame table valid
                r0, r1, c0, c1 = 0, 1, 0, 1
            rows = im[r0:r1]
            if rows.flags.c_contiguous:
                # rows x bytes: the column cut of a (rows, W * 3) view copies as one memcpy per row (the (h, w, 3) view of the
                # same pixels went element by element: 5 GB/s instead of 45)
                out_images.append(rows.reshape(r1 - r0, im.shape[1] * 3)[:, c0 * 3:c1 * 3])
            else:
                out_images.append(np.ascontiguousarray(rows[:, c0:c1]).reshape(r1 - r0, (c1 - c0) * 3))
            shapes.append((r1 - r0, c1 - c0, 3))
            out_diff[i][0], out_diff[i][1] = x0 + c0, y0 + r0
        return out_images, shapes, out_diff

    @staticmethod
    def _layout(images: Sequence[np.ndarray]):
        offs, total = [], 0
        for im in images:
            if not (isinstance(im, np.ndarray) and im.dtype == np.uint8 and (im.ndim == 2 or (im.ndim == 3 and im.shape[2] == 3))):
                raise AssertionError("aligner: images must be HxWx3 uint8 numpy arrays (or their rows x bytes views)")
            offs.append(total)
            total += (im.size + 15) // 16 * 16
        return offs, total

    def stage_crops_ring(self, images: Sequence[np.ndarray], dev):
        """crops -> a pinned ring slot (copy threads; numpy releases the GIL for these copies) -> one asynchronous H2D copy
        into the slot's device twin; returns (device bytes, per-frame byte offsets, slot)"""
        from . import _lib
        offs, total = self._layout(images)
        k = self._ring.acquire(total, dev)
        host = self._ring.host[k]
        # one C call per copy thread (af_stage_rows_u8: a memcpy per row; ctypes releases the GIL): numpy copies a column-cut view
        # with ~100 ns of iterator overhead per row - 8 000 rows per clip, half of the whole call
        rects = (_lib.StageRect * len(images))()
        keep = []              # every array whose address goes into `rects` - contiguous temporaries included - lives until the copies return
        for i, (im, o) in enumerate(zip(images, offs)):
            if im.ndim == 3:
                im = im.reshape(im.shape[0], im.shape[1] * 3) if im.flags.c_contiguous else np.ascontiguousarray(im).reshape(im.shape[0], -1)
            if im.strides[1] != 1:
                im = np.ascontiguousarray(im)
            keep.append(im)
            rects[i] = _lib.StageRect(im.ctypes.data, o, im.strides[0] if im.shape[0] > 1 else im.shape[1], im.shape[0], im.shape[1])
        base = host.data_ptr()
        # (staging in four chunks, each crossing PCIe while the next is copied, was tried: the extra pool round trips and small
        #  copies cost more than the overlap gained - host-inclusive 1 780 -> 860 clips/s)
        nt = min(_COPY_THREADS, len(images))
        if nt > 1 and total >= (1 << 20):
            n = len(images)
            cuts = [n * t // nt for t in range(nt + 1)]
            def part(t):
                _lib.check(_lib.lib.af_stage_rows_u8(C.c_void_p(base), C.byref(rects, cuts[t] * C.sizeof(_lib.StageRect)), cuts[t + 1] - cuts[t]),
                           "stage_rows_u8")
            list(_pool().map(part, range(nt)))
        else:
            _lib.check(_lib.lib.af_stage_rows_u8(C.c_void_p(base), rects, len(images)), "stage_rows_u8")
        del keep
        d = self._ring.dev[k]
        d[:total].copy_(host[:total], non_blocking=True)
        return d, offs, k

    @staticmethod
    def stage_crops(images: Sequence[np.ndarray], dev):
        """crops -> one fresh pinned host buffer -> one H2D copy; returns (device bytes, per-frame byte offsets, host buffer).
        (For callers that keep the crops resident, e.g. bench.py --model aligner; __call__ goes through the ring.)"""
        offs, total = FasterCropAlignXRay._layout(images)
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


class StreamingCropAligner:
    """The same alignment for a LIVE track (reference test/af_realtime.py:RealtimeAF.step, test/app_realtime.py:153: one call per
    captured frame; every `stride` frames the last `clip_size` crops of the track are aligned and classified).  The reference
    keeps the crops in a host deque and hands all 32 to FasterCropAlignXRay when the window closes - ~17 MB to stage and upload
    on the critical path of every window.  Here a crop goes to the GPU when its frame is captured (`push`: one copy into a pinned
    slot + one asynchronous H2D per frame, ~0.5 MB), so that closing a window (`align_last`) costs the similarity fit over the
    window's landmarks and ONE warp launch over crops that are already resident: enqueue -> score is fit + warp + forward.
    Same arithmetic as FasterCropAlignXRay (its fit, the same kernel): `align_last(n)` equals
    `FasterCropAlignXRay(size)(infos[-n:], crops[-n:])`."""

    def __init__(self, size: int = 224, capacity: int = 64, max_crop_pixels: int = 512 * 512, device: Optional[torch.device] = None):
        self.aligner = FasterCropAlignXRay(size, device=device)
        self.capacity = int(capacity)
        self.slot_bytes = (max_crop_pixels * 3 + 15) // 16 * 16
        dev = torch.device(device or torch.device("cuda", torch.cuda.current_device()))
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.host = torch.empty(self.capacity * self.slot_bytes, dtype=torch.uint8, pin_memory=True)
        self.dev = torch.empty(self.capacity * self.slot_bytes, dtype=torch.uint8, device=dev)
        self._hv = self.host.numpy()
        self.frames = []                   # (info, crop shape, slot) of the frames still resident, oldest first
        self.count = 0
        # slot reuse is ordered by events, not by assumptions about the caller's streams / pacing: the H2D copy out of a pinned
        # slot (recorded per slot), and the last warp launch that may still read the device twins (one event for all slots)
        self._h2d_done = [None] * self.capacity
        self._warp_done = None

    def push(self, info, crop: np.ndarray) -> None:
        """a captured frame of the track: landmark record ``(_, ldm5, ldm68, box)`` + its HxWx3 uint8 crop"""
        if not (isinstance(crop, np.ndarray) and crop.dtype == np.uint8 and crop.ndim == 3 and crop.shape[2] == 3):
            raise AssertionError("aligner: images must be HxWx3 uint8 numpy arrays")
        if crop.size > self.slot_bytes:
            raise ValueError("aligner: a %dx%d crop exceeds the slot size (max_crop_pixels)" % (crop.shape[1], crop.shape[0]))
        slot = self.count % self.capacity
        self.count += 1
        lo = slot * self.slot_bytes
        if self._h2d_done[slot] is not None:
            self._h2d_done[slot].synchronize()          # the H2D of `capacity` frames ago has left the pinned slot (normally long ago)
        np.copyto(self._hv[lo:lo + crop.size], crop.reshape(-1) if crop.flags.c_contiguous else np.ascontiguousarray(crop).reshape(-1))
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            if self._warp_done is not None:
                cur.wait_event(self._warp_done)         # a warp launched from another stream may still read the slot's old occupant
            self.dev[lo:lo + crop.size].copy_(self.host[lo:lo + crop.size], non_blocking=True)
            if self._h2d_done[slot] is None:
                self._h2d_done[slot] = torch.cuda.Event()
            self._h2d_done[slot].record(cur)
        self.frames.append((tuple(info[:4]), crop.shape, slot))
        if len(self.frames) > self.capacity - 1:
            self.frames.pop(0)

    def align_last(self, n: int = 32, out: Optional[torch.Tensor] = None):
        """(landmarks68 (n,68,2), aligned clip (n, size, size, 3) uint8 CUDA tensor) of the last n pushed frames; ``out``: write
        the clip there (e.g. the static input of a graph-replayed forward) instead of a new tensor"""
        if n > len(self.frames):
            raise ValueError("aligner: %d frames requested, %d resident" % (n, len(self.frames)))
        win = self.frames[-n:]
        al = self.aligner
        boxes = np.array([f[0][3] for f in win])
        five = np.array([f[0][1] for f in win])
        l68 = np.array([f[0][2] for f in win])
        left_top = boxes[:, :2].min(0)
        w, h = boxes[:, 2:].max(0) - left_top
        diff = boxes[:, :2] - left_top[None]
        tfm, trans = estimate_batch_transform(five + diff[:, None, :], al.std_points)
        t68 = _apply(trans, l68 + diff[:, None, :])
        if out is None:
            out = torch.empty((n, al.image_size, al.image_size, 3), dtype=torch.uint8, device=self.device)
        elif out.shape != (n, al.image_size, al.image_size, 3) or out.dtype != torch.uint8 or not out.is_contiguous() or out.device != self.device:
            raise ValueError("aligner: `out` must be a contiguous uint8 (%d,%d,%d,3) tensor on %s" % (n, al.image_size, al.image_size, self.device))
        shapes = [f[1] for f in win]
        for i, shp in enumerate(shapes):
            x, y = int(diff[i][0]), int(diff[i][1])
            if x < 0 or y < 0 or x + shp[1] > int(w) or y + shp[0] > int(h):
                raise ValueError("aligner: frame %d (%dx%d at %d,%d) does not fit the %dx%d canvas" % (i, shp[1], shp[0], x, y, int(w), int(h)))
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            for f in win:                                # crops pushed from another stream: their uploads come first
                if self._h2d_done[f[2]] is not None:
                    cur.wait_event(self._h2d_done[f[2]])
            al.launch_warps(self.dev, [f[2] * self.slot_bytes for f in win], shapes, diff, int(h), int(w), tfm, out)
            if self._warp_done is None:
                self._warp_done = torch.cuda.Event()
            self._warp_done.record(cur)
        return t68, out


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
