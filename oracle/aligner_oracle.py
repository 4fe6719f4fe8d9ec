"""TEST INFRASTRUCTURE - CPU restatement of the reference's clip aligner.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Restates, in numpy:
  * the similarity fit        altfreezing/test_tools/warp_for_xray.py:177-198 (tformfwd), :224-334
                              (findNonreflectiveSimilarity), :337-425 (findSimilarity), :428-529, :556-560
                              (estimiate_batch_transform), :573-576 (transform_landmarks), :532-549 (std points)
  * the clip aligner          altfreezing/test_tools/faster_crop_align_xray.py:11-88 (FasterCropAlignXRay)
  * cv2.warpAffine            default flags (INTER_LINEAR, BORDER_CONSTANT 0), 8-bit 3-channel, as called at
                              faster_crop_align_xray.py:84-86

PINNING.  The similarity fit and the landmark transform are pinned by tests/golden/f8_aligner.npz, produced by the
reference's own numpy code (oracle/gen_golden.py --aligner; `import cv2` at the top of that file is satisfied by an
empty stand-in module that is never called - the same device as the fvcore / timm stand-ins).
cv2.warpAffine itself is PARITY UNPINNED: OpenCV is a third-party dependency that is neither vendored in the
reference nor installed here, the reference pins no version (requirements: `opencv-python`), and it holds no aligned
frame as a fixture.  `warp_affine_u8` restates the published fixed-point algorithm of OpenCV 3.x - 4.10
(modules/imgproc/src/imgwarp.cpp: warpAffine -> WarpAffineInvoker -> remapBilinear<FixedPtCast<int, uchar, 15>>):
  - the 2x3 matrix is inverted in double exactly as warpAffine() does;
  - source coordinates are 1/32-pixel fixed point: X = (cvRound((M1*y + M2)*1024) + 16 + cvRound(M0*x*1024)) >> 5;
  - the four bilinear weights are (32-fx)*(32-fy)*32 ... (exact 15-bit integers that sum to 32768);
  - a tap outside the source image is the border constant 0; dst = (sum w*tap + 16384) >> 15.
OpenCV >= 4.11 replaced this path by a float interpolation for 8UC3; results there can differ by 1 LSB.
"""
import numpy as np

STD_POINTS_317 = np.array([[85.82991, 115.7792], [169.0532, 114.3381], [127.574, 167.0006],
                           [90.6964, 204.7014], [167.3069, 203.3733]]) + 30          # warp_for_xray.py:532-545
STD_POINTS_256 = STD_POINTS_317.copy()                                                 # :547-549
STD_POINTS_256[..., 0] -= 30
STD_POINTS_256[..., 1] -= 60


def tformfwd(trans, uv):
    """warp_for_xray.py:177-198"""
    uv1 = np.hstack((uv, np.ones((uv.shape[0], 1))))
    return np.dot(uv1, trans)[:, 0:-1]


def find_nonreflective_similarity(uv, xy):
    """warp_for_xray.py:224-334: r = X \\ U for [u v] = [x y 1] * [[sc -ss] [ss sc] [tx ty]]; returns (T, Tinv)"""
    m = xy.shape[0]
    x = xy[:, 0].reshape((-1, 1))
    y = xy[:, 1].reshape((-1, 1))
    top = np.hstack((x, y, np.ones((m, 1)), np.zeros((m, 1))))
    bot = np.hstack((y, -x, np.zeros((m, 1)), np.ones((m, 1))))
    X = np.vstack((top, bot))
    U = np.vstack((uv[:, 0].reshape((-1, 1)), uv[:, 1].reshape((-1, 1))))
    if np.linalg.matrix_rank(X) < 4:
        raise Exception("cp2tform:twoUniquePointsReq")
    r, _, _, _ = np.linalg.lstsq(X, U, rcond=-1)
    sc, ss, tx, ty = np.squeeze(r)
    tinv = np.array([[sc, -ss, 0], [ss, sc, 0], [tx, ty, 1]])
    t = np.linalg.inv(tinv)
    t[:, 2] = np.array([0, 0, 1])
    return t, tinv


def find_similarity(uv, xy):
    """warp_for_xray.py:337-425.  As in the reference, `xyR = xy` aliases the target array: the reflection negates
    xy's x column IN PLACE, so both residual norms are taken against the reflected targets."""
    trans1, trans1_inv = find_nonreflective_similarity(uv, xy)
    xy[:, 0] = -1 * xy[:, 0]                                  # (alias of xyR, :402-403)
    trans2r, _ = find_nonreflective_similarity(uv, xy)
    trans2 = np.dot(trans2r, np.array([[-1, 0, 0], [0, 1, 0], [0, 0, 1]]))
    norm1 = np.linalg.norm(tformfwd(trans1, uv) - xy)
    norm2 = np.linalg.norm(tformfwd(trans2, uv) - xy)
    if norm1 <= norm2:
        return trans1, trans1_inv
    return trans2, np.linalg.inv(trans2)


def estimate_batch_transform(all_src_pts, tgt_pts):
    """warp_for_xray.py:556-560 (+ :496-529): one similarity over the landmarks of ALL frames of the clip.
    Returns (tfm 2x3 for cv2.warpAffine, trans 3x3 row-vector form)."""
    tgt = np.repeat(tgt_pts[None, ...], len(all_src_pts), 0).reshape(-1, 2)
    src = np.array(all_src_pts).reshape(-1, 2)
    trans, _ = find_similarity(src, tgt)
    return trans[:, 0:2].T, trans


def transform_landmarks(landmarks, trans):
    """warp_for_xray.py:573-576"""
    return np.dot(np.hstack((landmarks, np.ones((landmarks.shape[0], 1)))), trans)[:, :2]


def invert_affine(m):
    """warpAffine(), imgwarp.cpp: the forward 2x3 matrix -> dst-to-src map, in double, in OpenCV's order of operations"""
    M = [float(v) for v in np.asarray(m, dtype=np.float64).reshape(6)]
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M


def _cv_round(v):
    """saturate_cast<int>(double): round half to even, saturated to int32"""
    return np.clip(np.rint(v), -2147483648.0, 2147483647.0).astype(np.int64)


def warp_affine_u8(src, m, size):
    """cv2.warpAffine(src, m, (size, size)) for an HxWxC uint8 image, INTER_LINEAR, BORDER_CONSTANT 0 (header)."""
    src = np.asarray(src)
    assert src.dtype == np.uint8 and src.ndim == 3
    h, w, _ = src.shape
    M = invert_affine(m)
    xs = np.arange(size, dtype=np.float64)
    adelta, bdelta = _cv_round(M[0] * xs * 1024.0), _cv_round(M[3] * xs * 1024.0)
    x0 = _cv_round((M[1] * xs + M[2]) * 1024.0) + 16                       # per output row y (xs doubles as y here)
    y0 = _cv_round((M[4] * xs + M[5]) * 1024.0) + 16
    # int32 wrap of the sums is not reachable for coordinates of an image; keep int64
    X = (x0[:, None] + adelta[None, :]) >> 5
    Y = (y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)  # saturate_cast<short>
    fx, fy = (X & 31).astype(np.int64), (Y & 31).astype(np.int64)
    acc = np.zeros((size, size, src.shape[2]), dtype=np.int64)
    for dy in (0, 1):
        for dx in (0, 1):
            wgt = ((fx if dx else 32 - fx) * (fy if dy else 32 - fy) * 32)[..., None]
            yy, xx = sy + dy, sx + dx
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            tap = np.where(ok[..., None], src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64), 0)
            acc += wgt * tap
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def crop_align(landmarks, images, size=224, return_ldm5=False):
    """FasterCropAlignXRay(size)(landmarks, images), jitter off (faster_crop_align_xray.py:21-88).
    landmarks: per frame (_, ldm5 (5,2), ldm68 (68,2), big box (x0,y0,x1,y1)); images: per frame HxWx3 uint8 crops."""
    std_points = STD_POINTS_256 * size / 256.0
    landmarks = [lm[:4] for lm in landmarks]
    ori_boxes = np.array([b for _, _, _, b in landmarks])
    five = np.array([l5 for _, l5, _, _ in landmarks])
    l68 = np.array([l for _, _, l, _ in landmarks])
    left_top = ori_boxes[:, :2].min(0)
    right_bottom = ori_boxes[:, 2:].max(0)
    w, h = right_bottom - left_top
    diff = ori_boxes[:, :2] - left_top[None, ...]
    new_five = five + diff[:, None, :]
    new_68 = l68 + diff[:, None, :]
    tfm, trans = estimate_batch_transform(new_five.copy(), std_points)
    t68 = np.array([transform_landmarks(l, trans) for l in new_68])
    t5 = np.array([transform_landmarks(l, trans) for l in new_five])
    if images is None:
        return (t5, t68) if return_ldm5 else t68
    out = []
    for image, d in zip(images, diff):
        canvas = np.zeros((h, w, 3), dtype=np.uint8)
        x, y = d
        ih, iw, _ = image.shape
        canvas[y:y + ih, x:x + iw] = image
        out.append(warp_affine_u8(canvas, tfm, size))
    out = np.stack(out)
    return (t5, t68, out) if return_ldm5 else (t68, out)
