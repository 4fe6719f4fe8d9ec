// Clip aligner warp: the per-frame
//     new_image = zeros(h, w, 3); new_image[y:y+ih, x:x+iw] = crop; cv2.warpAffine(new_image, tfm, (size, size))
// loop of FasterCropAlignXRay.process_single (reference altfreezing/test_tools/faster_crop_align_xray.py:75-88),
// for all frames of a clip in one launch, straight from the uploaded crops (the zero canvas is never built: a tap
// outside the pasted crop is 0 whether it falls on the canvas or beyond it - BORDER_CONSTANT 0).
//
// Arithmetic: OpenCV's fixed-point bilinear warp (imgwarp.cpp warpAffine / WarpAffineInvoker / remapBilinear, 3.x-4.10):
// the forward matrix is inverted in double on the host in OpenCV's order of operations; a destination pixel's source
// coordinate is X = (cvRound((M1*y + M2)*1024) + 16 + cvRound(M0*x*1024)) >> 5 in 1/32 pixel (doubles, no FMA
// contraction: __dmul_rn / __dadd_rn), weights (32-fx)(32-fy)*32 etc. (15-bit, sum 32768), dst = (sum + 16384) >> 15.
// Integer work: bit-exact against oracle/aligner_oracle.py; against cv2 itself the parity is unpinned (cv2 is absent
// from the build image and the reference holds no aligned frame).
// HBM-bound byte work: one thread per destination pixel, 4 x 3 source bytes in, 3 bytes out; a clip is ~5 MB.
#include "af_common.h"
#include <cmath>
#include <string.h>

namespace af {

struct AlignArgs {
    const unsigned char* crops;
    unsigned char* out;
    double m[6];                 // dst -> src map (already inverted)
    int size, n;
    int canvas_h, canvas_w;
    af_align_frame f[AF_ALIGN_MAX_FRAMES];
};

__device__ __forceinline__ int cv_round_sat(double v) {       // saturate_cast<int>(double): round half to even, saturated
    const double r = rint(v);
    return r <= -2147483648.0 ? (int)0x80000000 : r >= 2147483647.0 ? 0x7fffffff : (int)r;
}

__global__ __launch_bounds__(256) void warp_affine_clip_kernel(const AlignArgs a) {
    const int p = blockIdx.x * 256 + threadIdx.x, fr = blockIdx.y;
    if (p >= a.size * a.size) return;
    const int y = p / a.size, x = p - y * a.size;
    const af_align_frame f = a.f[fr];
    const int adelta = cv_round_sat(__dmul_rn(__dmul_rn(a.m[0], (double)x), 1024.0));
    const int bdelta = cv_round_sat(__dmul_rn(__dmul_rn(a.m[3], (double)x), 1024.0));
    const int X0 = cv_round_sat(__dmul_rn(__dadd_rn(__dmul_rn(a.m[1], (double)y), a.m[2]), 1024.0)) + 16;
    const int Y0 = cv_round_sat(__dmul_rn(__dadd_rn(__dmul_rn(a.m[4], (double)y), a.m[5]), 1024.0)) + 16;
    const long long X = ((long long)X0 + adelta) >> 5, Y = ((long long)Y0 + bdelta) >> 5;
    long long sxl = X >> 5, syl = Y >> 5;
    const int sx = (int)(sxl < -32768 ? -32768 : sxl > 32767 ? 32767 : sxl);      // saturate_cast<short>
    const int sy = (int)(syl < -32768 ? -32768 : syl > 32767 ? 32767 : syl);
    const int fx = (int)(X & 31), fy = (int)(Y & 31);
    int acc[3] = {0, 0, 0};
    const unsigned char* img = a.crops + f.offset;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int cx = sx + dx, cy = sy + dy;                     // canvas coordinates of the tap
            const int ix = cx - f.x, iy = cy - f.y;                   // crop coordinates
            if ((unsigned)cx < (unsigned)a.canvas_w && (unsigned)cy < (unsigned)a.canvas_h &&
                (unsigned)ix < (unsigned)f.iw && (unsigned)iy < (unsigned)f.ih) {
                const int w = (dx ? fx : 32 - fx) * (dy ? fy : 32 - fy) * 32;
                const unsigned char* s = img + ((long long)iy * f.iw + ix) * 3;
                acc[0] += w * s[0]; acc[1] += w * s[1]; acc[2] += w * s[2];
            }
        }
    unsigned char* o = a.out + (((long long)fr * a.size + y) * a.size + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int v = (acc[c] + (1 << 14)) >> 15;
        o[c] = (unsigned char)(v > 255 ? 255 : v);
    }
}

}  // namespace af

extern "C" int af_warp_affine_clip_u8(const void* crops, const af_align_frame* frames, int n_frames, int canvas_h, int canvas_w,
                                      const double* tfm, int size, void* out, void* stream) {
    using namespace af;
    AF_REQUIRE(crops && frames && tfm && out, "warp_affine_clip: null argument");
    AF_REQUIRE(n_frames >= 0 && n_frames <= AF_ALIGN_MAX_FRAMES, "warp_affine_clip: %d frames (at most %d per call)", n_frames, AF_ALIGN_MAX_FRAMES);
    AF_REQUIRE(size > 0 && size <= 4096 && canvas_h > 0 && canvas_w > 0 && canvas_h <= 32767 && canvas_w <= 32767,
               "warp_affine_clip: bad size %d / canvas %dx%d", size, canvas_h, canvas_w);
    if (n_frames == 0) return AF_OK;
    AlignArgs a;
    a.crops = (const unsigned char*)crops; a.out = (unsigned char*)out; a.size = size; a.n = n_frames;
    a.canvas_h = canvas_h; a.canvas_w = canvas_w;
    for (int i = 0; i < n_frames; ++i) {
        const af_align_frame& f = frames[i];
        // the reference pastes with new_image[y:y+ih, x:x+iw] = image, which numpy refuses unless the crop fits the canvas
        AF_REQUIRE(f.offset >= 0 && f.ih > 0 && f.iw > 0 && f.x >= 0 && f.y >= 0 && f.x + f.iw <= canvas_w && f.y + f.ih <= canvas_h,
                   "warp_affine_clip: frame %d (%dx%d at %d,%d) does not fit the %dx%d canvas", i, f.iw, f.ih, f.x, f.y, canvas_w, canvas_h);
        a.f[i] = f;
    }
    // warpAffine(): forward 2x3 matrix -> dst-to-src map, in double, same order of operations (volatile: no contraction)
    volatile double M[6];
    for (int i = 0; i < 6; ++i) M[i] = tfm[i];
    volatile double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1.0 / D : 0.0;
    volatile double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] = M[1] * -D; M[3] = M[3] * -D; M[4] = A22;
    volatile double t0 = -M[0] * M[2], t1 = M[1] * M[5], t2 = -M[3] * M[2], t3 = M[4] * M[5];
    volatile double b1 = t0 - t1, b2 = t2 - t3;
    M[2] = b1; M[5] = b2;
    for (int i = 0; i < 6; ++i) a.m[i] = M[i];
    const dim3 grid((unsigned)((size * size + 255) / 256), (unsigned)n_frames);
    hipLaunchKernelGGL(warp_affine_clip_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    AF_CHECK_LAUNCH("warp_affine_clip_kernel");
    return AF_OK;
}

// Host helper of the aligner: copies n rectangles of uint8 rows into one (pinned) staging buffer - memcpy per row, no device work.
// The Python caller cut each crop to the rows / columns the warp can sample; numpy copies such a strided view with ~100 ns of
// iterator overhead per row (8 000 rows per clip: half of the aligner call).  ctypes releases the GIL for the call.
extern "C" int af_align_plan_u8(const af_align_crop* crops, int n, int canvas_h, int canvas_w, const double* tfm, int size,
                                af_stage_rect* rects, af_align_frame* frames, int64_t* total_bytes, int32_t* bad_frame) {
    using namespace af;
    AF_REQUIRE(crops && n >= 0 && tfm && size > 0 && rects && frames && total_bytes, "align_plan_u8: bad argument");
    if (bad_frame) *bad_frame = -1;
    // the rectangle of the canvas the destination square can sample: dst = M [x y 1]^T  ->  src = M^-1 (dst - t) at its four corners
    const double det = tfm[0] * tfm[4] - tfm[1] * tfm[3];
    bool cut = std::isfinite(det) && std::fabs(det) >= 1e-12;      // singular map: OpenCV's D = 0 path samples around one point; keep everything
    long long ylo = 0, yhi = 0;
    if (cut) {
        const double s = (double)(size - 1), cx[4] = {0.0, s, 0.0, s}, cy[4] = {0.0, 0.0, s, s};
        double ymin = 0.0, ymax = 0.0;
        for (int k = 0; k < 4; ++k) {
            const double dx = cx[k] - tfm[2], dy = cy[k] - tfm[5];
            const double xs = (tfm[4] * dx - tfm[1] * dy) / det, ys = (-tfm[3] * dx + tfm[0] * dy) / det;
            if (!std::isfinite(xs) || !std::isfinite(ys)) { cut = false; break; }
            if (k == 0 || ys < ymin) ymin = ys;
            if (k == 0 || ys > ymax) ymax = ys;
        }
        if (cut) { ylo = (long long)std::floor(ymin) - 3; yhi = (long long)std::ceil(ymax) + 4; }
    }
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
        const af_align_crop& c = crops[i];
        AF_REQUIRE(c.src && c.h > 0 && c.w > 0 && c.pitch >= (int64_t)c.w * 3, "align_plan_u8: bad crop %d", i);
        if (c.x < 0 || c.y < 0 || (long long)c.x + c.w > canvas_w || (long long)c.y + c.h > canvas_h) {
            if (bad_frame) *bad_frame = i;
            return set_error(AF_ERR_ARG, "aligner: frame %d (%dx%d at %d,%d) does not fit the %dx%d canvas", i, c.w, c.h, c.x, c.y, canvas_w, canvas_h);
        }
        long long r0 = 0, r1 = c.h;
        if (cut) {
            r0 = ylo - c.y > 0 ? ylo - c.y : 0;
            r1 = yhi - c.y < c.h ? yhi - c.y : c.h;
        }
        int iw = c.w;
        if (r1 <= r0) { r0 = 0; r1 = 1; iw = 1; }                   // the warp never reaches this crop: one pixel keeps the frame table valid
        rects[i].src = (const char*)c.src + r0 * c.pitch;
        rects[i].dst_offset = total;
        rects[i].src_pitch = c.pitch;
        rects[i].rows = (int32_t)(r1 - r0);
        rects[i].row_bytes = iw * 3;
        if (rects[i].rows == 1) rects[i].src_pitch = rects[i].row_bytes;
        frames[i].offset = total;
        frames[i].ih = (int32_t)(r1 - r0);
        frames[i].iw = iw;
        frames[i].x = c.x;
        frames[i].y = (int32_t)(c.y + r0);
        total += ((int64_t)(r1 - r0) * iw * 3 + 15) / 16 * 16;
    }
    *total_bytes = total;
    return AF_OK;
}

extern "C" int af_stage_rows_u8(void* dst, const af_stage_rect* rects, int n) {
    using namespace af;
    AF_REQUIRE(dst && rects && n >= 0, "stage_rows_u8: bad argument");
    for (int i = 0; i < n; ++i) {
        const af_stage_rect& r = rects[i];
        AF_REQUIRE(r.src && r.rows >= 0 && r.row_bytes >= 0 && r.src_pitch >= r.row_bytes && r.dst_offset >= 0, "stage_rows_u8: bad rectangle %d", i);
        char* d = (char*)dst + r.dst_offset;
        const char* s_ = (const char*)r.src;
        if (r.src_pitch == r.row_bytes) memcpy(d, s_, (size_t)r.rows * r.row_bytes);
        else for (int y = 0; y < r.rows; ++y) memcpy(d + (size_t)y * r.row_bytes, s_ + (size_t)y * r.src_pitch, (size_t)r.row_bytes);
    }
    return AF_OK;
}
