"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- CPU restatement of the C3D VIDEO_DATA layer.

Follows feature_extration.prototxt:3-21 (new_height 128, new_width 171, crop_size 112, mirror false,
new_length 16, mean_file) of /root/reference/C3D/C3D-v1.0/examples/c3d_feature_extraction/
hollywood_feature_extraction/ and the window list of extract_C3D_features.py:667-684,866.

Parity status: UNPINNED for the resize.  The layer itself lives in the facebook/C3D v1.0 Caffe fork
(not vendored; OpenCV `cv::resize(INTER_LINEAR)` + test-phase centre crop + mean-cube subtraction).
cv2 is not installed here, so OpenCV's 11-bit fixed-point interpolation is restated in float and
rounded to the nearest 8-bit level; it can differ from OpenCV by one level at a few pixels.  When
the frames already are 128x171 the resize is the identity and the restatement is exact.
"""
import numpy as np

RH, RW, CROP, LENGTH = 128, 171, 112, 16


def resize_bilinear_u8(frame, out_h=RH, out_w=RW):
    """frame [H,W,C] uint8 -> [out_h,out_w,C] float32 holding integer levels (half-pixel centres,
    edge clamp, round half up)."""
    f = np.asarray(frame)
    h, w = f.shape[:2]
    sy, sx = np.float32(h) / np.float32(out_h), np.float32(w) / np.float32(out_w)
    fy = np.clip((np.arange(out_h, dtype=np.float32) + np.float32(0.5)) * sy - np.float32(0.5), 0, h - 1).astype(np.float32)
    fx = np.clip((np.arange(out_w, dtype=np.float32) + np.float32(0.5)) * sx - np.float32(0.5), 0, w - 1).astype(np.float32)
    y0, x0 = fy.astype(np.int64), fx.astype(np.int64)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    wy, wx = (fy - y0.astype(np.float32))[:, None, None], (fx - x0.astype(np.float32))[None, :, None]
    g = f.astype(np.float32)
    top = g[y0][:, x0] + wx * (g[y0][:, x1] - g[y0][:, x0])
    bot = g[y1][:, x0] + wx * (g[y1][:, x1] - g[y1][:, x0])
    return np.floor(top + wy * (bot - top) + np.float32(0.5)).astype(np.float32)


def video_data_layer(frames, window_starts, mean_cube=None):
    """frames [N,H,W,3] uint8, mean_cube [3,16,128,171] -> video [n,16,112,112,3] float32
    (channels last; Caffe's own blob is [n,3,16,112,112])."""
    oy, ox = (RH - CROP) // 2, (RW - CROP) // 2
    out = np.zeros((len(window_starts), LENGTH, CROP, CROP, 3), np.float32)
    for i, s in enumerate(window_starts):
        for z in range(LENGTH):
            r = resize_bilinear_u8(frames[s + z])[oy:oy + CROP, ox:ox + CROP]
            if mean_cube is not None:
                r = r - np.transpose(mean_cube[:, z, oy:oy + CROP, ox:ox + CROP], (1, 2, 0))
            out[i, z] = r
    return out
