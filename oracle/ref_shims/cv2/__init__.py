"""Import shim for OpenCV, used ONLY when the reference is imported in the build container
to generate golden vectors (tests/golden/make_golden.py).  cv2 is not installed there.

busca/tracking.py:1 imports cv2 and :71 calls cv2.resize(cutout, (w, h), interpolation=INTER_LINEAR).
The resize below is this repo's own restatement of OpenCV's 8-bit fixed-point bilinear resize
(oracle.geometry.resize_linear_u8); it is third-party arithmetic that the reference does not pin.
"""
INTER_LINEAR = 1


def resize(src, dsize, interpolation=INTER_LINEAR):
    from oracle.geometry import resize_linear_u8
    assert interpolation == INTER_LINEAR
    return resize_linear_u8(src, int(dsize[0]), int(dsize[1]))
