"""Decode the reference's sample frame (data/1023.jpg, 1920x1080 RGB) and store
its grayscale version as a lossless PNG fixture.

Gray uses OpenCV's 8-bit fixed-point BGR2GRAY weights
(R*4899 + G*9617 + B*1868 + 8192) >> 14, the conversion the reference's
pipeline would apply before feature detection.  Run in the build container:

    python tests/golden/make_frame_fixture.py /root/reference/data/1023.jpg
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def main(path):
    rgb = np.asarray(Image.open(path).convert("RGB")).astype(np.int32)
    gray = (rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14
    gray = np.clip(gray, 0, 255).astype(np.uint8)
    Image.fromarray(gray, mode="L").save(os.path.join(HERE, "frame_1023_gray.png"), optimize=True)
    print(gray.shape, gray.dtype, int(gray.sum()))


if __name__ == "__main__":
    main(sys.argv[1])
