"""Regression thumbnails: the oracle's RGBA8 frames of BASELINE configs 2 and 5 at thumbnail size,
stored as PNG so that a diff is something one can look at.  Linear scan, counter RNG, seed 2025.

    python tests/golden/make_thumbnails.py          # writes tests/golden/thumb_config2.png, thumb_config5.png

PNG codec: 8-bit RGB, zlib only (write_png / read_png below; tests import read_png).
"""
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SEED = 2025
HERE = os.path.dirname(os.path.abspath(__file__))


def thumbs():
    """name -> (objs, mats, camera kwargs)"""
    from tests.scenes import cornell_like
    from oracle import binding as ob
    rtow = ob.make_world_spheres(12345)
    cobjs, cmats, ckw = cornell_like()
    ckw = dict(ckw, image_width=128, samples_per_pixel=64)
    return {"thumb_config2": (rtow[0], rtow[1], dict(image_width=240, samples_per_pixel=32, max_depth=50)),
            "thumb_config5": (cobjs, cmats, ckw)}


def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def read_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 2)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert not raw[:, 0].any()  # filter type 0 on every scanline
    return raw[:, 1:].reshape(h, w, 3).copy()


def rgba_to_rgb(rgba):
    """0xAABBGGRR (RGBAColor, color.hpp:19-27) -> h x w x 3 bytes"""
    return np.stack([(rgba >> s) & 0xff for s in (0, 8, 16)], axis=-1).astype(np.uint8)


def main():
    from oracle import binding as ob
    for name, (objs, mats, kw) in thumbs().items():
        cam = ob.camera_setup(ob.camera_params(**kw))
        _, rgba = ob.render_rect_counter(cam, objs, mats, SEED, 0, 0, cam.img_width, cam.img_height,
                                         nthreads=int(os.environ.get("ORC_THREADS", "8")))
        path = os.path.join(HERE, name + ".png")
        write_png(path, rgba_to_rgb(rgba))
        assert np.array_equal(read_png(path), rgba_to_rgb(rgba))
        print(path, rgba.shape, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
