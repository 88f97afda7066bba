"""The frame container and the bitmap writer against the files the reference's own writer produced
for the same pixels (tests/golden/bitmaps.npz, written by tests/golden/make_golden.py from
reference: src/utils/image.c:25-66)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pixels_for(width, height):
    idx = np.arange(width * height, dtype=np.uint64)
    return ((idx * 2654435761) & 0xFFFFFFFF).astype(np.uint32)


@pytest.mark.parametrize("size", [(1, 1), (5, 3), (16, 9), (33, 2)])
def test_bitmap_file_is_the_reference_file(host, tmp_path, size):
    width, height = size
    golden = np.load(os.path.join(GOLDEN, "bitmaps.npz"))[f"bmp_{width}x{height}"]
    image = host.lib.rt_image_create(width, height)
    assert image and image.contents.bufferSize == width * height * 4
    pixels = pixels_for(width, height)
    C.memmove(image.contents.buffer, pixels.ctypes.data, pixels.nbytes)
    path = tmp_path / "frame.bmp"
    assert host.lib.rt_bitmap_save_image(str(path).encode(), image)
    data = path.read_bytes()
    assert data == golden.tobytes()
    # what the bytes mean: "BM", 54-byte headers, bottom-up rows written from the buffer's LAST row
    magic, file_size, _, offset = struct.unpack("<HIII", data[:14])
    assert (magic, file_size, offset) == (0x4D42, 54 + width * height * 4, 54)
    first = struct.unpack("<I", data[54:58])[0]
    src = int(pixels[(height - 1) * width])
    assert first == (src & 0xFF00FF00) | ((src >> 16) & 0xFF) | ((src & 0xFF) << 16)
    host.lib.rt_image_destroy(image)


def test_unwritable_path_is_reported(host, tmp_path):
    image = host.lib.rt_image_create(2, 2)
    assert not host.lib.rt_bitmap_save_image(str(tmp_path / "no_such_dir" / "frame.bmp").encode(), image)
    host.lib.rt_image_destroy(image)
    host.lib.rt_image_destroy(None)
