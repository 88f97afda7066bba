/*
 * rt_image.c -- the frame container the renderer writes into and its Windows-bitmap writer.
 *
 * Counterpart of the reference's image module (reference: src/utils/image.c:5-23 create/destroy,
 * :25-66 bitmap_save_image).  The file is byte-identical to the reference's for the same pixels:
 * a 14-byte file header and a 40-byte BITMAPINFOHEADER (32 bits per pixel, no compression, a
 * POSITIVE height, zero resolution), then the rows in the order y = height-1 ... 0 of the buffer
 * -- that is what the reference writes, so with a bottom-up bitmap the buffer's top row ends up
 * at the top of the picture -- each pixel turned from 0xAABBGGRR into 0xAARRGGBB.
 * Headers are serialised field by field in little-endian order, so the writer does not depend
 * on structure packing or on the host's byte order.
 */
#include "rt_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

Image* rt_image_create(uint32_t width, uint32_t height) {
    Image* image = malloc(sizeof(Image));
    if (!image) return NULL;
    image->width = width;
    image->height = height;
    image->bufferSize = (uint32_t)(sizeof(uint32_t) * width * height);
    image->buffer = malloc(sizeof(uint32_t) * (size_t)width * height + 1);   /* + 1: a 0 x 0 image still owns a buffer */
    if (!image->buffer) {
        free(image);
        return NULL;
    }
    return image;
}

void rt_image_destroy(Image* image) {
    if (!image) return;
    free(image->buffer);
    free(image);
}

static unsigned char* put16(unsigned char* at, uint32_t v) { at[0] = (unsigned char)v; at[1] = (unsigned char)(v >> 8); return at + 2; }
static unsigned char* put32(unsigned char* at, uint32_t v) { at = put16(at, v & 0xFFFFu); return put16(at, v >> 16); }

bool rt_bitmap_save_image(const char* path, const Image* image) {
    if (!path || !image || (!image->buffer && image->width * image->height != 0)) return false;
    enum { FILE_HEADER = 14, INFO_HEADER = 40 };
    const size_t pixels = (size_t)image->width * image->height;
    unsigned char* out = malloc(FILE_HEADER + INFO_HEADER + pixels * 4 + 1);
    if (!out) return false;
    unsigned char* at = out;
    at = put16(at, 0x4D42u);                                        /* "BM" */
    at = put32(at, (uint32_t)(FILE_HEADER + INFO_HEADER) + image->bufferSize);
    at = put32(at, 0u);
    at = put32(at, FILE_HEADER + INFO_HEADER);
    at = put32(at, INFO_HEADER);
    at = put32(at, image->width);
    at = put32(at, image->height);
    at = put16(at, 1u);                                             /* planes */
    at = put16(at, 32u);                                            /* bits per pixel */
    at = put32(at, 0u);                                             /* BI_RGB */
    at = put32(at, image->bufferSize);
    at = put32(at, 0u); at = put32(at, 0u);                         /* pixels per metre */
    at = put32(at, 0u); at = put32(at, 0u);                         /* palette */
    for (uint32_t row = image->height; row-- > 0;) {
        const uint32_t* src = image->buffer + (size_t)row * image->width;
        for (uint32_t x = 0; x < image->width; x++) {
            const uint32_t rgba = src[x];
            at = put32(at, (rgba & 0xFF00FF00u) | ((rgba >> 16) & 0xFFu) | ((rgba & 0xFFu) << 16));
        }
    }
    FILE* file = fopen(path, "wb");
    bool ok = file != NULL;
    if (ok) {
        ok = fwrite(out, 1, (size_t)(at - out), file) == (size_t)(at - out);
        ok = fclose(file) == 0 && ok;
    }
    free(out);
    return ok;
}
