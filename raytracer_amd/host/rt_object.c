/*
 * rt_object.c -- Wavefront OBJ ingestion: a triangle soup from the `v` and `f` records of a
 * file, and the three bulk edits the reference applies before adding an object to a scene.
 *
 * Counterpart of the reference's loader (reference: src/object.c:145-185 object_loadFromFile,
 * :187-210 scale / translate / materialIndex, :212-217 destroy; vertex ids are 1-based,
 * src/vertextable.c:19-22).  Given the same file it produces the same triangles, bit for bit:
 *   - a record is recognised by the first non-blank character of a line: `v` (but not `vn`, `vt`)
 *     is a position, `f` a face, everything else is skipped;
 *   - a coordinate is the longest run of the characters 0-9 . - e, converted with the C library
 *     (so "1.5e+3" reads as 1.5 followed by two empty numbers, as it does there);
 *   - a face index is the leading run of digits of each blank-separated token ("7/2/5" is 7);
 *   - a face of n corners becomes the fan (c0, c1, c2), (c0, c2, c3), ...; material index 0.
 * Where the reference asserts or reads out of bounds (a missing file, an index of 0 or beyond
 * the vertices read so far), this loader returns NULL and leaves a message for rt_object_lastError.
 */
#include "rt_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static _Thread_local char rt_object_error[160];

const char* rt_object_lastError(void) { return rt_object_error; }

typedef struct { Vec3* at; size_t count, room; } Positions;

static int is_blank(char c) { return c == ' ' || c == '\t'; }
static int is_line_end(char c) { return c == '\r' || c == '\n' || c == '\0'; }

/* the characters the reference accepts inside a coordinate (reference: src/object.c:55-57) */
static int is_coordinate_char(char c) { return (c >= '0' && c <= '9') || c == '.' || c == '-' || c == 'e'; }

/* Skips blanks AND line breaks: the reference does so between the coordinates of a vertex. */
static const char* skip_space(const char* p) {
    while (*p == ' ' || *p == '\t' || *p == '\n') p++;
    return p;
}

static float read_coordinate(const char** cursor) {
    const char* p = *cursor;
    char text[255];
    size_t n = 0;
    while (is_coordinate_char(*p)) {
        if (n < sizeof text - 1) text[n++] = *p;
        p++;
    }
    text[n] = '\0';
    *cursor = p;
    return (float)atof(text);
}

static uint32_t read_index(const char** cursor) {
    const char* p = *cursor;
    char text[255];
    size_t n = 0;
    while (*p >= '0' && *p <= '9') {
        if (n < sizeof text - 1) text[n++] = *p;
        p++;
    }
    text[n] = '\0';
    /* the rest of the token (texture and normal references) is not used */
    while (*p != '\0' && *p != '\n' && !is_blank(*p)) p++;
    while (is_blank(*p)) p++;
    *cursor = p;
    return (uint32_t)atoi(text);
}

static int push_position(Positions* v, Vec3 p) {
    if (v->count == v->room) {
        size_t room = v->room ? v->room * 2 : 256;
        Vec3* grown = realloc(v->at, room * sizeof(Vec3));
        if (!grown) return 0;
        v->at = grown;
        v->room = room;
    }
    v->at[v->count++] = p;
    return 1;
}

static int push_triangle(Object* o, Triangle t) {
    if (o->triangleCount == o->capacity) {
        uint32_t room = o->capacity ? o->capacity * 2 : 256;
        Triangle* grown = realloc(o->triangles, (size_t)room * sizeof(Triangle));
        if (!grown) return 0;
        o->triangles = grown;
        o->capacity = room;
    }
    o->triangles[o->triangleCount++] = t;
    return 1;
}

static int corner(const Positions* v, uint32_t id, Vec3* out, size_t line) {
    if (id == 0 || id > v->count) {
        snprintf(rt_object_error, sizeof rt_object_error, "line %zu: face refers to vertex %u of %zu", line, id, v->count);
        return 0;
    }
    *out = v->at[id - 1];
    return 1;
}

static char* read_whole_file(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    char* text = NULL;
    if (fseek(f, 0, SEEK_END) == 0) {
        long size = ftell(f);
        if (size >= 0 && fseek(f, 0, SEEK_SET) == 0 && (text = malloc((size_t)size + 1)) != NULL) {
            size_t got = fread(text, 1, (size_t)size, f);
            text[got] = '\0';
        }
    }
    fclose(f);
    return text;
}

Object* rt_object_loadFromFile(const char* filepath) {
    rt_object_error[0] = '\0';
    char* text = filepath ? read_whole_file(filepath) : NULL;
    if (!text) {
        snprintf(rt_object_error, sizeof rt_object_error, "cannot read %s", filepath ? filepath : "(null)");
        return NULL;
    }
    Object* object = calloc(1, sizeof(Object));
    Positions positions = {NULL, 0, 0};
    int ok = object != NULL;
    size_t line = 1;
    const char* p = text;
    while (ok && *p != '\0') {
        /* blank lines and leading blanks belong to no record */
        while (*p == ' ' || *p == '\t' || *p == '\n') { if (*p == '\n') line++; p++; }
        if (*p == 'v' && p[1] != 'n' && p[1] != 't') {
            Vec3 position;
            p = skip_space(p + 1);
            position.x = read_coordinate(&p);
            p = skip_space(p);
            position.y = read_coordinate(&p);
            p = skip_space(p);
            position.z = read_coordinate(&p);
            ok = push_position(&positions, position);
            if (!ok) snprintf(rt_object_error, sizeof rt_object_error, "out of memory");
        } else if (*p == 'f') {
            Vec3 first, previous, current;
            p = skip_space(p + 1);
            ok = corner(&positions, read_index(&p), &first, line);
            if (ok) { p = skip_space(p); ok = corner(&positions, read_index(&p), &previous, line); }
            while (ok && !is_line_end(*p)) {
                p = skip_space(p);
                ok = corner(&positions, read_index(&p), &current, line);
                if (!ok) break;
                Triangle t;
                t.materialIndex = 0;
                t.v0 = first; t.v1 = previous; t.v2 = current;
                ok = push_triangle(object, t);
                if (!ok) snprintf(rt_object_error, sizeof rt_object_error, "out of memory");
                previous = current;
            }
        }
        /* whatever is left of the line is not part of the record */
        while (*p != '\0' && *p != '\n') p++;
        if (*p == '\n') { p++; line++; }
    }
    free(positions.at);
    free(text);
    if (!ok) {
        rt_object_destroy(object);
        return NULL;
    }
    return object;
}

void rt_object_scale(Object* object, float factor) {
    for (uint32_t i = 0; i < object->triangleCount; i++) {
        Triangle* t = &object->triangles[i];
        Vec3* corners[3] = {&t->v0, &t->v1, &t->v2};
        for (int k = 0; k < 3; k++) {
            corners[k]->x = corners[k]->x * factor;
            corners[k]->y = corners[k]->y * factor;
            corners[k]->z = corners[k]->z * factor;
        }
    }
}

void rt_object_translate(Object* object, Vec3 translation) {
    for (uint32_t i = 0; i < object->triangleCount; i++) {
        Triangle* t = &object->triangles[i];
        Vec3* corners[3] = {&t->v0, &t->v1, &t->v2};
        for (int k = 0; k < 3; k++) {
            corners[k]->x = corners[k]->x + translation.x;
            corners[k]->y = corners[k]->y + translation.y;
            corners[k]->z = corners[k]->z + translation.z;
        }
    }
}

void rt_object_materialIndex(Object* object, uint32_t materialIndex) {
    for (uint32_t i = 0; i < object->triangleCount; i++) object->triangles[i].materialIndex = materialIndex;
}

void rt_object_destroy(Object* object) {
    if (!object) return;
    free(object->triangles);
    free(object);
}
