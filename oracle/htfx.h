/*
 * htfx.h -- tiny named-array container used for golden fixtures ("HTFX").
 *
 * TEST INFRASTRUCTURE ONLY (oracle/): used by oracle/ref_harness.cpp (writer) and
 * by the C oracle / Python tests (reader, see tests/htfx.py).  Not part of the product.
 *
 * Layout (little endian):
 *   char  magic[8] = "HTFX0001"
 *   u32   count
 *   repeat count times:
 *     char name[48]   (NUL padded)
 *     u32  dtype      0=f32 1=i32 2=u16 3=u8
 *     u32  ndim       (<=4)
 *     u32  dims[4]
 *     u64  nbytes
 *     u8   data[nbytes], then zero padding to a multiple of 8 bytes
 */
#ifndef HTFX_H
#define HTFX_H
#include <stdint.h>
#include <stdio.h>
#include <string.h>

enum { HTFX_F32 = 0, HTFX_I32 = 1, HTFX_U16 = 2, HTFX_U8 = 3 };

typedef struct htfx_writer { FILE *f; uint32_t count; } htfx_writer;

static inline int htfx_open(htfx_writer *w, const char *path)
{
	w->f = fopen(path, "wb");
	w->count = 0;
	if (!w->f) return -1;
	fwrite("HTFX0001", 1, 8, w->f);
	fwrite(&w->count, 4, 1, w->f);
	return 0;
}
static inline size_t htfx_elsize(uint32_t dtype) { return dtype == HTFX_F32 || dtype == HTFX_I32 ? 4 : dtype == HTFX_U16 ? 2 : 1; }
static inline void htfx_put(htfx_writer *w, const char *name, uint32_t dtype, uint32_t ndim, const uint32_t *dims, const void *data)
{
	char nm[48];
	uint32_t d[4] = { 1, 1, 1, 1 };
	uint64_t n = 1, pad = 0, zero = 0;
	memset(nm, 0, sizeof nm);
	strncpy(nm, name, 47);
	for (uint32_t i = 0; i < ndim && i < 4; i++) { d[i] = dims[i]; n *= dims[i]; }
	n *= htfx_elsize(dtype);
	fwrite(nm, 1, 48, w->f);
	fwrite(&dtype, 4, 1, w->f);
	fwrite(&ndim, 4, 1, w->f);
	fwrite(d, 4, 4, w->f);
	fwrite(&n, 8, 1, w->f);
	if (n) fwrite(data, 1, n, w->f);
	pad = (8 - (n & 7)) & 7;
	if (pad) fwrite(&zero, 1, pad, w->f);
	w->count++;
}
static inline void htfx_close(htfx_writer *w)
{
	fseek(w->f, 8, SEEK_SET);
	fwrite(&w->count, 4, 1, w->f);
	fclose(w->f);
	w->f = NULL;
}
#endif
