/*
 * ho_segment.c -- CPU restatement of the reference's hand segmentation (the step before the tracker for frames that are
 * not already 64x64).  TEST INFRASTRUCTURE ONLY: see ht_oracle.h.
 *
 *   HandSegmentVR                 include/handtrack.h:280-344
 *   DownSampleMin, camera / 2     include/misc_image.h:81-94,60,136
 *   Threshold, DistanceTransform  include/misc_image.h:179-195
 *   SampleD                       include/misc_image.h:154-162
 *   DCamera::deprojectz/projectz  include/misc_image.h:48-50
 *   quat_from_to, QuatFromAxisAngle  third_party/geometric.h:319-328,102
 *
 * atan2 at handtrack.h:326 is the unqualified C function (double); std::sin / std::cos in QuatFromAxisAngle are the float overloads.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "ht_oracle.h"

static f3 deprojectz(f2 focal, f2 principal, float px, float py, float d)      /* misc_image.h:48 */
{
	return scale3(F3((px - principal.x) / focal.x, (py - principal.y) / focal.y, 1.0f), d);
}

/* depth: [h][w]; cam12 = fx fy px py depth_scale pos3 quat4 (the pose of the input camera is not used by the reference).
 * tile: 64*64; camout12 likewise.  small_out (w/4*h/4 u16) and dt_out (same size, u8) may be NULL.  Returns 0, or 1 for the
 * pass-through case (input already 64x64: tile/camout are copies of the input). */
int ho_segment_vr(const uint16_t *depth, int w, int h, const float *cam12, int entry_options, float wrange_lo, float wrange_hi, float diam,
                  uint16_t *tile, float *camout12, uint16_t *small_out, unsigned char *dt_out)
{
	(void)wrange_lo;
	if (w == 64 && h == 64) { memcpy(tile, depth, 4096 * sizeof(uint16_t)); memcpy(camout12, cam12, 12 * sizeof(float)); return 1; }
	const f2 focal = { cam12[0], cam12[1] }, principal = { cam12[2], cam12[3] };
	const float depth_scale = cam12[4];
	/* two 2x2 min poolings; the camera is halved twice (misc_image.h:60) */
	const int w1 = w / 2, h1 = h / 2, sw = w1 / 2, sh = h1 / 2;
	uint16_t *half = (uint16_t *)malloc(sizeof(uint16_t) * w1 * h1), *small = (uint16_t *)malloc(sizeof(uint16_t) * sw * sh);
	unsigned char *dt = (unsigned char *)malloc((size_t)sw * sh);
	for (int y = 0; y < h; y += 2) for (int x = 0; x < w; x += 2)
	{
		uint16_t a = depth[y * w + x], b = depth[y * w + x + 1], c = depth[(y + 1) * w + x], d = depth[(y + 1) * w + x + 1];
		uint16_t m0 = b < a ? b : a, m1 = d < c ? d : c;
		half[(y / 2) * w1 + x / 2] = m1 < m0 ? m1 : m0;
	}
	for (int y = 0; y < h1; y += 2) for (int x = 0; x < w1; x += 2)
	{
		uint16_t a = half[y * w1 + x], b = half[y * w1 + x + 1], c = half[(y + 1) * w1 + x], d = half[(y + 1) * w1 + x + 1];
		uint16_t m0 = b < a ? b : a, m1 = d < c ? d : c;
		small[(y / 2) * sw + x / 2] = m1 < m0 ? m1 : m0;
	}
	const f2 sfocal = { focal.x / 2.0f / 2.0f, focal.y / 2.0f / 2.0f }, sprincipal = { principal.x / 2.0f / 2.0f, principal.y / 2.0f / 2.0f };
	const unsigned short wy = (unsigned short)(wrange_hi / depth_scale);           /* ushort2(wrange / depth_scale) */
	for (int i = 0; i < sw * sh; i++) dt[i] = small[i] < wy ? 255 : 0;
	/* Manhattan distance transform, two raster passes with clamped neighbours */
	for (int y = 0; y < sh; y++) for (int x = 0; x < sw; x++)
	{
		int l = dt[y * sw + (x > 0 ? x - 1 : 0)] + 1, u = dt[(y > 0 ? y - 1 : 0) * sw + x] + 1, c = dt[y * sw + x];
		int m = ho_mini(ho_mini(l, u), c);
		dt[y * sw + x] = (unsigned char)ho_mini(255, m);
	}
	for (int ry = 0; ry < sh; ry++) for (int rx = 0; rx < sw; rx++)
	{
		int x = sw - 1 - rx, y = sh - 1 - ry;
		int r = dt[y * sw + (x < sw - 1 ? x + 1 : sw - 1)] + 1, d = dt[(y < sh - 1 ? y + 1 : sh - 1) * sw + x] + 1, c = dt[y * sw + x];
		int m = ho_mini(ho_mini(r, d), c);
		dt[y * sw + x] = (unsigned char)ho_mini(255, m);
	}
	if (small_out) memcpy(small_out, small, sizeof(uint16_t) * sw * sh);
	if (dt_out) memcpy(dt_out, dt, (size_t)sw * sh);
	/* entry point: largest distance value along the enabled borders, first one wins (handtrack.h:291-295) */
	int ex = 0, ey = 0;
	if (entry_options == 1) { ex = sw / 2; ey = sh - 1; } else if (entry_options == 4) { ex = sw - 1; ey = sh / 2; } else if (entry_options == 8) { ex = 0; ey = sh / 2; }
#define DT(x, y) dt[(y) * sw + (x)]
	if (entry_options & 1) for (int x = 0; x < sw; x++) if (DT(x, sh - 1) > DT(ex, ey)) { ex = x; ey = sh - 1; }
	if (entry_options & 2) for (int x = 0; x < sw; x++) if (DT(x, 0) > DT(ex, ey)) { ex = x; ey = 0; }
	if (entry_options & 4) for (int y = 0; y < sh; y++) if (DT(sw - 1, y) > DT(ex, ey)) { ex = sw - 1; ey = y; }
	if (entry_options & 8) for (int y = 0; y < sh; y++) if (DT(0, y) > DT(ex, ey)) { ex = 0; ey = y; }
	float avgdepth = 0, comx = 0, comy = 0, wtotal = 0.0f;
	int count = 0;
	const int min_blob_radius = 2;
	for (int y = 0; y < sh; y++) for (int x = 0; x < sw; x++) if (DT(x, y) >= min_blob_radius)
	{
		const float dx = (float)(x - ex), dy = (float)(y - ey);
		const float wgt = sqrtf(dx * dx + dy * dy) + 0.00001f;
		wtotal += wgt;
		comx += (float)x * wgt; comy += (float)y * wgt;
		avgdepth += small[y * sw + x] * wgt;
		count++;
	}
	if (count && wtotal > 0.0f)
	{
		avgdepth *= depth_scale / wtotal;
		comx /= wtotal; comy /= wtotal;
	}
	float extx = (float)ex, exty = (float)ey;
	for (int y = 0; y < sh; y++) for (int x = 0; x < sw; x++) if (DT(x, y) >= min_blob_radius)
	{
		const float cx = comx - (float)ex, cy = comy - (float)ey;
		if (((float)x - (float)ex) * cx + ((float)y - (float)ey) * cy > (extx - (float)ex) * cx + (exty - (float)ey) * cy) { extx = (float)x; exty = (float)y; }
	}
#undef DT
	float angle = 0.0f;
	avgdepth = ho_clampf(avgdepth, 0.20f, 1.0f);
	if (count && wtotal > 0.0f && !(comx == (float)ex && comy == (float)ey))
	{
		angle = (float)atan2((double)((float)comx - ex), (double)((float)ey - comy));
		const float cx = comx - (float)ex, cy = comy - (float)ey;
		const float cl = sqrtf(cx * cx + cy * cy);
		const float nx = cx / cl, ny = cy / cl;
		const float exrad = (extx - comx) * nx + (exty - comy) * ny;
		const float shift = exrad - diam / 2.0f / avgdepth * sfocal.x;
		comx += nx * shift; comy += ny * shift;
	}
	const float dfocal = avgdepth * 64.0f / diam;
	const f2 df = { dfocal, dfocal }, dp = { 32.0f, 32.0f };      /* principal() = asfloat2(dim) * 0.5f */
	const f4 dq = qmul(quat_from_to(deprojectz(sfocal, sprincipal, sprincipal.x, sprincipal.y, 1.0f), deprojectz(sfocal, sprincipal, comx, comy, 1.0f)), quat_axis_angle(F3(0, 0, 1), angle));
	/* SampleD with background 4 m */
	const unsigned short background = (unsigned short)(4.0f / depth_scale);
	const f3 ppdir = add3(F3(0, 0, 0), qrot(dq, deprojectz(df, dp, dp.x, dp.y, 1.0f)));
	for (int y = 0; y < 64; y++) for (int x = 0; x < 64; x++)
	{
		const f3 dir = add3(F3(0, 0, 0), qrot(dq, deprojectz(df, dp, (float)x, (float)y, 1.0f)));
		const float u = dir.x / dir.z * focal.x + principal.x, v = dir.y / dir.z * focal.y + principal.y;      /* projectz: v.xy()/v.z*focal + principal */
		const int px = (int)u, py = (int)v;
		if (px >= 0 && px <= w - 1 && py >= 0 && py <= h - 1)
			tile[y * 64 + x] = (unsigned short)dot3(ppdir, deprojectz(focal, principal, (float)px, (float)py, (float)depth[py * w + px]));
		else tile[y * 64 + x] = background;
	}
	camout12[0] = dfocal; camout12[1] = dfocal; camout12[2] = 32.0f; camout12[3] = 32.0f; camout12[4] = depth_scale;
	camout12[5] = 0; camout12[6] = 0; camout12[7] = 0; camout12[8] = dq.x; camout12[9] = dq.y; camout12[10] = dq.z; camout12[11] = dq.w;
	free(half); free(small); free(dt);
	return 0;
}
