// ht_segment.hip -- the step before the tracker for full-size depth frames: hand segmentation to a 64x64 tile.
//
// Reference computations:
//   HandSegmentVR                 include/handtrack.h:280-344
//   DownSampleMin (x2), camera/2  include/misc_image.h:81-94,60,136
//   Threshold, DistanceTransform  include/misc_image.h:179-195
//   SampleD                       include/misc_image.h:154-162
//
// One 256-thread block per frame; the quarter-size image (80x60 for a 320x240 frame) and its distance transform stay in LDS.
//   1. 4x4 min pooling straight from the frame (two 2x2 min poolings compose exactly), threshold.
//   2. The reference's two raster passes of the Manhattan distance transform are min-plus recurrences over integers; each is
//      evaluated exactly as a row scan followed by a column scan (one thread per row / column).
//   3. Entry point and extreme point are "first maximum" searches in a fixed order: parallel arg-max on (value, order index).
//   4. The weighted centroid sums are float accumulations in raster order, so they stay sequential: the selected pixels are
//      compacted in raster order first, then four lanes each accumulate one of the four sums over that list.
//   5. Rotated resampling (SampleD), 16 output pixels per thread.
// atan2 at handtrack.h:326 is the unqualified C function, i.e. evaluated in double; sin/cos of the half angle are formed in double and
// rounded once (the reference calls sinf/cosf; the two agree except in rare half-ulp cases).
#include "ht_device.hpp"
#include "ht_launch.hpp"

#define SEG_THREADS 256
#define SEG_MAXSMALL 4800      // quarter-size pixels held in LDS (320x240 input)

__device__ __forceinline__ v3 seg_deproject(float fx, float fy, float px, float py, float x, float y, float d)      // DCamera::deprojectz, misc_image.h:48
{
	return V3((x - px) / fx, (y - py) / fy, 1.0f) * d;
}
// block-wide arg-max with the lowest order index winning ties; every thread returns the winner's order index
__device__ int seg_block_argmax(int val, int ord, int *red)
{
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1)
	{
		const int ov = __shfl_xor(val, o), oo = __shfl_xor(ord, o);
		if (ov > val || (ov == val && oo < ord)) { val = ov; ord = oo; }
	}
	__syncthreads();
	if (lane == 0) { red[2 * wave] = val; red[2 * wave + 1] = ord; }
	__syncthreads();
	int bv = red[0], bo = red[1];
	for (int k = 1; k < SEG_THREADS / 64; k++) if (red[2 * k] > bv || (red[2 * k] == bv && red[2 * k + 1] < bo)) { bv = red[2 * k]; bo = red[2 * k + 1]; }
	return bo;
}
__device__ int seg_block_argmax_f(float val, int ord, float *redf, int *redi)
{
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1)
	{
		const float ov = __shfl_xor(val, o); const int oo = __shfl_xor(ord, o);
		if (oo >= 0 && (ord < 0 || ov > val || (ov == val && oo < ord))) { val = ov; ord = oo; }
	}
	__syncthreads();
	if (lane == 0) { redf[wave] = val; redi[wave] = ord; }
	__syncthreads();
	float bv = redf[0]; int bo = redi[0];
	for (int k = 1; k < SEG_THREADS / 64; k++) if (redi[k] >= 0 && (bo < 0 || redf[k] > bv || (redf[k] == bv && redi[k] < bo))) { bv = redf[k]; bo = redi[k]; }
	return bo;
}

__global__ __launch_bounds__(SEG_THREADS) void k_segment(const uint16_t *__restrict__ depth, const float *__restrict__ cams, int w, int h, int entry_options, float wrange_hi, float diam,
                                                         uint16_t *__restrict__ tiles, float *__restrict__ cams_out)
{
	__shared__ uint16_t small[SEG_MAXSMALL];
	__shared__ unsigned char dt[SEG_MAXSMALL];
	__shared__ uint16_t sel[SEG_MAXSMALL];
	__shared__ int red[2 * SEG_THREADS / 64]; __shared__ float redf[SEG_THREADS / 64];
	__shared__ int wavecnt[SEG_THREADS / 64 + 1];
	__shared__ float sums[4];
	__shared__ float fin[8];      // dq (4), dfocal
	const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const uint16_t *src = depth + (size_t)b * w * h;
	const float *cam = cams + (size_t)b * HT_CAM;
	const float fx = cam[0], fy = cam[1], px = cam[2], py = cam[3], depth_scale = cam[4];
	const int sw = (w / 2) / 2, sh = (h / 2) / 2, ns = sw * sh;
	const unsigned short wy = (unsigned short)(wrange_hi / depth_scale);      // ushort2(wrange / depth_scale).y, handtrack.h:287
	// ---- 1. 4x4 min pooling + threshold ----
	for (int i = t; i < ns; i += SEG_THREADS)
	{
		const int sx = i % sw, sy = i / sw;
		unsigned m = 0xFFFFu;
#pragma unroll
		for (int r = 0; r < 4; r++)
		{
			const uint16_t *p = src + (size_t)(4 * sy + r) * w + 4 * sx;
			const unsigned a = p[0], bb = p[1], c = p[2], d = p[3];
			m = min(m, min(min(a, bb), min(c, d)));
		}
		small[i] = (uint16_t)m;
		dt[i] = m < wy ? 255 : 0;
	}
	__syncthreads();
	// ---- 2. distance transform: forward pass = row scan then column scan, backward pass likewise in reverse ----
	for (int y = t; y < sh; y += SEG_THREADS) { int run = 255 + 1; for (int x = 0; x < sw; x++) { const int v = min((int)dt[y * sw + x], run); dt[y * sw + x] = (unsigned char)min(v, 255); run = v + 1; } }
	__syncthreads();
	for (int x = t; x < sw; x += SEG_THREADS) { int run = 255 + 1; for (int y = 0; y < sh; y++) { const int v = min((int)dt[y * sw + x], run); dt[y * sw + x] = (unsigned char)min(v, 255); run = v + 1; } }
	__syncthreads();
	for (int y = t; y < sh; y += SEG_THREADS) { int run = 255 + 1; for (int x = sw - 1; x >= 0; x--) { const int v = min((int)dt[y * sw + x], run); dt[y * sw + x] = (unsigned char)min(v, 255); run = v + 1; } }
	__syncthreads();
	for (int x = t; x < sw; x += SEG_THREADS) { int run = 255 + 1; for (int y = sh - 1; y >= 0; y--) { const int v = min((int)dt[y * sw + x], run); dt[y * sw + x] = (unsigned char)min(v, 255); run = v + 1; } }
	__syncthreads();
	// ---- 3. entry point: first maximum over [start, bottom row, top row, right column, left column] (handtrack.h:290-295) ----
	int ex0 = 0, ey0 = 0;
	if (entry_options == 1) { ex0 = sw / 2; ey0 = sh - 1; } else if (entry_options == 4) { ex0 = sw - 1; ey0 = sh / 2; } else if (entry_options == 8) { ex0 = 0; ey0 = sh / 2; }
	const int ncand = 1 + 2 * sw + 2 * sh;
	int bestv = -1, bestk = 0x7fffffff;
	for (int k = t; k < ncand; k += SEG_THREADS)
	{
		int cx, cy; bool on = true;
		if (k == 0) { cx = ex0; cy = ey0; }
		else if (k <= sw) { cx = k - 1; cy = sh - 1; on = entry_options & 1; }
		else if (k <= 2 * sw) { cx = k - 1 - sw; cy = 0; on = entry_options & 2; }
		else if (k <= 2 * sw + sh) { cx = sw - 1; cy = k - 1 - 2 * sw; on = entry_options & 4; }
		else { cx = 0; cy = k - 1 - 2 * sw - sh; on = entry_options & 8; }
		const int v = on ? (int)dt[cy * sw + cx] : -1;
		if (v > bestv || (v == bestv && k < bestk)) { bestv = v; bestk = k; }
	}
	const int kwin = seg_block_argmax(bestv, bestk, red);
	int ex, ey;
	if (kwin == 0) { ex = ex0; ey = ey0; }
	else if (kwin <= sw) { ex = kwin - 1; ey = sh - 1; }
	else if (kwin <= 2 * sw) { ex = kwin - 1 - sw; ey = 0; }
	else if (kwin <= 2 * sw + sh) { ex = sw - 1; ey = kwin - 1 - 2 * sw; }
	else { ex = 0; ey = kwin - 1 - 2 * sw - sh; }
	// ---- 4. blob pixels (distance >= 2) compacted in raster order ----
	const int min_blob_radius = 2;
	int total = 0;
	for (int base = 0; base < ns; base += SEG_THREADS)
	{
		const int i = base + t;
		const bool on = i < ns && dt[i] >= min_blob_radius;
		const unsigned long long m = __ballot(on);
		if (lane == 0) wavecnt[wave] = __popcll(m);
		__syncthreads();
		int off = total;
		for (int k = 0; k < wave; k++) off += wavecnt[k];
		if (on) sel[off + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)i;
		for (int k = 0; k < SEG_THREADS / 64; k++) total += wavecnt[k];
		__syncthreads();
	}
	const int count = total;
	// the four sums of handtrack.h:303-311, each in raster order: lane c of wave 0 accumulates sum c
	if (t < 4)
	{
		float acc = 0.0f;
		for (int k = 0; k < count; k++)
		{
			const int i = sel[k], x = i % sw, y = i / sw;
			const float dx = (float)(x - ex), dy = (float)(y - ey);
			const float wgt = sqrtf(dx * dx + dy * dy) + 0.00001f;
			const float f = t == 0 ? 1.0f : t == 1 ? (float)x : t == 2 ? (float)y : (float)small[i];
			acc += f * wgt;
		}
		sums[t] = acc;
	}
	__syncthreads();
	float wtotal = sums[0], comx = sums[1], comy = sums[2], avgdepth = sums[3];
	if (count && wtotal > 0.0f) { avgdepth *= depth_scale / wtotal; comx /= wtotal; comy /= wtotal; }
	// extreme point: first maximum of dot(p - entry, com - entry) among the blob pixels, the entry itself being the start (handtrack.h:317-323)
	const float cex = comx - (float)ex, cey = comy - (float)ey;
	float bvf = 0.0f; int bkf = -1;
	for (int k = t; k < count; k += SEG_THREADS)
	{
		const int i = sel[k], x = i % sw, y = i / sw;
		const float d = ((float)x - (float)ex) * cex + ((float)y - (float)ey) * cey;
		if (bkf < 0 || d > bvf) { bvf = d; bkf = k; }
	}
	const int kext = seg_block_argmax_f(bvf, bkf, redf, red);
	if (t == 0)
	{
		float extx = (float)ex, exty = (float)ey;
		if (kext >= 0)
		{
			const int i = sel[kext], x = i % sw, y = i / sw;
			const float d = ((float)x - (float)ex) * cex + ((float)y - (float)ey) * cey;
			if (d > (extx - (float)ex) * cex + (exty - (float)ey) * cey) { extx = (float)x; exty = (float)y; }
		}
		const float sfx = fx / 2.0f / 2.0f, sfy = fy / 2.0f / 2.0f, spx = px / 2.0f / 2.0f, spy = py / 2.0f / 2.0f;      // camera of the quarter-size image
		float angle = 0.0f;
		avgdepth = clamp_std(avgdepth, 0.20f, 1.0f);
		if (count && wtotal > 0.0f && !(comx == (float)ex && comy == (float)ey))
		{
			angle = (float)atan2((double)(comx - ex), (double)((float)ey - comy));
			const float cl = sqrtf(cex * cex + cey * cey);
			const float nx = cex / cl, ny = cey / cl;
			const float exrad = (extx - comx) * nx + (exty - comy) * ny;
			const float shift = exrad - diam / 2.0f / avgdepth * sfx;
			comx += nx * shift; comy += ny * shift;
		}
		const float dfocal = avgdepth * 64.0f / diam;
		const v4 dq = qmul(quat_from_to(seg_deproject(sfx, sfy, spx, spy, spx, spy, 1.0f), seg_deproject(sfx, sfy, spx, spy, comx, comy, 1.0f)), quat_axis_angle(V3(0, 0, 1), angle));
		fin[0] = dq.x; fin[1] = dq.y; fin[2] = dq.z; fin[3] = dq.w; fin[4] = dfocal;
		float *co = cams_out + (size_t)b * HT_CAM;
		co[0] = dfocal; co[1] = dfocal; co[2] = 32.0f; co[3] = 32.0f; co[4] = depth_scale; co[5] = 0; co[6] = 0; co[7] = 0; co[8] = dq.x; co[9] = dq.y; co[10] = dq.z; co[11] = dq.w;
	}
	__syncthreads();
	// ---- 5. SampleD with a 4 m background ----
	const v4 dq = V4(fin[0], fin[1], fin[2], fin[3]); const float dfocal = fin[4];
	const unsigned short background = (unsigned short)(4.0f / depth_scale);
	const v3 ppdir = V3(0, 0, 0) + qrot(dq, seg_deproject(dfocal, dfocal, 32.0f, 32.0f, 32.0f, 32.0f, 1.0f));
	for (int i = t; i < 4096; i += SEG_THREADS)
	{
		const int x = i & 63, y = i >> 6;
		const v3 dir = V3(0, 0, 0) + qrot(dq, seg_deproject(dfocal, dfocal, 32.0f, 32.0f, (float)x, (float)y, 1.0f));
		const float u = dir.x / dir.z * fx + px, v = dir.y / dir.z * fy + py;
		const int sx = (int)u, sy = (int)v;
		unsigned short o = background;
		if (sx >= 0 && sx <= w - 1 && sy >= 0 && sy <= h - 1) o = (unsigned short)dot(ppdir, seg_deproject(fx, fy, px, py, (float)sx, (float)sy, (float)src[(size_t)sy * w + sx]));
		tiles[(size_t)b * 4096 + i] = o;
	}
}

bool ht_segment_supported(int w, int h) { return w >= 8 && h >= 8 && (w % 4) == 0 && (h % 4) == 0 && (w / 4) * (h / 4) <= SEG_MAXSMALL; }
void ht_launch_segment(const uint16_t *depth, const float *cams, int w, int h, int entry_options, float wrange_hi, float diam, uint16_t *tiles, float *cams_out, int B, hipStream_t s)
{
	hipLaunchKernelGGL(k_segment, dim3(B), dim3(SEG_THREADS), 0, s, depth, cams, w, h, entry_options, wrange_hi, diam, tiles, cams_out);
}
