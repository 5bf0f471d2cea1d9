// ht_cnn.hip -- CDNA4 (gfx950) kernels for the depth-tile CNN of the hand tracker.
//
// Reference computation: CNN::Eval (third_party/cnn.h:550-556) over the topology of PoseInitializerCNN
// (include/handtrack.h:108-118), preceded by the depth normalisation of handtrack.h:700 and the point-cloud
// extraction of misc_image.h:409-417 / physmodel.h:58-64 (both read the same depth tile, so one kernel does both).
//
//   k_prepare   u16 depth tile -> fp32 CNN input + order-preserving compacted point cloud  (HBM-bound, 8 KB in / 16 KB out per frame)
//   k_conv1     5x5x1->16 valid conv + 4x4 max-pool + tanh as an implicit GEMM on v_mfma_f32_16x16x4_f32 (one pooling window = one 16-row tile), input rows staged in LDS
//   k_conv2     4x4x16->64 valid conv as an implicit GEMM on v_mfma_f32_16x16x4_f32, + 2x2 max-pool + tanh
//   k_conv12    both of them in one launch for the 64x64 net (the first layer's pooled array in LDS is the second layer's input image)
//   k_fc        [B,K]x[K,N]+bias (+tanh) on v_mfma_f32_32x32x2_f32, 128x64 block tile on 8 waves, double-buffered LDS, register prefetch
//   k_fc144_pk  the last layer on a 64x144 tile (v_mfma_f32_16x16x4_f32), tiles by LDS-DMA, both operands in the instruction's order (128-bit fragment reads)
//   k_softmax_decode   chunked softmax (cnn.h:497-511) fused with the heat-map decode (handtrack.h:218-241)
//
// Numerics: the MFMA accumulator starts at the bias and sums k in ascending order, which is the reference's own order
// (cnn.h:219-226, 407-426); the only difference is one rounding per multiply-add instead of two.  max-pool commutes
// with the monotone tanh, so tanh is applied after pooling (16x fewer exponentials in conv1).
#include "ht_device.hpp"

// ------------------------------------------------------------------------------------------------- k_prepare
// one block per frame; thread t owns pixels [16t, 16t+16) so that a block-wide prefix sum keeps the row-major order
__global__ __launch_bounds__(256) void k_prepare(const uint16_t *__restrict__ depth, const float *__restrict__ cams, float drangey, int fraction,
                                                  float *__restrict__ cnn_in, float4 *__restrict__ pts, int *__restrict__ npts, int cap, ht_prepare_extra x)
{
	const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const float *cam = cams + (size_t)b * HT_CAM;
	if (x.cams_out && t < HT_CAM) x.cams_out[(size_t)b * HT_CAM + t] = cam[t];
	if (x.zero && b == 0 && t == 0) *x.zero = 0;
	if (x.start && t < x.nb)      // ht_tracker_reset for this frame: both models take the start pose, momenta and flags are cleared
	{
		const float *o = x.start + ((size_t)b * x.nb + t) * HT_POSE;
		float *s0 = x.state0 + ((size_t)b * x.nb + t) * HT_STATE_STRIDE, *s1 = x.state1 + ((size_t)b * x.nb + t) * HT_STATE_STRIDE;
		for (int k = 0; k < 7; k++) { s0[k] = o[k]; s1[k] = o[k]; }
		for (int k = 7; k < 13; k++) { s0[k] = 0.0f; s1[k] = 0.0f; }
		if (t == 0) { x.prev_err[b] = 0.0f; x.initializing[b] = 0; }
	}
	const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3], dscale = cam[4];
	const uint4 *src = reinterpret_cast<const uint4 *>(depth + (size_t)b * 4096 + 16 * t);
	uint4 r0 = src[0], r1 = src[1];
	unsigned w[8] = { r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w };
	float d[16], cin[16];
	int cnt = 0;
	unsigned mask = 0;
#pragma unroll
	for (int i = 0; i < 16; i++)
	{
		unsigned px = (w[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
		d[i] = (float)(int)px * dscale;
		cin[i] = clamp_std(1.0f - (d[i] - 0.1f) / (drangey - 0.1f), 0.0f, 1.0f);     // handtrack.h:700
		bool in = d[i] >= 0.1f && d[i] < drangey;                                     // FallsWithinRange misc_image.h:24
		mask |= in ? (1u << i) : 0u;
		cnt += in;
	}
	if (cnn_in)
	{
		float4 *dst = reinterpret_cast<float4 *>(cnn_in + (size_t)b * 4096 + 16 * t);
#pragma unroll
		for (int i = 0; i < 4; i++) dst[i] = make_float4(cin[4 * i], cin[4 * i + 1], cin[4 * i + 2], cin[4 * i + 3]);
	}
	// exclusive prefix of cnt over the 256 threads
	int incl = cnt;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
	__shared__ int wsum[4];
	if (lane == 63) wsum[wave] = incl;
	__syncthreads();
	int base = incl - cnt;
	for (int i = 0; i < wave; i++) base += wsum[i];
	if (pts)
	{
		int rank = base;
#pragma unroll
		for (int i = 0; i < 16; i++) if (mask & (1u << i))
		{
			if (rank % fraction == 0 && rank / fraction < cap)
			{
				int p = 16 * t + i;
				float x = (float)(p & 63), y = (float)(p >> 6);
				pts[(size_t)b * cap + rank / fraction] = make_float4(((x - cx) / fx) * d[i], ((y - cy) / fy) * d[i], 1.0f * d[i], 0.0f);   // deprojectz misc_image.h:48
			}
			rank++;
		}
	}
	if (npts && t == 255)
	{
		int total = base + cnt;
		int n = (total + fraction - 1) / fraction;
		npts[b] = n < cap ? n : cap;
	}
}

// Point cloud of a full-size frame (w x h, any size): takesubsample(PointCloud(dimage, {0.1, drangey}), fraction) of handtrack.h:703,751 --
// every fraction-th in-range pixel in row-major order, deprojected with the frame's own camera.  One block per frame; thread t owns
// the pixels [t*chunk, (t+1)*chunk), counts, the block forms the exclusive prefix, and a second walk emits.  *overflow counts frames with
// more than `cap` points (their cloud is truncated; the host API sizes cap so that this cannot happen and treats a count as an error).
__global__ __launch_bounds__(256) void k_prepare_frame(const uint16_t *__restrict__ depth, const float *__restrict__ cams, int w, int h, float drangey, int fraction,
                                                        float4 *__restrict__ pts, int *__restrict__ npts, int *__restrict__ overflow, int cap)
{
	const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const float *cam = cams + (size_t)b * HT_CAM;
	const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3], dscale = cam[4];
	const int npx = w * h, chunk = (npx + 255) / 256, p0 = min(npx, t * chunk), p1 = min(npx, p0 + chunk);
	const uint16_t *src = depth + (size_t)b * npx;
	int cnt = 0;
	for (int p = p0; p < p1; p++) { const float d = (float)(int)src[p] * dscale; cnt += (d >= 0.1f && d < drangey) ? 1 : 0; }
	int incl = cnt;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
	__shared__ int wsum[4];
	if (lane == 63) wsum[wave] = incl;
	__syncthreads();
	int rank = incl - cnt;
	for (int i = 0; i < wave; i++) rank += wsum[i];
	for (int p = p0; p < p1; p++)
	{
		const float d = (float)(int)src[p] * dscale;
		if (!(d >= 0.1f && d < drangey)) continue;
		if (rank % fraction == 0 && rank / fraction < cap)
		{
			const float x = (float)(p % w), y = (float)(p / w);
			pts[(size_t)b * cap + rank / fraction] = make_float4(((x - cx) / fx) * d, ((y - cy) / fy) * d, 1.0f * d, 0.0f);      // deprojectz misc_image.h:48
		}
		rank++;
	}
	if (t == 255)
	{
		const int n = (rank + fraction - 1) / fraction;
		npts[b] = n < cap ? n : cap;
		if (n > cap && overflow) atomicAdd(overflow, 1);
	}
}

// The CNN input of a frame of any size that is its own segment (handtrack.h:700 on the whole frame; BASELINE configs[4] end to end: the 128x128 frame
// feeds the 128x128-input net directly): eight pixels per thread, one 128-bit read and two 128-bit writes.  npx is a multiple of 8.
__global__ __launch_bounds__(256) void k_cnn_input(const uint16_t *__restrict__ depth, const float *__restrict__ cams, int npx, float drangey, float *__restrict__ cnn_in)
{
	const int b = blockIdx.y, i = (blockIdx.x * 256 + threadIdx.x) * 8;
	if (i >= npx) return;
	const float dscale = cams[(size_t)b * HT_CAM + 4];
	const uint4 r = *reinterpret_cast<const uint4 *>(depth + (size_t)b * npx + i);
	const unsigned w[4] = { r.x, r.y, r.z, r.w };
	float c[8];
#pragma unroll
	for (int k = 0; k < 8; k++)
	{
		const float d = (float)(int)((w[k >> 1] >> ((k & 1) * 16)) & 0xffffu) * dscale;
		c[k] = clamp_std(1.0f - (d - 0.1f) / (drangey - 0.1f), 0.0f, 1.0f);
	}
	float4 *dst = reinterpret_cast<float4 *>(cnn_in + (size_t)b * npx + i);
	dst[0] = make_float4(c[0], c[1], c[2], c[3]); dst[1] = make_float4(c[4], c[5], c[6], c[7]);
}

// ------------------------------------------------------------------------------------------------- k_conv1
// cnn.h:31.  The exponential is formed in double and rounded once (= the reference's expf except in rare half-ulp cases).  A plain float expf is
// one ulp off now and then; through MultiStepSim's hard-driven steps that doubled the pose deviation of the CNN-accepted frames (2.0e-3 on a
// quaternion against the 2e-3 tolerance), so the exponential stays a double one.
// The double exponential is a short one: x = k ln2 + r (k = nearest integer, ln2 in two parts), exp(r) by the Taylor polynomial of degree 11 (|r| <= 0.347:
// relative error below 2^-47), scaled by 2^k.  Rounded to float it differs from the correctly rounded expf for about one argument in 2^22 (the
// library's own double exp: one in 2^27), which is far inside what the reference's libm does, and it is a third of the library routine's instructions.
__device__ __forceinline__ float expf_via_double(float xf)
{
	const double x = (double)xf;
	const double k = __builtin_rint(x * 1.4426950408889634074);
	const double r = __builtin_fma(-k, 1.90821492927058770002e-10, __builtin_fma(-k, 6.93147180369123816490e-01, x));
	double p = 1.0 / 39916800.0;
	p = __builtin_fma(p, r, 1.0 / 3628800.0); p = __builtin_fma(p, r, 1.0 / 362880.0); p = __builtin_fma(p, r, 1.0 / 40320.0); p = __builtin_fma(p, r, 1.0 / 5040.0);
	p = __builtin_fma(p, r, 1.0 / 720.0); p = __builtin_fma(p, r, 1.0 / 120.0); p = __builtin_fma(p, r, 1.0 / 24.0); p = __builtin_fma(p, r, 1.0 / 6.0);
	p = __builtin_fma(p, r, 0.5); p = __builtin_fma(p, r, 1.0); p = __builtin_fma(p, r, 1.0);
	int ki = (int)k;
	ki = ki < -1100 ? -1100 : (ki > 1100 ? 1100 : ki);
	return (float)__builtin_ldexp(p, ki);
}
__device__ __forceinline__ float tanh_ref(float t) { float e = expf_via_double(2 * t); return (e - 1) / (e + 1); }
// The maximum of a pooling window (cnn.h:141-148 takes std::max) as one v_max_f32 instead of a compare and a select.  The two differ on NaNs (the layers' inputs are
// finite) and on which zero max(-0, +0) is -- and the pooled value goes through tanh_ref next, which maps either zero to +0.
// (Spelled as the instruction: through __builtin_fmaxf the compiler first quiets possible signalling NaNs of either operand with a v_max_f32 x, x, x each.)
__device__ __forceinline__ float max_pool(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float max_pool4(float a, float b, float c, float d) { float r; asm("v_max_f32 %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4" : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d)); return r; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 5x5x1->16 valid convolution + two 2x2 max-pools + tanh as an implicit GEMM on v_mfma_f32_16x16x4_f32: M = the 16 pixels of one 4x4 pooling
// window, N = the 16 output channels, K = the 25 taps (kx fastest, the reference's accumulation order, cnn.h:223-225) padded to 28 with zero weights.
// A block takes PR pooled rows of one frame; the 4*PR + 4 input rows they need are staged in LDS with a row stride of IW + 4 floats (rows 4 banks
// apart: the 4x4 pixels a 16-lane group reads fall on 16 different banks, and neighbouring taps re-read the same words, which is a broadcast).
// The seven weight fragments of a lane stay in registers; per window a wave issues 7 LDS reads and 7 MFMAs, the maximum over the window's 16 rows
// is four register maxima and two cross-lane steps, and tanh (an exponential in double) is applied once per pooled value by all threads at the end.
// 64x64 input: IW = 64, PW = 15, PR = 15 (one block per frame; bands of 8, 5 or 3 pooled rows, which would stagger the blocks' phases, were measured: no change).  128x128 input (BASELINE configs[4]): IW = 128, PW = 31, PR = 8 -> 4 bands of
// 36 rows (19 KB) instead of one 64 KB tile, so several blocks stay resident per CU.
template <int IW, int PW, int PR>
__global__ __launch_bounds__(256) void k_conv1(const float *__restrict__ cnn_in, const float *__restrict__ W1, const float *__restrict__ B1, float *__restrict__ act1)
{
	constexpr int TR = 4 * PR + 4, IWP = IW + 4;         // input rows a band touches, padded row stride
	static_assert(TR <= IW && (IW % 4) == 0, "band does not fit the image");
	__shared__ __attribute__((aligned(16))) float tile[TR * IWP];
	__shared__ float pooled[16 * PR * PW];
	const int b = blockIdx.x, band = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const int row0 = 4 * PR * band;
	const int nrows = min(TR, IW - row0);
	const int prows = min(PR, PW - PR * band);           // pooled rows this band really has
	const float4 *src = reinterpret_cast<const float4 *>(cnn_in + (size_t)b * IW * IW + (size_t)row0 * IW);
	for (int i = t; i < nrows * IW / 4; i += 256) { const int r = i / (IW / 4), c4 = i % (IW / 4); *reinterpret_cast<float4 *>(tile + r * IWP + 4 * c4) = src[i]; }
	// this lane's operands: B = weight of (tap k, channel n), A = pixel (py, px) of the window shifted by tap k; k = 4 * step + (lane >> 4)
	const int n = lane & 15, g = lane >> 4, px = lane & 3, py = (lane >> 2) & 3;
	float wreg[7]; int aoff[7];
#pragma unroll
	for (int s = 0; s < 7; s++)
	{
		const int k = 4 * s + g, kk = k < 25 ? k : 24;
		wreg[s] = k < 25 ? W1[n * 25 + k] : 0.0f;        // W index = kx + 5*(ky + 5*(ic + 1*oc)), cnn.h:45-47,227
		aoff[s] = (py + kk / 5) * IWP + px + kk % 5;
	}
	const float bias = B1[n];
	__syncthreads();
	// The window loop is bound by the SIMD's one vector issue port, not by the matrix pipe: 7 MFMAs hold the port for 56 of their 224 clocks, and everything else a window
	// issues has to fit the rest beside three other waves' windows.  So the wave's index is read into a scalar register -- the window's coordinates and its tile address
	// are then scalar arithmetic -- and a maximum is one v_max_f32 (a compare and a select cost three issue slots with their wait state).
	const int swave = __builtin_amdgcn_readfirstlane(wave);
	float *const pdst = pooled + n * PR * PW;
	for (int w = swave; w < prows * PW; w += 4)
	{
		const int ty = w / PW, tx = w % PW;
		const float *base = tile + 4 * ty * IWP + 4 * tx;
		f32x4 acc = { bias, bias, bias, bias };
#pragma unroll
		for (int s = 0; s < 7; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(base[aoff[s]], wreg[s], acc, 0, 0, 0);
		// C/D map 16x16: col = lane & 15 (channel), row = (lane >> 4) * 4 + r (pixel of the window): two 2x2 max-pools (cnn.h:141-148) = the maximum of the 16 rows
		float m = max_pool4(acc[0], acc[1], acc[2], acc[3]);
		// across the four 16-lane rows (the window's four image rows): v_permlane16_swap / v_permlane32_swap exchange rows between two copies of m
		{ const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false); m = max_pool(__uint_as_float(r[0]), __uint_as_float(r[1])); }
		{ const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false); m = max_pool(__uint_as_float(r[0]), __uint_as_float(r[1])); }
		if (lane < 16) pdst[ty * PW + tx] = m;
	}
	__syncthreads();
	for (int i = t; i < 16 * prows * PW; i += 256)
	{
		const int c = i / (prows * PW), r = i % (prows * PW), ty = r / PW, tx = r % PW;
		act1[(size_t)b * (16 * PW * PW) + c * (PW * PW) + (PR * band + ty) * PW + tx] = tanh_ref(pooled[(c * PR + ty) * PW + tx]);      // tanh after pooling (monotone)
	}
}

// ------------------------------------------------------------------------------------------------- k_conv2

// W2p: weights repacked to [k][oc] with k = (ky*4+kx)*16 + ic, i.e. the reference's accumulation order (cnn.h:223-225).
// A block takes BAND output rows of one frame (all 64 output channels, 16 per wave): implicit GEMM M = BAND*OW, N = 64, K = 256.
// 64x64 net: IWD = 15, OW = 12, BAND = 12 (M = 144, the whole frame).  128x128 net: IWD = 31, OW = 28, BAND = 4 (M = 112; 7 bands per frame,
// 14 KB of input rows + 28 KB of pre-pool outputs in LDS instead of 61 KB + 200 KB for the whole frame).
// The 16 rows of an MFMA tile are FOUR 2x2 pooling windows, a window's four pixels on consecutive rows: the C/D map of the instruction (row = (lane >> 4) * 4 + r) then puts
// a window's four pre-pool values into the four accumulator registers of one lane, the max-pool (cnn.h:141-148) is three register maxima, and every lane ends a tile with
// one finished output -- tanh and the store follow at once.  (Rows in image order needed 37 KB of LDS for the pre-pool values of a 64x64-net frame, a second pass and a
// barrier; with 51 KB per block a CU held three of the four blocks it gets at 1024 frames and ran the fourth alone: 56 us, now 14 KB per block and one round.)
template <int IWD, int OW, int BAND>
__global__ __launch_bounds__(256) void k_conv2(const float *__restrict__ act1, const float *__restrict__ W2p, const float *__restrict__ B2, float *__restrict__ act2)
{
	constexpr int IR = BAND + 3, PO = OW / 2, CH = IR * IWD, NW = (BAND / 2) * PO, MT = NW / 4;      // NW: pooling windows of the band
	static_assert(NW % 4 == 0 && BAND % 2 == 0 && OW % 2 == 0 && OW % BAND == 0, "band must hold whole MFMA tiles of whole pooling windows");
	__shared__ __attribute__((aligned(16))) float in[16 * CH];
	const int b = blockIdx.x, band = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const int oy0 = band * BAND;
	// the band's input rows in the order the matrix instruction reads them: [position][ic & 3][ic >> 2] -- a lane takes ic = 4 * j + (lane >> 4) for the four k-steps j of a
	// tap, so its four values are one 128-bit read (16 reads per tile, where [ic][position] took 64 values in 32 double reads)
	for (int i = t; i < 16 * CH; i += 256) { const int ic = i / CH, r = i % CH; in[(r * 4 + (ic & 3)) * 4 + (ic >> 2)] = act1[(size_t)b * (16 * IWD * IWD) + ic * (IWD * IWD) + oy0 * IWD + r]; }
	const int n = 16 * wave + (lane & 15);
	float breg[64];
#pragma unroll
	for (int ks = 0; ks < 64; ks++) breg[ks] = W2p[(4 * ks + (lane >> 4)) * 64 + n];
	const float bias = B2[n];
	__syncthreads();
	// per lane k decomposition is fixed per k-step: k = 4*ks + (lane>>4): tap p = k>>4 = ks>>2, ic = 4*(ks&3) + (lane>>4)
	const int icl = lane >> 4;
	const int arow = lane & 15, awin = arow >> 2, ady = (arow >> 1) & 1, adx = arow & 1;      // A operand: this lane supplies pixel (ady, adx) of window awin of the tile
	float *const dst = act2 + (size_t)b * (64 * PO * PO) + n * (PO * PO) + (oy0 / 2) * PO;       // index = x + PO*y + PO*PO*c, the layout LFull consumes
	for (int mt = 0; mt < MT; mt++)
	{
		const int win = 4 * mt + awin, wy = win / PO, wx = win % PO;
		f32x4 acc = { bias, bias, bias, bias };
		const float *base = in + (((2 * wy + ady) * IWD + 2 * wx + adx) * 4 + icl) * 4;
#pragma unroll
		for (int p = 0; p < 16; p++)      // tap p = (ky, kx); k = 16 * p + 4 * j + icl ascending
		{
			const float4 a = *reinterpret_cast<const float4 *>(base + ((p >> 2) * IWD + (p & 3)) * 16);
			acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, breg[4 * p + 0], acc, 0, 0, 0);
			acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, breg[4 * p + 1], acc, 0, 0, 0);
			acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, breg[4 * p + 2], acc, 0, 0, 0);
			acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, breg[4 * p + 3], acc, 0, 0, 0);
		}
		// C/D map 16x16: col = lane&15 (oc), row = (lane>>4)*4 + r = pixel r of window (lane>>4) of the tile, pixels in the order (0,0) (0,1) (1,0) (1,1)
		const float mm = max_pool4(acc[0], acc[1], acc[2], acc[3]);
		const int wo = 4 * mt + icl;
		dst[wo] = tanh_ref(mm);      // window wo of the band: pooled row wo / PO, column wo % PO
	}
}

// ------------------------------------------------------------------------------------------------- k_conv12
// Both convolution layers of the 64x64 net in one launch, a block per frame: k_conv1's window loop leaves the 16 x 15 x 15 pooled values in LDS, tanh is applied in place --
// which makes that array exactly the input image k_conv2 stages ([ic][15 x 15]) -- and k_conv2's tiles follow.  Saved against the two launches: conv2's load phase, 14.7 MB
// written and read back, a launch boundary with every block of the chip in step, and -- what the phases of a single launch cannot hide from each other (section 15) -- the 64 KB
// of conv2 weights every block loads into registers, which are asked for HERE before the first layer's window loop and arrive under it.  act1 is still written (the layer
// getter reads it), nobody waits for it.  Same arithmetic in the same order as the two kernels (the 128x128 net keeps them: its bands do not line up).
__global__ __launch_bounds__(256, 4) void k_conv12(const float *__restrict__ cnn_in, const float *__restrict__ W1, const float *__restrict__ B1, const float *__restrict__ W2p, const float *__restrict__ B2,
                                                float *__restrict__ act1, float *__restrict__ act2)
{
	constexpr int IW = 64, PW = 15, PR = 15, TR = 4 * PR + 4, IWP = IW + 4;
	constexpr int IWD = 15, OW = 12, PO = 6, CH = IWD * IWD, MT = 9;
	__shared__ __attribute__((aligned(16))) float tile[TR * IWP];
	__shared__ float pooled[16 * CH];      // layer 1's pooled values, then (tanh applied) layer 2's input [ic][15 x 15]
	const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const float4 *src = reinterpret_cast<const float4 *>(cnn_in + (size_t)b * IW * IW);
	for (int i = t; i < TR * IW / 4; i += 256) { const int r = i / (IW / 4), c4 = i % (IW / 4); *reinterpret_cast<float4 *>(tile + r * IWP + 4 * c4) = src[i]; }
	// layer 2's weight fragments, on their way while layer 1 runs
	const int n2 = 16 * wave + (lane & 15);
	float breg[64];
#pragma unroll
	for (int ks = 0; ks < 64; ks++) breg[ks] = W2p[(4 * ks + (lane >> 4)) * 64 + n2];
	const float bias2 = B2[n2];
	// ---- layer 1 (k_conv1)
	{
		const int n = lane & 15, g = lane >> 4, px = lane & 3, py = (lane >> 2) & 3;
		float wreg[7]; int aoff[7];
#pragma unroll
		for (int s = 0; s < 7; s++)
		{
			const int k = 4 * s + g, kk = k < 25 ? k : 24;
			wreg[s] = k < 25 ? W1[n * 25 + k] : 0.0f;
			aoff[s] = (py + kk / 5) * IWP + px + kk % 5;
		}
		const float bias = B1[n];
		__syncthreads();
		const int swave = __builtin_amdgcn_readfirstlane(wave);
		float *const pdst = pooled + n * CH;
		for (int w = swave; w < PR * PW; w += 4)
		{
			const int ty = w / PW, tx = w % PW;
			const float *base = tile + 4 * ty * IWP + 4 * tx;
			f32x4 acc = { bias, bias, bias, bias };
#pragma unroll
			for (int s = 0; s < 7; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(base[aoff[s]], wreg[s], acc, 0, 0, 0);
			float m = max_pool4(acc[0], acc[1], acc[2], acc[3]);
			{ const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false); m = max_pool(__uint_as_float(r[0]), __uint_as_float(r[1])); }
			{ const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false); m = max_pool(__uint_as_float(r[0]), __uint_as_float(r[1])); }
			if (lane < 16) pdst[ty * PW + tx] = m;
		}
		__syncthreads();
		// tanh after pooling (monotone), into the dead input tile in the order layer 2 reads it: [position][ic & 3][ic >> 2] -- a lane of the matrix instruction takes
		// ic = 4 * j + (lane >> 4) for the four k-steps j of a tap, so its four values are one 128-bit read (16 reads per tile instead of 64 values in 32 double reads)
		for (int i = t; i < 16 * CH; i += 256)
		{
			const float v = tanh_ref(pooled[i]);
			const int ic = i / CH, pos = i % CH;
			tile[(pos * 4 + (ic & 3)) * 4 + (ic >> 2)] = v;
			act1[(size_t)b * (16 * CH) + i] = v;
		}
		__syncthreads();
	}
	// ---- layer 2 (k_conv2<15, 12, 12>), its input from `tile`
	{
		const int icl = lane >> 4;
		const int arow = lane & 15, awin = arow >> 2, ady = (arow >> 1) & 1, adx = arow & 1;
		float *const dst = act2 + (size_t)b * (64 * PO * PO) + n2 * (PO * PO);
		for (int mt = 0; mt < MT; mt++)
		{
			const int win = 4 * mt + awin, wy = win / PO, wx = win % PO;
			f32x4 acc = { bias2, bias2, bias2, bias2 };
			const float *base = tile + (((2 * wy + ady) * IWD + 2 * wx + adx) * 4 + icl) * 4;
#pragma unroll
			for (int p = 0; p < 16; p++)      // tap p = (ky, kx); k = 16 * p + 4 * j + icl, ascending as in k_conv2
			{
				const float4 a = *reinterpret_cast<const float4 *>(base + ((p >> 2) * IWD + (p & 3)) * 16);
				acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, breg[4 * p + 0], acc, 0, 0, 0);
				acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, breg[4 * p + 1], acc, 0, 0, 0);
				acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, breg[4 * p + 2], acc, 0, 0, 0);
				acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, breg[4 * p + 3], acc, 0, 0, 0);
			}
			const float mm = max_pool4(acc[0], acc[1], acc[2], acc[3]);
			dst[4 * mt + icl] = tanh_ref(mm);
		}
	}
}

// ------------------------------------------------------------------------------------------------- k_fc
// C[M][N] = bias[N] + A[M][K] * W[K][N]  (W row-major as stored in the .cnnb, cnn.h:417)  [+ tanh]
// Block tile 128(M) x 64(N) x 32(K) on 8 waves laid out 4(M) x 2(N), one 32x32 accumulator tile (v_mfma_f32_32x32x2_f32) per wave: a CU
// holds two waves per SIMD, so one wave's LDS reads and the global prefetch of the next k-slab overlap the other's matrix instructions
// (an fp32 32x32x2 MFMA occupies the pipe for 64 cycles).  LDS tiles are double-buffered: one barrier per k-slab.
// Every output element accumulates k in ascending order from its bias, like the reference's loop (cnn.h:407-426).
#define FC_BK 32      // k-depth of a slab (64 was measured slower, here as in k_fc144: 116 against 100 us)
// WN = waves along N, WM = waves along M: the block tile is 32*WM x 32*WN.  N = 2048 uses WN = 2; N = 2304 used WN = 3 (192 blocks of
// 12 waves) before k_fc144 took that layer.  WM = 4: 128 rows, 8 waves, one block per CU at 1024 frames.  (Half tiles, WM = 2 with two blocks of
// four waves per CU so that one block's barrier does not idle the matrix pipes, were measured: 100 us either way.)
#ifdef HT_TUNING
__constant__ int ht_fc_dbg;
#define FC_MARK(k) if (fcst) { const long long tn = clock64(); fcc[k] += tn - ftm; ftm = tn; }
#else
#define FC_MARK(k)
#endif
// (This layer with ITS operands packed the same way -- act2 from k_conv2 in [k & 1][k >> 1] order within blocks of 8, a packed copy of W3, four + four 128-bit fragment reads per
// slab instead of 16 + 16 -- was built three ways: tiles by LDS-DMA with two buffers 113 us, with three buffers and the fragments read a slab ahead 100 us, this kernel's own
// register staging and slab order 100 us, 92 us once the 128-bit stores of A were spread over the banks: against this kernel's 93 us no gain for 19 + 103 MB of packed weights.)
// PACK16: the output feeds k_fc144_pk, which reads a lane's four k-values of a 16-k block with one 128-bit LDS read: column c of a row is stored at position
// (c & ~15) | (c & 3) << 2 | (c >> 2) & 3 -- within every block of 16 columns the order [k & 3][k >> 2] (ht_get_cnn_layers undoes it for callers that look at the layer)
template <bool TANH, int WN, int WM, bool FULL, bool PACK16 = false>      // FULL: M is a multiple of the tile's rows (no row test: the slab is then one branch-free region)
__global__ __launch_bounds__(64 * WM * WN) void k_fc(const float *__restrict__ A, const float *__restrict__ W, const float *__restrict__ bias, float *__restrict__ C, int M, int N, int K)
{
	constexpr int BM = 32 * WM, BN = 32 * WN, NT = 64 * WM * WN, LDA = BM + 1;
	__shared__ float As[2][FC_BK * LDA];
	__shared__ __attribute__((aligned(16))) float Bs[2][FC_BK * BN];
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave / WN, wn = wave % WN;
	const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
	// staging assignments: A tile = BM*BK/4 float4 (row = e / (BK/4), 4 consecutive k), B tile = BK*BN/4 float4 (k-row = e / (BN/4)).  Global loads run TWO
	// slabs ahead of the matrix instructions through two register sets (a slab's ~1000 MFMA cycles per wave are shorter than a loaded memory
	// system's latency: with one slab of distance the LDS store at the end of every slab waited for its loads).
	constexpr int KV = FC_BK / 4, NAV = BM * KV, NBV = FC_BK * BN / 4;      // float4 per tile row of A, per A tile, per B tile
	constexpr int NA = (NAV + NT - 1) / NT, NB = (NBV + NT - 1) / NT;
	float4 ra[2][NA], rb[2][NB];
	// every address of the staging is formed once, outside the slab loop (formed per slab they were ~80 instructions of every wave's 72 slabs, during which
	// the SIMD's matrix pipe had nothing to do: tools/fc_stats.py)
	const float *ag[NA], *bg[NB]; bool aon[NA], bon[NB]; int al[NA], bl[NB];
#pragma unroll
	for (int i = 0; i < NA; i++)
	{
		const int e = t + NT * i, row = e / KV, akc = (e % KV) * 4;
		aon[i] = e < NAV && (FULL || m0 + row < M);
		ag[i] = A + (size_t)(aon[i] ? m0 + row : 0) * K + (e < NAV ? akc : 0);
		al[i] = e < NAV ? akc * LDA + row : -1;
	}
#pragma unroll
	for (int i = 0; i < NB; i++)
	{
		const int e = t + NT * i, bk = e / (BN / 4), bnc = (e % (BN / 4)) * 4;
		bon[i] = e < NBV;
		bg[i] = W + (size_t)(bon[i] ? bk : 0) * N + n0 + (bon[i] ? bnc : 0);
		bl[i] = bk * BN + bnc;
	}
	auto gload = [&](float4 (&qa)[NA], float4 (&qb)[NB], int k0) {
#pragma unroll
		for (int i = 0; i < NA; i++) { const float4 q = *reinterpret_cast<const float4 *>(ag[i] + k0); qa[i] = (FULL && NAV == NA * NT) || aon[i] ? q : make_float4(0, 0, 0, 0); }
#pragma unroll
		for (int i = 0; i < NB; i++) { const float4 q = *reinterpret_cast<const float4 *>(bg[i] + (size_t)k0 * N); qb[i] = NBV == NB * NT || bon[i] ? q : make_float4(0, 0, 0, 0); }
	};
	auto lstore = [&](const float4 (&qa)[NA], const float4 (&qb)[NB], int buf) {
#pragma unroll
		for (int i = 0; i < NA; i++)
			if (NAV == NA * NT || al[i] >= 0) { float *a = As[buf] + al[i]; a[0] = qa[i].x; a[LDA] = qa[i].y; a[2 * LDA] = qa[i].z; a[3 * LDA] = qa[i].w; }
#pragma unroll
		for (int i = 0; i < NB; i++)
			if (NBV == NB * NT || bon[i]) *reinterpret_cast<float4 *>(Bs[buf] + bl[i]) = qb[i];
	};
	const float bv = bias[n0 + wn * 32 + (lane & 31)];
	f32x16 acc;
#pragma unroll
	for (int r = 0; r < 16; r++) acc[r] = bv;
#ifdef HT_TUNING
	const bool fcst = (ht_fc_dbg & 0x800000) != 0; long long fcc[5] = { 0, 0, 0, 0, 0 }, ftm = fcst ? clock64() : 0; const long long fc_t0 = ftm;
#endif
	gload(ra[0], rb[0], 0);
	if (FC_BK < K) gload(ra[1], rb[1], FC_BK);
	lstore(ra[0], rb[0], 0);
	FC_MARK(4)
	for (int k0 = 0; k0 < K; k0 += 2 * FC_BK)
	{
#pragma unroll
		for (int u = 0; u < 2; u++)      // slab k0/32 + u: its tile is in LDS buffer u, the next slab's loads are in flight in register set u ^ 1
		{
			const int kc = k0 + u * FC_BK;
			if (kc >= K) break;
			__syncthreads();                               // buffer u is complete; buffer u ^ 1 is free (its readers passed this barrier)
			FC_MARK(0)
			const float *ap = As[u] + (lane >> 5) * LDA + wm * 32 + (lane & 31);
			const float *bp = Bs[u] + (lane >> 5) * BN + wn * 32 + (lane & 31);
			// Order of a slab's instructions (fenced with sched_barrier): the operand fragments of its first half are read up front, the second half's reads are slotted
			// between the first matrix instructions (left to itself the scheduler issues each B pair right before the two MFMAs that use it: an LDS round trip
			// every 128 pipe cycles), and the GLOBAL loads of the slab after next are issued in the middle, behind eight matrix instructions: a wave spends ~300
			// cycles issuing its three loads and ~250 on the five stores of the next slab's tile (the CU's one address path serves all eight waves: tools/fc_stats.py), and issued first thing after the barrier, as
			// they were, both waves of a SIMD stood in that queue while the matrix pipe had nothing to do.  (The last slabs load an address again; unused.)
			float av[FC_BK / 2], bw[FC_BK / 2];
			constexpr int H = FC_BK / 4;      // matrix instructions per half slab
#pragma unroll
			for (int kk = 0; kk < H; kk++) { av[kk] = ap[2 * kk * LDA]; bw[kk] = bp[2 * kk * BN]; }
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int kk = 0; kk < H; kk++)      // first half, with the second half's fragments read behind each instruction
			{
				acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bw[kk], acc, 0, 0, 0);
				av[H + kk] = ap[2 * (H + kk) * LDA]; bw[H + kk] = bp[2 * (H + kk) * BN];
			}
			__builtin_amdgcn_sched_barrier(0);
			const int kn = kc + 2 * FC_BK < K ? kc + 2 * FC_BK : K - FC_BK;
			gload(ra[u], rb[u], kn);      // the slab after next (set u was stored one slab ago)
			__builtin_amdgcn_sched_barrier(0);
			acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[H], bw[H], acc, 0, 0, 0);
			acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[H + 1], bw[H + 1], acc, 0, 0, 0);
			__builtin_amdgcn_sched_barrier(0);
			lstore(ra[u ^ 1], rb[u ^ 1], u ^ 1);      // the next slab's tile (loaded a slab and a half ago) into the buffer whose readers passed this slab's barrier
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int kk = H + 2; kk < 2 * H; kk++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bw[kk], acc, 0, 0, 0);
			FC_MARK(2)
			FC_MARK(3)
		}
	}
#ifdef HT_TUNING
	if (fcst && lane == 0 && (wave == 0 || wave == 5) && (blockIdx.x + blockIdx.y * gridDim.x) % 64 == 0) printf("k_fc block %d,%d wave %d: barrier wait %lld, load issue %lld, reads+mfma issue %lld, store %lld, prologue %lld, whole loop %lld cycles (%d slabs)\n", blockIdx.x, blockIdx.y, wave, fcc[0], fcc[1], fcc[2], fcc[3], fcc[4], (long long)(clock64() - fc_t0), K / FC_BK);
#endif
	// C/D map 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
	const int colu = n0 + wn * 32 + (lane & 31), col = PACK16 ? ((colu & ~15) | ((colu & 3) << 2) | ((colu >> 2) & 3)) : colu;
#pragma unroll
	for (int r = 0; r < 16; r++)
	{
		const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
		const int g0 = m0 + wm * 32 + row;
		float v0 = acc[r];
		if (TANH) v0 = tanh_ref(v0);
		if (g0 < M) C[(size_t)g0 * N + col] = v0;
	}
}

// ------------------------------------------------------------------------------------------------- k_fc144
// The last layer (2048 -> 2304) on a 64 x 144 block tile: 1024 frames x 2304 outputs are then exactly 16 x 16 = 256 blocks, one per CU (the
// 128 x 96 tile of k_fc<.,3> makes 192 blocks and leaves a quarter of the chip idle; 128 x 64 makes 288 and a second, nearly empty round).
// 144 is nine 16-wide tiles, so the arithmetic is v_mfma_f32_16x16x4_f32: 12 waves as 4 (rows) x 3 (columns), a wave holds three 16 x 16
// accumulators that share one A fragment; double-buffered 32-deep slabs; accumulation starts from the bias and runs in ascending k.
// History of the kernel: tiles staged through registers with A stored transposed at stride 81 (105.5 us at 1024 frames; 64-deep slabs 136 us; k_fc's
// hand-ordered slab 135 us) -> the same tiles by LDS-DMA (100.6 us; 64-deep 102.7; the pieces asked for mid-slab: no change) -> operands packed for 128-bit fragment reads (89.2 us).
#define F2_BM 64
#define F2_BN 144
#define F2_BK 32
typedef const __attribute__((address_space(1))) void *ht_gptr;
typedef __attribute__((address_space(3))) void *ht_lptr;
// The last layer with BOTH operands in the order the matrix instruction takes them.  v_mfma_f32_16x16x4_f32 wants from lane l the element (row or column l & 15,
// k = 4 * step + (l >> 4)); with A and W row-major that is one 32-bit LDS read per operand and instruction, and the reads, their address arithmetic and waits share the
// SIMD's one issue port with the matrix instructions of the other waves: the pipe was ~75 % busy inside the matrix phase whatever the staging did.
// Here a lane's four k-values of a 16-k block are contiguous -- A comes from k_fc<PACK16> that way ([row][k >> 4][k & 3][(k >> 2) & 3]), W from ht_launch_pack_w4
// ([k >> 4][k & 3][n][(k >> 2) & 3]) -- so a 32-deep slab is TWO 128-bit reads of A and SIX of B per wave for its 24 matrix instructions, where it was 8 + 16 reads.
// Tiles by LDS-DMA (global_load_lds_dwordx4: a wave-instruction writes 64 x 16 bytes to LDS at a wave-uniform base + 16 * lane -- no staging registers, no ds_write pass;
// the SOURCE address is per lane; the compiler does not wait for such a load on its own, hence the explicit s_waitcnt): per slab 8 pieces of A (64 rows x 8 chunks, chunk slot = (h * 4 + g) ^ (row & 7): XOR-swizzled on the source side) and 18 of B
// (8 runs (h, g) of 144 columns x 16 bytes, contiguous in W4p); the sums run over k in the same order as before (k = 16 * kb + 4 * kk + g ascending in kb, kk).
__global__ __launch_bounds__(768) void k_fc144_pk(const float *__restrict__ Ap, const float *__restrict__ W4p, const float *__restrict__ bias, float *__restrict__ C, int M, int N, int K)
{
	constexpr int BK = F2_BK, NPA = 8, NPB = 18, NP = NPA + NPB, PW = 3;
	__shared__ __attribute__((aligned(1024))) float As[2][F2_BM * BK];
	__shared__ __attribute__((aligned(1024))) float Bs[2][BK * F2_BN];
	const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wm = wave / 3, wn = wave % 3;
	const int m0 = blockIdx.y * F2_BM, n0 = blockIdx.x * F2_BN;
	const float *src[PW]; int dst[PW]; size_t step[PW];
#pragma unroll
	for (int j = 0; j < PW; j++)
	{
		const int q = wave + 12 * j;
		if (q < NPA)
		{
			const int c = 64 * q + lane, row = c >> 3, hg = (c & 7) ^ (row & 7);      // chunk (h, g) = hg >> 2, hg & 3 of the row: 16 bytes at [kb = 2 * slab + h][g][0..3]
			const int grow = m0 + row < M ? m0 + row : M - 1;
			src[j] = Ap + (size_t)grow * K + 4 * hg; dst[j] = 256 * q; step[j] = BK;
		}
		else
		{
			const int c = 64 * (q - NPA) + lane, hg = c / F2_BN, nn = c % F2_BN;      // run (h, g), column nn of the block's 144
			src[j] = W4p + ((size_t)(hg < 8 ? hg : 0) * N + n0 + nn) * 4; dst[j] = 256 * (q - NPA); step[j] = (size_t)8 * N * 4;      // a slab is 2 x 4 runs of N x 4 floats
		}
	}
	auto dma = [&](int buf, int slab) {
#pragma unroll
		for (int j = 0; j < PW; j++)
		{
			const int q = wave + 12 * j;
			if (q >= NP) break;
			float *l = (q < NPA ? As[buf] : Bs[buf]) + dst[j];
			__builtin_amdgcn_global_load_lds((ht_gptr)(src[j] + (size_t)slab * step[j]), (ht_lptr)l, 16, 0, 0);
		}
	};
	f32x4 acc[3];
#pragma unroll
	for (int j = 0; j < 3; j++) { const float bv = bias[n0 + wn * 48 + j * 16 + (lane & 15)]; acc[j] = f32x4{ bv, bv, bv, bv }; }
	const int arow = wm * 16 + (lane & 15), g = lane >> 4;
	const int aoff0 = (arow * 8 + ((0 * 4 + g) ^ (arow & 7))) * 4, aoff1 = (arow * 8 + ((1 * 4 + g) ^ (arow & 7))) * 4;
	const int boff = (g * F2_BN + wn * 48 + (lane & 15)) * 4;
	const int nslab = K / BK;
#ifdef HT_TUNING
	const bool fcst = (ht_fc_dbg & 0x800000) != 0; long long fcc[5] = { 0, 0, 0, 0, 0 }, ftm = fcst ? clock64() : 0; const long long fc_t0 = ftm;
#endif
	// (Three buffers with the tiles asked for two slabs ahead and a slab's fragments read during the slab before it -- nothing between a slab's barrier and its first matrix
	// instruction -- were measured: 95.8 us against 89.2 for this plain order; three waves per SIMD cover the head of a slab, and the fences the other order needs cost more.)
	dma(0, 0);
	for (int s = 0; s < nslab; s++)
	{
		const int buf = s & 1;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of slab s have landed ...
		FC_MARK(0)
		__syncthreads();                                   // ... and everybody's; the other buffer's readers are through
		FC_MARK(1)
		if (s + 1 < nslab) dma(buf ^ 1, s + 1);
		FC_MARK(2)
		const float *ap = As[buf], *bp = Bs[buf] + boff;
		const float4 a0 = *reinterpret_cast<const float4 *>(ap + aoff0), a1 = *reinterpret_cast<const float4 *>(ap + aoff1);
		float4 b[2][3];
#pragma unroll
		for (int h = 0; h < 2; h++)
#pragma unroll
			for (int j = 0; j < 3; j++) b[h][j] = *reinterpret_cast<const float4 *>(bp + (h * 4 * F2_BN + j * 16) * 4);
#define F2_STEP(av, h, comp) \
		acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.comp, b[h][0].comp, acc[0], 0, 0, 0); \
		acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.comp, b[h][1].comp, acc[1], 0, 0, 0); \
		acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.comp, b[h][2].comp, acc[2], 0, 0, 0);
		F2_STEP(a0, 0, x) F2_STEP(a0, 0, y) F2_STEP(a0, 0, z) F2_STEP(a0, 0, w)
		F2_STEP(a1, 1, x) F2_STEP(a1, 1, y) F2_STEP(a1, 1, z) F2_STEP(a1, 1, w)
#undef F2_STEP
		FC_MARK(3)
	}
#ifdef HT_TUNING
	if (fcst && lane == 0 && (wave == 0 || wave == 4 || wave == 8) && (blockIdx.x + blockIdx.y * gridDim.x) % 128 == 0) printf("k_fc144_pk block %d,%d wave %d: wait for own pieces %lld, barrier %lld, DMA issue %lld, reads + matrix instructions %lld, whole loop %lld cycles (%d slabs)\n", blockIdx.x, blockIdx.y, wave, fcc[0], fcc[1], fcc[2], fcc[3], (long long)(clock64() - fc_t0), nslab);
#endif
	// C/D map 16x16: col = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
	for (int j = 0; j < 3; j++)
#pragma unroll
		for (int r = 0; r < 4; r++)
		{
			const int row = m0 + wm * 16 + 4 * (lane >> 4) + r;
			if (row < M) C[(size_t)row * N + n0 + wn * 48 + j * 16 + (lane & 15)] = acc[j][r];
		}
}
// W4p[((kb * 4 + g) * N + n) * 4 + kk] = W4[(16 * kb + 4 * kk + g) * N + n]      (W4 row-major [K = 2048][N = 2304] as stored in the .cnnb, cnn.h:417)
__global__ __launch_bounds__(256) void k_pack_w4(const float *__restrict__ W4, float *__restrict__ W4p, int K, int N)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one 16-byte chunk (kb, g, n) per thread
	if (i >= (size_t)K * N / 4) return;
	const int n = (int)(i % N), kg = (int)(i / N), kb = kg >> 2, g = kg & 3;
	float4 v;
	v.x = W4[(size_t)(16 * kb + 0 + g) * N + n]; v.y = W4[(size_t)(16 * kb + 4 + g) * N + n]; v.z = W4[(size_t)(16 * kb + 8 + g) * N + n]; v.w = W4[(size_t)(16 * kb + 12 + g) * N + n];
	reinterpret_cast<float4 *>(W4p)[i] = v;
}
void ht_launch_pack_w4(const float *W4, float *W4p, hipStream_t s)
{
	hipLaunchKernelGGL(k_pack_w4, dim3((unsigned)((HT_W4_COUNT / 4 + 255) / 256)), dim3(256), 0, s, W4, W4p, 2048, 2304);
}

// ------------------------------------------------------------------------------------------------- k_softmax_decode
// one wave per frame.  softmax chunks: 8 x 256 then 16 x 16 (handtrack.h:118); sums run in ascending order like cnn.h:503-505.
// analysis layout (HT_ANALYSIS floats): crays 8x4 | image_points 8x2 | confidence 8 | vals 16 | wristroll pitch tilt | palmq 4 | clenched 5
__device__ void decode_frame(const float *__restrict__ y, const float *__restrict__ cam, float *__restrict__ an, int lane, float sub)
{
	// hcam = camsub(cam,4) misc_image.h:60 (sub = 8 for the 128x128-input net: 16x16 heat-maps again)
	const float fx = cam[0] / sub, fy = cam[1] / sub, cx = cam[2] / sub, cy = cam[3] / sub;
	// ImageFindMax (misc_image.h:298-305: first maximum in a row-major scan) of the 8 heat-maps: 8 lanes per map take 32 consecutive values
	// each, then the partial results are merged in index order (a later segment only wins with a strictly larger value)
	int amax;
	{
		const int m = lane >> 3, seg = lane & 7;
		const float *base = y + 256 * m + 32 * seg;
		float best = base[0]; int bi = 32 * seg;
		for (int i = 1; i < 32; i++) { const float v = base[i]; if (v > best) { best = v; bi = 32 * seg + i; } }
#pragma unroll
		for (int o = 1; o < 8; o <<= 1)
		{
			const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
			if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
		}
		amax = __shfl(bi, lane * 8);      // lane m (< 8) picks up the result of map m
	}
	if (lane < 8)
	{
		const float *base = y + 256 * lane;
		const int mxx = amax & 15, mxy = amax >> 4;
		float wsum = 0.0f, vx = 0.0f, vy = 0.0f;
		for (int sy = max(0, mxy - 1); sy < min(16, mxy + 2); sy++) for (int sx = max(0, mxx - 1); sx < min(16, mxx + 2); sx++)
		{
			float w = base[sy * 16 + sx];
			vx = vx + (float)sx * w; vy = vy + (float)sy * w; wsum += w;                                                   // PeakSubPixel misc_image.h:312-325
		}
		float px, py;
		if (wsum == 0) { px = (float)mxx; py = (float)mxy; } else { px = vx / wsum; py = vy / wsum; }
		int ix = (int)(px + 0.5f), iy = (int)(py + 0.5f);
		float vol = 0.0f;
		for (int sy = max(0, iy - 1); sy < min(16, iy + 2); sy++) for (int sx = max(0, ix - 1); sx < min(16, ix + 2); sx++) vol += base[sy * 16 + sx];   // PeakVolume :328-336
		xf cp = XF(V3(cam[5], cam[6], cam[7]), V4(cam[8], cam[9], cam[10], cam[11]));
		v3 nrm = normalize(apply(cp, V3((px - cx) / fx, (py - cy) / fy, 1.0f) * 1.0f));
		an[4 * lane + 0] = nrm.x; an[4 * lane + 1] = nrm.y; an[4 * lane + 2] = nrm.z; an[4 * lane + 3] = base[16 * mxy + mxx];
		an[HT_AN_IMGPT + 2 * lane] = px; an[HT_AN_IMGPT + 2 * lane + 1] = py;
		an[HT_AN_CONF + lane] = vol;
	}
	else if (lane < 24)
	{
		const int row = lane - 8;
		const float *r = y + 2048 + 16 * row;
		int p = 0;
		for (int x = 1; x < 16; x++) if (r[p] < r[x]) p = x;                                                              // Peaks1D misc_image.h:389-399
		float v = 0.0f, wsum = 0.0f;
		for (int i = max(0, p - 1); i < min(16, p + 2); i++) { float w = r[i]; v += (float)i * w; wsum += w; }
		an[HT_AN_VALS + row] = ((wsum == 0) ? (float)p : v / wsum) / (float)(16 - 1);
	}
}
__device__ void calc_angles(float *an)      // handtrack.h:194-202
{
	const float *vals = an + HT_AN_VALS;
	float wristroll = vals[0] * 3.1415f * 2.0f + 3.1415f / 2.0f;
	float pitch = (vals[1] - 0.5f) * 3.1415f;
	float tilt = (vals[2] - 0.5f) * 3.1415f;
	v4 palmq = qmul(normalize(V4(1.0f, 0, 0, 1.0f)), qmul(quat_axis_angle(V3(-1, 0, 0), pitch), quat_axis_angle(V3(0, 0, 1), wristroll)));
	an[HT_AN_ANGLES + 0] = wristroll; an[HT_AN_ANGLES + 1] = pitch; an[HT_AN_ANGLES + 2] = tilt;
	an[HT_AN_PALMQ + 0] = palmq.x; an[HT_AN_PALMQ + 1] = palmq.y; an[HT_AN_PALMQ + 2] = palmq.z; an[HT_AN_PALMQ + 3] = palmq.w;
	for (int i = 0; i < 5; i++) an[HT_AN_CLENCH + i] = vals[3 + i] * 3.1415f;
}
// softmax=1: logits -> probabilities (written to cnn_out) then decode; softmax=0: decode an existing cnn_out
__global__ __launch_bounds__(64) void k_softmax_decode(const float *__restrict__ logits, float *__restrict__ cnn_out, const float *__restrict__ cams, float *__restrict__ analysis, int softmax, float sub)
{
	__shared__ float y[HT_CNN_OUT];
	__shared__ float an[HT_ANALYSIS];
	const int b = blockIdx.x, lane = threadIdx.x;
	if (softmax)
	{
		for (int i = lane; i < HT_CNN_OUT; i += 64) y[i] = (float)exp((double)logits[(size_t)b * HT_CNN_OUT + i]);
		__syncthreads();
		__shared__ float csum[24];
		if (lane < 24)       // chunk sums in ascending order like cnn.h:503-505 (one lane per chunk; 16 values are read per trip, then added one by one)
		{
			const int s = lane < 8 ? 256 : 16, base = lane < 8 ? 256 * lane : 2048 + 16 * (lane - 8);
			float sum = 0.0f;
			for (int i = base; i < base + s; i += 16)
			{
				float v[16];
#pragma unroll
				for (int k = 0; k < 16; k++) v[k] = y[i + k];
#pragma unroll
				for (int k = 0; k < 16; k++) sum += v[k];
			}
			csum[lane] = sum;
		}
		__syncthreads();
		for (int i = lane; i < HT_CNN_OUT; i += 64)
		{
			const float v = y[i] / csum[i < 2048 ? (i >> 8) : 8 + ((i - 2048) >> 4)];
			y[i] = v; cnn_out[(size_t)b * HT_CNN_OUT + i] = v;
		}
		__syncthreads();
	}
	else
	{
		for (int i = lane; i < HT_CNN_OUT; i += 64) y[i] = cnn_out[(size_t)b * HT_CNN_OUT + i];
		__syncthreads();
	}
	if (!analysis) return;
	decode_frame(y, cams + (size_t)b * HT_CAM, an, lane, sub);
	__syncthreads();
	if (lane == 0) calc_angles(an);
	__syncthreads();
	for (int i = lane; i < HT_ANALYSIS; i += 64) analysis[(size_t)b * HT_ANALYSIS + i] = an[i];
}

// ------------------------------------------------------------------------------------------------- k_voxel
// voxelsubsample<2048> (physmodel.h:66-118) of a frame's in-range points, for the main-thread cloud when HandTracker::subsample_voxel is set
// (handtrack.h:751): the points are summed per voxel in an open-addressing table of 2048 buckets (hash = dot of the integer cell with three
// primes, linear probing) in the order they come -- a float sum per bucket, so the order is the result -- and a full table flushes the home bucket
// of the point that found no room; then every bucket holding at least `min_count` points gives its mean, in table order.  One wave per frame: lane
// 0 inserts (a probe is one 128-bit LDS read of cell + count), all lanes clear the table and compact the output.  The reference converts the
// floor of a negative coordinate to unsigned (undefined behaviour); the compiled reference wraps, which is what the signed conversion here gives.
#define VOX_N 2048
__global__ __launch_bounds__(64) void k_voxel(const float4 *__restrict__ all, const int *__restrict__ nall, int cap, float size, int min_count, float4 *__restrict__ out, int *__restrict__ nout)
{
	__shared__ int4 key[VOX_N];        // cell x, y, z, count
	__shared__ float4 sum[VOX_N];
	const int b = blockIdx.x, lane = threadIdx.x;
	for (int i = lane; i < VOX_N; i += 64) { key[i] = make_int4(0, 0, 0, 0); sum[i] = make_float4(0, 0, 0, 0); }
	__syncthreads();
	const float4 *src = all + (size_t)b * cap;
	float4 *dst = out + (size_t)b * cap;
	const int n = nall[b];
	__shared__ int nflush;
	if (lane == 0)
	{
		const float ivs = 1.0f / size;
		int m = 0;
		for (int k = 0; k < n; k++)
		{
			const float4 pt = src[k];
			const int ix = (int)floorf(pt.x * ivs), iy = (int)floorf(pt.y * ivs), iz = (int)floorf(pt.z * ivs);
			const unsigned hash = (54851u * (unsigned)ix + 11909u * (unsigned)iy) + 24781u * (unsigned)iz;
			unsigned i = 0;
			for (; i < VOX_N; i++)
			{
				const unsigned s = (hash + i) & (VOX_N - 1);
				const int4 q = key[s];
				if (q.w == 0 || (q.x == ix && q.y == iy && q.z == iz))
				{
					const float4 a = sum[s];
					key[s] = make_int4(ix, iy, iz, q.w + 1);
					sum[s] = make_float4(a.x + pt.x, a.y + pt.y, a.z + pt.z, 0.0f);
					break;
				}
			}
			if (i == VOX_N)      // flush on collision
			{
				const unsigned s = hash & (VOX_N - 1);
				const float4 a = sum[s]; const float c = (float)key[s].w;
				if (m < cap) dst[m] = make_float4(a.x / c, a.y / c, a.z / c, 0.0f);
				m++;
				key[s] = make_int4(ix, iy, iz, 1); sum[s] = make_float4(pt.x, pt.y, pt.z, 0.0f);
			}
		}
		nflush = m;
	}
	__syncthreads();
	int m = nflush;
	for (int base = 0; base < VOX_N; base += 64)
	{
		const int4 q = key[base + lane];
		const bool keep = q.w >= min_count;
		const unsigned long long mask = __ballot(keep);
		if (keep)
		{
			const int o = m + __popcll(mask & ((1ull << lane) - 1ull));
			const float4 a = sum[base + lane]; const float c = (float)q.w;
			if (o < cap) dst[o] = make_float4(a.x / c, a.y / c, a.z / c, 0.0f);
		}
		m += __popcll(mask);
	}
	if (lane == 0) nout[b] = m < cap ? m : cap;
}

// ------------------------------------------------------------------------------------------------- host launchers
void ht_launch_voxel(const float4 *all, const int *nall, int cap, float size, int min_count, float4 *out, int *nout, int B, hipStream_t s)
{
	hipLaunchKernelGGL(k_voxel, dim3(B), dim3(64), 0, s, all, nall, cap, size, min_count, out, nout);
}
void ht_launch_prepare(const uint16_t *depth, const float *cams, float drangey, int fraction, float *cnn_in, float4 *pts, int *npts, int cap, int B, hipStream_t s, const ht_prepare_extra *extra)
{
	ht_prepare_extra x = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr };
	if (extra) x = *extra;
	hipLaunchKernelGGL(k_prepare, dim3(B), dim3(256), 0, s, depth, cams, drangey, fraction, cnn_in, pts, npts, cap, x);
}
void ht_launch_prepare_frame(const uint16_t *depth, const float *cams, int w, int h, float drangey, int fraction, float4 *pts, int *npts, int *overflow, int cap, int B, hipStream_t s)
{
	hipLaunchKernelGGL(k_prepare_frame, dim3(B), dim3(256), 0, s, depth, cams, w, h, drangey, fraction, pts, npts, overflow, cap);
}
// side = 64: PoseInitializerCNN's topology (handtrack.h:108-118); side = 128: the same layers on a 128x128 input (act1 [B][16*31*31], act2 [B][12544])
void ht_launch_cnn(const ht_cnn_weights &w, const float *cnn_in, float *act1, float *act2, float *act3, float *logits, int B, hipStream_t s, int side, bool beside_other_work)
{
	const dim3 g1(2048 / 64, (B + 127) / 128), t1(512);
#ifdef HT_TUNING
	{ static int done = 0; if (!done) { const int f = ht_tuning_flags(); (void)hipMemcpyToSymbol(HIP_SYMBOL(ht_fc_dbg), &f, sizeof(int)); done = 1; } }
#endif
	if (side == 128)
	{
		hipLaunchKernelGGL((k_conv1<128, 31, 8>), dim3(B, 4), dim3(256), 0, s, cnn_in, w.W1, w.B1, act1);
		hipLaunchKernelGGL((k_conv2<31, 28, 4>), dim3(B, 7), dim3(256), 0, s, act1, w.W2p, w.B2, act2);
		if (B % 128 == 0) hipLaunchKernelGGL((k_fc<true, 2, 4, true, true>), g1, t1, 0, s, act2, w.W3, w.B3, act3, B, 2048, 12544);
		else hipLaunchKernelGGL((k_fc<true, 2, 4, false, true>), g1, t1, 0, s, act2, w.W3, w.B3, act3, B, 2048, 12544);
	}
	else
	{
		// Alone on the chip the two convolution layers run as one launch (k_conv12: 90 us against 49 + 52).  Inside an update the carried pose's FitError runs beside the net on
		// a side stream (0.115 ms alone; the reset decision right behind the net needs it), and both want the whole chip: while the convolutions took 0.135 ms it ran under them
		// (conv2 82 us instead of 52); once they took 0.09 it reached into k_fc, whose one block per CU, all in step, lost 70 us to it (161 instead of 92).  Measured on one
		// device, same build, three runs each (ms per 1024-frame step, tuning build): two launches, FitError beside 5.694; two launches, FC layers waiting for it 5.713; one
		// launch, waiting 5.707; one launch, beside 5.737.  So an update keeps the two launches and a stand-alone evaluation takes the one: same results bit for bit.
		static const bool conv_split = ht_tuning_env("HT_CONV_SPLIT"), conv_fused = ht_tuning_env("HT_CONV_FUSED");      // measurement (-DHT_TUNING): force either arrangement
		if ((beside_other_work && !conv_fused) || conv_split)
		{
			hipLaunchKernelGGL((k_conv1<64, 15, 15>), dim3(B, 1), dim3(256), 0, s, cnn_in, w.W1, w.B1, act1);
			hipLaunchKernelGGL((k_conv2<15, 12, 12>), dim3(B, 1), dim3(256), 0, s, act1, w.W2p, w.B2, act2);
		}
		else hipLaunchKernelGGL(k_conv12, dim3(B), dim3(256), 0, s, cnn_in, w.W1, w.B1, w.W2p, w.B2, act1, act2);
		if (B % 128 == 0) hipLaunchKernelGGL((k_fc<true, 2, 4, true, true>), g1, t1, 0, s, act2, w.W3, w.B3, act3, B, 2048, 2304);
		else hipLaunchKernelGGL((k_fc<true, 2, 4, false, true>), g1, t1, 0, s, act2, w.W3, w.B3, act3, B, 2048, 2304);
	}
	// the last layer: act3 comes from k_fc in the packed column order and meets the packed copy of W4 (k_fc144_pk)
	hipLaunchKernelGGL(k_fc144_pk, dim3(2304 / F2_BN, (B + F2_BM - 1) / F2_BM), dim3(768), 0, s, act3, w.W4p, w.B4, logits, B, 2304, 2048);
}
void ht_launch_softmax_decode(const float *logits, float *cnn_out, const float *cams, float *analysis, int softmax, int B, hipStream_t s, int sub)
{
	hipLaunchKernelGGL(k_softmax_decode, dim3(B), dim3(64), 0, s, logits, cnn_out, cams, analysis, softmax, (float)sub);
}
void ht_launch_cnn_input(const uint16_t *depth, const float *cams, int npx, float drangey, float *cnn_in, int B, hipStream_t s)
{
	hipLaunchKernelGGL(k_cnn_input, dim3((npx / 8 + 255) / 256, B), dim3(256), 0, s, depth, cams, npx, drangey, cnn_in);
}
