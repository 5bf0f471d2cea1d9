// ht_train.hip -- one SGD step of the pose-initialiser CNN per sample.
//
// Reference computations (third_party/cnn.h): CNN::Train :558-580 (forward keeping every layer's output, E = y - t, backward() of
// layers 10..1 with the weights as they are, then update() of every layer), LConv :205-279, LMaxPool :141-164, LFull :405-445,
// LActivation<TanH> :457-469 (df = 1 - y*y), LSoftMaxChunked :497-526.  Topology: include/handtrack.h:108-118.
//
// Batch-1 SGD is sequential over samples, so the parallelism is inside one sample, and the two fully connected layers (2 x 18.9 MB of
// weights) make a step HBM-bound: the forward product streams each matrix once (rows split 32 ways, partial sums reduced in a fixed
// order), and the backward product and the weight update of a layer share one pass over its matrix (row i yields d[i] = W[i,:].E with
// the old weights and is then rewritten as W[i,:] - x[i]*E*alpha): 113 MB of traffic per sample.  Everything else is fused around
// those four passes so that a step is 9 launches:
//   conv1+tanh+pool+pool | conv2+tanh+pool | FC1 partial | (reduce+tanh of FC1) FC2 partial | reduce+softmax+loss+softmax' |
//   FC2 back+update (+bias, tanh', loss value) | FC1 back+update (+bias) | (pool', tanh') conv2 back |
//   (pool', pool', tanh') conv1 update, (pool', tanh') conv2 update + the MFMA-packed copy of conv2 the inference kernel reads
// The error that reaches conv1's update is non-zero at one position per 4x4 pooling window only (the two max-pools route it to the first
// maximum), so it is handled as a compact (position, value) list instead of three dense maps.
// Sums are reduced in parallel (fixed order, deterministic), so results agree with the reference to float rounding, not bit for bit.
#include "ht_device.hpp"
#include "ht_launch.hpp"

__device__ __forceinline__ float t_tanh(float t) { float e = (float)exp((double)(2 * t)); return (e - 1) / (e + 1); }      // TanH::f cnn.h:31

// first maximum of a 2x2 window in the reference's scan order (x then y, strict >: cnn.h:150-160)
__device__ __forceinline__ int first_max4(float a, float b, float c, float d, float &m)
{
	int k = 0; m = a;
	if (b > m) { m = b; k = 1; }
	if (c > m) { m = c; k = 2; }
	if (d > m) { m = d; k = 3; }
	return k;
}
__device__ __forceinline__ float max4(float a, float b, float c, float d) { return fmax_std(fmax_std(fmax_std(a, b), c), d); }
__device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ float seg16_sum(float v) { for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o); return v; }

// conv1 (5x5, 1 -> 16 channels, 64x64 -> 60x60) + tanh + the two max-pools (-> 30x30 -> 15x15).  Block (py, oz): four output rows of one
// channel, one thread per output, taps in the reference's order (cnn.h:226-228); the first 15 threads then pool the 4x60 strip.
__global__ __launch_bounds__(256) void k_t_conv1_tanh_pool(const float *__restrict__ in, const float *__restrict__ W, const float *__restrict__ B, float *__restrict__ a1, float *__restrict__ a3)
{
	__shared__ float s_o[4][60];
	const int py = blockIdx.x, oz = blockIdx.y, t = threadIdx.x;
	if (t < 240)
	{
		const int r = t / 60, x = t % 60, y = 4 * py + r;
		float acc = B[oz];
#pragma unroll
		for (int ky = 0; ky < 5; ky++)
#pragma unroll
			for (int kx = 0; kx < 5; kx++) acc += in[(y + ky) * 64 + x + kx] * W[kx + 5 * (ky + 5 * oz)];
		const float o = t_tanh(acc);
		a1[oz * 3600 + y * 60 + x] = o; s_o[r][x] = o;
	}
	__syncthreads();
	if (t < 15)
	{
		float q[4];
#pragma unroll
		for (int k = 0; k < 4; k++) { const int r = 2 * (k >> 1), c = 4 * t + 2 * (k & 1); q[k] = max4(s_o[r][c], s_o[r][c + 1], s_o[r + 1][c], s_o[r + 1][c + 1]); }
		a3[oz * 225 + py * 15 + t] = max4(q[0], q[1], q[2], q[3]);
	}
}
// conv2 (4x4, 16 -> 64 channels, 15x15 -> 12x12) + tanh + max-pool (-> 6x6): one block per output channel, input and taps in LDS
__global__ __launch_bounds__(192) void k_t_conv2_tanh_pool(const float *__restrict__ a3, const float *__restrict__ W, const float *__restrict__ B, float *__restrict__ a5, float *__restrict__ a6)
{
	__shared__ float s_in[3600], s_w[256], s_o[144];
	const int oz = blockIdx.x, t = threadIdx.x;
	for (int i = t; i < 3600; i += 192) s_in[i] = a3[i];
	for (int i = t; i < 256; i += 192) s_w[i] = W[oz * 256 + i];
	__syncthreads();
	if (t < 144)
	{
		const int x = t % 12, y = t / 12;
		float acc = B[oz];
		for (int ky = 0; ky < 4; ky++) for (int kx = 0; kx < 4; kx++)
#pragma unroll
			for (int iz = 0; iz < 16; iz++) acc += s_in[iz * 225 + (y + ky) * 15 + x + kx] * s_w[kx + 4 * (ky + 4 * iz)];
		const float o = t_tanh(acc);
		a5[oz * 144 + t] = o; s_o[t] = o;
	}
	__syncthreads();
	if (t < 36) { const float *p = s_o + (2 * (t / 6)) * 12 + 2 * (t % 6); a6[oz * 36 + t] = max4(p[0], p[1], p[12], p[13]); }
}
// y[j] = B[j] + sum_i x[i] W[i][j].  Block (jb, ks): rows [ks*M/32, (ks+1)*M/32) of 256 consecutive columns; wave w takes every fourth
// row of the slab, a lane four columns (128-bit loads); the four waves' sums are added in wave order.  part[ks][j].
// XPART: x is the tanh of the previous layer's output, still in partial sums xpart[32][M] + xb: the block reduces the slab's rows itself
// (and column-block 0 keeps them in xout for the backward pass).
#define T_KSPLIT 32
template <bool XPART> __global__ __launch_bounds__(256) void k_t_fc_partial(const float *__restrict__ x, const float *__restrict__ xpart, const float *__restrict__ xb, float *__restrict__ xout,
                                                                           const float *__restrict__ W, float *__restrict__ part, int M, int N)
{
	__shared__ float4 s_acc[4][64];
	__shared__ float s_x[72];
	const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, j = blockIdx.x * 256 + 4 * lane, ks = blockIdx.y, rows = M / T_KSPLIT;
	const int i0 = ks * rows;
	if (threadIdx.x < rows)
	{
		const int i = i0 + threadIdx.x;
		float v;
		if (XPART)
		{
			float acc = xb[i];
#pragma unroll 8
			for (int k = 0; k < T_KSPLIT; k++) acc += xpart[(size_t)k * M + i];
			v = t_tanh(acc);
			if (blockIdx.x == 0) xout[i] = v;
		}
		else v = x[i];
		s_x[threadIdx.x] = v;
	}
	__syncthreads();
	float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll 6
	for (int r = w; r < rows; r += 4)
	{
		const float xi = s_x[r];
		const float4 ww = *reinterpret_cast<const float4 *>(W + (size_t)(i0 + r) * N + j);
		acc.x += xi * ww.x; acc.y += xi * ww.y; acc.z += xi * ww.z; acc.w += xi * ww.w;
	}
	s_acc[w][lane] = acc;
	__syncthreads();
	if (w == 0)
	{
		for (int k = 1; k < 4; k++) { const float4 o = s_acc[k][lane]; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
		*reinterpret_cast<float4 *>(part + (size_t)ks * N + j) = acc;
	}
}
// Output layer: reduce FC2's partial sums, chunked softmax forward (cnn.h:497-511), E = y - t, softmax backward (cnn.h:512-526).
// Blocks 0..7 own the eight 256-wide chunks (a thread per element), block 8 the sixteen 16-wide ones; sqp[block] = its sum of E^2.
__global__ __launch_bounds__(256) void k_t_softmax_loss(const float *__restrict__ part, const float *__restrict__ B, const float *__restrict__ target, float *__restrict__ e9, float *__restrict__ sqp)
{
	__shared__ float red[3][4];
	const int t = threadIdx.x, w = t >> 6, lane = t & 63, i = blockIdx.x * 256 + t;
	const bool wide = blockIdx.x < 8;
	float acc = B[i];
#pragma unroll 8
	for (int k = 0; k < T_KSPLIT; k++) acc += part[(size_t)k * HT_CNN_OUT + i];
	const float v = (float)exp((double)acc);
	float cs = wide ? wave_sum(v) : seg16_sum(v);
	if (lane == 0) red[0][w] = cs;
	__syncthreads();
	if (wide) cs = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
	const float y = v / cs, d = y - target[i];
	float cd = wide ? wave_sum(d * y) : seg16_sum(d * y);
	const float sq = wave_sum(d * d);
	if (lane == 0) { red[1][w] = cd; red[2][w] = sq; }
	__syncthreads();
	if (wide) cd = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
	e9[i] = y * (d - cd);
	if (t == 0) sqp[blockIdx.x] = ((red[2][0] + red[2][1]) + red[2][2]) + red[2][3];
}
// One wave per row i of a fully connected layer: d = W[i,:].E with the old weights, then W[i,:] -= X[i]*E*alpha (cnn.h:430-445).
// OUTPUT (the layer under the softmax): X is the output of a tanh layer whose backward is folded in, D[i] = (1 - X[i]^2) d, and the
// first thread finishes the loss value CNN::Train returns.  The first blocks also step the bias.
template <bool OUTPUT> __global__ __launch_bounds__(256) void k_t_fc_back_update(float *__restrict__ W, float *__restrict__ B, const float *__restrict__ X, const float *__restrict__ E, float *__restrict__ D, int M, int N, float alpha,
                                                                                const float *__restrict__ sqp, float *__restrict__ mse_out)
{
	const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	{ const int j = blockIdx.x * 256 + threadIdx.x; if (j < N) B[j] -= E[j] * alpha; }
	if (OUTPUT && blockIdx.x == 0 && threadIdx.x == 0) { float m = 0.0f; for (int k = 0; k < 9; k++) m += sqp[k]; *mse_out = m / (float)HT_CNN_OUT; }
	if (i >= M) return;
	float *w = W + (size_t)i * N;
	const float xi = X[i];
	float acc = 0.0f;
#pragma unroll 3
	for (int j = 4 * lane; j < N; j += 256)
	{
		const float4 ww = *reinterpret_cast<const float4 *>(w + j), ee = *reinterpret_cast<const float4 *>(E + j);
		acc += ww.x * ee.x; acc += ww.y * ee.y; acc += ww.z * ee.z; acc += ww.w * ee.w;
		*reinterpret_cast<float4 *>(w + j) = make_float4(ww.x - xi * ee.x * alpha, ww.y - xi * ee.y * alpha, ww.z - xi * ee.z * alpha, ww.w - xi * ee.w * alpha);
	}
	acc = wave_sum(acc);
	if (lane == 0) D[i] = OUTPUT ? (1.0f - xi * xi) * acc : acc;
}
// third max-pool backward + conv2's tanh backward for pooled element p of channel oz: the first maximum of the 2x2 window takes the
// error (cnn.h:150-164); writes the window's four entries of the channel's 12x12 error map, times scale
__device__ __forceinline__ void pool3_back_tanh(const float *__restrict__ a5, const float *__restrict__ e6, int oz, int p, float scale, float *__restrict__ e4c)
{
	const int base = (2 * (p / 6)) * 12 + 2 * (p % 6);
	const float *a = a5 + oz * 144 + base;
	float m;
	const int k = first_max4(a[0], a[1], a[12], a[13], m);
	const float d = scale * ((1.0f - m * m) * e6[oz * 36 + p]);
	e4c[base] = k == 0 ? d : 0.0f; e4c[base + 1] = k == 1 ? d : 0.0f; e4c[base + 12] = k == 2 ? d : 0.0f; e4c[base + 13] = k == 3 ? d : 0.0f;
}
// LConv::backward of conv2 as a gather (cnn.h:236-250).  Block (iz, g): input channel iz, output channels [4g, 4g+4), whose error maps
// are rebuilt in LDS from the pooled error; a thread owns one input position and walks output channel, output y, output x ascending.
#define T_CB_GROUPS 16
__global__ __launch_bounds__(256) void k_t_conv2_back(const float *__restrict__ a5, const float *__restrict__ e6, const float *__restrict__ W, float *__restrict__ part3)
{
	__shared__ float s_e[4 * 144], s_w[4 * 16];
	const int iz = blockIdx.x, g = blockIdx.y, t = threadIdx.x;
	if (t < 144) pool3_back_tanh(a5, e6, 4 * g + t / 36, t % 36, 1.0f, s_e + (t / 36) * 144);
	if (t < 64) { const int oz = t >> 4, k = t & 15; s_w[t] = W[k + 16 * (iz + 16 * (4 * g + oz))]; }
	__syncthreads();
	if (t >= 225) return;
	const int x = t % 15, y = t / 15;
	const int oy0 = max(0, y - 3), oy1 = min(11, y), ox0 = max(0, x - 3), ox1 = min(11, x);
	float acc = 0.0f;
	for (int oz = 0; oz < 4; oz++) for (int oy = oy0; oy <= oy1; oy++) for (int ox = ox0; ox <= ox1; ox++)
		acc += s_w[oz * 16 + (x - ox) + 4 * (y - oy)] * s_e[oz * 144 + oy * 12 + ox];
	part3[g * 3600 + iz * 225 + t] = acc;
}
// LConv::update (cnn.h:252-279) of both convolutions in one launch.
// blocks 0..63: conv2, output channel = block, a thread per tap walks the 144 output positions in the reference's order, then the
//               MFMA-packed copy W2p[(ky*4+kx)*16 + ic][oc] k_conv2 reads is refreshed;
// blocks 64..79: conv1, output channel = block - 64.  The two max-pools pass conv1's error to one position per 4x4 window only, so the
//               block first finds those 225 (position, value after tanh') pairs, then eight lanes per tap share them.
__global__ __launch_bounds__(256) void k_t_conv_update(const float *__restrict__ x0, const float *__restrict__ a1, const float *__restrict__ a3, const float *__restrict__ a5, const float *__restrict__ e6, const float *__restrict__ part3,
                                                       float *__restrict__ W1, float *__restrict__ B1, float *__restrict__ W2, float *__restrict__ B2, float *__restrict__ W2p, float alpha)
{
	__shared__ float s_x[4096], s_e[256];
	__shared__ int s_i[256];
	const int t = threadIdx.x;
	if (blockIdx.x < 64)
	{
		const int oz = blockIdx.x;
		for (int i = t; i < 3600; i += 256) s_x[i] = a3[i];
		if (t < 36) pool3_back_tanh(a5, e6, oz, t, -alpha, s_e);
		__syncthreads();
		const int kx = t & 3, ky = (t >> 2) & 3, iz = t >> 4;
		float w = W2[oz * 256 + t];
		const float *xp = s_x + iz * 225 + ky * 15 + kx;
		for (int y = 0; y < 12; y++)
#pragma unroll
			for (int x = 0; x < 12; x++) w += xp[y * 15 + x] * s_e[y * 12 + x];
		W2[oz * 256 + t] = w;
		W2p[(size_t)((ky * 4 + kx) * 16 + iz) * 64 + oz] = w;
		if (t == 0) { float bb = B2[oz]; for (int p = 0; p < 144; p++) bb += s_e[p]; B2[oz] = bb; }
	}
	else
	{
		const int oz = blockIdx.x - 64;
		for (int i = t; i < 4096; i += 256) s_x[i] = x0[i];
		if (t < 225)
		{
			const int x = t % 15, y = t / 15, i = oz * 225 + t;
			const float *p = a1 + oz * 3600 + (4 * y) * 60 + 4 * x;
			float q[4], m;
#pragma unroll
			for (int k = 0; k < 4; k++) { const float *r = p + (k >> 1) * 120 + (k & 1) * 2; q[k] = max4(r[0], r[1], r[60], r[61]); }
			const int k2 = first_max4(q[0], q[1], q[2], q[3], m);
			const float *r = p + (k2 >> 1) * 120 + (k2 & 1) * 2;
			const int k1 = first_max4(r[0], r[1], r[60], r[61], m);
			float e3 = 0.0f;
#pragma unroll
			for (int g = 0; g < T_CB_GROUPS; g++) e3 += part3[g * 3600 + i];
			s_i[t] = (4 * y + 2 * (k2 >> 1) + (k1 >> 1)) * 64 + 4 * x + 2 * (k2 & 1) + (k1 & 1);
			s_e[t] = -alpha * ((1.0f - m * m) * e3);
		}
		__syncthreads();
		const int tap = t >> 3, sub = t & 7;      // taps 0..24, tap 25 = bias, 26..31 idle
		const int off = tap < 25 ? (tap / 5) * 64 + tap % 5 : 0;
		float acc = 0.0f;
		if (tap < 25) for (int p = sub; p < 225; p += 8) acc += s_x[s_i[p] + off] * s_e[p];
		else if (tap == 25) for (int p = sub; p < 225; p += 8) acc += s_e[p];
		for (int o = 4; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
		if (sub == 0 && tap < 25) W1[oz * 25 + tap] += acc;
		if (sub == 0 && tap == 25) B1[oz] += acc;
	}
}

// act: layer outputs kept for the backward pass; err: errors and conv2-backward partial sums; part: FC1's and FC2's partial sums
void ht_launch_train_step(float *w, float *W2p, const float *x, const float *target, float alpha, float *act, float *err, float *part, float *mse_out, hipStream_t s)
{
	float *a1 = act, *a3 = a1 + 57600, *a5 = a3 + 3600, *a6 = a5 + 9216, *a8 = a6 + 2304;
	float *e9 = err, *e7 = e9 + 2304, *e6 = e7 + 2048, *part3 = e6 + 2304, *sqp = part3 + T_CB_GROUPS * 3600;
	float *part1 = part, *part2 = part + (size_t)T_KSPLIT * 2048;
	float *W1 = w, *B1 = W1 + 400, *W2 = B1 + 16, *B2 = W2 + 16384, *W3 = B2 + 64, *B3 = W3 + (size_t)2304 * 2048, *W4 = B3 + 2048, *B4 = W4 + (size_t)2048 * 2304;
	// forward
	hipLaunchKernelGGL(k_t_conv1_tanh_pool, dim3(15, 16), dim3(256), 0, s, x, W1, B1, a1, a3);
	hipLaunchKernelGGL(k_t_conv2_tanh_pool, dim3(64), dim3(192), 0, s, a3, W2, B2, a5, a6);
	hipLaunchKernelGGL(k_t_fc_partial<false>, dim3(2048 / 256, T_KSPLIT), dim3(256), 0, s, a6, nullptr, nullptr, nullptr, W3, part1, 2304, 2048);
	hipLaunchKernelGGL(k_t_fc_partial<true>, dim3(2304 / 256, T_KSPLIT), dim3(256), 0, s, nullptr, part1, B3, a8, W4, part2, 2048, 2304);
	hipLaunchKernelGGL(k_t_softmax_loss, dim3(9), dim3(256), 0, s, part2, B4, target, e9, sqp);
	// backward with the old weights, each layer's update in the same pass
	hipLaunchKernelGGL(k_t_fc_back_update<true>, dim3(2048 / 4), dim3(256), 0, s, W4, B4, a8, e9, e7, 2048, 2304, alpha, sqp, mse_out);
	hipLaunchKernelGGL(k_t_fc_back_update<false>, dim3(2304 / 4), dim3(256), 0, s, W3, B3, a6, e7, e6, 2304, 2048, alpha, nullptr, nullptr);
	hipLaunchKernelGGL(k_t_conv2_back, dim3(16, T_CB_GROUPS), dim3(256), 0, s, a5, e6, W2, part3);
	hipLaunchKernelGGL(k_t_conv_update, dim3(80), dim3(256), 0, s, x, a1, a3, a5, e6, part3, W1, B1, W2, B2, W2p, alpha);
}
size_t ht_train_act_floats() { return 57600 + 3600 + 9216 + 2304 + 2048; }
size_t ht_train_err_floats() { return 2304 + 2048 + 2304 + T_CB_GROUPS * 3600 + 16; }
size_t ht_train_part_floats() { return (size_t)T_KSPLIT * (2048 + 2304); }
