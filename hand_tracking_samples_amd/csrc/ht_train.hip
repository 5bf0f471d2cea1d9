// ht_train.hip -- one SGD step of the pose-initialiser CNN per sample.
//
// Reference computations (third_party/cnn.h): CNN::Train :558-580 (forward keeping every layer's output, E = y - t, backward() of
// layers 10..1 with the weights as they are, then update() of every layer), LConv :205-279, LMaxPool :141-164, LFull :405-445,
// LActivation<TanH> :457-469 (df = 1 - y*y), LSoftMaxChunked :497-526.  Topology: include/handtrack.h:108-118.
//
// Batch-1 SGD is sequential over samples, so the parallelism is inside one sample and the two fully connected layers (2 x 18.9 MB
// of weights) make the step HBM-bound: the forward product streams each matrix once (split over K, partial sums reduced in k order),
// and the backward product and the weight update of a layer share one pass over its matrix (row i yields d[i] = W[i,:].E with the old
// weights and is then rewritten as W[i,:] - x[i]*E*alpha), i.e. about 113 MB of traffic per sample.
// Where it is cheap the reference's accumulation order is kept (convolutions, pooling, activations, every weight update, which is
// element-wise); the long dot products of the fully connected layers are reduced in parallel, so results agree with the reference
// to float rounding rather than bit for bit.
#include "ht_device.hpp"
#include "ht_launch.hpp"

__device__ __forceinline__ float t_tanh(float t) { float e = (float)exp((double)(2 * t)); return (e - 1) / (e + 1); }      // TanH::f cnn.h:31

// valid convolution, one thread per output; taps in the reference's order (kx fastest, then ky, input channel innermost of a tap: cnn.h:226-228)
__global__ void k_t_conv(const float *__restrict__ in, const float *__restrict__ W, const float *__restrict__ B, float *__restrict__ out, int iw, int ih, int ic, int kw, int kh, int oc)
{
	const int ow = iw - kw + 1, oh = ih - kh + 1, n = ow * oh * oc;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int x = i % ow, y = (i / ow) % oh, oz = i / (ow * oh);
	float acc = B[oz];
	for (int ky = 0; ky < kh; ky++) for (int kx = 0; kx < kw; kx++) for (int iz = 0; iz < ic; iz++)
		acc += in[iz * iw * ih + (y + ky) * iw + x + kx] * W[kx + kw * (ky + kh * (iz + ic * oz))];
	out[i] = acc;
}
__global__ void k_t_tanh(const float *__restrict__ x, float *__restrict__ y, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) y[i] = t_tanh(x[i]); }
__global__ void k_t_pool(const float *__restrict__ in, float *__restrict__ out, int w, int h, int c)
{
	const int ow = w / 2, oh = h / 2, i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= ow * oh * c) return;
	const int x = i % ow, y = (i / ow) % oh, z = i / (ow * oh);
	const float *p = in + z * w * h + (2 * y) * w + 2 * x;
	out[i] = fmax_std(fmax_std(fmax_std(p[0], p[1]), p[w]), p[w + 1]);
}
// y[j] = B[j] + sum_i x[i] W[i][j]: block (jb, ks) sums rows [ks*rows, (ks+1)*rows) for 256 consecutive outputs
#define T_KSPLIT 16
__global__ __launch_bounds__(256) void k_t_fc_partial(const float *__restrict__ x, const float *__restrict__ W, float *__restrict__ part, int M, int N)
{
	const int j = blockIdx.x * 256 + threadIdx.x, ks = blockIdx.y, rows = (M + T_KSPLIT - 1) / T_KSPLIT;
	if (j >= N) return;
	const int i0 = ks * rows, i1 = min(M, i0 + rows);
	float acc = 0.0f;
	for (int i = i0; i < i1; i++) acc += x[i] * W[(size_t)i * N + j];
	part[(size_t)ks * N + j] = acc;
}
__global__ void k_t_fc_reduce(const float *__restrict__ part, const float *__restrict__ B, float *__restrict__ y, int N)
{
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= N) return;
	float acc = B[j];
	for (int k = 0; k < T_KSPLIT; k++) acc += part[(size_t)k * N + j];
	y[j] = acc;
}
// chunked softmax forward (cnn.h:497-511), E = y - t with the squared error, and the softmax backward (cnn.h:512-526); one block
__global__ __launch_bounds__(256) void k_t_softmax_loss(const float *__restrict__ logits, const float *__restrict__ target, float *__restrict__ y, float *__restrict__ e10, float *__restrict__ e9, float *__restrict__ mse_out)
{
	__shared__ float v[HT_CNN_OUT], e[HT_CNN_OUT], cs[24], cd[24];
	const int t = threadIdx.x;
	for (int i = t; i < HT_CNN_OUT; i += 256) v[i] = (float)exp((double)logits[i]);
	__syncthreads();
	if (t < 24) { const int s = t < 8 ? 256 : 16, base = t < 8 ? 256 * t : 2048 + 16 * (t - 8); float sum = 0.0f; for (int i = base; i < base + s; i++) sum += v[i]; cs[t] = sum; }
	__syncthreads();
	for (int i = t; i < HT_CNN_OUT; i += 256) { const float yy = v[i] / cs[i < 2048 ? (i >> 8) : 8 + ((i - 2048) >> 4)]; v[i] = yy; y[i] = yy; const float d = yy - target[i]; e[i] = d; e10[i] = d; }
	__syncthreads();
	if (t < 24) { const int s = t < 8 ? 256 : 16, base = t < 8 ? 256 * t : 2048 + 16 * (t - 8); float dp = 0.0f; for (int i = base; i < base + s; i++) dp += e[i] * v[i]; cd[t] = dp; }
	if (t == 32) { float m = 0.0f; for (int i = 0; i < HT_CNN_OUT; i++) m += e[i] * e[i]; *mse_out = m / (float)HT_CNN_OUT; }      // in index order like the reference's transform
	__syncthreads();
	for (int i = t; i < HT_CNN_OUT; i += 256) e9[i] = v[i] * (e[i] - cd[i < 2048 ? (i >> 8) : 8 + ((i - 2048) >> 4)]);
}
// one wave per row i of a fully connected layer: D[i] = W[i,:].E (old weights), then W[i,:] -= X[i]*E*alpha (cnn.h:430-445)
__global__ __launch_bounds__(256) void k_t_fc_back_update(float *__restrict__ W, const float *__restrict__ X, const float *__restrict__ E, float *__restrict__ D, int M, int N, float alpha)
{
	const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (i >= M) return;
	float *w = W + (size_t)i * N;
	const float xi = X[i];
	float acc = 0.0f;
	for (int j = 4 * lane; j < N; j += 256)
	{
		const float4 ww = *reinterpret_cast<const float4 *>(w + j), ee = *reinterpret_cast<const float4 *>(E + j);
		acc += ww.x * ee.x; acc += ww.y * ee.y; acc += ww.z * ee.z; acc += ww.w * ee.w;
		*reinterpret_cast<float4 *>(w + j) = make_float4(ww.x - xi * ee.x * alpha, ww.y - xi * ee.y * alpha, ww.z - xi * ee.z * alpha, ww.w - xi * ee.w * alpha);
	}
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
	if (D && lane == 0) D[i] = acc;
}
__global__ void k_t_bias_update(float *__restrict__ B, const float *__restrict__ E, int n, float alpha) { const int j = blockIdx.x * blockDim.x + threadIdx.x; if (j < n) B[j] -= E[j] * alpha; }
__global__ void k_t_tanh_back(const float *__restrict__ Y, const float *__restrict__ E, float *__restrict__ D, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) D[i] = (1.0f - Y[i] * Y[i]) * E[i]; }
// LMaxPool::backward: the first maximum of the window (x then y) takes the error, the other three entries are 0
__global__ void k_t_pool_back(const float *__restrict__ X, const float *__restrict__ E, float *__restrict__ D, int w, int h, int c)
{
	const int ow = w / 2, oh = h / 2, i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= ow * oh * c) return;
	const int x = i % ow, y = (i / ow) % oh, z = i / (ow * oh);
	const int base = z * w * h + (2 * y) * w + 2 * x;
	const int q[4] = { base, base + 1, base + w, base + w + 1 };
	int m = 0;
	for (int k = 1; k < 4; k++) if (X[q[k]] > X[q[m]]) m = k;
	for (int k = 0; k < 4; k++) D[q[k]] = k == m ? E[i] : 0.0f;
}
// LConv::backward as a gather, contributions in the reference's order (output channel, then output y, then output x ascending)
__global__ void k_t_conv_back(const float *__restrict__ E, const float *__restrict__ W, float *__restrict__ D, int iw, int ih, int ic, int kw, int kh, int oc)
{
	const int ow = iw - kw + 1, oh = ih - kh + 1, i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= iw * ih * ic) return;
	const int x = i % iw, y = (i / iw) % ih, iz = i / (iw * ih);
	float acc = 0.0f;
	for (int oz = 0; oz < oc; oz++)
		for (int oy = max(0, y - kh + 1); oy <= min(oh - 1, y); oy++) for (int ox = max(0, x - kw + 1); ox <= min(ow - 1, x); ox++)
			acc += W[(x - ox) + kw * ((y - oy) + kh * (iz + ic * oz))] * E[oz * ow * oh + oy * ow + ox];
	D[i] = acc;
}
// LConv::update: one thread per weight walks the output positions in order (x fastest); one thread per output channel does the bias
__global__ void k_t_conv_update(const float *__restrict__ X, const float *__restrict__ E, float *__restrict__ W, float *__restrict__ B, int iw, int ih, int ic, int kw, int kh, int oc, float alpha)
{
	const int ow = iw - kw + 1, oh = ih - kh + 1, nw = kw * kh * ic * oc, i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nw)
	{
		const int kx = i % kw, ky = (i / kw) % kh, iz = (i / (kw * kh)) % ic, oz = i / (kw * kh * ic);
		float w = W[i];
		for (int y = 0; y < oh; y++) for (int x = 0; x < ow; x++) w += X[iz * iw * ih + (y + ky) * iw + x + kx] * (-alpha * E[oz * ow * oh + y * ow + x]);
		W[i] = w;
	}
	else if (i < nw + oc)
	{
		const int oz = i - nw;
		float bb = B[oz];
		for (int p = 0; p < ow * oh; p++) bb -= E[oz * ow * oh + p] * alpha;
		B[oz] = bb;
	}
}
// conv2 weights repacked to [k][oc], k = (ky*4+kx)*16 + ic, for k_conv2 (same layout ht_cnn_load_weights builds on the host)
__global__ void k_t_repack_conv2(const float *__restrict__ W2, float *__restrict__ W2p)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= 16384) return;
	const int kx = i % 4, ky = (i / 4) % 4, ic = (i / 16) % 16, oc = i / 256;
	W2p[(size_t)((ky * 4 + kx) * 16 + ic) * 64 + oc] = W2[i];
}

#define T_GRID(n) dim3(((n) + 255) / 256), dim3(256)
// buffers: act = 11 layer outputs, err = 11 layer errors (sizes of CNN layer outputs), part = T_KSPLIT x 2304 partial sums
void ht_launch_train_step(float *w, float *W2p, const float *x, const float *target, float alpha, float *act, float *err, float *part, float *mse_out, hipStream_t s)
{
	static const int NL[11] = { 57600, 57600, 14400, 3600, 9216, 9216, 2304, 2048, 2048, 2304, 2304 };
	float *o[11], *e[11]; { size_t off = 0; for (int i = 0; i < 11; i++) { o[i] = act + off; e[i] = err + off; off += NL[i]; } }
	float *W1 = w, *B1 = W1 + 400, *W2 = B1 + 16, *B2 = W2 + 16384, *W3 = B2 + 64, *B3 = W3 + (size_t)2304 * 2048, *W4 = B3 + 2048, *B4 = W4 + (size_t)2048 * 2304;
	// forward
	hipLaunchKernelGGL(k_t_conv, T_GRID(57600), 0, s, x, W1, B1, o[0], 64, 64, 1, 5, 5, 16);
	hipLaunchKernelGGL(k_t_tanh, T_GRID(57600), 0, s, o[0], o[1], 57600);
	hipLaunchKernelGGL(k_t_pool, T_GRID(14400), 0, s, o[1], o[2], 60, 60, 16);
	hipLaunchKernelGGL(k_t_pool, T_GRID(3600), 0, s, o[2], o[3], 30, 30, 16);
	hipLaunchKernelGGL(k_t_conv, T_GRID(9216), 0, s, o[3], W2, B2, o[4], 15, 15, 16, 4, 4, 64);
	hipLaunchKernelGGL(k_t_tanh, T_GRID(9216), 0, s, o[4], o[5], 9216);
	hipLaunchKernelGGL(k_t_pool, T_GRID(2304), 0, s, o[5], o[6], 12, 12, 64);
	hipLaunchKernelGGL(k_t_fc_partial, dim3(2048 / 256, T_KSPLIT), dim3(256), 0, s, o[6], W3, part, 2304, 2048);
	hipLaunchKernelGGL(k_t_fc_reduce, T_GRID(2048), 0, s, part, B3, o[7], 2048);
	hipLaunchKernelGGL(k_t_tanh, T_GRID(2048), 0, s, o[7], o[8], 2048);
	hipLaunchKernelGGL(k_t_fc_partial, dim3(2304 / 256, T_KSPLIT), dim3(256), 0, s, o[8], W4, part, 2048, 2304);
	hipLaunchKernelGGL(k_t_fc_reduce, T_GRID(2304), 0, s, part, B4, o[9], 2304);
	hipLaunchKernelGGL(k_t_softmax_loss, dim3(1), dim3(256), 0, s, o[9], target, o[10], e[10], e[9], mse_out);
	// backward (old weights) fused with the updates of the two big layers; the remaining updates follow once no backward needs their weights
	hipLaunchKernelGGL(k_t_fc_back_update, dim3(2048 / 4), dim3(256), 0, s, W4, o[8], e[9], e[8], 2048, 2304, alpha);
	hipLaunchKernelGGL(k_t_bias_update, T_GRID(2304), 0, s, B4, e[9], 2304, alpha);
	hipLaunchKernelGGL(k_t_tanh_back, T_GRID(2048), 0, s, o[8], e[8], e[7], 2048);
	hipLaunchKernelGGL(k_t_fc_back_update, dim3(2304 / 4), dim3(256), 0, s, W3, o[6], e[7], e[6], 2304, 2048, alpha);
	hipLaunchKernelGGL(k_t_bias_update, T_GRID(2048), 0, s, B3, e[7], 2048, alpha);
	hipLaunchKernelGGL(k_t_pool_back, T_GRID(2304), 0, s, o[5], e[6], e[5], 12, 12, 64);
	hipLaunchKernelGGL(k_t_tanh_back, T_GRID(9216), 0, s, o[5], e[5], e[4], 9216);
	hipLaunchKernelGGL(k_t_conv_back, T_GRID(3600), 0, s, e[4], W2, e[3], 15, 15, 16, 4, 4, 64);
	hipLaunchKernelGGL(k_t_pool_back, T_GRID(3600), 0, s, o[2], e[3], e[2], 30, 30, 16);
	hipLaunchKernelGGL(k_t_pool_back, T_GRID(14400), 0, s, o[1], e[2], e[1], 60, 60, 16);
	hipLaunchKernelGGL(k_t_tanh_back, T_GRID(57600), 0, s, o[1], e[1], e[0], 57600);
	hipLaunchKernelGGL(k_t_conv_update, T_GRID(400 + 16), 0, s, x, e[0], W1, B1, 64, 64, 1, 5, 5, 16, alpha);
	hipLaunchKernelGGL(k_t_conv_update, T_GRID(16384 + 64), 0, s, o[3], e[4], W2, B2, 15, 15, 16, 4, 4, 64, alpha);
	hipLaunchKernelGGL(k_t_repack_conv2, T_GRID(16384), 0, s, W2, W2p);
}
size_t ht_train_act_floats() { return 57600 + 57600 + 14400 + 3600 + 9216 + 9216 + 2304 + 2048 + 2048 + 2304 + 2304; }
size_t ht_train_part_floats() { return (size_t)T_KSPLIT * 2304; }
