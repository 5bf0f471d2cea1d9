// conv1_probe.hip -- where k_conv1's time goes (52.7 us at 1024 frames, 21 us of it matrix instructions): the kernel of csrc/ht_cnn.hip with clock stamps at its phase
// boundaries, and variants that take one thing away at a time.  Random input and weights (timing only).
//   hipcc --offload-arch=gfx950 -O3 -I hand_tracking_samples_amd/csrc tools/probe/conv1_probe.hip -o build_alt/conv1_probe && build_alt/conv1_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "ht_math.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
__device__ __forceinline__ float max_pool(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float max_pool4(float a, float b, float c, float d) { float r; asm("v_max_f32 %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4" : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d)); return r; }

// VAR: 4 the wave's index and everything derived from it (window, tile address) in scalar registers; 0 the kernel as shipped; 1 no tanh (the pooled value is stored); 2 no matrix loop (pooled = the tile's first words); 3 two windows per trip (two independent accumulators)
template <int IW, int PW, int PR, int VAR>
__global__ __launch_bounds__(256) void k_conv1(const float *__restrict__ cnn_in, const float *__restrict__ W1, const float *__restrict__ B1, float *__restrict__ act1, long long *stamps)
{
	constexpr int TR = 4 * PR + 4, IWP = IW + 4;
	__shared__ __attribute__((aligned(16))) float tile[TR * IWP];
	__shared__ float pooled[16 * PR * PW];
	const long long t0 = wall_clock64();
	const int b = blockIdx.x, band = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const int row0 = 4 * PR * band;
	const int nrows = min(TR, IW - row0);
	const int prows = min(PR, PW - PR * band);
	const float4 *src = reinterpret_cast<const float4 *>(cnn_in + (size_t)b * IW * IW + (size_t)row0 * IW);
	for (int i = t; i < nrows * IW / 4; i += 256) { const int r = i / (IW / 4), c4 = i % (IW / 4); *reinterpret_cast<float4 *>(tile + r * IWP + 4 * c4) = src[i]; }
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
	const long long t1 = wall_clock64(); const long long c1 = clock64();
	if (VAR == 2)
	{
		for (int i = t; i < 16 * prows * PW; i += 256) pooled[i] = tile[i % (TR * IWP)];
	}
	else if (VAR == 3)
	{
		for (int w = wave; w < prows * PW; w += 8)
		{
			const int w2 = w + 4 < prows * PW ? w + 4 : w;
			const int ty = w / PW, tx = w % PW, ty2 = w2 / PW, tx2 = w2 % PW;
			const float *base = tile + 4 * ty * IWP + 4 * tx, *base2 = tile + 4 * ty2 * IWP + 4 * tx2;
			f32x4 acc = { bias, bias, bias, bias }, acc2 = acc;
#pragma unroll
			for (int s = 0; s < 7; s++) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(base[aoff[s]], wreg[s], acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(base2[aoff[s]], wreg[s], acc2, 0, 0, 0); }
			float m = fmax_std(fmax_std(fmax_std(acc[0], acc[1]), acc[2]), acc[3]), m2 = fmax_std(fmax_std(fmax_std(acc2[0], acc2[1]), acc2[2]), acc2[3]);
			{ const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false); m = fmax_std(__uint_as_float(r[0]), __uint_as_float(r[1])); }
			{ const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m2), __float_as_uint(m2), false, false); m2 = fmax_std(__uint_as_float(r[0]), __uint_as_float(r[1])); }
			{ const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false); m = fmax_std(__uint_as_float(r[0]), __uint_as_float(r[1])); }
			{ const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m2), __float_as_uint(m2), false, false); m2 = fmax_std(__uint_as_float(r[0]), __uint_as_float(r[1])); }
			if (lane < 16) { pooled[(n * PR + ty) * PW + tx] = m; pooled[(n * PR + ty2) * PW + tx2] = m2; }
		}
	}
	else if (VAR == 5 || VAR == 6)
	{
		// 5: scalar wave index and v_max_f32 maxima (the product after this round's change); 6: the same with the next window's seven reads issued ahead of the current window's matrix instructions
		const int swave = __builtin_amdgcn_readfirstlane(wave);
		float *const pdst = pooled + n * PR * PW;
		float an[7];
		if (VAR == 6) { const float *base = tile + 4 * (swave / PW) * IWP + 4 * (swave % PW);
#pragma unroll
			for (int s = 0; s < 7; s++) an[s] = base[aoff[s]]; }
		for (int w = swave; w < prows * PW; w += 4)
		{
			const int ty = w / PW, tx = w % PW;
			const float *base = tile + 4 * ty * IWP + 4 * tx;
			f32x4 acc = { bias, bias, bias, bias };
			float ac[7];
			if (VAR == 6)
			{
				const int wn = w + 4 < prows * PW ? w + 4 : w;
				const float *bn = tile + 4 * (wn / PW) * IWP + 4 * (wn % PW);
#pragma unroll
				for (int s = 0; s < 7; s++) { ac[s] = an[s]; an[s] = bn[aoff[s]]; }
				__builtin_amdgcn_sched_barrier(0);
			}
			else
			{
#pragma unroll
				for (int s = 0; s < 7; s++) ac[s] = base[aoff[s]];
			}
#pragma unroll
			for (int s = 0; s < 7; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[s], wreg[s], acc, 0, 0, 0);
			float m = max_pool4(acc[0], acc[1], acc[2], acc[3]);
			{ const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false); m = max_pool(__uint_as_float(r[0]), __uint_as_float(r[1])); }
			{ const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false); m = max_pool(__uint_as_float(r[0]), __uint_as_float(r[1])); }
			if (lane < 16) pdst[ty * PW + tx] = m;
		}
	}
	else if (VAR == 4)
	{
		const int swave = __builtin_amdgcn_readfirstlane(wave);
		float *const pdst = pooled + n * PR * PW;
		for (int w = swave; w < prows * PW; w += 4)
		{
			const int ty = w / PW, tx = w % PW;
			const float *base = tile + 4 * ty * IWP + 4 * tx;
			f32x4 acc = { bias, bias, bias, bias };
#pragma unroll
			for (int s = 0; s < 7; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(base[aoff[s]], wreg[s], acc, 0, 0, 0);
			float m = fmax_std(fmax_std(fmax_std(acc[0], acc[1]), acc[2]), acc[3]);
			{ const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false); m = fmax_std(__uint_as_float(r[0]), __uint_as_float(r[1])); }
			{ const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false); m = fmax_std(__uint_as_float(r[0]), __uint_as_float(r[1])); }
			if (lane < 16) pdst[ty * PW + tx] = m;
		}
	}
	else
	{
		for (int w = wave; w < prows * PW; w += 4)
		{
			const int ty = w / PW, tx = w % PW;
			const float *base = tile + 4 * ty * IWP + 4 * tx;
			f32x4 acc = { bias, bias, bias, bias };
#pragma unroll
			for (int s = 0; s < 7; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(base[aoff[s]], wreg[s], acc, 0, 0, 0);
			float m = fmax_std(fmax_std(fmax_std(acc[0], acc[1]), acc[2]), acc[3]);
			{ const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false); m = fmax_std(__uint_as_float(r[0]), __uint_as_float(r[1])); }
			{ const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false); m = fmax_std(__uint_as_float(r[0]), __uint_as_float(r[1])); }
			if (lane < 16) pooled[(n * PR + ty) * PW + tx] = m;
		}
	}
	__syncthreads();
	const long long t2 = wall_clock64(); const long long c2 = clock64();
	for (int i = t; i < 16 * prows * PW; i += 256)
	{
		const int c = i / (prows * PW), r = i % (prows * PW), ty = r / PW, tx = r % PW;
		const float v = pooled[(c * PR + ty) * PW + tx];
		act1[(size_t)b * (16 * PW * PW) + c * (PW * PW) + (PR * band + ty) * PW + tx] = VAR == 1 ? v : tanh_ref(v);
	}
	__syncthreads();
	if (t == 0) { long long *o = stamps + 4 * (size_t)blockIdx.x; o[0] = t0; o[1] = t1; o[2] = t2; o[3] = wall_clock64(); if (blockIdx.x == 7) o[0] = t0, stamps[4 * (size_t)gridDim.x] = c2 - c1, stamps[4 * (size_t)gridDim.x + 1] = t2 - t1; }
}

template <int VAR> static void run(const char *name, const float *in, const float *W, const float *Bv, float *out, long long *stamps, int B)
{
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	float best = 1e9f;
	for (int rep = 0; rep < 6; rep++)
	{
		(void)hipEventRecord(e0, 0);
		hipLaunchKernelGGL((k_conv1<64, 15, 15, VAR>), dim3(B, 1), dim3(256), 0, 0, in, W, Bv, out, stamps);
		(void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep > 0 && ms < best) best = ms;
	}
	std::vector<long long> h(4 * (size_t)B + 2);
	(void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
	const double mhz = h[4 * (size_t)B + 1] > 0 ? (double)h[4 * (size_t)B] / (h[4 * (size_t)B + 1] * 0.01) : 0.0;
	long long first = h[0], last = 0; double p1 = 0, p2 = 0, p3 = 0;
	for (int b = 0; b < B; b++) { first = std::min(first, h[4 * b]); last = std::max(last, h[4 * b + 3]); p1 += h[4 * b + 1] - h[4 * b]; p2 += h[4 * b + 2] - h[4 * b + 1]; p3 += h[4 * b + 3] - h[4 * b + 2]; }
	long long latest_start = 0; for (int b = 0; b < B; b++) latest_start = std::max(latest_start, h[4 * b] - first);
	// wall_clock64 ticks at 100 MHz
	printf("%-34s %7.1f us; first start to last end %6.1f us, last block starts at %5.1f us; mean per block: load %5.1f us, windows %5.1f us, tanh + store %5.1f us; shader clock in the windows phase %.0f MHz\n", name, best * 1e3, (last - first) * 0.01, latest_start * 0.01,
	       p1 / B * 0.01, p2 / B * 0.01, p3 / B * 0.01, mhz);
}

int main(int argc, char **argv)
{
	const int B = argc > 1 ? atoi(argv[1]) : 1024;
	std::vector<float> in((size_t)B * 4096), W(400), Bv(16);
	for (auto &v : in) v = (float)rand() / RAND_MAX;
	for (auto &v : W) v = 0.2f * ((float)rand() / RAND_MAX - 0.5f);
	for (auto &v : Bv) v = 0.1f * ((float)rand() / RAND_MAX - 0.5f);
	float *d_in, *d_W, *d_B, *d_out; long long *d_st;
	(void)hipMalloc(&d_in, in.size() * 4); (void)hipMalloc(&d_W, 1600); (void)hipMalloc(&d_B, 64); (void)hipMalloc(&d_out, (size_t)B * 3600 * 4); (void)hipMalloc(&d_st, (size_t)B * 32 + 64);
	(void)hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(d_W, W.data(), 1600, hipMemcpyHostToDevice); (void)hipMemcpy(d_B, Bv.data(), 64, hipMemcpyHostToDevice);
	run<0>("as shipped", d_in, d_W, d_B, d_out, d_st, B);
	run<1>("without tanh", d_in, d_W, d_B, d_out, d_st, B);
	run<2>("without the matrix loop", d_in, d_W, d_B, d_out, d_st, B);
	run<3>("two windows per trip", d_in, d_W, d_B, d_out, d_st, B);
	run<4>("wave index in a scalar register", d_in, d_W, d_B, d_out, d_st, B);
	run<5>("... and v_max_f32 maxima", d_in, d_W, d_B, d_out, d_st, B);
	run<6>("... and reads one window ahead", d_in, d_W, d_B, d_out, d_st, B);
	return 0;
}
