// fp32 MFMA issue probe: TFLOP/s of v_mfma_f32_32x32x2_f32 with one dependent accumulator chain per wave against two independent chains, at 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_probe.hip -o tools/probe/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS> __global__ void k(int iters, float *sink)
{
	float x = threadIdx.x * 1e-3f; f32x16 a0, a1; for (int i = 0; i < 16; i++) { a0[i] = x; a1[i] = -x; }
	for (int i = 0; i < iters; i++)
	{
#pragma unroll
		for (int k = 0; k < 8; k++)
		{
			a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 1.0f, a0, 0, 0, 0);
			if (CHAINS == 2) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 1.0f, a1, 0, 0, 0); else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 1.0f, a0, 0, 0, 0);
		}
	}
	if (a0[3] == 1.5f || a1[2] == 7.0f) sink[0] = a0[0] + a1[0];
}
int main()
{
	float *s; (void)hipMalloc(&s, 4);
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const int iters = 20000;
	for (int chains = 1; chains <= 2; chains++)
		for (int wps = 1; wps <= 4; wps *= 2)
		{
			float best = 1e9f;
			for (int rep = 0; rep < 3; rep++)
			{
				(void)hipEventRecord(e0, 0);
				if (chains == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * wps), 0, 0, iters, s); else hipLaunchKernelGGL(k<2>, dim3(256), dim3(256 * wps), 0, 0, iters, s);
				(void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
				float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
			}
			const double flop = 256.0 * 4 * wps * iters * 16.0 * 4096.0;
			printf("%d chain(s), %d wave(s) per SIMD: %.3f ms  %.1f TFLOP/s\n", chains, wps, best, flop / best * 1e-9);
		}
	return 0;
}
