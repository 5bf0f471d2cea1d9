// Measures the shader clock a kernel actually runs at: s_memtime (core clock) against the 100 MHz constant counter, for a VALU loop and an fp32 MFMA loop.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/clock_probe.hip -o tools/probe/clock_probe && ./tools/probe/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k_probe(int mode, int iters, long long *out, float *sink)
{
	const long long c0 = clock64(), w0 = wall_clock64();
	float x = threadIdx.x * 1e-3f; f32x16 acc; for (int i = 0; i < 16; i++) acc[i] = x;
	for (int i = 0; i < iters; i++)
	{
		if (mode == 0) { for (int k = 0; k < 64; k++) x = __fmaf_rn(x, 1.0001f, 0.5f); }
		else { for (int k = 0; k < 16; k++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, 1.0f, acc, 0, 0, 0); }
	}
	const long long c1 = clock64(), w1 = wall_clock64();
	if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
	if (x == 12345.0f || acc[3] == 1.5f) sink[0] = x + acc[0];
}
int main()
{
	long long *d; float *s; hipMalloc(&d, 16); hipMalloc(&s, 4);
	for (int mode = 0; mode < 2; mode++)
		for (int rep = 0; rep < 3; rep++)
		{
			hipLaunchKernelGGL(k_probe, dim3(256 * 8), dim3(256), 0, 0, mode, mode ? 4000 : 20000, d, s);
			long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
			printf("%s loop on 2048 blocks: %lld core cycles in %lld ticks of 100 MHz -> %.0f MHz\n", mode ? "fp32 MFMA" : "VALU fma", h[0], h[1], (double)h[0] / ((double)h[1] / 100.0));
		}
	return 0;
}
