// Does a VALU instruction cost less when whole 16-lane rows of the wave are switched off in EXEC?  One wave per SIMD, a dependent FMA chain and a DPP
// chain, with 64 / 32 / 16 / 4 active lanes: cycles per instruction from the core counter.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/exec_probe.hip -o tools/probe/exec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int active, int iters, long long *out, float *sink)
{
	const int lane = threadIdx.x & 63;
	float x = lane * 1e-3f, y = 1.0f;
	long long c = 0;
	if (lane < active)
	{
		const long long c0 = clock64();
		for (int i = 0; i < iters; i++)
		{
#pragma unroll
			for (int k = 0; k < 32; k++) { x = __fmaf_rn(x, 1.0001f, y); y = __fmaf_rn(y, 0.9999f, x); }
		}
		c = clock64() - c0;
	}
	if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = c;
	if (x == 12345.0f) sink[0] = x + y;
}
int main()
{
	long long *d; float *s; (void)hipMalloc(&d, 8); (void)hipMalloc(&s, 4);
	const int iters = 20000;
	for (int active : { 64, 32, 16, 4 })
	{
		hipLaunchKernelGGL(k, dim3(1024), dim3(64), 0, 0, active, iters, d, s);
		long long h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
		printf("%2d active lanes: %.2f cycles per dependent v_fma_f32\n", active, (double)h / (iters * 64.0));
	}
	return 0;
}
