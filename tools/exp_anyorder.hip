// exp_anyorder.hip -- experiment: do kernels launched with hipExtAnyOrderLaunch (AQL packets without the barrier bit) run beside the previous kernel of the SAME
// stream on gfx950, and does the next ordinary launch wait for all of them?  (hip_ext.h says the flag is "not supported on GFX9xx" for one of its entry points.)
// If so, the rows of a solver step (cloud rows, chamber rows, contacts) can run side by side without the cross-queue fork and join of two side streams.
//   hipcc --offload-arch=gfx950 -O2 tools/exp_anyorder.hip -o /tmp/exp_anyorder && /tmp/exp_anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>

__global__ void spin(long cycles, int *out, int tag, const int *in)
{
	const long t0 = wall_clock64();      // 100 MHz
	int seen = in ? *in : 0;
	while (wall_clock64() - t0 < cycles) {}
	if (threadIdx.x == 0 && blockIdx.x == 0) *out = tag + seen;
}

static float run(hipStream_t s, int mode, int *d, int reps)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipEventRecord(a, s);
	for (int r = 0; r < reps; r++)
	{
		const unsigned f = mode ? hipExtAnyOrderLaunch : 0;
		hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, 0, 10000L, d + 0, 1, (const int *)(d + 3));      // 100 us, ordered: waits for the previous "solve"
		hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, f, 10000L, d + 1, 2, (const int *)(d + 3));
		hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, f, 10000L, d + 2, 3, (const int *)(d + 3));
		hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, 0, 10000L, d + 3, 4, (const int *)(d + 1));      // the "solve": must see all three
	}
	hipEventRecord(b, s); hipEventSynchronize(b);
	float ms = 0; hipEventElapsedTime(&ms, a, b);
	return ms / reps;
}

int main()
{
	int *d; hipMalloc(&d, 16); hipMemset(d, 0, 16);
	hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	run(s, 0, d, 2);
	const float t0 = run(s, 0, d, 20);
	const float t1 = run(s, 1, d, 20);
	// two side streams, as the library does it today
	hipStream_t u, v; hipStreamCreateWithFlags(&u, hipStreamNonBlocking); hipStreamCreateWithFlags(&v, hipStreamNonBlocking);
	hipEvent_t ef, e1, e2, a, b; hipEventCreateWithFlags(&ef, hipEventDisableTiming); hipEventCreateWithFlags(&e1, hipEventDisableTiming); hipEventCreateWithFlags(&e2, hipEventDisableTiming);
	hipEventCreate(&a); hipEventCreate(&b);
	float t2 = 0;
	for (int pass = 0; pass < 2; pass++)
	{
		hipEventRecord(a, s);
		for (int r = 0; r < 20; r++)
		{
			hipEventRecord(ef, s); hipStreamWaitEvent(u, ef, 0); hipStreamWaitEvent(v, ef, 0);
			hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, 10000L, d + 0, 1, (const int *)(d + 3));
			hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, u, 10000L, d + 1, 2, (const int *)(d + 3));
			hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, v, 10000L, d + 2, 3, (const int *)(d + 3));
			hipEventRecord(e1, u); hipStreamWaitEvent(s, e1, 0); hipEventRecord(e2, v); hipStreamWaitEvent(s, e2, 0);
			hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, 10000L, d + 3, 4, (const int *)(d + 1));
		}
		hipEventRecord(b, s); hipEventSynchronize(b);
		hipEventElapsedTime(&t2, a, b); t2 /= 20;
	}
	int h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
	printf("per round of 4 x 100 us kernels: in order %.3f ms, any-order middle two %.3f ms, two side streams %.3f ms (ideal 0.200); last values %d %d %d %d\n", t0, t1, t2, h[0], h[1], h[2], h[3]);
	return 0;
}
