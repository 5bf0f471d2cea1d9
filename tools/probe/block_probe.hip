// Micro-costs behind the blocked two-body phases of k_solve (round 5), on a lone wave per SIMD and with two:
//   (a) ds_add_f32 with k lanes adding to ONE LDS word: clocks per instruction, and the order in which the lanes' terms are summed
//       (compared with an ascending-lane, a descending-lane and a pairwise sum of the same terms)
//   (b) the resolve step of a 32-row block -- v_med3, exec shift, v_readlane, v_fmac -- clocks per row
//   (c) ds_bpermute_b32 round trip
//   hipcc --offload-arch=gfx950 -O2 tools/probe/block_probe.hip -o tools/probe/block_probe && ./tools/probe/block_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>

__global__ __launch_bounds__(64) void k_atomic(int k, int reps, const float *terms, float *sums, long long *cyc)
{
	__shared__ float acc[64];
	const int lane = threadIdx.x;
	acc[lane] = 0.0f;
	__syncthreads();
	const float t = terms[lane];
	float *dst = &acc[lane < k ? 0 : lane];      // lanes [0, k) meet on word 0, the others on words of their own
	__syncthreads();
	const long long c0 = clock64();
	for (int r = 0; r < reps; r++)
	{
		__hip_atomic_fetch_add(dst, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	}
	const long long c1 = clock64();
	__syncthreads();
	if (lane == 0 && blockIdx.x == 0) { sums[0] = acc[0]; cyc[0] = c1 - c0; }
}

// one resolve pass over a 32-row block, forward (lanes 0..31), couplings in registers g[0..31]; statement per 16 steps
#define RS_STEP(i, G) \
	"v_med3_f32 %0, %1, %2, %3\n\t" \
	"s_lshl_b32 exec_lo, exec_lo, 1\n\t" \
	"v_readlane_b32 %4, %0, " #i "\n\t" \
	"v_fmac_f32 %1, %4, " G "\n\t"
__global__ __launch_bounds__(64) void k_resolve(int reps, const float *in, float *out, long long *cyc)
{
	const int lane = threadIdx.x;
	float g[32];
	for (int i = 0; i < 32; i++) g[i] = in[(lane * 32 + i) & 1023] * 1e-3f;
	float x = in[lane], lo = -1.0f, hi = 1.0f, imp = 0.0f, total = 0.0f;
	const long long c0 = clock64();
	for (int r = 0; r < reps; r++)
	{
		int tmp;
		asm volatile("s_mov_b32 exec_hi, 0\n\t"
		             RS_STEP(0, "%5") RS_STEP(1, "%6") RS_STEP(2, "%7") RS_STEP(3, "%8") RS_STEP(4, "%9") RS_STEP(5, "%10") RS_STEP(6, "%11") RS_STEP(7, "%12")
		             RS_STEP(8, "%13") RS_STEP(9, "%14") RS_STEP(10, "%15") RS_STEP(11, "%16") RS_STEP(12, "%17") RS_STEP(13, "%18") RS_STEP(14, "%19") RS_STEP(15, "%20")
		             "s_mov_b64 exec, -1"
		             : "+v"(imp), "+v"(x), "+v"(lo), "+v"(hi), "=&s"(tmp)
		             : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]), "v"(g[8]), "v"(g[9]), "v"(g[10]), "v"(g[11]), "v"(g[12]), "v"(g[13]), "v"(g[14]), "v"(g[15]));
		asm volatile("s_mov_b32 exec_hi, 0\n\t"
		             "s_mov_b32 exec_lo, 0xffff0000\n\t"
		             RS_STEP(16, "%5") RS_STEP(17, "%6") RS_STEP(18, "%7") RS_STEP(19, "%8") RS_STEP(20, "%9") RS_STEP(21, "%10") RS_STEP(22, "%11") RS_STEP(23, "%12")
		             RS_STEP(24, "%13") RS_STEP(25, "%14") RS_STEP(26, "%15") RS_STEP(27, "%16") RS_STEP(28, "%17") RS_STEP(29, "%18") RS_STEP(30, "%19") RS_STEP(31, "%20")
		             "s_mov_b64 exec, -1"
		             : "+v"(imp), "+v"(x), "+v"(lo), "+v"(hi), "=&s"(tmp)
		             : "v"(g[16]), "v"(g[17]), "v"(g[18]), "v"(g[19]), "v"(g[20]), "v"(g[21]), "v"(g[22]), "v"(g[23]), "v"(g[24]), "v"(g[25]), "v"(g[26]), "v"(g[27]), "v"(g[28]), "v"(g[29]), "v"(g[30]), "v"(g[31]));
		total += imp;
		x = x * 0.5f + total * 1e-3f;
	}
	const long long c1 = clock64();
	out[blockIdx.x * 64 + lane] = total;
	if (lane == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}

__global__ __launch_bounds__(64) void k_bperm(int reps, float *out, long long *cyc)
{
	const int lane = threadIdx.x;
	int v = lane * 7 + 1;
	const long long c0 = clock64();
	for (int r = 0; r < reps; r++) v = __builtin_amdgcn_ds_bpermute(((lane * 5 + 3) & 63) << 2, v) + 1;
	const long long c1 = clock64();
	out[blockIdx.x * 64 + lane] = (float)v;
	if (lane == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}

int main()
{
	float *d_terms, *d_sums, *d_out; long long *d_cyc;
	hipMalloc(&d_terms, 64 * 4); hipMalloc(&d_sums, 64 * 4); hipMalloc(&d_cyc, 64); hipMalloc(&d_out, 8192 * 64 * 4);
	float h_in[1024];
	for (int i = 0; i < 1024; i++) h_in[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
	float *d_in; hipMalloc(&d_in, sizeof h_in); hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
	// (a) terms whose float sum depends on the order: alternating magnitudes
	float terms[64];
	for (int i = 0; i < 64; i++) terms[i] = (i % 3 == 0 ? 1.0f : 1e-7f) * (1.0f + 0.013f * i) * ((i % 5) == 2 ? -1.0f : 1.0f);
	hipMemcpy(d_terms, terms, sizeof terms, hipMemcpyHostToDevice);
	for (int k : { 1, 2, 4, 8, 16, 32, 64 })
	{
		hipLaunchKernelGGL(k_atomic, dim3(1), dim3(64), 0, 0, k, 1, d_terms, d_sums, d_cyc);
		float s; hipMemcpy(&s, d_sums, 4, hipMemcpyDeviceToHost);
		float asc = 0, desc = 0; for (int i = 0; i < k; i++) asc += terms[i]; for (int i = k - 1; i >= 0; i--) desc += terms[i];
		std::vector<float> t(terms, terms + k); while (t.size() > 1) { std::vector<float> u; for (size_t i = 0; i + 1 < t.size(); i += 2) u.push_back(t[i] + t[i + 1]); if (t.size() & 1) u.push_back(t.back()); t = u; }
		bool same = true;
		for (int rep = 0; rep < 20; rep++) { hipLaunchKernelGGL(k_atomic, dim3(1), dim3(64), 0, 0, k, 1, d_terms, d_sums, d_cyc); float s2; hipMemcpy(&s2, d_sums, 4, hipMemcpyDeviceToHost); same = same && memcmp(&s, &s2, 4) == 0; }
		hipLaunchKernelGGL(k_atomic, dim3(1), dim3(64), 0, 0, k, 1000, d_terms, d_sums, d_cyc);
		long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
		hipLaunchKernelGGL(k_atomic, dim3(1024), dim3(64), 0, 0, k, 1000, d_terms, d_sums, d_cyc);
		long long c4; hipMemcpy(&c4, d_cyc, 8, hipMemcpyDeviceToHost);
		printf("ds_add_f32, %2d lanes on one word: %.1f clocks per instruction alone, %.1f with 4 waves per CU; sum %.9g (ascending %.9g%s, descending %.9g%s, pairwise %.9g%s), repeatable: %s\n",
		       k, c / 1000.0, c4 / 1000.0, s, asc, s == asc ? " =" : "", desc, s == desc ? " =" : "", t[0], s == t[0] ? " =" : "", same ? "yes" : "NO");
	}
	for (int blocks : { 1, 1024, 2048 })
	{
		hipLaunchKernelGGL(k_resolve, dim3(blocks), dim3(64), 0, 0, 2000, d_in, d_out, d_cyc);
		long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
		printf("resolve of a 32-row block, %4d blocks: %.1f clocks per row\n", blocks, c / 2000.0 / 32.0);
	}
	for (int blocks : { 1, 1024 })
	{
		hipLaunchKernelGGL(k_bperm, dim3(blocks), dim3(64), 0, 0, 2000, d_out, d_cyc);
		long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
		printf("ds_bpermute_b32 dependent chain, %4d blocks: %.1f clocks per hop\n", blocks, c / 2000.0);
	}
	return 0;
}
