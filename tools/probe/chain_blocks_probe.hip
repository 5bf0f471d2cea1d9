// chain_blocks_probe.hip -- experiment of round 4; ADOPTED in round 5 (csrc/ht_quad.hpp: quad_blocks_run) once the real frames' shape was put beside it -- a depth frame puts a
// third to a half of its single-body rows on ONE body, where the synthetic frame below has a quarter, which is the break-even: what the single-body rows of a solve cost row by row on sixteen quads (quad_chain_run, the
// product's) and four at a time on four DPP rows (quad_blocks_run, below),
// on synthetic chains of a main pass's shape: 17 bodies, 1010 rows, the longest chain 250; one wave per frame, every frame its own 64 KB of records, 20 sweeps, 36 KB of
// LDS per wave (four waves per CU) as in k_solve's build for 1024 frames.
//   hipcc --offload-arch=gfx950 -O3 -I hand_tracking_samples_amd/csrc tools/probe/chain_blocks_probe.hip -o build_alt/chain_blocks_probe && build_alt/chain_blocks_probe [frames]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "ht_quad.hpp"

// ---- single-body rows four at a time (round 4) -----------------------------------------------------------------------------------
// Consecutive rows of one body depend on each other only through the momenta, and linearly: with M the momenta before row 0 of a block of four rows,
//   vn_j / effmass_j = c_j . (M + sum_{i<j} d_i imp_i) = c_j . M + sum_{i<j} G(j,i) imp_i,      G(j,i) = c_j . d_i   (c_j = slots x, z of row j, d_i = slots w, y of row i)
// and G does not change during a PhysicsUpdate any more than the records do.  So the four quads of a DPP row (16 lanes) take the four rows of a block TOGETHER: every quad
// holds the body's momenta (the same values), forms its row's c_j . M side by side with the others (the expensive part: two products, a three-lane sum), then the impulses
// are resolved in row order -- imp_0 = clamp(x_0); x_j += -G(j,0) imp_0; imp_1 = clamp(x_1); ... : two dependent instructions per row -- and all quads add all four
// d_i imp_i to their momenta.  Same rows, same order, same clamps as LimitLinear::Iter (physics.h:289-307); one more association order of the same sums (G imp in place of
// c . (d imp)).  A block costs ~30 instructions where four single rows cost ~80, and the wave's four DPP rows walk FOUR bodies' chains at a time with the bodies dealt out
// so that the rows carry about equal numbers of blocks (k_solve's prologue), where sixteen quads on sixteen bodies waited for the longest chain.
// G travels in an array of its own in chain order, 16 bytes per row (-G(j,0), -G(j,1), -G(j,2), 0; zero where i >= j), written once per solve by the prologue.
//
// One block: a = this lane's slot of its quad's row, g = the row's couplings, sum = the row's impulse sum.  Returns the new impulse sum (all four lanes of the quad).
// Lane 3's column runs through the same instructions with the scalars of its slot: what it computes in p, t, s, ul, ua is never used.
template <bool POST>
__device__ __forceinline__ float quad_block_step(quad_body &B, const float4 a, const float4 g, const float sum)
{
	float ns, p, t, s, x, lo, hi, imp, ul, ua;
	asm volatile("v_mul_f32 %[p], %[ax], %[l]\n\t"
	             "v_fmac_f32 %[p], %[az], %[av]\n\t"                                                                        // lanes 0-2: (n[c]*massinv*P[c] + b[c]*L[c]) / effective mass
	             "v_sub_f32 %[lo], %[az], %[sum]\n\t"                                                                       // lane 3: fmin*dt - sum
	             "v_sub_f32 %[hi], %[aw], %[sum]\n\t"                                                                       //         fmax*dt - sum
	             "v_add_f32_dpp %[t], %[p], %[p] quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"           // lane 1: p0 + p1
	             "s_nop 1\n\t"
	             "v_add_f32_dpp %[s], %[t], %[p] quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"           // lane 2: (p0 + p1) + p2
	             "s_nop 1\n\t"
	             "v_subrev_f32_dpp %[x], %[s], -%[ts] quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      // lane 3: (-targetspeed - c.M) / effective mass
	             "v_med3_f32 %[imp], %[x], %[lo], %[hi]\n\t"                                                                // row 0's impulse is final
	             "s_nop 1\n\t"
	             "v_fmac_f32_dpp %[x], %[imp], %[g0] row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            // rows 1-3: x += -G(j,0) * imp_0
	             "v_med3_f32 %[imp], %[x], %[lo], %[hi]\n\t"                                                                // row 1's
	             "s_nop 1\n\t"
	             "v_fmac_f32_dpp %[x], %[imp], %[g1] row_newbcast:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
	             "v_med3_f32 %[imp], %[x], %[lo], %[hi]\n\t"                                                                // row 2's
	             "s_nop 1\n\t"
	             "v_fmac_f32_dpp %[x], %[imp], %[g2] row_newbcast:11 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
	             "v_med3_f32 %[imp], %[x], %[lo], %[hi]\n\t"                                                                // row 3's
	             "s_nop 1\n\t"
	             "v_mul_f32_dpp %[ul], %[imp], %[aw] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       // n[c] * imp_j
	             "v_mul_f32_dpp %[ua], %[imp], %[ay] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       // g[c] * imp_j
	             "v_add_f32_dpp %[ns], %[imp], %[sum] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      // the row's new impulse sum
	             // the four rows' contributions, summed the same way in every quad: (u_j + u_(j+2)) + (u_(j+1) + u_(j+3)) -- additions commute, so the four copies of the
	             // momenta stay equal bit for bit
	             "v_add_f32_dpp %[ul], %[ul], %[ul] row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
	             "v_add_f32_dpp %[ua], %[ua], %[ua] row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
	             "s_nop 0\n\t"
	             "v_add_f32_dpp %[ul], %[ul], %[ul] row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
	             "v_add_f32_dpp %[ua], %[ua], %[ua] row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
	             "v_add_f32 %[l], %[l], %[ul]\n\t"
	             "v_add_f32 %[av], %[av], %[ua]"
	             : [l] "+v"(B.l), [av] "+v"(B.av), [ns] "=&v"(ns), [p] "=&v"(p), [t] "=&v"(t), [s] "=&v"(s), [x] "=&v"(x), [lo] "=&v"(lo), [hi] "=&v"(hi), [imp] "=&v"(imp),
	               [ul] "=&v"(ul), [ua] "=&v"(ua)
	             : [ax] "v"(a.x), [ay] "v"(a.y), [az] "v"(a.z), [aw] "v"(a.w), [ts] "v"(POST ? a.y : a.x), [g0] "v"(g.x), [g1] "v"(g.y), [g2] "v"(g.z), [sum] "v"(sum));
	return ns;
}
// A DPP row's walk: blocks [0, nblk) of its segment of the chain lists (entries e0 .. e0 + 4*nblk: row j of block t is entry e0 + 4t + j), the bodies of the segment one
// after the other.  idx / G / sums are indexed by entry, recs by record index; c = lane within the quad, j = quad within the DPP row.  Which body a block belongs to: the
// segment starts with body `head`, a body b has cblk[b] blocks and is followed by body cnext[b]; the momenta of the body in hand live in registers (all four quads hold
// the same values) and go back to lin_w / ang_w when the row moves on.  Eight register sets rotate: a block's record is asked for eight blocks ahead of its use (its
// index another eight blocks earlier), its couplings eight blocks ahead, its impulse sum two.  Reads run up to 16 blocks of indices, 8 of records (of valid indices) and
// couplings and 2 of sums past the segment's end: into the next row's segment, or the slack behind the last (QUAD_BLOCK_SLACK entries, indices naming the no-op record).
#define QUAD_BLOCK_SLACK 64
template <bool POST, class IDX>
__device__ __forceinline__ void quad_blocks_run_(const float *recs, const IDX *idx, const float4 *G, float *sums, int e0, int nblk, int c, int j, float *lin_w, float *ang_w, int head,
                                                 const unsigned char *cnext, const unsigned short *cblk)
{
	const float4 *pa = reinterpret_cast<const float4 *>(recs) + c;
	const IDX *px = idx + e0 + j;
	const float4 *pg = G + e0 + j;
	float *ps = sums + e0 + j;
	quad_body B = { 0.0f, 0.0f };
	int cur = -1, nxt = head, left = 0;
#define QB_LX(i, blk) x##i = (unsigned)px[4 * (blk)]; __builtin_amdgcn_sched_barrier(0)
#define QB_LA(i) a##i = pa[4 * x##i]; __builtin_amdgcn_sched_barrier(0)
#define QB_LG(i, blk) g##i = pg[4 * (blk)]; __builtin_amdgcn_sched_barrier(0)
#define QB_LS(i, blk) s##i = ps[4 * (blk)]; __builtin_amdgcn_sched_barrier(0)
	float4 a0, a1, a2, a3, a4, a5, a6, a7, g0, g1, g2, g3, g4, g5, g6, g7; float s0, s1, s2, s3, s4, s5, s6, s7;
	unsigned x0, x1, x2, x3, x4, x5, x6, x7;
	QB_LX(0, 0); QB_LX(1, 1); QB_LX(2, 2); QB_LX(3, 3); QB_LX(4, 4); QB_LX(5, 5); QB_LX(6, 6); QB_LX(7, 7);
	QB_LA(0); QB_LA(1); QB_LA(2); QB_LA(3); QB_LA(4); QB_LA(5); QB_LA(6); QB_LA(7);
	QB_LG(0, 0); QB_LG(1, 1); QB_LG(2, 2); QB_LG(3, 3); QB_LG(4, 4); QB_LG(5, 5); QB_LG(6, 6); QB_LG(7, 7);
	QB_LX(0, 8); QB_LX(1, 9); QB_LX(2, 10); QB_LX(3, 11); QB_LX(4, 12); QB_LX(5, 13); QB_LX(6, 14); QB_LX(7, 15);
	QB_LS(0, 0); QB_LS(1, 1);
	// the row moves on to its next body: the momenta in hand go back, the next body's come in (and are waited for inside the branch: the waits the compiler
	// derives behind the branch are then those of the blocks that do not change body)
#define QB_BODY() \
	if (left == 0) \
	{ \
		if (c < 3 && cur >= 0) { lin_w[4 * cur + c] = B.l; ang_w[4 * cur + c] = B.av; } \
		cur = nxt; B.l = lin_w[4 * cur + c]; B.av = ang_w[4 * cur + c]; left = cblk[cur]; nxt = cnext[cur]; \
		__builtin_amdgcn_s_waitcnt(0xC07F); \
	} \
	left--; __builtin_amdgcn_sched_barrier(0)
#define QB_STEP(i, blk) ps[4 * (blk)] = quad_block_step<POST>(B, a##i, g##i, s##i); __builtin_amdgcn_sched_barrier(0)
	// Block i of the trip.  Ahead of its arithmetic (so that the reads have the block's ~200 clocks to come back before the next block waits on the LDS queue): the record
	// and the couplings eight blocks on for the register set the PREVIOUS block has just freed (its index came in during the last trip), the index sixteen blocks on for
	// that set, and the impulse sum two blocks on (set k).  The last block of a trip asks for its own set's refill behind its arithmetic.
#define QB_BLK(i, p, k) QB_BODY(); QB_LA(p); QB_LG(p, 8 + p); QB_LX(p, 16 + p); QB_LS(k, 2 + i); QB_STEP(i, i)
	int t = 0;
	for (; t + 8 <= nblk; t += 8)
	{
		QB_BODY(); QB_LS(2, 2); QB_STEP(0, 0);
		QB_BLK(1, 0, 3); QB_BLK(2, 1, 4); QB_BLK(3, 2, 5); QB_BLK(4, 3, 6); QB_BLK(5, 4, 7); QB_BLK(6, 5, 0); QB_BLK(7, 6, 1);
		QB_LA(7); QB_LG(7, 15); QB_LX(7, 23);
		px += 32; pg += 32; ps += 32;
	}
	const int rest = nblk - t;      // 0..7 blocks: their records and couplings are in the register sets, the sums of the first two too
#define QB_TAIL(i, k) if (rest > i) { QB_BODY(); QB_LS(k, 2 + i); QB_STEP(i, i); } __builtin_amdgcn_sched_barrier(0)
	QB_TAIL(0, 2); QB_TAIL(1, 3); QB_TAIL(2, 4); QB_TAIL(3, 5); QB_TAIL(4, 6); QB_TAIL(5, 7); QB_TAIL(6, 0);
	if (c < 3 && cur >= 0) { lin_w[4 * cur + c] = B.l; ang_w[4 * cur + c] = B.av; }
#undef QB_LX
#undef QB_LA
#undef QB_LG
#undef QB_LS
#undef QB_BODY
#undef QB_STEP
#undef QB_BLK
#undef QB_TAIL
}
template <class IDX>
__device__ __forceinline__ void quad_blocks_run(const float *recs, const IDX *idx, const float4 *G, float *sums, int e0, int nblk, int c, int j, int post, float *lin_w, float *ang_w, int head,
                                                const unsigned char *cnext, const unsigned short *cblk)
{
	if (post) quad_blocks_run_<true>(recs, idx, G, sums, e0, nblk, c, j, lin_w, ang_w, head, cnext, cblk);
	else quad_blocks_run_<false>(recs, idx, G, sums, e0, nblk, c, j, lin_w, ang_w, head, cnext, cblk);
}
// the couplings of row j of a block with the rows before it (aj[c] = slot c of row j's record, ai[c] = slot c of row i's): c_j . d_i
__device__ __forceinline__ float quad_coupling(const float4 aj[3], const float4 ai[3])
{
	float gsum = aj[0].x * ai[0].w;
	gsum = __fmaf_rn(aj[0].z, ai[0].y, gsum);
	gsum = __fmaf_rn(aj[1].x, ai[1].w, gsum); gsum = __fmaf_rn(aj[1].z, ai[1].y, gsum);
	gsum = __fmaf_rn(aj[2].x, ai[2].w, gsum); gsum = __fmaf_rn(aj[2].z, ai[2].y, gsum);
	return gsum;
}


#define NBODY 17
#define STRIDE 1280      // entries / records per frame
#define NOOP (STRIDE - 1)

struct frame_lists
{
	int ccnt[32], cstart[32];                 // row by row: chain of body b (bodies 0..15; the 17th body's rows ride behind body 15's)
	int seg[4], nblk[4], head[4];             // four at a time: segment (first entry), blocks, first body of every DPP row
	unsigned char cnext[32]; unsigned short cblk[32];
};

extern "C" __global__ void __launch_bounds__(64) k_rows(const float *recs, const unsigned short *idx, const frame_lists *fl, float *out, int sweeps)
{
	extern __shared__ float lds[];
	float *sums = lds, *lin_w = lds + STRIDE, *ang_w = lin_w + 128;
	unsigned short *lidx = reinterpret_cast<unsigned short *>(ang_w + 128);
	const int b = blockIdx.x, lane = threadIdx.x, quad = lane >> 2, c = lane & 3;
	const frame_lists &F = fl[b];
	for (int i = lane; i < STRIDE; i += 64) { sums[i] = 0.0f; lidx[i] = idx[(size_t)b * STRIDE + i]; }
	for (int i = lane; i < 128; i += 64) { lin_w[i] = 0.001f * i; ang_w[i] = 0.002f * i; }
	__syncthreads();
	const float *R = recs + (size_t)b * STRIDE * CREC;
	for (int s = 0; s < sweeps; s++)
	{
		const int cnt = F.ccnt[quad], start = F.cstart[quad];
		if (cnt > 0)
		{
			quad_body qb = { lin_w[4 * quad + c], ang_w[4 * quad + c] };
			quad_chain_run(qb, R, lidx + start, sums + start, cnt, c, s >= 16);
			if (c < 3) { lin_w[4 * quad + c] = qb.l; ang_w[4 * quad + c] = qb.av; }
		}
		__syncthreads();
	}
	if (lane < 64) out[(size_t)b * 128 + lane] = lin_w[lane] + ang_w[lane];
}

extern "C" __global__ void __launch_bounds__(64) k_blocks(const float *recs, const unsigned short *idx, const float4 *G, const frame_lists *fl, float *out, int sweeps)
{
	extern __shared__ float lds[];
	float *sums = lds, *lin_w = lds + STRIDE, *ang_w = lin_w + 128;
	unsigned short *lidx = reinterpret_cast<unsigned short *>(ang_w + 128);
	__shared__ unsigned char cnext[32]; __shared__ unsigned short cblk[32];
	const int b = blockIdx.x, lane = threadIdx.x, row = lane >> 4, j = (lane >> 2) & 3, c = lane & 3;
	const frame_lists &F = fl[b];
	for (int i = lane; i < STRIDE; i += 64) { sums[i] = 0.0f; lidx[i] = idx[(size_t)b * STRIDE + i]; }
	for (int i = lane; i < 128; i += 64) { lin_w[i] = 0.001f * i; ang_w[i] = 0.002f * i; }
	if (lane < 32) { cnext[lane] = F.cnext[lane]; cblk[lane] = F.cblk[lane]; }
	__syncthreads();
	const float *R = recs + (size_t)b * STRIDE * CREC;
	const float4 *Gf = G + (size_t)b * STRIDE;
	for (int s = 0; s < sweeps; s++)
	{
		const int nblk = F.nblk[row];
		if (nblk > 0) quad_blocks_run(R, lidx, Gf, sums, F.seg[row], nblk, c, j, s >= 16, lin_w, ang_w, F.head[row], cnext, cblk);
		__syncthreads();
	}
	if (lane < 64) out[(size_t)b * 128 + lane] = lin_w[lane] + ang_w[lane];
}

static float frand() { return (float)rand() / RAND_MAX - 0.5f; }

int main(int argc, char **argv)
{
	const int B = argc > 1 ? atoi(argv[1]) : 1024, sweeps = 20;
	const int lens[NBODY] = { 250, 90, 80, 70, 60, 60, 50, 50, 40, 40, 40, 40, 30, 30, 30, 30, 20 };
	int nrows = 0; for (int k = 0; k < NBODY; k++) nrows += lens[k];
	std::vector<float> recs((size_t)STRIDE * CREC), g((size_t)STRIDE * 4);
	for (int i = 0; i < STRIDE - 1; i++)
	{
		float *r = &recs[(size_t)i * CREC];
		for (int s = 0; s < 3; s++) { r[4 * s] = 0.01f * frand(); r[4 * s + 1] = 0.1f * frand(); r[4 * s + 2] = 0.01f * frand(); r[4 * s + 3] = frand(); }
		r[12] = 0.01f * frand(); r[13] = 0.01f * frand(); r[14] = -0.05f; r[15] = 0.05f;
	}
	for (auto &v : g) v = 0.01f * frand();
	// row by row: bodies 0..15 on their quads, body 16's rows appended to body 15's
	frame_lists F; memset(&F, 0, sizeof F);
	std::vector<unsigned short> idx_rows(STRIDE, NOOP), idx_blk(STRIDE, NOOP);
	{
		int pos = 0, rec = 0;
		for (int k = 0; k < 16; k++) { const int n = lens[k] + (k == 15 ? lens[16] : 0); F.cstart[k] = pos; F.ccnt[k] = n; for (int i = 0; i < n; i++) idx_rows[pos++] = (unsigned short)rec++; }
	}
	// four at a time: longest first onto the DPP row with the fewest blocks
	{
		int order[NBODY]; for (int k = 0; k < NBODY; k++) order[k] = k;
		std::sort(order, order + NBODY, [&](int a, int b) { return lens[a] > lens[b]; });
		int load[4] = { 0, 0, 0, 0 }, last[4] = { -1, -1, -1, -1 }, rowof[NBODY], startblk[NBODY];
		for (int k = 0; k < 32; k++) F.cnext[k] = 255;
		for (int o = 0; o < NBODY; o++)
		{
			const int k = order[o], nb = (lens[k] + 3) / 4;
			int r = 0; for (int q = 1; q < 4; q++) if (load[q] < load[r]) r = q;
			rowof[k] = r; startblk[k] = load[r]; load[r] += nb; F.cblk[k] = (unsigned short)nb;
			if (last[r] >= 0) F.cnext[last[r]] = (unsigned char)k; else F.head[r] = k;
			last[r] = k;
		}
		int seg = 0;
		for (int r = 0; r < 4; r++) { F.seg[r] = 4 * seg; F.nblk[r] = load[r]; seg += load[r]; }
		int rec = 0;
		for (int k = 0; k < NBODY; k++) { const int e = F.seg[rowof[k]] + 4 * startblk[k]; for (int i = 0; i < lens[k]; i++) idx_blk[e + i] = (unsigned short)rec++; }
		printf("rows %d; blocks per DPP row %d %d %d %d (longest chain alone: %d blocks)\n", nrows, load[0], load[1], load[2], load[3], (lens[0] + 3) / 4);
	}
	float *d_recs, *d_out; unsigned short *d_idx_rows, *d_idx_blk; float4 *d_g; frame_lists *d_fl;
	hipMalloc(&d_recs, (size_t)B * STRIDE * CREC * 4 + 4096); hipMalloc(&d_g, (size_t)B * STRIDE * 16 + 4096); hipMalloc(&d_out, (size_t)B * 128 * 4);
	hipMalloc(&d_idx_rows, (size_t)B * STRIDE * 2 + 4096); hipMalloc(&d_idx_blk, (size_t)B * STRIDE * 2 + 4096); hipMalloc(&d_fl, (size_t)B * sizeof F);
	for (int b = 0; b < B; b++)
	{
		hipMemcpy(d_recs + (size_t)b * STRIDE * CREC, recs.data(), recs.size() * 4, hipMemcpyHostToDevice);
		hipMemcpy(d_g + (size_t)b * STRIDE, g.data(), g.size() * 4, hipMemcpyHostToDevice);
		hipMemcpy(d_idx_rows + (size_t)b * STRIDE, idx_rows.data(), STRIDE * 2, hipMemcpyHostToDevice);
		hipMemcpy(d_idx_blk + (size_t)b * STRIDE, idx_blk.data(), STRIDE * 2, hipMemcpyHostToDevice);
		hipMemcpy(d_fl + b, &F, sizeof F, hipMemcpyHostToDevice);
	}
	const size_t smem = 36 * 1024;
	hipFuncSetAttribute(reinterpret_cast<const void *>(k_rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
	hipFuncSetAttribute(reinterpret_cast<const void *>(k_blocks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int variant = 0; variant < 2; variant++)
	{
		float best = 1e9f;
		for (int rep = 0; rep < 5; rep++)
		{
			hipEventRecord(e0, 0);
			if (variant == 0) hipLaunchKernelGGL(k_rows, dim3(B), dim3(64), smem, 0, d_recs, d_idx_rows, d_fl, d_out, sweeps);
			else hipLaunchKernelGGL(k_blocks, dim3(B), dim3(64), smem, 0, d_recs, d_idx_blk, d_g, d_fl, d_out, sweeps);
			hipEventRecord(e1, 0); hipEventSynchronize(e1);
			float ms = 0; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
		}
		const hipError_t err = hipDeviceSynchronize();
		printf("%s: %d frames, %d sweeps: %.3f ms (%s)\n", variant == 0 ? "row by row, 16 quads   " : "four at a time, 4 DPP rows", B, sweeps, best, hipGetErrorString(err));
	}
	return 0;
}
