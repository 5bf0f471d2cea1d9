// ht_quad.hpp -- "quad layout" helpers shared by the constraint-solver kernels: a body's momenta live one component per lane in a
// quad of 4 lanes (x, y, z, spare); cross and dot products reach the other components through DPP quad permutes.
#pragma once
#include "ht_device.hpp"

// ---- DPP helpers (quad layout) -------------------------------------------------------------------
#define QP_BC0 0x00        // quad_perm:[0,0,0,0]
#define QP_BC1 0x55        // quad_perm:[1,1,1,1]
#define QP_BC2 0xAA        // quad_perm:[2,2,2,2]
#define QP_BC3 0xFF        // quad_perm:[3,3,3,3]
#define QP_ROT1 0xC9       // quad_perm:[1,2,0,3]: lane c reads component (c+1)%3
#define QP_ROT2 0xD2       // quad_perm:[2,0,1,3]: lane c reads component (c+2)%3
#define DPP_ROW_SHL4 0x104 // lane i reads lane i+4 of its 16-lane row
#define DPP_ROW_SHR4 0x114 // lane i reads lane i-4
template <int CTRL> __device__ __forceinline__ float dpp(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true)); }
// value of the same component held by the other quad of a lane pair (quads 2p and 2p+1)
__device__ __forceinline__ float pair_swap(float v)
{
	int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), DPP_ROW_SHL4, 0xF, 0x5, false);       // quads 0 and 2 of a row read lane+4
	t = __builtin_amdgcn_update_dpp(t, __float_as_int(v), DPP_ROW_SHR4, 0xF, 0xA, false);           // quads 1 and 3 read lane-4
	return __int_as_float(t);
}
// The other quad's value of something all four lanes of a quad agree on (a quad's reduced sum): the mirror image within the pair's eight lanes is a lane of the
// other quad -- one DPP operand of the instruction that uses it, where pair_swap is a zero and two masked moves.
__device__ __forceinline__ float pair_other_uniform(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141 /* row_half_mirror */, 0xF, 0xF, true)); }
// max(lo, min(hi, x)) for lo <= hi in one instruction; equals the reference's std::min/std::max pair except for the sign of a zero
// result and NaN operands (a NaN impulse ends in the SanityCheck reset either way)
__device__ __forceinline__ float clamp_med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
// x / y as the IEEE fp32 division expands (reciprocal estimate, one Newton step, quotient with two residual corrections)
__device__ __forceinline__ float div_ieee(float x, float y)
{
	float r = __builtin_amdgcn_rcpf(y);
	const float e = __fmaf_rn(-y, r, 1.0f);
	r = __fmaf_rn(e, r, r);
	float q = x * r;
	float rem = __fmaf_rn(-y, q, x);
	q = __fmaf_rn(rem, r, q);
	rem = __fmaf_rn(-y, q, x);
	return __fmaf_rn(rem, r, q);
}


// refined reciprocal of the IEEE division above: depends on the divisor only, so rows whose divisor is sweep-invariant store it once
__device__ __forceinline__ float rcp_refined(float y)
{
	float r = __builtin_amdgcn_rcpf(y);
	const float e = __fmaf_rn(-y, r, 1.0f);
	return __fmaf_rn(e, r, r);
}
// x / y given r = rcp_refined(y): the remaining five operations of div_ieee, bit for bit the same quotient
__device__ __forceinline__ float div_ieee_r(float x, float y, float r)
{
	float q = x * r;
	float rem = __fmaf_rn(-y, q, x);
	q = __fmaf_rn(rem, r, q);
	rem = __fmaf_rn(-y, q, x);
	return __fmaf_rn(rem, r, q);
}

// ---- single-body row chains ------------------------------------------------------------------------------------------------------
// Arithmetic of a row (round 3).  LimitLinear::Iter (physics.h:289-307) forms the anchor velocity from the momenta every time it runs:
//   vn = dot(cross(Iinv*L, r1) + P*massinv, n)                 (L, P: angular / linear momentum of the body, r1: lever arm, n: row direction)
// which is linear in the momenta with coefficients that do not change during one PhysicsUpdate (orientation and Iinv are only re-made by
// rbupdatepose at its end).  The same number in "Jacobian form":
//   vn = dot(b, L) + dot(n*massinv, P)       with  g = cross(r1, n),  b = Iinv*g   (Iinv is symmetric)
//   impulse = (-targetspeed - vn) * (1 / effective mass), clamped to [fmin*dt - sum, fmax*dt - sum]
//   P += n*impulse,  L += g*impulse          (= ApplyImpulse, physics.h:222-226: cross(r1, n*impulse) = g*impulse)
// g, b and the reciprocal are pre-computed once per row with the reference's own expressions (the effective mass exactly as physics.h:299-300
// writes it); a sweep then costs 8 dependent operations per row instead of 24, with fused multiply-adds.  It is the reference's algorithm (same
// rows, same order, same clamps) evaluated in another association order: results differ from the reference's IEEE build by rounding, measured at
// <= 2e-7 m / 6e-6 (quaternion) per solver stage on the golden frames and, over the whole unit of work on the 256 bench frames, by LESS than the
// reference itself moves between its IEEE and its FMA-contracted build (tests/golden/ref_flag_spread.py, DESIGN.md section 4 "Numerics").
//
// Record of one single-body row (LimitLinear with rb0 == NULL), 4 x 16 bytes, laid out by the lane that reads each part (written by the kernel that makes
// the row: k_cloud_rows for the cloud rows, k_solve's prologue for the landmark-ray, boundary-plane and caller-built rows).  The reciprocal of the effective
// mass is folded into the coefficients, so the dot product IS vn / effective mass and the impulse is one subtraction away:
//   slot c (c = 0,1,2)  lane c of the quad:  n[c]*massinv/effmass, g[c], b[c]/effmass, n[c]
//   slot 3              lane 3:  targetspeed/effmass, the same after RemoveBias (physics.h:288: min(ts, ts_nobias)), fmin*dt, fmax*dt
// The records are read-only during the sweeps; the one value a row changes, its impulse sum, lives in an LDS array beside them (a store into
// the record would sit in the same in-order memory queue as the reads of the rows ahead and hold them back until it is acknowledged).
// Lane c < 3 carries component c of the body's momenta; lane 3 does the scalar part of the row (impulse, clamp, impulse sum) and hands the impulse to
// the others through a DPP operand.  Per row each lane issues ONE 16-byte read (the texture path moves 64 B per clock per CU, i.e. 16 clocks per
// such wave instruction: with four to eight waves per CU walking chains that path, not the arithmetic, is what a second read per row would
// saturate), one LDS read and one LDS write.
#define CREC HT_CREC       // floats per record
#define QP_PREV 0x90       // quad_perm:[0,0,1,2]: lane i reads lane i-1 of its quad
struct quad_body { float l, av; };      // this lane's component of the linear / angular momentum

// One LimitLinear::Iter: a = this lane's slot, sum = the row's impulse sum, POST = sweeps after RemoveBias.  Returns the new impulse sum (on
// every lane of the quad).
template <bool POST>
__device__ __forceinline__ float quad_row_step(quad_body &B, const float4 a, const float sum)
{
	const float p = __fmaf_rn(a.z, B.av, a.x * B.l);                    // (n[c]*massinv*P[c] + b[c]*L[c]) / effective mass
	const float t = dpp<QP_PREV>(p) + p;                                 // lane 1: p0 + p1
	const float s = dpp<QP_PREV>(t) + p;                                 // lane 2: (p0 + p1) + p2 = vn / effective mass
	const float x = -(POST ? a.y : a.x) - dpp<QP_PREV>(s);               // lane 3: (-targetspeed - vn) / effective mass
	const float impulse = clamp_med3(x, a.z - sum, a.w - sum);           // lane 3: a.z = fmin*dt, a.w = fmax*dt
	// lane 3's impulse to all four lanes: the broadcast rides on the three instructions that use it (a DPP operand each) instead of a move of its own; the two
	// wait states between the clamp and the first DPP read of its result are spelled out, the compiler does not look into the block
	float ns;
	asm volatile("s_nop 1\n\t"
	             "v_fmac_f32_dpp %0, %3, %4 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      // P += n * impulse
	             "v_fmac_f32_dpp %1, %3, %5 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      // L += g * impulse
	             "v_add_f32_dpp %2, %3, %6 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1"             // the row's new impulse sum
	             : "+v"(B.l), "+v"(B.av), "=&v"(ns) : "v"(impulse), "v"(a.w), "v"(a.y), "v"(sum));
	return ns;      // the same value on all four lanes (they read the same sum): the quad's four writes to one address agree
}
// Applies rows [0, cnt) of one chain in order.  The records of a frame lie where their producers wrote them (a cloud row's record at its point's index:
// k_cloud_rows writes it; the other single-body rows behind them: k_solve's prologue), and a chain is a list of record indices: recs = the frame's first
// record, idx = the chain's first index (LDS u16 or HBM u32), sums = the chain's first impulse sum, c = lane within the quad, post = 1 after RemoveBias.
// Eight register sets rotate: the read of a record is issued eight rows ahead of its use (a row is ~95 clocks and the records stream from L2, the
// Infinity Cache or HBM: 200 / 550 / 900 clocks away), its index is read another eight rows earlier, its impulse sum four rows ahead (LDS).  The loop trips of
// the quads of a wave differ, the compiler masks finished quads off.
// Reads run up to 23 indices, 15 records (of valid indices) and 11 sums past the chain's end: the caller's arrays have more slack than that, zero-filled.
#define QUAD_CHAIN_SLACK 48
// A quad can walk TWO bodies' chains back to back (k_solve, models with more than 16 bodies: the 17th body's rows follow the host body's, which are
// padded to a multiple of 8 with rows that change nothing): at row `kswitch` (a multiple of 8, or < 0 for none) the momenta go back to body `bodyA`'s
// slots of lin_w / ang_w and body `bodyB`'s state is taken up.
__device__ __forceinline__ void quad_switch_body(quad_body &B, int c, float *lin_w, float *ang_w, int bodyA, int bodyB)
{
	if (c < 3) { lin_w[4 * bodyA + c] = B.l; ang_w[4 * bodyA + c] = B.av; }
	B.l = lin_w[4 * bodyB + c]; B.av = ang_w[4 * bodyB + c];
}
template <bool POST, class IDX>
__device__ __forceinline__ void quad_chain_run_(quad_body &B, const float *recs, const IDX *idx, float *sums, int cnt, int c,
                                                int kswitch, float *lin_w, float *ang_w, int bodyA, int bodyB)
{
	// Round 5: EIGHT register sets (the record of a row is asked for eight rows ahead of its use, its index another eight rows earlier, its impulse sum four rows ahead).
	// Sixteen sets had measured no faster than eight when they were introduced (dead end (b)); the 48 registers they held are what lets the blocked
	// two-body phases keep their couplings in registers at two waves per SIMD.  Reads run up to 23 indices, 15 records (of valid indices) and 11 sums past the chain's end.
	const float4 *pa = reinterpret_cast<const float4 *>(recs) + c;
	const IDX *px = idx;
	float *ps = sums;
#define QC_LX(i, row) x##i = (unsigned)px[row]; __builtin_amdgcn_sched_barrier(0)
#define QC_LA(i) a##i = pa[4 * x##i]; __builtin_amdgcn_sched_barrier(0)
#define QC_LS(i, row) s##i = ps[row]; __builtin_amdgcn_sched_barrier(0)
	float4 a0, a1, a2, a3, a4, a5, a6, a7; float s0, s1, s2, s3, s4, s5, s6, s7;
	unsigned x0, x1, x2, x3, x4, x5, x6, x7;
	QC_LX(0, 0); QC_LX(1, 1); QC_LX(2, 2); QC_LX(3, 3); QC_LX(4, 4); QC_LX(5, 5); QC_LX(6, 6); QC_LX(7, 7);
	QC_LA(0); QC_LA(1); QC_LA(2); QC_LA(3); QC_LA(4); QC_LA(5); QC_LA(6); QC_LA(7);
	QC_LX(0, 8); QC_LX(1, 9); QC_LX(2, 10); QC_LX(3, 11); QC_LX(4, 12); QC_LX(5, 13); QC_LX(6, 14); QC_LX(7, 15);
	QC_LS(0, 0); QC_LS(1, 1); QC_LS(2, 2); QC_LS(3, 3);
	int k = 0;
	// The scheduling barriers keep every row's instructions between its own pair: left alone, the ILP-first scheduler hoists the first use of the
	// record that was requested last to the top of the trip as a hazard filler, which turns the wait for it into a wait for every outstanding
	// read (s_waitcnt vmcnt(0)), i.e. one full memory round trip per trip.
#define QC_STEP(i) ps[i] = quad_row_step<POST>(B, a##i, s##i)
	// row i of the trip: apply it, then ask for the record eight rows on (same register set; its index came in during the last trip), the index
	// sixteen rows on and the impulse sum four rows on (set i + 4)
#define QC_ROW(i, j) QC_STEP(i); QC_LA(i); QC_LX(i, 16 + i); QC_LS(j, 4 + i)
	for (; k + 8 <= cnt; k += 8)
	{
		if (k == kswitch) quad_switch_body(B, c, lin_w, ang_w, bodyA, bodyB);
		QC_ROW(0, 4); QC_ROW(1, 5); QC_ROW(2, 6); QC_ROW(3, 7); QC_ROW(4, 0); QC_ROW(5, 1); QC_ROW(6, 2); QC_ROW(7, 3);
		px += 8; ps += 8;
	}
	if (k == kswitch) quad_switch_body(B, c, lin_w, ang_w, bodyA, bodyB);
	const int left = cnt - k;      // 0..7 rows: their records are in the register sets, the sums of the first four too
#define QC_TAIL(i) if (left > i) { QC_STEP(i); } __builtin_amdgcn_sched_barrier(0)
#define QC_TAIL_S(i, j) if (left > i) { QC_STEP(i); QC_LS(j, 4 + i); } __builtin_amdgcn_sched_barrier(0)
	QC_TAIL_S(0, 4); QC_TAIL_S(1, 5); QC_TAIL_S(2, 6); QC_TAIL(3); QC_TAIL(4); QC_TAIL(5); QC_TAIL(6);
#undef QC_LX
#undef QC_LA
#undef QC_LS
#undef QC_STEP
#undef QC_ROW
#undef QC_TAIL
#undef QC_TAIL_S
}
// the sweeps before and after RemoveBias are two instances of the loop (the target-speed slot is chosen at compile time, not per row)
template <class IDX>
__device__ __forceinline__ void quad_chain_run(quad_body &B, const float *recs, const IDX *idx, float *sums, int cnt, int c, int post,
                                               int kswitch = -1, float *lin_w = nullptr, float *ang_w = nullptr, int bodyA = 0, int bodyB = 0)
{
	if (post) quad_chain_run_<true>(B, recs, idx, sums, cnt, c, kswitch, lin_w, ang_w, bodyA, bodyB);
	else quad_chain_run_<false>(B, recs, idx, sums, cnt, c, kswitch, lin_w, ang_w, bodyA, bodyB);
}
// ---- single-body rows FOUR at a time on the four quads of a DPP row (round 5; measured in round 4 on a synthetic frame, tools/probe/chain_blocks_probe.hip) -----------
// Consecutive rows of one body depend on each other only through the momenta, and linearly: with M the momenta before row 0 of a block of four rows,
//   vn_j / effmass_j = c_j . (M + sum_{i<j} d_i imp_i) = c_j . M + sum_{i<j} G(j,i) imp_i,      G(j,i) = c_j . d_i   (c_j = slots x, z of row j, d_i = slots w, y of row i)
// and G does not change during a PhysicsUpdate any more than the records do.  So the four quads of a DPP row (16 lanes) take the four rows of a block TOGETHER: every quad
// holds the body's momenta (the same values), forms its row's c_j . M side by side with the others (the expensive part: two products, a three-lane sum), then the impulses
// are resolved in row order -- imp_0 = clamp(x_0); x_j += -G(j,0) imp_0; imp_1 = clamp(x_1); ... : two dependent instructions per row -- and all quads add all four
// d_i imp_i to their momenta.  Same rows, same order, same clamps as LimitLinear::Iter (physics.h:289-307); one more association order of the same sums (G imp in place of
// c . (d imp)).  The wave's four DPP rows walk FOUR bodies' chains at a time, the bodies dealt out longest first so that the rows carry about equal numbers of blocks
// (k_solve's prologue), where sixteen quads on sixteen bodies wait for the longest chain: a frame's phase takes (its rows / 16) blocks instead of (its longest chain)
// rows -- and a depth frame puts a third to a half of its points on one bone.
// G travels in an array of its own, ten floats per block: -G(3,0) -G(3,1) -G(3,2) | -G(2,0) -G(2,1) 0 | -G(1,0) 0 0 | 0 -- the quad of row j reads three floats from its own
// offset (QUAD_G_OFF: 0, 3, 6 and, for row 0, 7: three zeros), i.e. its couplings with a zero wherever i >= j; written once per solve by the prologue.  In k_solve's build
// for 1024 frames the array lives in LDS (the records of the 128 frames resident on an XCD then fit its L2 as they did before the couplings existed).
//
// One block: a = this lane's slot of its quad's row, g = the row's couplings, sum = the row's impulse sum.  Returns the new impulse sum (all four lanes of the quad).
// Lane 3's column runs through the same instructions with the scalars of its slot: what it computes in p, t, s, ul, ua is never used.
#define QUAD_G_BLOCK 10      // floats of couplings per block of four rows
#define QUAD_G_OFF(j) ((j) == 3 ? 0 : (j) == 2 ? 3 : (j) == 1 ? 6 : 7)
struct __attribute__((packed, aligned(4))) quad_g3 { float x, y, z; };
template <bool POST>
__device__ __forceinline__ float quad_block_step(quad_body &B, const float4 a, const quad_g3 g, const float sum)
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
// the same values) and go back to lin_w / ang_w when the row moves on.  Four register sets rotate: a block's record and couplings are asked for four blocks ahead of its
// use, its index another four blocks earlier, its impulse sum two blocks ahead.  Reads run up to 8 blocks of indices, 4 of records (of valid indices) and couplings and 2
// of sums past the segment's end: into the next row's segment, or the slack behind the last (QUAD_BLOCK_SLACK entries, indices naming the no-op record).
#define QUAD_BLOCK_SLACK 32
template <bool POST, class IDX>
__device__ __forceinline__ void quad_blocks_run_(const float *recs, const IDX *idx, const float *G, float *sums, int e0, int nblk, int c, int j, float *lin_w, float *ang_w, int head,
                                                 const signed char *cnext, const int *cblk)
{
	const float4 *pa = reinterpret_cast<const float4 *>(recs) + c;
	const IDX *px = idx + e0 + j;
	const float *pg = G + QUAD_G_BLOCK * (e0 >> 2) + QUAD_G_OFF(j);
	float *ps = sums + e0 + j;
	quad_body B = { 0.0f, 0.0f };
	int cur = -1, nxt = head, left = 0;
#define QB_LX(i, blk) x##i = (unsigned)px[4 * (blk)]; __builtin_amdgcn_sched_barrier(0)
#define QB_LA(i) a##i = pa[4 * x##i]; __builtin_amdgcn_sched_barrier(0)
#define QB_LG(i, blk) g##i = *reinterpret_cast<const quad_g3 *>(pg + QUAD_G_BLOCK * (blk)); __builtin_amdgcn_sched_barrier(0)
#define QB_LS(i, blk) s##i = ps[4 * (blk)]; __builtin_amdgcn_sched_barrier(0)
	float4 a0, a1, a2, a3; quad_g3 g0, g1, g2, g3; float s0, s1, s2, s3;
	unsigned x0, x1, x2, x3;
	QB_LX(0, 0); QB_LX(1, 1); QB_LX(2, 2); QB_LX(3, 3);
	QB_LA(0); QB_LA(1); QB_LA(2); QB_LA(3);
	QB_LG(0, 0); QB_LG(1, 1); QB_LG(2, 2); QB_LG(3, 3);
	QB_LX(0, 4); QB_LX(1, 5); QB_LX(2, 6); QB_LX(3, 7);
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
	// block i of the trip: apply it; then the record and the couplings four blocks on for its register set (their index came in during the last trip), the index eight
	// blocks on, and the impulse sum two blocks on (set k)
#define QB_BLK(i, k) QB_BODY(); QB_STEP(i, i); QB_LA(i); QB_LG(i, 4 + i); QB_LX(i, 8 + i); QB_LS(k, 2 + i)
	int t = 0;
	for (; t + 4 <= nblk; t += 4)
	{
		QB_BLK(0, 2); QB_BLK(1, 3); QB_BLK(2, 0); QB_BLK(3, 1);
		px += 16; pg += 4 * QUAD_G_BLOCK; ps += 16;
	}
	const int rest = nblk - t;      // 0..3 blocks: their records and couplings are in the register sets, the sums of the first two too
#define QB_TAIL(i) if (rest > i) { QB_BODY(); QB_STEP(i, i); } __builtin_amdgcn_sched_barrier(0)
	if (rest > 0) { QB_BODY(); QB_STEP(0, 0); if (rest > 2) { QB_LS(2, 2); } } __builtin_amdgcn_sched_barrier(0);
	QB_TAIL(1); QB_TAIL(2);
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
__device__ __forceinline__ void quad_blocks_run(const float *recs, const IDX *idx, const float *G, float *sums, int e0, int nblk, int c, int j, int post, float *lin_w, float *ang_w, int head,
                                                const signed char *cnext, const int *cblk)
{
	if (post) quad_blocks_run_<true>(recs, idx, G, sums, e0, nblk, c, j, lin_w, ang_w, head, cnext, cblk);
	else quad_blocks_run_<false>(recs, idx, G, sums, e0, nblk, c, j, lin_w, ang_w, head, cnext, cblk);
}
// ---- ONE body's rows sixteen at a time (round 4: the single-body solves of k_reset, where one quad walked the whole cloud row by row while the batch waited) --------
// Consecutive rows of a body depend on each other only through the momenta, and linearly: with M the momenta before row 0 of a block of sixteen rows,
//   vn_j / effmass_j = c_j . (M + sum_{i<j} d_i imp_i) = c_j . M + sum_{i<j} G(j,i) imp_i,      G(j,i) = c_j . d_i   (c_j = slots x, z of row j's record, d_i = slots w, y of row i's)
// and G changes during a PhysicsUpdate no more than the records do.  So the sixteen quads of a wave take sixteen rows TOGETHER: every quad holds the body's momenta
// (the same values), forms its row's c_j . M side by side with the others (the expensive part: two products and a three-lane sum), then the impulses are resolved in
// row order -- imp_i = clamp(x_i), read from its lane into a scalar register, x_j += -G(j,i) imp_i for the rows behind it: three dependent instructions per row -- and
// every quad adds all sixteen d_i imp_i to its momenta.  Same rows, same order, same clamps as LimitLinear::Iter (physics.h:289-307); another association order of the
// same sums (G imp in place of c . (d imp); the sixteen contributions added as a tree), like the step from the reference's expression to quad_row_step was.
// ~80 instructions per sixteen rows where sixteen single rows issue ~300, the dependent chain ~55 instead of ~130.  (For k_solve's chains, many bodies per frame and 1024
// frames in flight, a four-row variant of this was measured and bought 10 %: DESIGN.md section 15.  Here one body has all the rows and eight blocks have the chip.)
// G travels with the records: 16 floats per row, -G(j,i) at position i, zero for i >= j.
// One block: a = this lane's slot of its quad's row, g = the row's sixteen couplings, sum = the row's impulse sum.  Returns the new impulse sum (all lanes of the quad).
// Lane 3's column runs through the vector instructions with the scalars of its slot: what it computes in p, t, s, ul, ua is never used.
template <bool POST>
__device__ __forceinline__ float quad_block16_step(quad_body &B, const float4 a, const float4 (&g)[4], const float sum)
{
	const float p = __fmaf_rn(a.z, B.av, a.x * B.l);
	const float t = dpp<QP_PREV>(p) + p;
	const float s = dpp<QP_PREV>(t) + p;
	float x = -(POST ? a.y : a.x) - dpp<QP_PREV>(s);                     // lane 3: (-targetspeed - c.M) / effective mass
	const float lo = a.z - sum, hi = a.w - sum;
	float imp;
	// row i's impulse is final once the rows before it have been taken into x_i; it goes to the rows behind it through a scalar register (quad i's lane 3 is lane 4i + 3)
#define QB16(i, G) imp = clamp_med3(x, lo, hi); x = __fmaf_rn(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(imp), 4 * (i) + 3)), G, x)
	QB16(0, g[0].x); QB16(1, g[0].y); QB16(2, g[0].z); QB16(3, g[0].w); QB16(4, g[1].x); QB16(5, g[1].y); QB16(6, g[1].z); QB16(7, g[1].w);
	QB16(8, g[2].x); QB16(9, g[2].y); QB16(10, g[2].z); QB16(11, g[2].w); QB16(12, g[3].x); QB16(13, g[3].y); QB16(14, g[3].z);
#undef QB16
	imp = clamp_med3(x, lo, hi);
	const float impq = dpp<QP_BC3>(imp);
	float ul = a.w * impq, ua = a.y * impq;                              // n[c] * imp_j, g[c] * imp_j
	// the sixteen rows' contributions, summed the same way in every quad (each step adds two values that both partners hold: additions commute, so the sixteen copies of
	// the momenta stay equal bit for bit): quads j and j + 2 of a DPP row, then j and j + 1, then the rows pairwise, then the pairs
	ul += dpp<0x128>(ul); ua += dpp<0x128>(ua);                          // row_ror:8
	ul += dpp<0x124>(ul); ua += dpp<0x124>(ua);                          // row_ror:4
	{ const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(ul), __float_as_uint(ul), false, false); ul = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
	{ const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(ua), __float_as_uint(ua), false, false); ua = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
	{ const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(ul), __float_as_uint(ul), false, false); ul = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
	{ const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(ua), __float_as_uint(ua), false, false); ua = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
	B.l += ul; B.av += ua;
	return sum + impq;
}
// Applies rows [0, 16 * nblk) in order, a whole wave on one body: recs = the rows' records back to back (16 floats each), G = their couplings (16 floats each), sums =
// their impulse sums; lane = lane of the wave.  Two register sets: a block's operands are read while the block before it is applied; reads run one block past the end.
#define QUAD_B16_SLACK 32      // rows the arrays hold behind the last block (read ahead, never applied)
template <bool POST>
__device__ __forceinline__ void quad_blocks16_run_(quad_body &B, const float *recs, const float *G, float *sums, int nblk, int lane)
{
	const int jq = lane >> 2, c = lane & 3;
	const float4 *pa = reinterpret_cast<const float4 *>(recs) + 4 * jq + c;
	const float4 *pg = reinterpret_cast<const float4 *>(G) + 4 * jq;
	float *ps = sums + jq;
	float4 a0 = pa[0], a1, g0[4] = { pg[0], pg[1], pg[2], pg[3] }, g1[4];
	float s0 = ps[0], s1;
	for (int blk = 0; blk < nblk; blk += 2)
	{
		a1 = pa[64]; g1[0] = pg[64]; g1[1] = pg[65]; g1[2] = pg[66]; g1[3] = pg[67]; s1 = ps[16];
		ps[0] = quad_block16_step<POST>(B, a0, g0, s0);
		if (blk + 1 >= nblk) break;
		a0 = pa[128]; g0[0] = pg[128]; g0[1] = pg[129]; g0[2] = pg[130]; g0[3] = pg[131]; s0 = ps[32];
		ps[16] = quad_block16_step<POST>(B, a1, g1, s1);
		pa += 128; pg += 128; ps += 32;
	}
}
__device__ __forceinline__ void quad_blocks16_run(quad_body &B, const float *recs, const float *G, float *sums, int nblk, int lane, int post)
{
	if (post) quad_blocks16_run_<true>(B, recs, G, sums, nblk, lane); else quad_blocks16_run_<false>(B, recs, G, sums, nblk, lane);
}
// -G(j,i): what row i's impulse adds to row j's (target speed - velocity) / effective mass; rj, ri = the rows' records
__device__ __forceinline__ float quad_coupling16(const float4 *rj, const float4 *ri)
{
	float gsum = rj[0].x * ri[0].w;
	gsum = __fmaf_rn(rj[0].z, ri[0].y, gsum);
	gsum = __fmaf_rn(rj[1].x, ri[1].w, gsum); gsum = __fmaf_rn(rj[1].z, ri[1].y, gsum);
	gsum = __fmaf_rn(rj[2].x, ri[2].w, gsum); gsum = __fmaf_rn(rj[2].z, ri[2].y, gsum);
	return -gsum;
}
// fills one record (see the layout above); r1 = lever arm in the world frame, n = row direction, Iinv = the body's world inverse inertia, minv = its inverse
// mass, y = effective mass (physics.h:299-300, formed by the caller with the reference's expression)
__device__ __forceinline__ void quad_write_record(float *rec, v3 r1, v3 n, const m3 &Iinv, float minv, float ts, float ts_post, float y, float fmin_dt, float fmax_dt)
{
	float4 *o = reinterpret_cast<float4 *>(rec);
	const v3 g = cross(r1, n), b = mul(Iinv, g);
	const float rinv = 1.0f / y;
	o[0] = make_float4((n.x * minv) * rinv, g.x, b.x * rinv, n.x); o[1] = make_float4((n.y * minv) * rinv, g.y, b.y * rinv, n.y); o[2] = make_float4((n.z * minv) * rinv, g.z, b.z * rinv, n.z);
	o[3] = make_float4(ts * rinv, ts_post * rinv, fmin_dt, fmax_dt);
}
// a record that changes nothing (zero direction, zero limits: impulse 0)
__device__ __forceinline__ void quad_write_noop(float *rec)
{
	float4 *o = reinterpret_cast<float4 *>(rec);
	o[0] = o[1] = o[2] = o[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
