// ht_solver.hip -- the rigid-body constraint solve of one fit step on CDNA4, one wavefront per frame.
//
// Reference computations:
//   PhysModel::FitPointCloud            include/physmodel.h:345-356   (row assembly order: caller rows, cloud rows, joint rows, [contacts])
//   PhysicsUpdate                       third_party/physics.h:543-587 (16 Gauss-Seidel sweeps, RK4 pose, RemoveBias, 4 sweeps, commit)
//   LimitLinear::Iter / LimitAngular::Iter   physics.h:289-307, 251-265
//   ConstrainPositionNailed / ConstrainAngularRange(W) / ConstrainAngularDrive / ConstrainConeAngle / ...Deadzone   physics.h:313-414
//   ConstrainContacts                   physics.h:463-489
//   rbinitvelocity / rbcalcnextpose / rkupdateq / rbupdatepose   physics.h:202-218, 500-541
//   HandModelEnhancements, CNNOutputAnalysis::ApplyAngles, the landmark-ray rows of MultiStepSim   include/handtrack.h:406-441, 203-216, 666-676
//   SanityCheck                         include/physmodel.h:437-442
//
// Order-preserving parallelism.  The reference applies rows strictly in vector order; two rows commute exactly when they touch disjoint
// bodies.  (1) Rows with rb0 == NULL touch one body only and form a prefix of the row vector (chamber / landmark-ray rows, then
// cloud rows): the prefix is stably partitioned by body and pre-computed into 64-byte records (ht_quad.hpp); the chains of different
// bodies run side by side.
// (2) The two-body tail (joint rows, contact triples; then all angular rows) is list-scheduled: level(row) = 1 + the highest level
// of an earlier row sharing a body, so conflicting rows keep their order and rows of one level touch disjoint bodies.
//
// Lane mapping of a sweep ("quad layout").  A body's momenta live one component per lane in a quad of 4 lanes (x, y, z, spare):
// every 3-vector operation is one scalar instruction per lane, dot products meet through DPP quad permutes (no LDS, no extra instruction
// when the permute folds into the consumer).  Chains: quad q walks body q, then a body >= 16 it hosts.  Two-body rows: lanes 8p..8p+7
// (two quads = the two bodies) take the p-th group of the current step; the sides meet through DPP row shifts.
//
// Arithmetic of a row (round 3): Jacobian form.  What LimitLinear::Iter / LimitAngular::Iter (physics.h:289-307, 251-265) recompute from
// the momenta every time -- r = R*position, cross(Iinv*L, r), the effective mass -- does not change during one PhysicsUpdate (orientation and
// Iinv are only re-made by rbupdatepose at its end), so every row is reduced ONCE, by the prologue, to the coefficients of
//     vn = dot(b, L) + dot(n*massinv, P),  impulse = clamp((-targetspeed - vn) / effective mass),  P += n*impulse,  L += g*impulse
// with g = cross(r, n), b = Iinv*g (Iinv symmetric), all formed with the reference's expressions.  A sweep then issues ~20 instructions per
// row instead of ~80 and the dependent chain through the momenta is 8-11 operations.  Same rows, same order, same clamps as the reference;
// another association order of the floating-point operations: see ht_quad.hpp and DESIGN.md (Numerics) for the measured effect.
#include "ht_solve_shared.hpp"

template <int NGRP_, int NSUM_, int NANG_, int NIDX_, int AS_ = 2, int NCG_ = 0> struct lds_t
{
	static constexpr int NGRP = NGRP_, NSUM = NSUM_, NANG = NANG_, NIDX = NIDX_, MAXA2 = MAXA2_OF(AS_), NCG = NCG_;
	static constexpr int LIDLE = MAXG - 1;                      // slot of the idle entry in lorder
	float pool[NGRP * LGRP] __attribute__((aligned(16)));      // two-body linear groups; first member: group addresses then fit the short offsets of two-address LDS reads
	// body state in 16-byte records: component c of body b is word 4*b + c
	float4 lin4[HT_MAXNB];                 // xyz linear momentum, w = massinv
	float4 ang4[HT_MAXNB];                 // xyz angular momentum, w = friction
	float pos[HT_MAXNB][3], q[HT_MAXNB][4];
	float csum[NSUM];                      // impulse sum of every single-body row, in chain order (+ read-ahead slack)
	unsigned short cidx[NIDX > 0 ? NIDX : 2];   // the chains: record index of every single-body row, in chain order (+ read-ahead slack); in HBM when the build has no room
	float cg[NCG];                         // the couplings of the single-body rows' blocks of four (ht_quad.hpp: QUAD_G_BLOCK floats per block); only the build for 1024 frames has room, else in HBM
	int ccnt[HT_MAXNB], cstart[HT_MAXNB];  // chain of body b: rows [cstart, cstart+ccnt) of the partitioned single-body stream
	signed char cextra[HT_MAXNB];          // body b < 16 hosts the chain of this body >= 16 on its quad (-1: none): it follows b's rows, padded to a multiple of 8
	union
	{
		unsigned lorder[MAXG];             // two-body linear groups sorted by step: group | rb0 << 16 | rb1 << 24; last slot = the idle entry
		struct { unsigned short abody[128]; unsigned etmp[64]; } blk;      // frames whose two-body rows are resolved in blocks (ht_block.hpp) have no level schedule: body pair of every angular row (rb0 | rb1 << 8, 255 = none), and the edge sort's table
	};
	unsigned short lstart[MAXG + 2];       // step L = lorder[lstart[L] .. lstart[L+1]); a step is a level, split so that it holds <= 8 groups
	unsigned aorder[MAXA2 + 1];            // angular row groups (runs of consecutive rows on the same body pair): first row | count << 8 | rb0 << 16 | rb1 << 24
	unsigned short astart[MAXA2 + 2];
	int nlev_lin, nlev_ang, nray;
	union
	{
		struct      // prologue only
		{
			float jr[HT_MAXNJ][6];                 // joint ranges after HandModelEnhancements
			float ray[36][HT_ROW];                 // landmark-ray rows: 4 per ray (MultiStepSim: 5 rays; slowfit: 8 rays + 3 nail rows)
			int aprefix[HT_MAXNJ + 1], rprefix[HT_MAXNJ + 1];
			unsigned char rowj[MAXA2];             // joint of every joint-range row
			unsigned char lrb[MAXG][2], arb[MAXA2][2];     // body pair of every group / angular row (255 = none), for the level schedule
			unsigned char gst[MAXA_RUNS + 1];      // first row of every angular run
			float4 I4[HT_MAXNB][3];                // columns of the world inverse inertia (w unused): only the row builders need it
		};
		float arec[(NANG + 4) * AROW] __attribute__((aligned(16)));      // sweeps: angular records (written once the prologue scratch is dead) + the idle record + read-ahead slack
	};
};


// Level schedule of one solve's groups (two-body linear groups, or runs of angular rows), on the whole wave.
//   level(group) = 1 + the highest level of an earlier group that shares a body with it, so conflicting rows keep the reference's order;
//   groups are counting-sorted by level (stable), and every level is cut into steps of at most 8 groups (one per lane pair).
// Group g is spoken for by lane g & 63 (slot g >> 6) with its bodies b0, b1 (255 = none).  The levels come from one pass in group order whose
// state -- the level of the latest group on each body -- sits in the lanes (lane k = body k) and is read and written with v_readlane / compare-
// select; a group's place in the sorted order and its step follow from counting, every lane over all groups.  (One lane walking LDS arrays took
// 52 k cycles per launch for ~20 groups and ~30 runs, 7 % of the kernel.)
//   put_order(pos, g, slot): group g takes place pos;  put_start(step, pos): step `step` (from 1) begins at place pos; returns the number of steps,
//   and start[steps + 1] = start[steps + 2] = n as the sweeps' read-ahead expects.
template <int NS, class Order, class Start>
__device__ __forceinline__ int level_schedule(int n, int lane, const int (&b0)[2], const int (&b1)[2], Order put_order, Start put_start)
{
	auto rdl = [](int v, int l) -> int { return __builtin_amdgcn_readlane(v, l); };
	int last = 0, lev[2] = { 0, 0 };
	for (int g = 0; g < n; g++)
	{
		const int l = g & 63;
		const bool hi = NS > 1 && g >= 64;
		const int x0 = hi ? rdl(b0[1], l) : rdl(b0[0], l), x1 = hi ? rdl(b1[1], l) : rdl(b1[0], l);
		const int l0 = x0 != 255 ? rdl(last, x0 & 63) : 0, l1 = x1 != 255 ? rdl(last, x1 & 63) : 0;
		const int lv = (l0 > l1 ? l0 : l1) + 1;
		last = (lane == x0 || lane == x1) ? lv : last;
		lev[0] = (lane == l && !hi) ? lv : lev[0];
		if (NS > 1) lev[1] = (lane == l && hi) ? lv : lev[1];
	}
	int lower[2] = { 0, 0 }, before[2] = { 0, 0 };
	for (int h = 0; h < n; h++)
	{
		const int lh = (NS > 1 && h >= 64) ? rdl(lev[1], h & 63) : rdl(lev[0], h & 63);
#pragma unroll
		for (int s = 0; s < NS; s++) { lower[s] += lh < lev[s] ? 1 : 0; before[s] += (lh == lev[s] && h < lane + 64 * s) ? 1 : 0; }
	}
	int key[2], sb[2] = { 0, 0 };
#pragma unroll
	for (int s = 0; s < 2; s++) key[s] = (s < NS && lane + 64 * s < n) ? lev[s] * 2 + ((before[s] & 7) == 0 ? 1 : 0) : 0x7ffffffe;
	for (int h = 0; h < n; h++)
	{
		const int kh = (NS > 1 && h >= 64) ? rdl(key[1], h & 63) : rdl(key[0], h & 63);
#pragma unroll
		for (int s = 0; s < NS; s++) sb[s] += ((kh >> 1) < lev[s]) ? (kh & 1) : 0;
	}
	int steps = 0;
#pragma unroll
	for (int s = 0; s < NS; s++)
	{
		const int g = lane + 64 * s;
		const bool valid = g < n, first = valid && (before[s] & 7) == 0;
		const int pos = lower[s] + before[s];
		if (valid) put_order(pos, g, s);
		if (first) put_start(1 + sb[s] + (before[s] >> 3), pos);
		steps += __popcll(__ballot(first));
	}
	if (lane == 0) { put_start(steps + 1, n); put_start(steps + 2, n); }
	return steps;
}

// EXACT (tests only, ht_debug_exact_solver): the rows are built as always, but the sweeps are the reference's own -- every row's Iter (physics.h:251-265,
// 289-307) in the reference's row order and association order, no fused multiply-adds, one lane -- instead of the Jacobian-form sweeps.  With it the whole
// update reproduces the restatement bit for bit, which isolates the Jacobian-form arithmetic as the solver's only difference from the reference.
#define EX_LIN HT_EX_LIN   // two-body linear rows a frame can have in the exact instantiation (a.exact_lin [B][EX_LIN][HT_ROW])
template <int NGRP_, int NSUM_, int NANG_, int NIDX_, bool EXACT = false, int AS = 2, int NCG_ = 0>
__global__ __launch_bounds__(64, 2) void k_solve(ht_model_dev M, ht_physics_dev ph, solve_args a)      // two waves per SIMD: the register budget (256 with the accumulator file) of eight frames per CU in the small build
{
	__shared__ lds_t<NGRP_, NSUM_, NANG_, NIDX_, AS, NCG_> S;
	constexpr int ASLOTS = AS, MAXA2 = MAXA2_OF(AS), MAXA_LDS = MAXA_CAP_OF(AS);
	const int lane = threadIdx.x;
	const int b = a.frame_order ? a.frame_order[blockIdx.x] : (int)blockIdx.x;      // which frame: results do not depend on it, only when the frame's turn comes
	if (a.active_flag && !a.active_flag[b]) return;                // a launch never touches another launch's frames
	const long long t_cost = a.cost_out ? clock64() : 0;
	const long long t_entry = HT_DBG(a.dbg, 2048) ? clock64() : 0;
	const int nb = M.nb, nj = M.nj;
	float *st = a.state + (size_t)b * nb * HT_STATE_STRIDE;
	const float dt = ph.deltaT;
	// ---- round 6: the solve's tables come ready from k_solve_prep (ht_solve_shared.hpp, csrc/ht_prep.hip) -- the joints' groups, the angular records, the blocks' couplings
	//      and edges, the chain lists and their blocks' couplings -- unless the frame is one the blocked form does not hold (then everything below runs as it always did).
	//      The flags are the same on every lane.  What is left of the prologue for such a frame: the bodies, the contacts' groups and their couplings.
	const float *const T = a.tables ? a.tables + (size_t)b * TB_WORDS : nullptr;
	bool fast_pose = false, fast_chain = false;      // the pose-only tables (joints' groups, angular records, the blocks' couplings and edges) / the chain tables (lists, dealing, four-row couplings, landmark rays): k_solve_prep may have made either
	int t_na = 0, t_npre = 0, t_total = 0, t_e0 = 0, t_nblk = 0, t_head = 0;
	if constexpr (!EXACT)
	{
		if (T && !a.two_body_levels && !a.lin_tail && !a.ang_user && !a.no_model_rows && !(a.sf_refpose && a.sf_hold) && a.sf_select < 0 && a.sf_ncray == 0 && !a.rows_cloud)
		{
			const int h = lane < 32 ? reinterpret_cast<const int *>(T + TB_HDR)[lane] : 0;
			const int nc0 = (a.contacts && ph.use_collision) ? a.ncontacts[b] : 0;
			const int R = lane >> 4;
			t_na = __builtin_amdgcn_readlane(h, TH_NA); t_npre = __builtin_amdgcn_readlane(h, TH_NPRE); t_total = __builtin_amdgcn_readlane(h, TH_TOTAL);
			t_e0 = R == 0 ? __builtin_amdgcn_readlane(h, TH_E0) : R == 1 ? __builtin_amdgcn_readlane(h, TH_E0 + 1) : R == 2 ? __builtin_amdgcn_readlane(h, TH_E0 + 2) : __builtin_amdgcn_readlane(h, TH_E0 + 3);
			t_nblk = R == 0 ? __builtin_amdgcn_readlane(h, TH_NBLK) : R == 1 ? __builtin_amdgcn_readlane(h, TH_NBLK + 1) : R == 2 ? __builtin_amdgcn_readlane(h, TH_NBLK + 2) : __builtin_amdgcn_readlane(h, TH_NBLK + 3);
			t_head = R == 0 ? __builtin_amdgcn_readlane(h, TH_HEAD) : R == 1 ? __builtin_amdgcn_readlane(h, TH_HEAD + 1) : R == 2 ? __builtin_amdgcn_readlane(h, TH_HEAD + 2) : __builtin_amdgcn_readlane(h, TH_HEAD + 3);
			fast_pose = __builtin_amdgcn_readlane(h, TH_OK) != 0 && 3 * nj + 3 * (nc0 > HT_MAXCONTACT ? HT_MAXCONTACT : nc0) <= 4 * BLK_LROWS && !HT_DBG(a.dbg, 65536) && !HT_DBG(a.dbg, 1 << 22);
			fast_pose = __builtin_amdgcn_readfirstlane((int)fast_pose) != 0;
			fast_chain = fast_pose && __builtin_amdgcn_readlane(h, TH_CHAIN_OK) != 0;
			fast_chain = __builtin_amdgcn_readfirstlane((int)fast_chain) != 0;
		}
	}

	// ---- load state, rbinitvelocity (physics.h:500-519) ----
	if (lane < nb)
	{
		const float *s = st + lane * HT_STATE_STRIDE;
		const float *bc = M.bodyc + lane * HT_BC;
		v3 lin = V3(s[7], s[8], s[9]), ang = V3(s[10], s[11], s[12]);
		const float damp = bc[HT_BC_DAMPLEFT];
		lin = lin * damp; ang = ang * damp;
		// gravity is (0,0,0) for the tracker (handtrack.h:837) and gravscale 0: force*dt and torque*dt are exact zeros
		lin = lin + V3(0, 0, 0); ang = ang + V3(0, 0, 0);
		for (int i = 0; i < 3; i++) S.pos[lane][i] = s[i];
		for (int i = 0; i < 4; i++) S.q[lane][i] = s[3 + i];
		S.lin4[lane] = make_float4(lin.x, lin.y, lin.z, bc[HT_BC_MASSINV]);
		S.ang4[lane] = make_float4(ang.x, ang.y, ang.z, bc[HT_BC_FRICTION]);
		m3 I = world_inertia(V4(s[3], s[4], s[5], s[6]), LM(bc + HT_BC_TINV), bc[HT_BC_MASSINV]);
		S.I4[lane][0] = make_float4(I.x.x, I.x.y, I.x.z, 0.0f); S.I4[lane][1] = make_float4(I.y.x, I.y.y, I.y.z, 0.0f); S.I4[lane][2] = make_float4(I.z.x, I.z.y, I.z.z, 0.0f);
	}
	if (lane < nj && !fast_pose) for (int i = 0; i < 6; i++) S.jr[lane][i] = M.jointc[lane * HT_JC + HT_JC_RMIN + i];
	if (lane == 0) S.nray = fast_chain && a.ray_rows ? t_npre : 0;
	__syncthreads();

	// ---- HandModelEnhancements (handtrack.h:417-420, 434-440); acos()/cos() are the C double overloads there ----
	if (nb >= 17 && !a.no_model_rows && !fast_pose)
	{
		if (lane < 4)
		{
			const int bb = 7 + 3 * lane;
			float c = clamp_std(dot(qzdir(L4(S.q[bb - 2])), qzdir(L4(S.q[bb - 1]))), 0.0f, 1.0f);
			float ang = (float)(acos((double)c) * (double)180.0f / (double)3.14159f / (double)2.0f);
			S.jr[bb - 1][3] = ang; S.jr[bb - 1][0] = ang;
		}
		else if (lane < 8)
		{
			const int k = lane - 4;
			const int kb = k == 0 ? 14 : k == 1 ? 11 : k == 2 ? 8 : 5;
			const float r0 = k == 0 ? -30.0f : -10.0f, r1 = k == 3 ? 20.0f : 10.0f;          // handtrack.h:434
			bool up = (double)dot(qydir(L4(S.q[1])), qydir(L4(S.q[kb]))) > ph.cos40d;
			S.jr[kb - 1][1] = up ? r0 : -0.0f;
			S.jr[kb - 1][4] = up ? r1 : 0.0f;
		}
	}
	// ---- landmark-ray rows of MultiStepSim (handtrack.h:666-676): 2 dead-zone pairs per open finger ----
	if (fast_chain) {}
	else if (a.ray_rows && a.sf_ncray + (a.sf_select >= 0) > 0 && lane == 8)
	{
		// slowfit (handtrack.h:803-810): dead-zone pairs along the two axes perpendicular to each landmark ray, rays from the origin, then the nail
		int k = 0;
		for (int i = 0; i < a.sf_ncray && i < 8; i++)
		{
			const float *cr = a.sf_crays + ((size_t)b * 8 + i) * 4;
			v4 q = quat_from_to(V3(0, 0, 1), L3(cr));
			const int rb = FEATURE_BONE[i];
			const v3 off = V3(FEATURE_OFF[i][0], FEATURE_OFF[i][1], FEATURE_OFF[i][2]);
			for (int ax = 0; ax < 2; ax++)
			{
				const v3 axis = ax == 0 ? qxdir(q) : qydir(q);
				const float base = dot(anchor_world(S, rb, off) - V3(0, 0, 0), axis);       // ConstrainAlongDirectionDeadzone physics.h:332-340
				for (int sd = 0; sd < 2; sd++)
				{
					float *o = S.ray[k++];
					o[0] = -1.0f; o[1] = (float)rb; o[2] = 0.0f; o[3] = 0.0f; o[4] = 0.0f; o[5] = off.x; o[6] = off.y; o[7] = off.z;
					o[8] = axis.x; o[9] = axis.y; o[10] = axis.z; o[11] = sd == 0 ? base + 0.01f : base - 0.01f; o[12] = 0.0f;
					o[13] = sd == 0 ? fmin_std(0.0f, 100000.0f) : fmin_std(-100000.0f, 0.0f); o[14] = sd == 0 ? fmax_std(0.0f, 100000.0f) : fmax_std(-100000.0f, 0.0f); o[15] = 0.0f;
				}
			}
		}
		if (a.sf_select >= 0)      // ConstrainPositionNailed(NULL, spoint, selectrb, rbpoint) physics.h:342-346
		{
			const int rb = a.sf_select;
			const v3 sp = V3(a.sf_spoint[0], a.sf_spoint[1], a.sf_spoint[2]), rp = V3(a.sf_rbpoint[0], a.sf_rbpoint[1], a.sf_rbpoint[2]);
			const v3 d = anchor_world(S, rb, rp) - sp;
			for (int ax = 0; ax < 3; ax++)
			{
				float *o = S.ray[k++];
				o[0] = -1.0f; o[1] = (float)rb; o[2] = sp.x; o[3] = sp.y; o[4] = sp.z; o[5] = rp.x; o[6] = rp.y; o[7] = rp.z;
				o[8] = ax == 0 ? 1.0f : 0.0f; o[9] = ax == 1 ? 1.0f : 0.0f; o[10] = ax == 2 ? 1.0f : 0.0f; o[11] = ax == 0 ? d.x : ax == 1 ? d.y : d.z; o[12] = 0.0f;
				o[13] = -FLT_MAX; o[14] = FLT_MAX; o[15] = 0.0f;
			}
		}
		S.nray = k;
	}
	else if (a.ray_rows && lane == 8)
	{
		const float *an = a.analysis + (size_t)b * HT_ANALYSIS;
		const float *cam = a.cams + (size_t)b * HT_CAM;
		const v3 campos = V3(cam[5], cam[6], cam[7]);
		int k = 0;
		for (int i = (a.steps_keyangles ? 3 : 0); i < 8; i++)
			if (i >= 3 && an[HT_AN_CLENCH + i - 3] < 3.14f / 2.0f && an[HT_AN_CRAYS + 4 * i + 3] >= a.min_cray_prob)
			{
				v4 q = quat_from_to(V3(0, 0, 1), L3(an + HT_AN_CRAYS + 4 * i));
				const int rb = FEATURE_BONE[i];
				const v3 off = V3(FEATURE_OFF[i][0], FEATURE_OFF[i][1], FEATURE_OFF[i][2]);
				for (int ax = 0; ax < 2; ax++)
				{
					const v3 axis = ax == 0 ? qxdir(q) : qydir(q);
					const float base = dot(anchor_world(S, rb, off) - campos, axis);       // ConstrainAlongDirectionDeadzone physics.h:332-340
					for (int side = 0; side < 2; side++)
					{
						float *o = S.ray[k++];
						o[0] = -1.0f; o[1] = (float)rb; o[2] = campos.x; o[3] = campos.y; o[4] = campos.z; o[5] = off.x; o[6] = off.y; o[7] = off.z;
						o[8] = axis.x; o[9] = axis.y; o[10] = axis.z; o[11] = side == 0 ? base + 0.01f : base - 0.01f; o[12] = 0.0f;
						o[13] = side == 0 ? fmin_std(0.0f, 100000.0f) : fmin_std(-100000.0f, 0.0f); o[14] = side == 0 ? fmax_std(0.0f, 100000.0f) : fmax_std(-100000.0f, 0.0f); o[15] = 0.0f;
					}
				}
			}
		S.nray = k;
	}
	__syncthreads();
	if (HT_DBG(a.dbg, 256)) return;
	const long long t_m1 = HT_DBG(a.dbg, 2048) ? clock64() : 0;
	// ---- angular rows: [ApplyAngles 12] [arm cone 1] [joint ranges], generated by the lane that owns them ----
	// slowfit's RelativeAngularConstraints (physmodel.h:423-432, filter handtrack.h:799): one row per ranged axis of every joint that passes
	const bool rel = a.sf_refpose && a.sf_hold;
	const int na_user = a.ang_user ? (a.n_ang_user[b] < MAXA_LDS ? a.n_ang_user[b] : MAXA_LDS) : 0;      // the caller's rows lead the list (PhysModel::FitPointCloud appends its own, physmodel.h:351)
	// row counts per joint (lane j = joint j), their prefixes over the joints by a scalar walk through the lanes' registers, and the owner table of the range rows
	int na_pre, na;
	const int na_fix = na_user + (a.apply_angles ? 12 : 0) + (a.arm_cone ? 1 : 0);      // [caller's rows | ApplyAngles, arm cone | relative rows | joint ranges]
	if (fast_pose) { na_pre = na_fix; na = t_na; }
	else
	{
		const int acnt = (lane < nj && !a.no_model_rows) ? angular_range_count(L3(S.jr[lane]), L3(S.jr[lane] + 3)) : 0;
		int rcnt = 0;
		if (rel && lane < nj && ((lane != 0 && a.sf_hold == 2) || lane > 3)) for (int ax = 0; ax < 3; ax++) rcnt += S.jr[lane][ax] != S.jr[lane][3 + ax];
		int rpre = 0, apre = 0, rtot = 0, atot = 0;
		for (int j = 0; j < nj; j++)
		{
			const int rc = __builtin_amdgcn_readlane(rcnt, j), ac = __builtin_amdgcn_readlane(acnt, j);
			rpre += j < lane ? rc : 0; apre += j < lane ? ac : 0; rtot += rc; atot += ac;
		}
		na_pre = na_fix + rtot; na = na_pre + atot;
		if (lane <= nj) { S.rprefix[lane] = na_fix + rpre; S.aprefix[lane] = na_pre + apre; }
		for (int k = 0; k < acnt; k++) if (na_pre + apre + k < MAXA2) S.rowj[na_pre + apre + k] = (unsigned char)lane;
	}
	__syncthreads();
	if (na > MAXA_LDS && a.caps && lane == 0) atomicAdd(a.caps, 1);      // more angular rows than the kernel holds: the excess is dropped, and reported
	if (na > MAXA2) na = MAXA2;
	arow AR[ASLOTS];
#pragma unroll
	for (int s = 0; s < ASLOTS; s++)
	{
		const int r = lane + 64 * s;
		arow &R = AR[s];
		R.rb0 = -1; R.rb1 = -1; R.axis = V3(0, 0, 1); R.targetspin = -FLT_MAX; R.mn = 0; R.mx = 0; R.s2t = 0; R.torque = 0; R.mintorque = 0; R.lev = 0;
		if (r < na && !fast_pose)
		{
			float row[8];
			if (r < na_user)
			{
				const float *u = a.ang_user + ((size_t)b * a.ang_user_stride + r) * HT_AROW;
				put_ang(row, (int)u[0], (int)u[1], V3(u[2], u[3], u[4]), u[5], u[6], u[7]);
			}
			else if (r < na_fix)
			{
				const float *cam = a.cams + (size_t)b * HT_CAM;
				const v4 camq = V4(cam[8], cam[9], cam[10], cam[11]);
				const int ra = a.apply_angles ? r - na_user : 12;          // index into the ApplyAngles list, 12 = the arm cone
				if (ra == 12) cone_angle(ph, S, -1, qrot(camq, V3(0, -1, 0)), 0, V3(0, 0, 1), 70.0f, row);             // handtrack.h:426, 684
				else
				{
					const float *an = a.analysis + (size_t)b * HT_ANALYSIS;
					const float *fc = an + HT_AN_CLENCH;
					if (ra < 3)
					{
						float tmp[3][8];
						angular_drive(ph, S, -1, 1, qmul(camq, V4(an[HT_AN_PALMQ], an[HT_AN_PALMQ + 1], an[HT_AN_PALMQ + 2], an[HT_AN_PALMQ + 3])), a.drive_force, tmp);      // handtrack.h:206
						for (int k = 0; k < 8; k++) row[k] = ra == 0 ? tmp[0][k] : ra == 1 ? tmp[1][k] : tmp[2][k];
					}
					else if (ra == 3) { float th = fc[0]; cone_angle(ph, S, 1, V3((float)cos((double)th), 0, (float)sin((double)th)), 4, V3(0, 0, 1), 10.0f, row); }
					else
					{
						const int finger = 1 + (ra - 4) / 2;
						const float aa = fc[finger];
						if (((ra - 4) & 1) == 0) cone_angle(ph, S, 1, V3(0, (float)(-sin((double)aa)), (float)cos((double)aa)), 3 + finger * 3, V3(0, 0, 1), 10.0f, row);
						else
						{
							v4 jf = L4(M.jointc + (1 + finger * 3) * HT_JC + HT_JC_FRAME);
							v3 inner = V3(0, (float)(-sin((double)(aa / 2.0f))), (float)cos((double)(aa / 2.0f)));
							cone_angle(ph, S, 1, qrot(jf, qrot(jf, inner)), 2 + finger * 3, V3(0, 0, 1), 10.0f, row);
						}
					}
				}
			}
			else if (r < na_pre)      // RelativeAngularConstraints (physmodel.h:423-432): keep the joint's rotation relative to the reference pose
			{
				int j = 0;
				while (j + 1 < nj && S.rprefix[j + 1] <= r) j++;
				int sub = r - S.rprefix[j], ax = 0;
				for (int k = 0; k < 3; k++) if (S.jr[j][k] != S.jr[j][3 + k]) { if (sub == 0) { ax = k; sub = -1; } else if (sub > 0) sub--; }
				const float *jc = M.jointc + j * HT_JC;
				const int rb0 = (int)jc[HT_JC_RB0], rb1 = (int)jc[HT_JC_RB1];
				const float *r0 = a.sf_refpose + ((size_t)b * nb + rb0) * HT_POSE, *r1 = a.sf_refpose + ((size_t)b * nb + rb1) * HT_POSE;
				const xf ref0 = XF(L3(r0), L4(r0 + 3)), ref1 = XF(L3(r1), L4(r1 + 3));
				const xf dq = mul(mul(inverse(mul(inverse(ref0), ref1)), inverse(body_xf(S, rb0))), body_xf(S, rb1));
				const m3 R0 = qmat(L4(S.q[rb0]));
				const v3 axis = ax == 0 ? R0.x : ax == 1 ? R0.y : R0.z;
				const float qa = ax == 0 ? dq.q.x : ax == 1 ? dq.q.y : dq.q.z;
				put_ang(row, rb0, rb1, axis, -qa * 2.0f / dt, -FLT_MAX, FLT_MAX);
			}
			else
			{
				const int j = S.rowj[r];
				const int sub = r - S.aprefix[j];
				const float *jc = M.jointc + j * HT_JC;
				const int rb0 = (int)jc[HT_JC_RB0], rb1 = (int)jc[HT_JC_RB1];
				const v4 jf = L4(jc + HT_JC_FRAME);
				angular_range_row(ph, rb0, rb0 >= 0 ? qmul(L4(S.q[rb0]), jf) : jf, rb1, rb1 >= 0 ? L4(S.q[rb1]) : V4(0, 0, 0, 1), L3(S.jr[j]), L3(S.jr[j] + 3), sub, row);
			}
			R.rb0 = __float_as_int(row[0]); R.rb1 = __float_as_int(row[1]); R.axis = V3(row[2], row[3], row[4]); R.targetspin = row[5];
			const float mintorque = row[6], maxtorque = row[7];
			// physics.h:256-259: Iinv is invariant during the update, so 1/(axis.Iinv0.axis + axis.Iinv1.axis) is computed once
			R.s2t = 1.0f / (((R.rb0 >= 0) ? dot(R.axis, mul(body_I(S, R.rb0), R.axis)) : 0.0f) + ((R.rb1 >= 0) ? dot(R.axis, mul(body_I(S, R.rb1), R.axis)) : 0.0f));
			R.mn = mintorque * dt; R.mx = maxtorque * dt; R.mintorque = mintorque; R.torque = 0.0f;
			if constexpr (EXACT) { float *e = a.exact_ang + ((size_t)b * 256 + r) * 8; for (int k = 0; k < 8; k++) e[k] = row[k]; }
			S.arb[r][0] = (unsigned char)(R.rb0 >= 0 ? R.rb0 : 255); S.arb[r][1] = (unsigned char)(R.rb1 >= 0 ? R.rb1 : 255);
		}
	}

	if (HT_DBG(a.dbg, 512)) return;
	const long long t_m2 = HT_DBG(a.dbg, 2048) ? clock64() : 0;
	// ---- two-body linear rows: joints (physmodel.h:328-334) then contacts (physics.h:463-489), each row reduced by one lane to its part of the group record ----
	int nc = (a.contacts && ph.use_collision) ? a.ncontacts[b] : 0;
	if (nc > HT_MAXCONTACT) nc = HT_MAXCONTACT;
	const int njg = a.no_model_rows ? 0 : nj;                                             // joint groups
	const int nt = a.lin_tail ? a.n_lin_tail[b] : 0, ngt = a.lin_tail ? a.n_tail_groups[b] : 0;      // the caller's rows from its first two-body row on, and the groups the host packed them into
	const int n2 = nt + 3 * njg + 3 * nc, ng2 = ngt + njg + nc;
	const bool seq_lin = ng2 > MAXG - 1;      // more groups than the level schedule has tables for (more than 96 contacts: physics.h:451-462 keeps every contact): one group per step, in row order
	const int rec_cap = a.scratch_stride - HT_SCRATCH_TAIL;                               // rows of the frame's scratch slot that hold chain records
	float *scr = a.scratch + (size_t)b * a.scratch_stride * CREC;
	// the three arrays a build may be too small for (see the top of the file): in LDS when the frame fits, else in the tail of its scratch slot
	const bool pool_lds = ng2 + 1 <= S.NGRP;
	float *const gpool = scr + (size_t)rec_cap * CREC, *const garec = gpool + MAXG_CAP * LGRP;
	float *const pool = pool_lds ? S.pool : gpool;
	if (ngt > 0)      // a caller's group may hold fewer than three rows: the slots no row fills change nothing (zero direction, zero limits)
	{
		for (int i = lane; i < ngt * LGRP; i += 64) pool[i] = ((i % LGRP) >= LG_RINV && (i % LGRP) < LG_RINV + 3) ? 1.0f : 0.0f;
		__threadfence_block();
		__syncthreads();
	}
	if (fast_pose)      // the joints' groups as k_solve_prep made them (the same statements as the loop below, csrc/ht_prep.hip)
	{
		const float4 *src = reinterpret_cast<const float4 *>(T + TB_POOL);
		for (int i = lane; i < njg * (LGRP / 4); i += 64) reinterpret_cast<float4 *>(pool)[i] = src[i];
	}
	for (int r = (fast_pose ? 3 * njg : 0) + lane; r < n2; r += 64)
	{
		int rb0, rb1, meta = 0, g, kk;
		v3 p0, p1, n; float targetdist, tsnb, fmn, fmx;
		if (r < nt)      // a caller's row (LimitLinear, physics.h:267-287) in the layout of ht_stage_cloud_rows
		{
			const float *u = a.lin_tail + ((size_t)b * a.lin_tail_stride + r) * HT_ROW;
			const unsigned pos = a.lin_tail_pos[(size_t)b * a.lin_tail_stride + r];
			g = (int)((pos & 0x7FFF) >> 2); kk = (int)(pos & 3);
			rb0 = (int)u[0]; rb1 = (int)u[1]; p0 = L3(u + 2); p1 = L3(u + 5); n = L3(u + 8); targetdist = u[11]; tsnb = u[12]; fmn = u[13]; fmx = u[14];
			if (pos & 0x8000) meta = kk == 0 ? LM_NORMAL : LM_FRIC;
		}
		else if (r < nt + 3 * njg)
		{
			const int j = (r - nt) / 3, ax = (r - nt) % 3;
			g = ngt + j; kk = ax;
			const float *jc = M.jointc + j * HT_JC;
			rb0 = (int)jc[HT_JC_RB0]; rb1 = (int)jc[HT_JC_RB1];
			p0 = L3(jc + HT_JC_P0) - L3(M.bodyc + rb0 * HT_BC + HT_BC_COM); p1 = L3(jc + HT_JC_P1) - L3(M.bodyc + rb1 * HT_BC + HT_BC_COM);
			const v3 d = anchor_world(S, rb1, p1) - anchor_world(S, rb0, p0);                       // ConstrainPositionNailed physics.h:342-346
			n = V3(ax == 0 ? 1.0f : 0.0f, ax == 1 ? 1.0f : 0.0f, ax == 2 ? 1.0f : 0.0f);
			targetdist = ax == 0 ? d.x : ax == 1 ? d.y : d.z; tsnb = 0.0f; fmn = -FLT_MAX; fmx = FLT_MAX;
		}
		else
		{
			const int ci = (r - nt - 3 * njg) / 3, k = (r - nt - 3 * njg) % 3;
			g = ngt + njg + ci; kk = k;
			const float *c = a.contacts + ((size_t)b * HT_MAXCONTACT + ci) * HT_CONTACT;
			rb0 = (int)c[0]; rb1 = (int)c[1];
			const v3 normal = L3(c + 2), p0w = L3(c + 5), p1w = L3(c + 8);
			const float separation = c[11];
			p0 = apply(inverse(body_xf(S, rb0)), p0w); p1 = apply(inverse(body_xf(S, rb1)), p1w);          // PhysContact physics.h:431-432
			if (k == 0)
			{
				const v3 r0w = p0w - L3(S.pos[rb0]), r1w = p1w - L3(S.pos[rb1]);
				const v3 v0 = cross(spin_of(S, rb0), r0w) + F3(S.lin4[rb0]) * S.lin4[rb0].w;
				const v3 v1 = cross(spin_of(S, rb1), r1w) + F3(S.lin4[rb1]) * S.lin4[rb1].w;
				const v3 v = v0 - v1;
				const float minsep = ph.driftmax * 0.25f;
				const float bouncevel = fmax_std(0.0f, (-dot(normal, v) - ph.gravity_len * ph.falltime_to_ballistic) * ph.restitution);
				n = -normal; targetdist = fmin_std((separation - minsep) * ph.biasfactorpositive, separation); tsnb = -bouncevel; fmn = 0; fmx = FLT_MAX;
				meta = LM_NORMAL;
			}
			else
			{
				v4 q = quat_from_to(V3(0, 0, 1), -normal);
				n = k == 1 ? qydir(q) : qxdir(q);           // row order: normal, binormal (friction_master -1), tangent (-2)
				targetdist = 0; tsnb = 0; fmn = 0;
				fmx = 0;
				meta = LM_FRIC;
			}
		}
		if constexpr (EXACT)
		{
			if (r < EX_LIN)
			{
				float *e = a.exact_lin + ((size_t)b * EX_LIN + r) * HT_ROW;
				e[0] = (float)rb0; e[1] = (float)rb1; e[2] = p0.x; e[3] = p0.y; e[4] = p0.z; e[5] = p1.x; e[6] = p1.y; e[7] = p1.z; e[8] = n.x; e[9] = n.y; e[10] = n.z;
				e[11] = targetdist; e[12] = tsnb; e[13] = fmin_std(fmn, fmx); e[14] = fmax_std(fmn, fmx); e[15] = (meta & LM_FRIC) ? (float)-kk : 0.0f;      // friction_master (physics.h:477-478)
			}
		}
		// a side without a body (rb == NULL in the reference, physics.h:293-300): its lever arm is the anchor itself, it adds nothing to the effective mass and moves nothing
		const m3 Z = { V3(0, 0, 0), V3(0, 0, 0), V3(0, 0, 0) };
		const v3 r0 = rb0 >= 0 ? qrot(L4(S.q[rb0]), p0) : p0, r1 = rb1 >= 0 ? qrot(L4(S.q[rb1]), p1) : p1;
		const m3 I0 = rb0 >= 0 ? body_I(S, rb0) : Z, I1 = rb1 >= 0 ? body_I(S, rb1) : Z;
		const float impulsed = (rb0 >= 0 ? S.lin4[rb0].w + dot(cross(mul(I0, cross(r0, n)), r0), n) : 0.0f) + (rb1 >= 0 ? S.lin4[rb1].w + dot(cross(mul(I1, cross(r1, n)), r1), n) : 0.0f);      // physics.h:299-300
		const float ts = targetdist / dt;
		const v3 g0 = -cross(r0, n), g1 = cross(r1, n), b0 = mul(I0, g0), b1 = mul(I1, g1);          // rb0 receives -impulse and contributes -v0: its sign rides on g and b
		float *o = pool + g * LGRP;
		float *os = o + LG_S + 4 * kk;
		os[0] = ts; os[1] = fmin_std(ts, tsnb); os[2] = fmin_std(fmn, fmx) * dt; os[3] = fmax_std(fmn, fmx) * dt;
		if (meta & LM_FRIC) os[3] = fmax_std(rb0 >= 0 ? S.ang4[rb0].w : 0.0f, rb1 >= 0 ? S.ang4[rb1].w : 0.0f);       // mu of physics.h:292; the limits are formed from the normal row's impulse every sweep
		o[LG_RINV + kk] = 1.0f / impulsed; o[LG_SUM + kk] = 0.0f;
		if (kk == 0) o[LG_META] = __int_as_float(meta | (rb0 & 255) | ((rb1 & 255) << 8));
		o[LG_N + 3 * kk] = n.x; o[LG_N + 3 * kk + 1] = n.y; o[LG_N + 3 * kk + 2] = n.z;
		float *og = o + LG_GB + 12 * kk;
		og[0] = g0.x; og[1] = b0.x; og[2] = g0.y; og[3] = b0.y; og[4] = g0.z; og[5] = b0.z;
		og[6] = g1.x; og[7] = b1.x; og[8] = g1.y; og[9] = b1.y; og[10] = g1.z; og[11] = b1.z;
		if (kk == 0 && g < MAXG) { S.lrb[g][0] = (unsigned char)(rb0 >= 0 ? rb0 : 255); S.lrb[g][1] = (unsigned char)(rb1 >= 0 ? rb1 : 255); }
	}
	if (lane < LGRP) pool[ng2 * LGRP + lane] = (lane >= LG_RINV && lane < LG_RINV + 3) ? 1.0f : lane == LG_META ? __int_as_float(IDLE_BODY | (IDLE_BODY << 8)) : 0.0f;      // idle group: zero direction, zero limits
	if (lane == 0)
	{
		S.lin4[IDLE_BODY] = make_float4(0, 0, 0, 0); S.ang4[IDLE_BODY] = make_float4(0, 0, 0, 0);
		S.I4[IDLE_BODY][0] = S.I4[IDLE_BODY][1] = S.I4[IDLE_BODY][2] = make_float4(0, 0, 0, 0);
		S.lorder[S.LIDLE] = (unsigned)ng2 | ((unsigned)IDLE_BODY << 16) | ((unsigned)IDLE_BODY << 24);
	}
	__threadfence_block();
	__syncthreads();
	const long long t_m2b = HT_DBG(a.dbg, 2048) ? clock64() : 0;
	// ---- level schedule, once per solve.  The unit is a group: the 3 consecutive rows of a joint or of a contact (same two bodies, same lever
	//      arms), respectively a run of consecutive angular rows on the same body pair.  level(group) = 1 + max level of an earlier group sharing
	//      a body, so conflicting rows keep the reference's order; a group's own rows run back to back in one lane pair with the momenta in
	//      registers.  Groups are counting-sorted by level into steps of at most 8 groups (one per lane pair). ----
	if (na > MAXA_LDS) na = MAXA_LDS;
	// ---- which way the two-body rows go (round 5).  BLOCKED (ht_block.hpp): the model's own joint and contact triples (at most 4 blocks of ten) and up to 128 angular
	//      rows, resolved a block at a time in the reference's row order; everything else -- a caller's two-body linear rows, more rows than the blocks hold, an angular
	//      row that RemoveBias switches on (its gain differs between the sweeps before and after, and a block's couplings carry the gain) -- keeps the level schedule
	//      below.  The choice follows the frame's rows, never the build or the launch, so every build returns the same bits.
	bool blocked = false;
	if constexpr (!EXACT)
	{
		bool sw = false;
#pragma unroll
		for (int s = 0; s < ASLOTS; s++) sw = sw || (lane + 64 * s < na && AR[s].targetspin == -FLT_MAX && AR[s].mintorque < 0);
		blocked = !a.two_body_levels && ngt == 0 && n2 <= 4 * BLK_LROWS && na <= 128 && __ballot(sw) == 0ull;
		blocked = __builtin_amdgcn_readfirstlane((int)blocked) != 0;
		if (fast_pose) blocked = true;      // k_solve_prep looked at the angular rows (no row RemoveBias switches on, no more than the builds keep), the row count was checked at the top
	}
	int nga = 0;
	if (!blocked)
	for (int base = 0; base < na; base += 64)          // heads of the angular runs, found one row per lane
	{
		const int r = base + lane;
		const bool head = r < na && (r == 0 || S.arb[r][0] != S.arb[r - 1][0] || S.arb[r][1] != S.arb[r - 1][1]);
		const unsigned long long m = __ballot(head);
		const int k = nga + __popcll(m & ((1ull << lane) - 1ull));
		if (head && k <= MAXA_RUNS) S.gst[k] = (unsigned char)r;      // entry MAXA_RUNS, if there is one, is where the rows that are kept end
		nga += __popcll(m);
	}
	__syncthreads();
	if (nga > MAXA_RUNS)      // more runs than the schedule holds (only a caller's many short runs can do that): the rows from run MAXA_RUNS on are dropped, and reported
	{
		if (a.caps && lane == 0) atomicAdd(a.caps, 1);
		na = S.gst[MAXA_RUNS]; nga = MAXA_RUNS;
	}
	if (blocked) { if (lane == 0) { S.nlev_lin = 0; S.nlev_ang = 0; } }
	else
	{
		// two-body linear groups: lane g (and g + 64) speaks for group g
		int gb0[2], gb1[2];
		const int ngs = seq_lin ? 0 : ng2;
#pragma unroll
		for (int s = 0; s < 2; s++) { const int g = lane + 64 * s; gb0[s] = g < ngs ? S.lrb[g][0] : 255; gb1[s] = g < ngs ? S.lrb[g][1] : 255; }
		auto put_lorder = [&](int pos, int g, int s) { S.lorder[pos] = (unsigned)g | ((unsigned)(gb0[s] == 255 ? IDLE_BODY : gb0[s]) << 16) | ((unsigned)(gb1[s] == 255 ? IDLE_BODY : gb1[s]) << 24); };
		auto put_lstart = [&](int step, int v) { S.lstart[step] = (unsigned short)v; };
		int nl = ngs <= 64 ? level_schedule<1>(ngs, lane, gb0, gb1, put_lorder, put_lstart) : level_schedule<2>(ngs, lane, gb0, gb1, put_lorder, put_lstart);
		if (seq_lin) nl = ng2;      // step L applies group L - 1 (linear_phase's entry())
		// angular runs
		int ar[2], ac[2];
#pragma unroll
		for (int s = 0; s < 2; s++)
		{
			const int g = lane + 64 * s;
			ar[s] = g < nga ? (int)S.gst[g] : 0;
			ac[s] = (g + 1 < nga ? (int)S.gst[g + 1] : na) - ar[s];
			gb0[s] = g < nga ? S.arb[ar[s]][0] : 255; gb1[s] = g < nga ? S.arb[ar[s]][1] : 255;
		}
		auto put_aorder = [&](int pos, int g, int s) { S.aorder[pos] = (unsigned)ar[s] | ((unsigned)ac[s] << 8) | ((unsigned)(gb0[s] == 255 ? IDLE_BODY : gb0[s]) << 16) | ((unsigned)(gb1[s] == 255 ? IDLE_BODY : gb1[s]) << 24); };
		auto put_astart = [&](int step, int v) { S.astart[step] = (unsigned short)v; };
		const int nla = nga <= 64 ? level_schedule<1>(nga, lane, gb0, gb1, put_aorder, put_astart) : level_schedule<2>(nga, lane, gb0, gb1, put_aorder, put_astart);
		if (lane == 0)
		{
			S.nlev_lin = nl; S.nlev_ang = nla;
			S.aorder[MAXA2] = (unsigned)na | (1u << 8) | ((unsigned)IDLE_BODY << 16) | ((unsigned)IDLE_BODY << 24);
		}
	}
	__syncthreads();
	const int nlev_lin = S.nlev_lin, nlev_ang = S.nlev_ang;
	if (HT_DBG(a.dbg, 64)) return;
	const long long t_m3 = HT_DBG(a.dbg, 2048) ? clock64() : 0;

	// ---- single-body prefix: [landmark-ray / boundary-plane / caller's rows] then the cloud rows.  Every row is reduced to a 64-byte record (ht_quad.hpp) where
	//      its producer stands: a cloud row's record was written by k_cloud_rows at its point's index (a.cloud_body holds the rows' bodies), the others are
	//      written here, behind the cloud's slots.  A chain is a list of record indices in the reference's row order (stable partition by body). ----
	const int npre_g = a.rows_pre ? a.n_pre[b] : 0;
	const int npre = a.ray_rows ? S.nray : npre_g;
	const int ncl = (a.rows_cloud || a.cloud_body) ? a.n_cloud[b] : 0;
	const int n1 = npre + ncl;
	const int pre_base = M.pts_cap, noop_idx = rec_cap - 1;      // record indices: cloud row j -> j, other single-body row i -> pre_base + i, the record that changes nothing -> the last
	const int npad_max = 7 * (nb > 16 ? nb - 16 : 0);      // entries that may be added to pad host chains (below)
	// Two layouts of the chain lists (ht_quad.hpp): sixteen quads on sixteen bodies row by row, or the four quads of a DPP row on four consecutive rows of one body, four
	// bodies at a time (round 5).  The second one's phase takes (rows / 16) blocks where the first takes (longest chain) rows: CHAIN4 per frame (never per build) below.
	const int nlist = n1 + (npad_max > 3 * nb ? npad_max : 3 * nb) + (QUAD_CHAIN_SLACK > QUAD_BLOCK_SLACK ? QUAD_CHAIN_SLACK : QUAD_BLOCK_SLACK);      // chain entries incl. padding and read-ahead slack, either layout
	const bool sums_lds = nlist <= S.NSUM;
	const bool idx_lds = sums_lds && nlist <= S.NIDX && rec_cap <= 65536;
	float *const gsum = a.scratch + (size_t)a.batch * a.scratch_stride * CREC + (size_t)b * a.scratch_stride;      // this frame's sums in HBM, behind all frames' records
	unsigned *const gidx = reinterpret_cast<unsigned *>(a.scratch + (size_t)a.batch * a.scratch_stride * (CREC + 1)) + (size_t)b * a.scratch_stride;      // and its chain lists behind those
	float *const gG = a.scratch + (size_t)a.batch * a.scratch_stride * (CREC + 2) + (size_t)b * a.scratch_stride * 4;      // and, behind those, the couplings of its rows' blocks of four when the build's LDS has no room for them (QUAD_G_BLOCK floats per block; the region has 16 B per chain entry)
	if (sums_lds) { for (int i = lane; i < nlist; i += 64) S.csum[i] = 0.0f; }
	else for (int i = lane; i < nlist && i < a.scratch_stride; i += 64) gsum[i] = 0.0f;
	if (fast_chain) { if (idx_lds) for (int i = lane; i < nlist; i += 64) S.cidx[i] = (unsigned short)gidx[i]; }      // the lists as k_solve_prep placed them (in HBM for the builds that walk them there)
	else if (idx_lds) { for (int i = lane; i < nlist; i += 64) S.cidx[i] = (unsigned short)noop_idx; }
	else for (int i = lane; i < nlist && i < a.scratch_stride; i += 64) gidx[i] = (unsigned)noop_idx;
	if (lane == 0 && !fast_chain) quad_write_noop(scr + (size_t)noop_idx * CREC);
	auto pre_ptr = [&](int i) -> const float * { return a.ray_rows ? S.ray[i] : a.rows_pre + ((size_t)b * a.pre_stride + i) * HT_ROW; };
	auto body_of = [&](int i) -> int {
		if (i < npre) return (int)pre_ptr(i)[1];
		if (a.cloud_body) return (int)a.cloud_body[(size_t)b * M.pts_cap + (i - npre)];
		return (int)a.rows_cloud[((size_t)b * M.pts_cap + (i - npre)) * HT_ROW + 1];
	};
	__syncthreads();
	int mycnt = 0;                                     // lane bb counts the rows of body bb
	if (!fast_chain)
	for (int base = 0; base < n1; base += 64)          // pass A: rows per body
	{
		const int i = base + lane;
		int body = (i < n1) ? body_of(i) : -1;
		unsigned long long todo = __ballot(body >= 0);
		while (todo)
		{
			const int leader = __ffsll((long long)todo) - 1;      // wave-uniform (todo is a ballot): the body comes out of its lane into a scalar (v_readlane: no LDS round trip as with a shuffle)
			const int bb = __builtin_amdgcn_readlane(body, leader);
			const unsigned long long m = __ballot(body == bb);
			if (lane == bb) mycnt += __popcll(m);
			todo &= ~m;
		}
	}
	const long long t_l1 = HT_DBG(a.dbg, 262144) ? clock64() : 0;      // HT_DEBUG_SKIP += 262144 (with 2048): the chain lists' parts in the places of the prologue's first four
	// CHAIN4: the bodies' chains in blocks of four rows, dealt longest first to the wave's four DPP rows (each body to the row with the fewest blocks so far); a row's
	// segment of the lists = its bodies' blocks one after the other, a chain padded to whole blocks with the record that changes nothing
	const bool chain4 = !EXACT && !HT_DBG(a.dbg, 65536);      // HT_DEBUG_SKIP += 65536 (-DHT_TUNING): the row-by-row walk, for an A/B
	int c4_e0 = 0, c4_nblk = 0, c4_head = 0, c4_start = 0, c4_next = -1, c4_total = 0;
	int myblk = (lane < nb && !HT_DBG(a.dbg, 1)) ? (mycnt + 3) >> 2 : 0;
	if (fast_chain)
	{
		c4_e0 = t_e0; c4_nblk = t_nblk; c4_head = t_head; c4_total = t_total;
		if (lane < HT_MAXNB) { myblk = reinterpret_cast<const int *>(T + TB_CCNT)[lane]; c4_next = reinterpret_cast<const int *>(T + TB_CNEXT)[lane]; }
	}
	else if (chain4)
	{
		// everything below is the same on every lane: the per-body values are read out of their lanes into scalars (v_readlane with a scalar lane number), so the
		// dealing loop is scalar arithmetic
		int myrank = 0;
		for (int k = 0; k < nb; k++) { const int o = __builtin_amdgcn_readlane(myblk, k); myrank += (o > myblk || (o == myblk && k < lane)) ? 1 : 0; }
		if (lane < nb) S.cstart[myrank] = lane;      // the bodies in dealing order (the table takes the chains' starts further down)
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		__builtin_amdgcn_wave_barrier();
		const int sorted = lane < nb ? S.cstart[lane] : 0;
		int ld0 = 0, ld1 = 0, ld2 = 0, ld3 = 0, la0 = -1, la1 = -1, la2 = -1, la3 = -1, hd0 = 0, hd1 = 0, hd2 = 0, hd3 = 0, myrow = 0, mypos = 0;
		for (int r = 0; r < nb; r++)
		{
			const int bb = __builtin_amdgcn_readlane(sorted, r);
			const int blk = __builtin_amdgcn_readlane(myblk, bb);
			if (blk == 0) break;                       // sorted: the rest have no rows
			int row = 0, best = ld0;
			if (ld1 < best) { row = 1; best = ld1; }
			if (ld2 < best) { row = 2; best = ld2; }
			if (ld3 < best) { row = 3; best = ld3; }
			const int prev = row == 0 ? la0 : row == 1 ? la1 : row == 2 ? la2 : la3;
			if (lane == bb) { myrow = row; mypos = best; }
			if (lane == prev) c4_next = bb;
			if (prev < 0) { if (row == 0) hd0 = bb; else if (row == 1) hd1 = bb; else if (row == 2) hd2 = bb; else hd3 = bb; }
			if (row == 0) { la0 = bb; ld0 += blk; } else if (row == 1) { la1 = bb; ld1 += blk; } else if (row == 2) { la2 = bb; ld2 += blk; } else { la3 = bb; ld3 += blk; }
		}
		const int sg1 = 4 * ld0, sg2 = sg1 + 4 * ld1, sg3 = sg2 + 4 * ld2;
		c4_total = ld0 + ld1 + ld2 + ld3;
		c4_start = (myrow == 0 ? 0 : myrow == 1 ? sg1 : myrow == 2 ? sg2 : sg3) + 4 * mypos;
		const int R = lane >> 4;                       // this lane's DPP row in the sweeps
		c4_e0 = R == 0 ? 0 : R == 1 ? sg1 : R == 2 ? sg2 : sg3; c4_nblk = R == 0 ? ld0 : R == 1 ? ld1 : R == 2 ? ld2 : ld3; c4_head = R == 0 ? hd0 : R == 1 ? hd1 : R == 2 ? hd2 : hd3;
	}
	// Sixteen quads walk the chains.  A body beyond the 16th rides on the quad of one of the first sixteen: it goes to the one with the fewest rows that
	// does not host yet, its entries follow the host's in the list, and the host's entries are padded to a multiple of 8 with the record that changes nothing
	// (zero direction, zero limits: impulse 0), so that the quad changes body at a block boundary of the chain walk (quad_chain_run).
	int myextra = -1, myhost = -1;
	if (!fast_chain)
	{
		int avail = (lane < 16 && lane < nb) ? mycnt : 0x7fffff;
		for (int e = 16; e < nb; e++)
		{
			int key = (avail << 6) | lane;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_xor(key, o); key = v < key ? v : key; }
			const int h = key & 63;
			if (lane == h) { myextra = e; avail = 0x7fffff; }
			if (lane == e) myhost = h;
		}
	}
	const int extracnt = __shfl(mycnt, myextra >= 0 ? myextra : lane);
	const int mylen = lane >= 16 ? 0 : (myextra >= 0 && extracnt > 0 ? ((mycnt + 7) & ~7) + extracnt : mycnt);      // entries of this lane's slot of the list
	if (myextra >= 0 && extracnt == 0) myextra = -1;                                                                  // nothing to host
	int mystart = mylen;                               // exclusive prefix over lanes = segment start of body `lane`
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(mystart, o); if (lane >= o) mystart += v; }
	mystart -= mylen;
	{
		const int hs = __shfl(mystart, myhost >= 0 ? myhost : lane), hc = __shfl(mycnt, myhost >= 0 ? myhost : lane);
		if (myhost >= 0) mystart = hs + ((hc + 7) & ~7);      // an extra body's entries start behind its host's padded ones (which already name the no-op record)
	}
	if (chain4) { mystart = c4_start; if (lane < HT_MAXNB) { S.ccnt[lane] = myblk; S.cstart[lane] = mystart; S.cextra[lane] = (signed char)c4_next; } }      // ccnt: blocks, cextra: the next body of the DPP row
	else if (lane < HT_MAXNB) { S.ccnt[lane] = HT_DBG(a.dbg, 1) ? 0 : mycnt; S.cstart[lane] = mystart; S.cextra[lane] = (signed char)(HT_DBG(a.dbg, 1) ? -1 : myextra); }
	int myrun = 0;
	const long long t_l2 = HT_DBG(a.dbg, 262144) ? clock64() : 0;
	if (!fast_chain)
	for (int base = 0; base < n1; base += 64)          // pass B: placement in stable order; records of the rows that have none yet
	{
		const int i = base + lane;
		int body = (i < n1) ? body_of(i) : -1;
		int dst = -1;
		unsigned long long todo = __ballot(body >= 0);
		while (todo)
		{
			const int leader = __ffsll((long long)todo) - 1;
			const int bb = __builtin_amdgcn_readlane(body, leader);
			const unsigned long long m = __ballot(body == bb);
			const int segbase = __builtin_amdgcn_readlane(mystart + myrun, bb);
			if (body == bb) dst = segbase + __popcll(m & ((1ull << lane) - 1ull));
			if (lane == bb) myrun += __popcll(m);
			todo &= ~m;
		}
		if (body >= 0)
		{
			const bool made = i >= npre && a.cloud_body;                                  // k_cloud_rows wrote this row's record
			const int src = i < npre ? pre_base + i : i - npre;
			if (src < noop_idx && dst < nlist - QUAD_CHAIN_SLACK)
			{
				if (idx_lds) S.cidx[dst] = (unsigned short)src; else if (dst < a.scratch_stride) gidx[dst] = (unsigned)src;
				if (!made)
				{
					const float *r = i < npre ? pre_ptr(i) : a.rows_cloud + ((size_t)b * M.pts_cap + (i - npre)) * HT_ROW;
					const v3 p1 = L3(r + 5), n = L3(r + 8);
					const v3 r1 = qrot(L4(S.q[body]), p1);
					const m3 Ib = body_I(S, body);
					const float impulsed = S.lin4[body].w + dot(cross(mul(Ib, cross(r1, n)), r1), n);       // 0 + (...) for the NULL side
					const float ts = r[11] / dt;
					quad_write_record(scr + (size_t)src * CREC, r1, n, Ib, S.lin4[body].w, ts, fmin_std(ts, r[12]), impulsed, r[13] * dt, r[14] * dt);
				}
			}
		}
	}
	const long long t_l3 = HT_DBG(a.dbg, 262144) ? clock64() : 0;
	// the last use of the inverse inertias, which share LDS with the angular records: every row's Iinv*axis, while the table is still there
	v3 ABA0[ASLOTS], ABA1[ASLOTS];
#pragma unroll
	for (int s = 0; s < ASLOTS; s++)
	{
		const arow &R = AR[s];
		const bool on = lane + 64 * s < na && !fast_pose;
		ABA0[s] = (on && R.rb0 >= 0) ? -mul(body_I(S, R.rb0), R.axis) : V3(0, 0, 0);
		ABA1[s] = (on && R.rb1 >= 0) ? mul(body_I(S, R.rb1), R.axis) : V3(0, 0, 0);
	}
	__threadfence_block();      // the records and lists are read back by other lanes of this wave
	__syncthreads();
	const bool g_lds = idx_lds && S.NCG > 0 && QUAD_G_BLOCK * (c4_total + 4) <= S.NCG;      // the couplings in LDS (the walk reads four blocks ahead); only beside LDS chain lists: one instance of the walk less
	if (fast_chain)
	{
		if (g_lds) { for (int i = lane; i < QUAD_G_BLOCK * c4_total; i += 64) S.cg[i] = gG[i]; __threadfence_block(); __syncthreads(); }      // k_solve_prep left them in the frame's HBM slot
	}
	else if (chain4 && !HT_DBG(a.dbg, 131072))      // HT_DEBUG_SKIP += 131072 (-DHT_TUNING, wrong results): without this loop, to see what it costs
	{
		// the couplings of every block's rows with the rows before them: a quad per block, lane c of it takes slot c of the block's four records; -G(j,i) = -(c_j . d_i),
		// the three lanes' shares summed (p0 + p1) + p2 on lane 2, which writes the rows' entries
		const int quad_ = lane >> 2, c_ = lane & 3;
		// (the read-ahead slack behind the last block is loaded by the walk but never applied: its couplings are not written)
		for (int blk0 = quad_; blk0 < c4_total; blk0 += 64)      // four blocks per quad and trip: their sixteen record reads are in flight together (the records come from L2 or further)
		{
			float4 r_[4][4];
#pragma unroll
			for (int u = 0; u < 4; u++)
#pragma unroll
				for (int j = 0; j < 4; j++)
				{
					const int blk = blk0 + 16 * u, e = 4 * blk + j;
					const unsigned x = (blk < c4_total) ? (idx_lds ? (unsigned)S.cidx[e] : gidx[e]) : (unsigned)noop_idx;
					r_[u][j] = reinterpret_cast<const float4 *>(scr + (size_t)x * CREC)[c_];
				}
#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				const int blk = blk0 + 16 * u;
				auto coup = [&](int j, int i) -> float {
					const float p = __fmaf_rn(r_[u][j].z, r_[u][i].y, r_[u][j].x * r_[u][i].w);
					const float t = dpp<QP_PREV>(p) + p;
					return -(dpp<QP_PREV>(t) + p);
				};
				const float g10 = coup(1, 0), g20 = coup(2, 0), g21 = coup(2, 1), g30 = coup(3, 0), g31 = coup(3, 1), g32 = coup(3, 2);
				if (c_ == 2 && blk < c4_total && 4 * blk + 3 < a.scratch_stride)
				{
					float2 *o = reinterpret_cast<float2 *>((g_lds ? S.cg : gG) + QUAD_G_BLOCK * blk);      // row 3's three | row 2's two, 0 | row 1's, 0, 0 | 0 (the quad of row 0 reads the last three)
					o[0] = make_float2(g30, g31); o[1] = make_float2(g32, g20); o[2] = make_float2(g21, 0.0f); o[3] = make_float2(g10, 0.0f); o[4] = make_float2(0.0f, 0.0f);
				}
			}
		}
		__threadfence_block();
		__syncthreads();
	}
	const long long t_l4 = HT_DBG(a.dbg, 262144) ? clock64() : 0;
	v3 pos_next = V3(0, 0, 0); v4 q_next = V4(0, 0, 0, 1);
	auto calc_next_pose = [&]() {
		// rbcalcnextpose physics.h:522-531 with rkupdateq :211-218 (momentum-preserving RK4 on the quaternion)
		const float *bc = M.bodyc + lane * HT_BC;
		const float minv = S.lin4[lane].w;
		pos_next = L3(S.pos[lane]) + (F3(S.lin4[lane]) * minv) * dt;
		const m3 tinv = LM(bc + HT_BC_TINV) * minv;
		const v3 angm = F3(S.ang4[lane]);
		const v4 s = L4(S.q[lane]);
		auto diffq = [&](v4 o) -> v4 {
			v4 sn = normalize(o);
			m3 Mx = qmat(sn);
			m3 Ii = mul(Mx, mul(tinv, transpose(Mx)));
			v3 hs = mul(Ii, angm) * 0.5f;
			return qmul(V4(hs.x, hs.y, hs.z, 0), sn);
		};
		v4 d1 = diffq(s), d2 = diffq(s + d1 * (dt / 2)), d3 = diffq(s + d2 * (dt / 2)), d4 = diffq(s + d3 * dt);
		v4 o = normalize((((s + d1 * (dt / 6)) + d2 * (dt / 3)) + d3 * (dt / 3)) + d4 * (dt / 6));
		if (o.x < FLT_EPSILON / 4.0f && o.x > -FLT_EPSILON / 4.0f) o.x = 0.0f;
		if (o.y < FLT_EPSILON / 4.0f && o.y > -FLT_EPSILON / 4.0f) o.y = 0.0f;
		if (o.z < FLT_EPSILON / 4.0f && o.z > -FLT_EPSILON / 4.0f) o.z = 0.0f;
		q_next = o;
	};
	if constexpr (EXACT)
	{
		// ---- the reference's own sweeps (PhysicsUpdate physics.h:556-581), row by row on one lane.  Linears = [landmark-ray / boundary-plane rows][cloud rows]
		//      [joint rows][contact rows] (physmodel.h:348-350, physics.h:549-551), then the angular rows, all in the reference's order; a row's impulse sum /
		//      accumulated torque lives in the frame's sums array in HBM.
		const int nlin2 = n2 < EX_LIN ? n2 : EX_LIN;
		float *const sums = gsum;
		for (int i = lane; i < n1 + nlin2 + na && i < a.scratch_stride; i += 64) sums[i] = 0.0f;
		__threadfence_block();
		__syncthreads();
		auto sweeps = [&](int count, bool post) {
			auto linear = [&](int rb0, int rb1, v3 p0, v3 p1, v3 n, float ts, float fx, float fy, float *sum) {
				const m3 Z = { V3(0, 0, 0), V3(0, 0, 0), V3(0, 0, 0) };
				const v3 r0 = rb0 >= 0 ? qrot(L4(S.q[rb0]), p0) : p0, r1 = rb1 >= 0 ? qrot(L4(S.q[rb1]), p1) : p1;
				const v3 v0 = rb0 >= 0 ? cross(spin_of(S, rb0), r0) + F3(S.lin4[rb0]) * S.lin4[rb0].w : V3(0, 0, 0);
				const v3 v1 = rb1 >= 0 ? cross(spin_of(S, rb1), r1) + F3(S.lin4[rb1]) * S.lin4[rb1].w : V3(0, 0, 0);
				const float vn = dot(v1 - v0, n);
				const float impulsen = -ts - vn;
				const float impulsed = (rb0 >= 0 ? S.lin4[rb0].w + dot(cross(mul(body_I(S, rb0), cross(r0, n)), r0), n) : 0.0f)
				                     + (rb1 >= 0 ? S.lin4[rb1].w + dot(cross(mul(body_I(S, rb1), cross(r1, n)), r1), n) : 0.0f);
				float impulse = impulsen / impulsed;
				const float isum = *sum;
				impulse = fmin_std(fy * dt - isum, impulse);
				impulse = fmax_std(fx * dt - isum, impulse);
				if (rb0 >= 0)       // ApplyImpulse(rb0, r0, normal * -impulse) physics.h:222-226
				{
					const v3 im = n * -impulse;
					const v3 l = F3(S.lin4[rb0]) + im, av = F3(S.ang4[rb0]) + cross(r0, im);
					S.lin4[rb0].x = l.x; S.lin4[rb0].y = l.y; S.lin4[rb0].z = l.z; S.ang4[rb0].x = av.x; S.ang4[rb0].y = av.y; S.ang4[rb0].z = av.z;
				}
				if (rb1 >= 0)
				{
					const v3 im = n * impulse;
					const v3 l = F3(S.lin4[rb1]) + im, av = F3(S.ang4[rb1]) + cross(r1, im);
					S.lin4[rb1].x = l.x; S.lin4[rb1].y = l.y; S.lin4[rb1].z = l.z; S.ang4[rb1].x = av.x; S.ang4[rb1].y = av.y; S.ang4[rb1].z = av.z;
				}
				*sum = isum + impulse;
				(void)Z;
			};
			for (int sw = 0; sw < count; sw++)
			{
				for (int i = 0; i < n1; i++)
				{
					const float *r = i < npre ? pre_ptr(i) : a.rows_cloud + ((size_t)b * M.pts_cap + (i - npre)) * HT_ROW;
					const float ts = r[11] / dt;
					linear((int)r[0], (int)r[1], L3(r + 2), L3(r + 5), L3(r + 8), post ? fmin_std(ts, r[12]) : ts, r[13], r[14], sums + i);
				}
				for (int i = 0; i < nlin2; i++)
				{
					const float *r = a.exact_lin + ((size_t)b * EX_LIN + i) * HT_ROW;
					const int rb0 = (int)r[0], rb1 = (int)r[1], fm = (int)r[15];
					const float ts = r[11] / dt;
					float fx = r[13], fy = r[14];
					if (fm)      // physics.h:292: the friction rows' limits follow the normal row's impulse sum
					{
						fy = fmax_std(rb0 >= 0 ? S.ang4[rb0].w : 0.0f, rb1 >= 0 ? S.ang4[rb1].w : 0.0f) * sums[n1 + i + fm] / dt;
						fx = -fy;
					}
					linear(rb0, rb1, L3(r + 2), L3(r + 5), L3(r + 8), post ? fmin_std(ts, r[12]) : ts, fx, fy, sums + n1 + i);
				}
				for (int i = 0; i < na; i++)      // LimitAngular::Iter physics.h:251-265
				{
					const float *r = a.exact_ang + ((size_t)b * 256 + i) * 8;
					const int rb0 = __float_as_int(r[0]), rb1 = __float_as_int(r[1]);
					const v3 axis = L3(r + 2);
					const float mintorque = r[6], maxtorque = r[7];
					const float targetspin = post ? ((mintorque < 0) ? 0 : fmin_std(r[5], 0.0f)) : r[5];      // RemoveBias physics.h:250
					if (targetspin == -FLT_MAX) continue;
					const float currentspin = ((rb1 >= 0) ? dot(spin_of(S, rb1), axis) : 0.0f) - ((rb0 >= 0) ? dot(spin_of(S, rb0), axis) : 0.0f);
					const float dspin = targetspin - currentspin;
					const float spintotorque = 1.0f / (((rb0 >= 0) ? dot(axis, mul(body_I(S, rb0), axis)) : 0.0f) + ((rb1 >= 0) ? dot(axis, mul(body_I(S, rb1), axis)) : 0.0f));
					float dtorque = dspin * spintotorque;
					float *tq = sums + n1 + nlin2 + i;
					const float torque = *tq;
					dtorque = fmin_std(dtorque, maxtorque * dt - torque);
					dtorque = fmax_std(dtorque, mintorque * dt - torque);
					if (rb0 >= 0) { const v3 av = F3(S.ang4[rb0]) - axis * dtorque; S.ang4[rb0].x = av.x; S.ang4[rb0].y = av.y; S.ang4[rb0].z = av.z; }
					if (rb1 >= 0) { const v3 av = F3(S.ang4[rb1]) + axis * dtorque; S.ang4[rb1].x = av.x; S.ang4[rb1].y = av.y; S.ang4[rb1].z = av.z; }
					*tq = torque + dtorque;
				}
			}
		};
		if (lane == 0) sweeps(ph.iterations, false);
		__threadfence_block();
		__syncthreads();
		if (lane < nb) calc_next_pose();
		__syncthreads();
		if (lane == 0) sweeps(ph.iterations_post, true);
		__threadfence_block();
		__syncthreads();
	}
	else
	{
	// ---- the prologue scratch is dead now: angular rows move from their builder lanes into their records ----
	const bool arec_lds = na <= S.NANG;
	float *const arec = arec_lds ? S.arec : garec;
	if (fast_pose)      // the records, the idle record and the slack as k_solve_prep wrote them (the prologue scratch they share LDS with is dead: the contacts' groups are made)
	{
		const float4 *src = reinterpret_cast<const float4 *>(T + TB_AREC);
		for (int i = lane; i < (na + 4) * (AROW / 4); i += 64) reinterpret_cast<float4 *>(arec)[i] = src[i];
		const unsigned w = reinterpret_cast<const unsigned *>(T + TB_ABODY)[lane];
		S.blk.abody[2 * lane] = (unsigned short)(w & 0xFFFFu); S.blk.abody[2 * lane + 1] = (unsigned short)(w >> 16);
	}
	else
	{
#pragma unroll
	for (int s = 0; s < ASLOTS; s++)
	{
		const int r = lane + 64 * s;
		if (r < na)
		{
			const arow &R = AR[s];
			float *o = arec + r * AROW;
			const float ts_post = (R.mintorque < 0) ? 0 : fmin_std(R.targetspin, 0.0f);                      // RemoveBias physics.h:250
			o[AR_S] = R.targetspin; o[AR_S + 1] = ts_post; o[AR_S + 2] = R.mn; o[AR_S + 3] = R.mx;
			// a row whose target spin is -FLT_MAX is skipped by LimitAngular::Iter (physics.h:252): gain 0 = no torque; RemoveBias can turn it on (target 0)
			o[AR_GAIN] = R.targetspin == -FLT_MAX ? 0.0f : R.s2t; o[AR_TORQUE] = 0.0f;
			o[AR_AXIS] = R.axis.x; o[AR_AXIS + 1] = R.axis.y; o[AR_AXIS + 2] = R.axis.z; o[AR_AXIS + 3] = ts_post == -FLT_MAX ? 0.0f : R.s2t;
			const v3 ba0 = ABA0[s], ba1 = ABA1[s];
			o[AR_BA] = ba0.x; o[AR_BA + 1] = ba0.y; o[AR_BA + 2] = ba0.z; o[AR_BA + 3] = ba1.x; o[AR_BA + 4] = ba1.y; o[AR_BA + 5] = ba1.z;
		}
	}
	if (lane < 4 * AROW) arec[na * AROW + lane] = 0.0f;      // idle record + read-ahead slack
	if (blocked)
	{
#pragma unroll
		for (int s = 0; s < 2; s++)
		{
			const int r = lane + 64 * s;
			S.blk.abody[r] = (unsigned short)(r < na ? ((AR[s].rb0 >= 0 ? AR[s].rb0 : 255) | ((AR[s].rb1 >= 0 ? AR[s].rb1 : 255) << 8)) : 0xFFFF);
		}
	}
	}      // !fast_pose
	__threadfence_block();
	__syncthreads();

	// ---- two-body rows in blocks (ht_block.hpp), once per solve: every row's couplings to the rows before it in its block, and every block's edges sorted by body ----
	float GL[32], GA[32];                                  // coupling registers of the linear and of the angular rows: lane m holds -k_j c_j . D_i of its forward row (block 2h, row m) with row i in register i < m, of its backward row (block 2h + 1, row 31 - m) with row i in register 31 - i > m
	unsigned emL0 = 0, emL1 = 0, emL2 = 0, emL3 = 0, emA0 = 0, emA1 = 0, emA2 = 0, emA3 = 0;      // edge words of the blocks (ht_block.hpp)
	unsigned lbod = 0xFFFFFFFFu, abod = 0xFFFFFFFFu;      // body pairs (rb0 | rb1 << 8, 255 = none) of this lane's linear / angular rows: forward block's row in the low half, backward block's in the high half
#pragma unroll
	for (int i = 0; i < 32; i++) { GL[i] = 0.0f; GA[i] = 0.0f; }
	const int nbl = blocked ? (n2 + BLK_LROWS - 1) / BLK_LROWS : 0, nba = blocked ? (na + 31) >> 5 : 0;
	long long t_c0 = 0, t_c1 = 0, t_c2 = 0;
	if (blocked)
	{
		if (HT_DBG(a.dbg, 2048)) t_c0 = clock64();
		const int m = lane & 31, hh = lane >> 5;
		{
			auto lbodies = [&](int Q) -> unsigned {
				const int p = (Q & 1) ? 31 - m : m, row = BLK_LROWS * Q + p;
				const bool on = p < BLK_LROWS && row < n2;
				return on ? ((unsigned)__float_as_int(pool[(row / 3) * LGRP + LG_META]) & 0xFFFFu) : 0xFFFFu;
			};
			auto abodies = [&](int Q) -> unsigned { const int row = 32 * Q + ((Q & 1) ? 31 - m : m); return row < na ? (unsigned)S.blk.abody[row] : 0xFFFFu; };
			lbod = lbodies(2 * hh) | (lbodies(2 * hh + 1) << 16);
			abod = abodies(2 * hh) | (abodies(2 * hh + 1) << 16);
		}
		// Both loops go over the coupling registers rho = 0..31: a lane's forward row (position m of block 2h) takes register rho < m for the row at position rho, its backward
		// row (position 31 - m of block 2h + 1) takes register rho > m for the row at position 31 - rho.  The partner's record is read by every lane of the half-wave (one address
		// per half: a broadcast), eight (four) partners ahead of their use; nothing in the loop is conditional but the final select, so the reads overlap the arithmetic.
		// Angular rows: row r of the list sits in block r / 32; w_j = ba0_j . L(rb0) + ba1_j . L(rb1), a unit torque of row i adds -axis_i to L(rb0_i) and +axis_i to L(rb1_i):
		//   G(j,i) = -gain_j (ba0_j . axis_i ([rb0_j = rb1_i] - [rb0_j = rb0_i]) + ba1_j . axis_i ([rb1_j = rb1_i] - [rb1_j = rb0_i]))      (a missing body has ba = 0)
		auto ang_couplings = [&](const auto arec_, const bool bwd) {
			const int r = bwd ? 64 * hh + 63 - m : 64 * hh + m;
			const float *R = arec_ + (r < na ? r : na) * AROW;
			const float ng = -R[AR_GAIN];
			const v3 b0 = L3(R + AR_BA) * ng, b1 = L3(R + AR_BA + 3) * ng;
			const int bo = (int)S.blk.abody[r], a0 = bo & 255, a1 = bo >> 8;
#pragma unroll
			for (int c8 = 0; c8 < 32; c8 += 8)
			{
				float ax[8], ay[8], az[8]; int pb[8];
#pragma unroll
				for (int u = 0; u < 8; u++)
				{
					const int rp = bwd ? 64 * hh + 63 - (c8 + u) : 64 * hh + c8 + u;
					const float *P = arec_ + (rp < na ? rp : na) * AROW + AR_AXIS;
					ax[u] = P[0]; ay[u] = P[1]; az[u] = P[2]; pb[u] = (int)S.blk.abody[rp];
				}
#pragma unroll
				for (int u = 0; u < 8; u++)
				{
					const int rho = c8 + u, p0 = pb[u] & 255, p1 = pb[u] >> 8;
					const float t0 = (b0.x * ax[u] + b0.y * ay[u]) + b0.z * az[u], t1 = (b1.x * ax[u] + b1.y * ay[u]) + b1.z * az[u];
					float g = (a0 == p1 ? t0 : a0 == p0 ? -t0 : 0.0f) + (a1 == p1 ? t1 : a1 == p0 ? -t1 : 0.0f);
					asm volatile("" : "+v"(g));      // computed by every lane: the select below must not become a branch around the arithmetic (and its reads)
					GA[rho] = (bwd ? rho > m : rho < m) ? g : GA[rho];
				}
			}
		};
		if (fast_pose)      // the angular blocks' couplings, and the linear blocks' couplings among the joints' rows, as k_solve_prep left them (register 4k + c of lane l at (k * 64 + l) * 4 + c)
		{
#pragma unroll
			for (int k = 0; k < 8; k++)
			{
				const float4 ga = reinterpret_cast<const float4 *>(T + TB_GA)[k * 64 + lane], gl = reinterpret_cast<const float4 *>(T + TB_GL)[k * 64 + lane];
				GA[4 * k] = ga.x; GA[4 * k + 1] = ga.y; GA[4 * k + 2] = ga.z; GA[4 * k + 3] = ga.w;
				GL[4 * k] = gl.x; GL[4 * k + 1] = gl.y; GL[4 * k + 2] = gl.z; GL[4 * k + 3] = gl.w;
			}
		}
		else if (arec_lds) { ang_couplings(S.arec, false); if (nba > 1) ang_couplings(S.arec, true); }
		else { ang_couplings(garec, false); if (nba > 1) ang_couplings(garec, true); }
		// Linear rows: row r of the joint and contact triples sits in block r / 30; w_j = b0_j . L(rb0) - n_j minv0 . P(rb0) + b1_j . L(rb1) + n_j minv1 . P(rb1), a unit impulse
		// of row i adds -n_i to P(rb0_i), g0_i to L(rb0_i), n_i to P(rb1_i), g1_i to L(rb1_i) (the sides' signs ride on g and b: the group record's layout):
		//   G(j,i) = -rinv_j (b0_j . ([rb0_j = rb0_i] g0_i + [rb0_j = rb1_i] g1_i) + b1_j . ([rb1_j = rb0_i] g0_i + [rb1_j = rb1_i] g1_i)
		//                     + n_j . n_i (minv0 ([rb0_j = rb0_i] - [rb0_j = rb1_i]) + minv1 ([rb1_j = rb1_i] - [rb1_j = rb0_i])))
		auto lin_couplings = [&](const auto pool_, const bool bwd) {
			const int blk = 2 * hh + (bwd ? 1 : 0), pos = bwd ? 31 - m : m, row = BLK_LROWS * blk + pos;
			const bool on = pos < BLK_LROWS && row < n2;
			const int g_ = on ? row / 3 : ng2, k_ = on ? row - 3 * g_ : 0;
			const float *R = pool_ + g_ * LGRP;
			const float nr = on ? -R[LG_RINV + k_] : 0.0f;
			const v3 n = L3(R + LG_N + 3 * k_) * nr;
			const float *og = R + LG_GB + 12 * k_;
			const v3 b0 = V3(og[1], og[3], og[5]) * nr, b1 = V3(og[7], og[9], og[11]) * nr;
			const unsigned bo = bwd ? lbod >> 16 : lbod & 0xFFFFu;
			const int a0 = (int)(bo & 255u), a1 = (int)(bo >> 8);
			const float ma = S.lin4[a0 < HT_MAXNB ? a0 : IDLE_BODY].w, mb = S.lin4[a1 < HT_MAXNB ? a1 : IDLE_BODY].w;
#pragma unroll
			for (int c4 = 0; c4 < 32; c4 += 4)
			{
				float pn[4][3], q0[4][3], q1[4][3]; int pb[4];
#pragma unroll
				for (int u = 0; u < 4; u++)
				{
					const int ppos = bwd ? 31 - (c4 + u) : c4 + u, prow = BLK_LROWS * blk + ppos;       // static position: the group and the row within it are compile-time numbers but for the block
					const bool pon = ppos >= 0 && ppos < BLK_LROWS && prow < n2;
					const int pg = pon ? 10 * blk + ppos / 3 : ng2, pk = pon ? ppos % 3 : 0;
					const float *P = pool_ + pg * LGRP;
					const float *pgb = P + LG_GB + 12 * pk;
					pn[u][0] = P[LG_N + 3 * pk]; pn[u][1] = P[LG_N + 3 * pk + 1]; pn[u][2] = P[LG_N + 3 * pk + 2];
					q0[u][0] = pgb[0]; q0[u][1] = pgb[2]; q0[u][2] = pgb[4]; q1[u][0] = pgb[6]; q1[u][1] = pgb[8]; q1[u][2] = pgb[10];
					pb[u] = __float_as_int(P[LG_META]);
				}
#pragma unroll
				for (int u = 0; u < 4; u++)
				{
					const int rho = c4 + u, p0 = pb[u] & 255, p1 = (pb[u] >> 8) & 255;
					const bool daa = a0 == p0, dab = a0 == p1, dba = a1 == p0, dbb = a1 == p1;
					const float t00 = (b0.x * q0[u][0] + b0.y * q0[u][1]) + b0.z * q0[u][2], t01 = (b0.x * q1[u][0] + b0.y * q1[u][1]) + b0.z * q1[u][2];
					const float t10 = (b1.x * q0[u][0] + b1.y * q0[u][1]) + b1.z * q0[u][2], t11 = (b1.x * q1[u][0] + b1.y * q1[u][1]) + b1.z * q1[u][2];
					const float nn = (n.x * pn[u][0] + n.y * pn[u][1]) + n.z * pn[u][2];
					const float mm = (daa ? ma : dab ? -ma : 0.0f) + (dbb ? mb : dba ? -mb : 0.0f);
					float g = ((daa ? t00 : dab ? t01 : 0.0f) + (dba ? t10 : dbb ? t11 : 0.0f)) + nn * mm;
					asm volatile("" : "+v"(g));
					GL[rho] = (bwd ? rho > m : rho < m) ? g : GL[rho];
				}
			}
		};
		// With tables only the CONTACTS' rows are left: a contact row couples to the rows before it (joints among them), never a joint row to a contact.  A call serves the lower
		// half-wave's block and the upper's in lockstep, so it runs when either holds a contact row; a joint row's lane then forms the values it already has once more.
		const int njr = 3 * njg;      // the joints' rows lead the list
		auto has_contacts = [&](int Q) -> bool { return n2 > njr && BLK_LROWS * Q < n2 && BLK_LROWS * (Q + 1) > njr; };
		const bool fwd = !fast_pose || has_contacts(0) || has_contacts(2), bwd = !fast_pose || has_contacts(1) || has_contacts(3);
		if (pool_lds) { if (nbl > 0 && fwd) lin_couplings(S.pool, false); if (nbl > 1 && bwd) lin_couplings(S.pool, true); }
		else { if (nbl > 0 && fwd) lin_couplings(gpool, false); if (nbl > 1 && bwd) lin_couplings(gpool, true); }
		if (HT_DBG(a.dbg, 2048)) t_c1 = clock64();
		// edge words: a block's (row, side) pairs sorted by body, one per lane (ht_block.hpp)
		auto edge_word = [&](bool valid, int ba, int bb) -> unsigned {
			const int ka = (valid && ba < nb) ? ba : 255, kb = (valid && bb < nb) ? bb : 255;
			const unsigned long long lt = (1ull << lane) - 1ull;
			int r0 = -1, r1 = -1, base = 0;
			// the bodies the block touches, in ascending order (a wave-wide OR of the lanes' body bits, then one round per set bit)
			unsigned pm = (ka < 32 ? 1u << ka : 0u) | (kb < 32 ? 1u << kb : 0u);
			pm |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)pm, 0x111, 0xF, 0xF, false); pm |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)pm, 0x112, 0xF, 0xF, false);
			pm |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)pm, 0x114, 0xF, 0xF, false); pm |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)pm, 0x118, 0xF, 0xF, false);
			pm |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)pm, 0x142, 0xA, 0xF, false); pm |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)pm, 0x143, 0xC, 0xF, false);
			pm = (unsigned)__builtin_amdgcn_readlane((int)pm, 63);
			while (pm)
			{
				const int k = __ffs((int)pm) - 1;
				pm &= pm - 1u;
				const unsigned long long m0 = __ballot(ka == k), m1 = __ballot(kb == k);
				const int below = __popcll((m0 | m1) & lt);
				if (ka == k) r0 = base + below;
				if (kb == k) r1 = base + below;
				base += __popcll(m0) + __popcll(m1);
			}
			__syncthreads();
			S.blk.etmp[lane] = 0u;
			__threadfence_block();
			__syncthreads();
			if (r0 >= 0) S.blk.etmp[r0] = (unsigned)lane | BLK_E_VALID | ((unsigned)ka << 16);
			if (r1 >= 0) S.blk.etmp[r1] = (unsigned)lane | BLK_E_SIDE | BLK_E_VALID | ((unsigned)kb << 16);
			__threadfence_block();
			__syncthreads();
			const unsigned e = S.blk.etmp[lane];
			const bool v = (e & BLK_E_VALID) != 0;
			auto same = [&](int other) -> bool { const unsigned o = S.blk.etmp[other & 63]; return v && other >= 0 && other < 64 && (o & BLK_E_VALID) != 0 && (o >> 16) == (e >> 16); };
			const int rs = lane & ~15;
			unsigned w = e;
			if (v && !same(lane + 1)) w |= BLK_E_TAIL;
			if (lane - 1 >= rs && same(lane - 1)) w |= 1u << 9;
			if (lane - 2 >= rs && same(lane - 2)) w |= 1u << 10;
			if (lane - 4 >= rs && same(lane - 4)) w |= 1u << 11;
			if (lane - 8 >= rs && same(lane - 8)) w |= 1u << 12;
			if ((lane & 16) && same(rs - 1)) w |= 1u << 13;      // rows 1 and 3 of the wave: row_bcast15 brings the lane before the row
			if (lane >= 32 && same(31)) w |= 1u << 14;           // rows 2 and 3: row_bcast31 brings lane 31
			return w;
		};
		auto lin_edges = [&](int Q) -> unsigned {
			const int p = (Q & 1) ? 31 - m : m, row = BLK_LROWS * Q + p;
			const bool on = hh == (Q >> 1) && p < BLK_LROWS && row < n2;
			const unsigned bo = (Q & 1) ? lbod >> 16 : lbod & 0xFFFFu;
			return edge_word(on, (int)(bo & 255u), (int)(bo >> 8));
		};
		auto ang_edges = [&](int Q) -> unsigned {
			const int p = (Q & 1) ? 31 - m : m, row = 32 * Q + p;
			const bool on = hh == (Q >> 1) && row < na;
			const unsigned bo = (Q & 1) ? abod >> 16 : abod & 0xFFFFu;
			return edge_word(on, (int)(bo & 255u), (int)(bo >> 8));
		};
		// a block without a contact row has the edges k_solve_prep sorted for it; the angular blocks all have
		auto tab_edges = [&](int off, int Q) -> unsigned { return reinterpret_cast<const unsigned *>(T + off)[Q * 64 + lane]; };
		if (nbl > 0) emL0 = fast_pose && !has_contacts(0) ? tab_edges(TB_EML, 0) : lin_edges(0);
		if (nbl > 1) emL1 = fast_pose && !has_contacts(1) ? tab_edges(TB_EML, 1) : lin_edges(1);
		if (nbl > 2) emL2 = fast_pose && !has_contacts(2) ? tab_edges(TB_EML, 2) : lin_edges(2);
		if (nbl > 3) emL3 = fast_pose && !has_contacts(3) ? tab_edges(TB_EML, 3) : lin_edges(3);
		if (nba > 0) emA0 = fast_pose ? tab_edges(TB_EMA, 0) : ang_edges(0);
		if (nba > 1) emA1 = fast_pose ? tab_edges(TB_EMA, 1) : ang_edges(1);
		if (nba > 2) emA2 = fast_pose ? tab_edges(TB_EMA, 2) : ang_edges(2);
		if (nba > 3) emA3 = fast_pose ? tab_edges(TB_EMA, 3) : ang_edges(3);
		__syncthreads();
		if (HT_DBG(a.dbg, 2048)) t_c2 = clock64();
	}

	if (HT_DBG(a.dbg, 128)) return;
	// ---- Gauss-Seidel sweeps ----
	const bool stats = HT_DBG(a.dbg, 2048) != 0;          // timing experiments: per-frame cycle counts accumulated in the last scratch record
	long long cyc_chain = 0, cyc_lin = 0, cyc_ang = 0, t_mark = stats ? clock64() : 0;
	long long cyc_bh = 0, cyc_br = 0, cyc_bg = 0;      // blocked phases: head (loads, velocity terms), resolve, gather (impulses to the momenta)
	const long long t_begin = t_mark;
	const int total_sweeps = ph.iterations + ph.iterations_post;
	const float inv_dt = 1.0f / dt;
	const int quad = lane >> 2, c = lane & 3, cc = c < 3 ? c : 2;         // lane 3 of a quad shadows component z; its vector results are never stored
	const int c1 = (cc + 1) % 3, c2 = (cc + 2) % 3;
	const int side = quad & 1;                                            // two-body rows: even quad = rb0, odd quad = rb1
	const int sidesign = side ? 0 : (int)0x80000000;                      // rb0 receives -impulse, and contributes -v0 to v1 - v0
	float *const lin_w = reinterpret_cast<float *>(S.lin4), *const ang_w = reinterpret_cast<float *>(S.ang4);
	const int pslot = lane >> 3;
	const int ls_lin = S.lstart[lane], ls_ang = S.astart[lane];      // step boundaries of the first 63 steps, read back with v_readlane
	// ---- the two phases of the two-body tail, written once and instantiated for records in LDS (the frame fits the build) and in HBM (it does not) ----
	// (both phases are compiled once for the sweeps before RemoveBias and once for those after it, like the chains: which target speed a row uses -- and,
	// for an angular row, whether it is disabled -- is then a compile-time choice)
	auto linear_phase = [&](const auto pool_, const auto post_c) {
		constexpr bool post = decltype(post_c)::value;
		struct lset { unsigned e; int meta; float n0, n1, n2, g0, g1, g2, b0, b1, b2, minv; float4 s0, s1, s2; float q0, q1, q2, i0, i1, i2; };
		auto entry = [&](int L) -> unsigned {
			if (L > nlev_lin) return S.lorder[S.LIDLE];
			if (seq_lin)      // no schedule: the first lane pair takes group L - 1, its bodies from the group's own record; the other pairs idle
			{
				const int meta = __float_as_int(pool_[(L - 1) * LGRP + LG_META]);
				const unsigned r0 = (unsigned)meta & 255u, r1 = ((unsigned)meta >> 8) & 255u;
				const unsigned e = (unsigned)(L - 1) | ((r0 == 255u ? (unsigned)IDLE_BODY : r0) << 16) | ((r1 == 255u ? (unsigned)IDLE_BODY : r1) << 24);
				return pslot == 0 ? e : S.lorder[S.LIDLE];
			}
			int lo, hi;
			if (L < 63) { lo = __builtin_amdgcn_readlane(ls_lin, L); hi = __builtin_amdgcn_readlane(ls_lin, L + 1); }
			else { lo = S.lstart[L]; hi = S.lstart[L + 1]; }
			const int idx = lo + pslot;
			return S.lorder[idx < hi ? idx : S.LIDLE];
		};
		auto fetch = [&](lset &r, unsigned e) {
			r.e = e;
			const float *R = pool_ + (int)(e & 0xFFFF) * LGRP;
			const int body = side ? (int)(e >> 24) : (int)((e >> 16) & 255);
			r.s0 = *reinterpret_cast<const float4 *>(R + LG_S); r.s1 = *reinterpret_cast<const float4 *>(R + LG_S + 4); r.s2 = *reinterpret_cast<const float4 *>(R + LG_S + 8);
			const float4 qm = *reinterpret_cast<const float4 *>(R + LG_RINV), is = *reinterpret_cast<const float4 *>(R + LG_SUM);
			r.q0 = qm.x; r.q1 = qm.y; r.q2 = qm.z; r.meta = __float_as_int(qm.w); r.i0 = is.x; r.i1 = is.y; r.i2 = is.z;
			r.n0 = __int_as_float(__float_as_int(R[LG_N + cc]) ^ sidesign); r.n1 = __int_as_float(__float_as_int(R[LG_N + 3 + cc]) ^ sidesign); r.n2 = __int_as_float(__float_as_int(R[LG_N + 6 + cc]) ^ sidesign);      // rb0: -n, rb1: n
			const float *G = R + LG_GB + (3 * side + cc) * 2;
			const float2 gb0 = *reinterpret_cast<const float2 *>(G), gb1 = *reinterpret_cast<const float2 *>(G + 12), gb2 = *reinterpret_cast<const float2 *>(G + 24);
			r.g0 = gb0.x; r.b0 = gb0.y; r.g1 = gb1.x; r.b1 = gb1.y; r.g2 = gb2.x; r.b2 = gb2.y;
			r.minv = lin_w[4 * body + 3];
		};
		// the momenta of a step's bodies are read FIRST (behind the previous step's stores), the next step's record after them: LDS answers a wave in
		// order, so a momenta read queued behind a dozen record reads would wait for all of them
		auto momenta = [&](const lset &r, float &l, float &av) {
			const int body = side ? (int)(r.e >> 24) : (int)((r.e >> 16) & 255);
			l = lin_w[4 * body + c]; av = ang_w[4 * body + c];
		};
		auto step = [&](const lset &r, float l, float av) {
			const int body = side ? (int)(r.e >> 24) : (int)((r.e >> 16) & 255);
			auto row = [&](float n, float g, float bq, float ts, float fmn, float fmx, float rinv, float isum) -> float {
				const float p = __fmaf_rn(bq, av, (n * r.minv) * l);                                      // this side's share of vn, component c
				const float sp = (dpp<QP_BC0>(p) + dpp<QP_BC1>(p)) + dpp<QP_BC2>(p);
				const float vn = sp + pair_other_uniform(sp);                                              // v1.n - v0.n
				float impulse = (-ts - vn) * rinv;
				impulse = clamp_med3(impulse, fmn - isum, fmx - isum);
				l = __fmaf_rn(n, impulse, l);
				av = __fmaf_rn(g, impulse, av);
				return isum + impulse;
			};
			const float ns0 = row(r.n0, r.g0, r.b0, post ? r.s0.y : r.s0.x, r.s0.z, r.s0.w, r.q0, r.i0);
			float f1n = r.s1.z, f1x = r.s1.w, f2n = r.s2.z, f2x = r.s2.w;
			if (r.meta & LM_NORMAL)       // a contact: the friction rows are limited by the normal row's impulse sum (physics.h:292); their fmax slot holds mu
			{
				const float lim1 = (f1x * ns0) * inv_dt; f1x = lim1 * dt; f1n = (-lim1) * dt;
				const float lim2 = (f2x * ns0) * inv_dt; f2x = lim2 * dt; f2n = (-lim2) * dt;
			}
			const float ns1 = row(r.n1, r.g1, r.b1, post ? r.s1.y : r.s1.x, f1n, f1x, r.q1, r.i1);
			const float ns2 = row(r.n2, r.g2, r.b2, post ? r.s2.y : r.s2.x, f2n, f2x, r.q2, r.i2);
			if (c < 3) { if (body != IDLE_BODY) { lin_w[4 * body + c] = l; ang_w[4 * body + c] = av; } }      // a side without a body moves nothing (and the idle body stays at rest)
			else if (side == 0)
			{
				float *R = pool_ + (int)(r.e & 0xFFFF) * LGRP + LG_SUM;
				R[0] = ns0; R[1] = ns1; R[2] = ns2;
			}
		};
		lset A, Bs;
		fetch(A, entry(1));
		unsigned e_b = entry(2), e_a;
		float l, av;
		for (int L = 1; L <= nlev_lin; L += 2)
		{
			momenta(A, l, av);
			__builtin_amdgcn_sched_barrier(0);
			fetch(Bs, e_b); e_a = entry(L + 2);
			__builtin_amdgcn_sched_barrier(0);
			step(A, l, av);
			if (L + 1 > nlev_lin) break;
			__builtin_amdgcn_wave_barrier();
			momenta(Bs, l, av);
			__builtin_amdgcn_sched_barrier(0);
			fetch(A, e_a); e_b = entry(L + 3);
			__builtin_amdgcn_sched_barrier(0);
			step(Bs, l, av);
			__builtin_amdgcn_wave_barrier();
		}
	};
	auto angular_phase = [&](const auto arec_, const auto post_c) {
		constexpr bool POST = decltype(post_c)::value;
		struct aset { unsigned e; float ax, ba, ts, mn, mx, gain, torque; };
		auto entry = [&](int L) -> unsigned {
			if (L > nlev_ang) return S.aorder[MAXA2];
			int lo, hi;
			if (L < 63) { lo = __builtin_amdgcn_readlane(ls_ang, L); hi = __builtin_amdgcn_readlane(ls_ang, L + 1); }
			else { lo = S.astart[L]; hi = S.astart[L + 1]; }
			const int idx = lo + pslot;
			return S.aorder[idx < hi ? idx : MAXA2];
		};
		auto load = [&](aset &r, const float *R) {
			const float4 sv = *reinterpret_cast<const float4 *>(R + AR_S); const float2 gt = *reinterpret_cast<const float2 *>(R + AR_GAIN);
			r.ts = POST ? sv.y : sv.x; r.mn = sv.z; r.mx = sv.w; r.gain = POST ? R[AR_AXIS + 3] : gt.x; r.torque = gt.y;
			r.ax = R[AR_AXIS + cc]; r.ba = R[AR_BA + 3 * side + cc];
		};
		auto fetch = [&](aset &r, unsigned e) { r.e = e; load(r, arec_ + (int)(e & 0xFF) * AROW); };
		auto momentum = [&](const aset &r) -> float { const int body = side ? (int)(r.e >> 24) : (int)((r.e >> 16) & 255); return ang_w[4 * body + c]; };
		// one row of a run: R = its record (the accumulated torque is written back)
		auto apply = [&](const aset &r, float *R, float &av) {
			const float axs = __int_as_float(__float_as_int(r.ax) ^ sidesign);                            // rb0: -axis, rb1: axis
			const float gain = r.gain;                                                                     // 0 for a disabled row (physics.h:252), decided when the record was written
			const float p = r.ba * av;
			const float sp = (dpp<QP_BC0>(p) + dpp<QP_BC1>(p)) + dpp<QP_BC2>(p);                           // this side's signed spin about the axis
			const float currentspin = sp + pair_other_uniform(sp);                                         // spin1 - spin0
			float dtorque = (r.ts - currentspin) * gain;
			dtorque = clamp_med3(dtorque, r.mn - r.torque, r.mx - r.torque);
			av = __fmaf_rn(axs, dtorque, av);                                                              // rb0: a - axis*dtorque, rb1: a + axis*dtorque
			R[AR_TORQUE] = r.torque + dtorque;                                                             // the same value from every lane of the pair
		};
		// a run: the records of its rows rotate through three register sets, each read two rows ahead of its use (a copy from a "next" set into the
		// current one would wait for the read at the end of every row); reads run up to two records past the run
		auto step = [&](const aset &r0, float av) {
			const int body = side ? (int)(r0.e >> 24) : (int)((r0.e >> 16) & 255);
			const int cnt = HT_DBG(a.dbg, 16384) ? 1 : (int)((r0.e >> 8) & 255);      // timing experiment (-DHT_TUNING, wrong results): only the first row of every run -- what a level costs without its rows
			const bool bv = body != IDLE_BODY;       // a missing body: its ba is an exact zero, so it contributes exactly 0 and its (zero) momenta are never stored
			float *R = arec_ + (int)(r0.e & 0xFF) * AROW;
			aset ra = r0, rb, rc;
			load(rb, R + AROW);
			for (int k = 0;;)
			{
				load(rc, R + 2 * AROW); apply(ra, R, av); if (++k >= cnt) break;
				load(ra, R + 3 * AROW); apply(rb, R + AROW, av); if (++k >= cnt) break;
				load(rb, R + 4 * AROW); apply(rc, R + 2 * AROW, av); if (++k >= cnt) break;
				R += 3 * AROW;
			}
			if (c < 3 && bv) ang_w[4 * body + c] = av;
		};
		aset A, Bs;
		fetch(A, entry(1));
		unsigned e_b = entry(2), e_a;
		for (int L = 1; L <= nlev_ang; L += 2)
		{
			float av = momentum(A);
			__builtin_amdgcn_sched_barrier(0);
			fetch(Bs, e_b); e_a = entry(L + 2);
			__builtin_amdgcn_sched_barrier(0);
			step(A, av);
			if (L + 1 > nlev_ang) break;
			__builtin_amdgcn_wave_barrier();
			av = momentum(Bs);
			__builtin_amdgcn_sched_barrier(0);
			fetch(A, e_a); e_b = entry(L + 3);
			__builtin_amdgcn_sched_barrier(0);
			step(Bs, av);
			__builtin_amdgcn_wave_barrier();
		}
	};
	// ---- the same two phases a block at a time (ht_block.hpp): every lane of a block's half-wave holds one row.  Per block: the rows' velocity terms against the momenta as
	//      they stand (LDS), the resolve in row order, the impulse sums / torques stored, the momenta brought up to date through the block's sorted edges ----
	const int idt_bits = __builtin_amdgcn_readfirstlane(__float_as_int(inv_dt)), dt_bits = __builtin_amdgcn_readfirstlane(__float_as_int(dt));
	auto blocked_linear = [&](const auto pool_, const bool post) {
		for (int Q = 0; Q < nbl; Q++)
		{
			const long long tb0 = stats ? clock64() : 0;
			const int m = lane & 31;
			const int p = (Q & 1) ? 31 - m : m, row = BLK_LROWS * Q + p;
			const bool act = (lane >> 5) == (Q >> 1) && p < BLK_LROWS && row < n2;
			const int g = act ? row / 3 : ng2, k = act ? row - 3 * g : 0;
			auto *const R = pool_ + g * LGRP;
			const float4 sv = *reinterpret_cast<const float4 *>(R + LG_S + 4 * k);
			const float rinv = R[LG_RINV + k], sum = R[LG_SUM + k], fms = R[LG_SUM];
			const v3 n = L3(R + LG_N + 3 * k);
			const float4 o0 = *reinterpret_cast<const float4 *>(R + LG_GB + 12 * k), o1 = *reinterpret_cast<const float4 *>(R + LG_GB + 12 * k + 4), o2 = *reinterpret_cast<const float4 *>(R + LG_GB + 12 * k + 8);
			const v3 g0 = V3(o0.x, o0.z, o1.x), b0 = V3(o0.y, o0.w, o1.y), g1 = V3(o1.z, o2.x, o2.z), b1 = V3(o1.w, o2.y, o2.w);
			const unsigned bo = act ? ((Q & 1) ? lbod >> 16 : lbod & 0xFFFFu) : 0xFFFFu;
			const int ba = (bo & 255u) < HT_MAXNB ? (int)(bo & 255u) : IDLE_BODY, bb = (bo >> 8) < HT_MAXNB ? (int)(bo >> 8) : IDLE_BODY;
			const bool contact = g >= njg && g < ng2;                  // the joints' triples come first (physmodel.h:350, physics.h:549-551)
			const float4 Pa = S.lin4[ba], La = S.ang4[ba], Pb = S.lin4[bb], Lb = S.ang4[bb];
			// vn = v1.n - v0.n (physics.h:296-297) against the momenta before the block
			float w = b0.x * La.x; w = __fmaf_rn(b0.y, La.y, w); w = __fmaf_rn(b0.z, La.z, w);
			w = __fmaf_rn(b1.x, Lb.x, w); w = __fmaf_rn(b1.y, Lb.y, w); w = __fmaf_rn(b1.z, Lb.z, w);
			const float na_ = -Pa.w, nb_ = Pb.w;
			w = __fmaf_rn(n.x * na_, Pa.x, w); w = __fmaf_rn(n.y * na_, Pa.y, w); w = __fmaf_rn(n.z * na_, Pa.z, w);
			w = __fmaf_rn(n.x * nb_, Pb.x, w); w = __fmaf_rn(n.y * nb_, Pb.y, w); w = __fmaf_rn(n.z * nb_, Pb.z, w);
			float x = (-(post ? sv.y : sv.x) - w) * rinv;
			const bool fric = act && k > 0 && contact;      // limited by the normal row's impulse sum (physics.h:292): set behind that row's step
			float lo = fric ? 0.0f : sv.z - sum, hi = fric ? 0.0f : sv.w - sum;
			const int mp = fric ? p - k : 255;
			float imp = 0.0f;
			const int nrows = n2 - BLK_LROWS * Q;
			long long tb1 = 0; if (stats) { asm volatile("" :: "v"(x), "v"(lo), "v"(hi)); tb1 = clock64(); }
			// per half of the block: its triples, which of them are contacts; a half of five joint triples takes the plain statement
			int cwl[2]; bool gen[2];
#pragma unroll
			for (int hf = 0; hf < 2; hf++)
			{
				const int g0_ = 10 * Q + 5 * hf;
				int np = ng2 - g0_, fc = njg - g0_;                                  // triples the block has from here on, joints among them
				np = np < 0 ? 0 : np > 5 ? 5 : np; fc = fc < 0 ? 0 : fc > 5 ? 5 : fc;
				const int pres = (1 << np) - 1;
				cwl[hf] = __builtin_amdgcn_readfirstlane((pres << 8) | (pres & ~((1 << fc) - 1)));
				gen[hf] = __builtin_amdgcn_readfirstlane((int)(fc < 5 || np < 5)) != 0;
			}
#define BLK_LHALF(QQ, HF) if (gen[HF]) blk_resolve15<QQ, HF, true>(x, lo, hi, imp, GL, fms, sv.w, sum, mp, idt_bits, dt_bits, cwl[HF]); else blk_resolve15<QQ, HF, false>(x, lo, hi, imp, GL, fms, sv.w, sum, mp, idt_bits, dt_bits, 0);
#define BLK_LCASE(QQ) case QQ: BLK_LHALF(QQ, 0) if (nrows > 15) { BLK_LHALF(QQ, 1) } break;
			switch (Q) { BLK_LCASE(0) BLK_LCASE(1) BLK_LCASE(2) default: BLK_LCASE(3) }
#undef BLK_LHALF
#undef BLK_LCASE
			long long tb2 = 0; if (stats) { asm volatile("" :: "v"(imp)); tb2 = clock64(); }
			if (act) R[LG_SUM + k] = sum + imp;
			// the momenta: -n d to P(rb0), g0 d to L(rb0), n d to P(rb1), g1 d to L(rb1) (ApplyImpulse physics.h:222-226), summed per body over the block's sorted edges
			const unsigned e = Q == 0 ? emL0 : Q == 1 ? emL1 : Q == 2 ? emL2 : emL3;
			const int src = (int)(e & 63);
			const bool side = (e & BLK_E_SIDE) != 0;
			const float cf = (e & BLK_E_VALID) ? (side ? 1.0f : -1.0f) : 0.0f, cv = (e & BLK_E_VALID) ? 1.0f : 0.0f;
			const float dpx = n.x * imp, dpy = n.y * imp, dpz = n.z * imp;
			const float d0x = g0.x * imp, d0y = g0.y * imp, d0z = g0.z * imp, d1x = g1.x * imp, d1y = g1.y * imp, d1z = g1.z * imp;
			float vpx = blk_pull(src, dpx) * cf, vpy = blk_pull(src, dpy) * cf, vpz = blk_pull(src, dpz) * cf;
			const float q0x = blk_pull(src, d0x), q0y = blk_pull(src, d0y), q0z = blk_pull(src, d0z), q1x = blk_pull(src, d1x), q1y = blk_pull(src, d1y), q1z = blk_pull(src, d1z);
			float vlx = (side ? q1x : q0x) * cv, vly = (side ? q1y : q0y) * cv, vlz = (side ? q1z : q0z) * cv;
			const blk_scan_mul sm = blk_scan_multipliers(e);
			blk_seg_scan3(vpx, vpy, vpz, sm);
			blk_seg_scan3(vlx, vly, vlz, sm);
			if (e & BLK_E_TAIL)
			{
				const int body = (int)(e >> 16);
				float4 P = S.lin4[body], L = S.ang4[body];
				P.x += vpx; P.y += vpy; P.z += vpz; L.x += vlx; L.y += vly; L.z += vlz;
				S.lin4[body] = P; S.ang4[body] = L;
			}
			__builtin_amdgcn_wave_barrier();
			if (stats) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long tb3 = clock64(); cyc_bh += tb1 - tb0; cyc_br += tb2 - tb1; cyc_bg += tb3 - tb2; }
		}
	};
	auto blocked_angular = [&](const auto arec_, const bool post) {
		for (int Q = 0; Q < nba; Q++)
		{
			const int m = lane & 31;
			const int p = (Q & 1) ? 31 - m : m, row = 32 * Q + p;
			const bool act = (lane >> 5) == (Q >> 1) && row < na;
			auto *const R = arec_ + (act ? row : na) * AROW;      // the record behind the last row is all zeros
			const float4 sv = *reinterpret_cast<const float4 *>(R + AR_S), gt = *reinterpret_cast<const float4 *>(R + AR_GAIN), ab = *reinterpret_cast<const float4 *>(R + AR_AXIS + 2), bt = *reinterpret_cast<const float4 *>(R + AR_BA + 2);
			// gt = gain, torque, axis.x, axis.y;  ab = axis.z, gain after RemoveBias (the same: a frame with a row that differs takes the level schedule), ba0.x, ba0.y;  bt = ba0.z, ba1.xyz
			const unsigned bo = act ? ((Q & 1) ? abod >> 16 : abod & 0xFFFFu) : 0xFFFFu;
			const int ba = (bo & 255u) < HT_MAXNB ? (int)(bo & 255u) : IDLE_BODY, bb = (bo >> 8) < HT_MAXNB ? (int)(bo >> 8) : IDLE_BODY;
			const float4 La = S.ang4[ba], Lb = S.ang4[bb];
			float w = ab.z * La.x; w = __fmaf_rn(ab.w, La.y, w); w = __fmaf_rn(bt.x, La.z, w);
			w = __fmaf_rn(bt.y, Lb.x, w); w = __fmaf_rn(bt.z, Lb.y, w); w = __fmaf_rn(bt.w, Lb.z, w);      // spin1.axis - spin0.axis (physics.h:253-254): ba0 carries rb0's minus sign
			float x = ((post ? sv.y : sv.x) - w) * gt.x;
			const float lo = sv.z - gt.y, hi = sv.w - gt.y;
			float imp = 0.0f;
			const int nrows = na - 32 * Q;
			// a half with all sixteen rows takes the plain statement, the last one stops behind its rows (tested every four)
			auto cw16 = [](int nr) -> int { return (nr > 4 ? 2 : 0) | (nr > 8 ? 4 : 0) | (nr > 12 ? 8 : 0); };
			const int cwa0 = __builtin_amdgcn_readfirstlane(cw16(nrows)), cwa1 = __builtin_amdgcn_readfirstlane(cw16(nrows - 16));
#define BLK_ACASE(QQ) case QQ: if (nrows >= 16) blk_resolve16<QQ, 0, false>(x, lo, hi, imp, GA, 0); else blk_resolve16<QQ, 0, true>(x, lo, hi, imp, GA, cwa0); \
	if (nrows >= 32) blk_resolve16<QQ, 1, false>(x, lo, hi, imp, GA, 0); else if (nrows > 16) blk_resolve16<QQ, 1, true>(x, lo, hi, imp, GA, cwa1); break;
			switch (Q) { BLK_ACASE(0) BLK_ACASE(1) BLK_ACASE(2) default: BLK_ACASE(3) }
#undef BLK_ACASE
			if (act) R[AR_TORQUE] = gt.y + imp;
			// rb0: L -= axis d, rb1: L += axis d (physics.h:262-263)
			const unsigned e = Q == 0 ? emA0 : Q == 1 ? emA1 : Q == 2 ? emA2 : emA3;
			const int src = (int)(e & 63);
			const float cf = (e & BLK_E_VALID) ? ((e & BLK_E_SIDE) ? 1.0f : -1.0f) : 0.0f;
			const float tx = gt.z * imp, ty = gt.w * imp, tz = ab.x * imp;
			float vx = blk_pull(src, tx) * cf, vy = blk_pull(src, ty) * cf, vz = blk_pull(src, tz) * cf;
			const blk_scan_mul sm = blk_scan_multipliers(e);
			blk_seg_scan3(vx, vy, vz, sm);
			if (e & BLK_E_TAIL)
			{
				const int body = (int)(e >> 16);
				float4 L = S.ang4[body];
				L.x += vx; L.y += vy; L.z += vz;
				S.ang4[body] = L;
			}
			__builtin_amdgcn_wave_barrier();
		}
	};
	for (int sweep = 0; sweep < total_sweeps; sweep++)
	{
		const bool post = sweep >= ph.iterations;
		const int tsoff = post ? 1 : 0;                                   // RemoveBias (physics.h:288): ts_post = min(ts, ts_nobias) was stored next to ts
		// (1) chains: quad q applies the single-body rows of body q (then those of a body >= 16 it hosts) in order; momenta, inertia row and mass stay in registers.
		//     Lane c < 3 of the quad reads r1[c] and n[c] of a record, lane 3 reads its target speed (ts or ts_post).
		if (chain4)
		{
			if (c4_nblk > 0)
			{
				const int jq = (lane >> 2) & 3;
				if (idx_lds && g_lds) quad_blocks_run(scr, S.cidx, S.cg, S.csum, c4_e0, c4_nblk, c, jq, tsoff, lin_w, ang_w, c4_head, S.cextra, S.ccnt);
				else if (idx_lds) quad_blocks_run(scr, S.cidx, gG, S.csum, c4_e0, c4_nblk, c, jq, tsoff, lin_w, ang_w, c4_head, S.cextra, S.ccnt);
				else if (sums_lds) quad_blocks_run(scr, gidx, gG, S.csum, c4_e0, c4_nblk, c, jq, tsoff, lin_w, ang_w, c4_head, S.cextra, S.ccnt);
				else quad_blocks_run(scr, gidx, gG, gsum, c4_e0, c4_nblk, c, jq, tsoff, lin_w, ang_w, c4_head, S.cextra, S.ccnt);
			}
		}
		else
		{
			const int body = quad;
			const bool has = body < nb;
			const int cnt = has ? S.ccnt[body] : 0, start = has ? S.cstart[body] : 0;
			const int ex = has ? S.cextra[body] : -1;                      // a body >= 16 whose rows follow this body's padded ones
			const int kswitch = ex >= 0 ? (cnt + 7) & ~7 : -1;
			const int total = ex >= 0 ? kswitch + S.ccnt[ex] : cnt;
			if (total > 0)
			{
				const float l = lin_w[4 * body + c], av = ang_w[4 * body + c];       // lane 3 carries massinv / friction here and never stores
				quad_body qb = { l, av };
				// RemoveBias (physics.h:288): lane 3 switches to the ts_post slot
				if (idx_lds) quad_chain_run(qb, scr, S.cidx + start, S.csum + start, total, c, tsoff, kswitch, lin_w, ang_w, body, ex);
				else if (sums_lds) quad_chain_run(qb, scr, gidx + start, S.csum + start, total, c, tsoff, kswitch, lin_w, ang_w, body, ex);
				else quad_chain_run(qb, scr, gidx + start, gsum + start, total, c, tsoff, kswitch, lin_w, ang_w, body, ex);
				const int last = ex >= 0 ? ex : body;
				if (c < 3) { lin_w[4 * last + c] = qb.l; ang_w[4 * last + c] = qb.av; }
			}
		}
		__builtin_amdgcn_wave_barrier();
		if (stats) { const long long t = clock64(); cyc_chain += t - t_mark; t_mark = t; }
		// (2) two-body linear rows (LimitLinear::Iter physics.h:289-307), one group per lane pair and step: the 3 rows of a joint (x, y, z) or
		//     of a contact (normal, two friction rows) share their bodies and are applied back to back with the momenta in registers.  Pairs
		//     without a group work on the idle group / idle body, so a step is branch-free.  Three-stage software pipeline: the sort entry is
		//     fetched two steps ahead, the group's record one step ahead; only the momenta are read after the previous step's stores.
		if (blocked) { if (!HT_DBG(a.dbg, 2)) { if (pool_lds) blocked_linear(S.pool, post); else blocked_linear(gpool, post); } }
		else if (!HT_DBG(a.dbg, 2) && nlev_lin > 0)
		{
			if (post) { if (pool_lds) linear_phase(S.pool, ht_true{}); else linear_phase(gpool, ht_true{}); }
			else { if (pool_lds) linear_phase(S.pool, ht_false{}); else linear_phase(gpool, ht_false{}); }
		}
		__builtin_amdgcn_wave_barrier();
		if (stats) { const long long t = clock64(); cyc_lin += t - t_mark; t_mark = t; }
		// (3) angular rows (LimitAngular::Iter physics.h:251-265): one run of consecutive rows on the same body pair per lane pair and step,
		//     same pipeline; inside a run the next row's record is read while the current row is applied
		if (blocked) { if (!HT_DBG(a.dbg, 4)) { if (arec_lds) blocked_angular(S.arec, post); else blocked_angular(garec, post); } }
		else if (!HT_DBG(a.dbg, 4) && nlev_ang > 0)
		{
			if (post) { if (arec_lds) angular_phase(S.arec, ht_true{}); else angular_phase(garec, ht_true{}); }
			else { if (arec_lds) angular_phase(S.arec, ht_false{}); else angular_phase(garec, ht_false{}); }
		}
		__syncthreads();
		if (stats) { const long long t = clock64(); cyc_ang += t - t_mark; t_mark = t; }
		if (sweep + 1 == ph.iterations && lane < nb) calc_next_pose();
	}

	if (stats && lane == 0)
	{
		float *o = scr + (size_t)(a.scratch_stride - 1) * CREC;
		int mc = 0; for (int k = 0; k < nb; k++) if (S.ccnt[k] > mc) mc = S.ccnt[k];
		o[0] += 1.0f; o[1] += (float)cyc_chain; o[2] += (float)cyc_lin; o[3] += (float)cyc_ang; o[4] += (float)(clock64() - t_begin);
		o[5] += blocked ? (float)(t_c1 - t_c0) : (float)nlev_lin; o[6] += blocked ? (float)(t_c2 - t_c1) : (float)nlev_ang;      /* blocked frames: cycles of the couplings / of the edge words */ o[7] += (float)mc; o[8] += (float)n1; o[9] += (float)n2; o[10] += (float)na; o[11] += (float)(t_begin - t_entry);
		o[12] += (float)(t_m1 - t_entry); o[13] += (float)(t_m2 - t_m1); o[14] += (float)(t_m2b - t_m2); o[15] += (float)(t_m3 - t_m2b);
		if (HT_DBG(a.dbg, 262144)) { o[12] += (float)(t_l1 - t_m3) - (float)(t_m1 - t_entry); o[13] += (float)(t_l2 - t_l1) - (float)(t_m2 - t_m1); o[14] += (float)(t_l3 - t_l2) - (float)(t_m2b - t_m2); o[15] += (float)(t_l4 - t_l3) - (float)(t_m3 - t_m2b); }      /* the chain lists' parts: set-up + counting pass | dealing | placement pass + records | Iinv*axis + four-row couplings */
		if (HT_DBG(a.dbg, 32768)) { o[12] += (float)cyc_bh - (float)(t_m1 - t_entry); o[13] += (float)cyc_br - (float)(t_m2 - t_m1); o[14] += (float)cyc_bg - (float)(t_m2b - t_m2); }      /* HT_DEBUG_SKIP += 32768: the blocked linear phase's head / resolve / gather cycles instead of the first three prologue parts */
	}
	}      // !EXACT
	// ---- rbupdatepose (physics.h:533-541), SanityCheck (physmodel.h:437-442), optional momentum reset (handtrack.h:686-687) ----
	if (lane < nb)
	{
		float *s = st + lane * HT_STATE_STRIDE;
		const float *bc = M.bodyc + lane * HT_BC;
		v3 pos = pos_next; v4 q = q_next;
		v3 lin = F3(S.lin4[lane]), ang = F3(S.ang4[lane]);
		const bool bad = isnan(lin.x) || isnan(lin.y) || isnan(lin.z) || isnan(pos.x) || isnan(pos.y) || isnan(pos.z) || isnan(ang.x) || isnan(ang.y) || isnan(ang.z)
		              || isnan(q.x) || isnan(q.y) || isnan(q.z) || isnan(q.w);
		if (bad) { pos = L3(bc + HT_BC_POS0); q = L4(bc + HT_BC_Q0); lin = V3(0, 0, 0); ang = V3(0, 0, 0); }
		if (a.zero_momenta) { lin = V3(0, 0, 0); ang = V3(0, 0, 0); }
		s[0] = pos.x; s[1] = pos.y; s[2] = pos.z; s[3] = q.x; s[4] = q.y; s[5] = q.z; s[6] = q.w;
		s[7] = lin.x; s[8] = lin.y; s[9] = lin.z; s[10] = ang.x; s[11] = ang.y; s[12] = ang.z;
		if (a.out_poses)      // the update's last solve delivers the user poses too: GetPoseUser (physmodel.h:434) and the "initializing = 50" rule (handtrack.h:781-782)
		{
			const v3 pu = apply(XF(pos, q), -L3(bc + HT_BC_COM));
			float *o = a.out_poses + ((size_t)b * nb + lane) * HT_POSE;
			o[0] = pu.x; o[1] = pu.y; o[2] = pu.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
			if (lane == 0 && a.out_npts[b] < a.out_min_point_num) a.out_initializing[b] = 50;
		}
	}
	if (a.cost_out && lane == 0) a.cost_out[b] = (int)((clock64() - t_cost) >> 4);      // what the frame took: the next update's launch order (ht_launch_rank_desc)
}

// order[slot][f0 + r] = the frame of segment [f0, f0 + 4096) with the r-th largest work[slot][.] (ties by index): ranks by counting, the works in LDS; a block per slot
// whose bit is set in `slots` and segment.  For launches whose blocks outnumber the resident ones several times: taken in this order the launch ends on short frames.
__global__ __launch_bounds__(1024) void k_rank_desc(const int *__restrict__ work, int *__restrict__ order, int Ball, int stride, unsigned slots)
{
	__shared__ int rk_w[4096];
	if (!((slots >> blockIdx.x) & 1u)) return;
	const int f0 = blockIdx.y * 4096, B = Ball - f0 < 4096 ? Ball - f0 : 4096, B4 = (B + 3) & ~3;
	const int *w = work + (size_t)blockIdx.x * stride + f0;
	int *o = order + (size_t)blockIdx.x * stride + f0;
	for (int i = threadIdx.x; i < B4; i += 1024) rk_w[i] = i < B ? w[i] : -1;
	__syncthreads();
	for (int i = threadIdx.x; i < B; i += 1024)
	{
		const int wi = rk_w[i];
		int rank = 0;
		for (int j = 0; j < B4; j += 4)
		{
			const int4 k = *reinterpret_cast<const int4 *>(rk_w + j);
			rank += ((k.x > wi || (k.x == wi && j < i)) ? 1 : 0) + ((k.y > wi || (k.y == wi && j + 1 < i)) ? 1 : 0) + ((k.z > wi || (k.z == wi && j + 2 < i)) ? 1 : 0) + ((k.w > wi || (k.w == wi && j + 3 < i)) ? 1 : 0);
		}
		o[rank] = f0 + i;
	}
}
void ht_launch_rank_desc(const int *work, int *order, int B, int stride, unsigned slots, int nslots, hipStream_t s)
{
	hipLaunchKernelGGL(k_rank_desc, dim3(nslots, (B + 4095) / 4096), dim3(1024), 0, s, work, order, B, stride, slots);
}

#define SOLVE_ONLY_NCG 1440      // floats of single-body-row couplings the build for 1024 frames keeps in LDS: what is left of a quarter of a CU's 160 KB (144 blocks of four rows)
static_assert(sizeof(lds_t<66, 1024, 126, 1024, 2, SOLVE_ONLY_NCG>) <= 40960, "the build for 1024 frames must fit four times into a CU's LDS");
static_assert(sizeof(lds_t<34, 584, 84, 0>) <= 20480, "the small build must leave room for eight frames per CU (160 KB of LDS)");
static_assert(HT_SCRATCH_TAIL * HT_CREC >= MAXG_CAP * LGRP + (MAXA_CAP_OF(4) + 4) * AROW + HT_CREC, "the tail of a frame's scratch slot must hold its linear groups, its angular records and the tuning record");
void ht_launch_solve(const ht_model_dev &M, const ht_physics_dev &ph, const solve_args &a, int B, hipStream_t s)
{
	// the build by what the host knows of the launch: the model's joints and the most points a frame of this call can carry.  A frame that exceeds
	// the chosen build's arrays (contacts, points: device-side data) keeps the array in question in HBM (top of the file); nothing is relaunched.
	const int pts = M.pts_bound > 0 ? M.pts_bound : M.pts_cap;
	const bool tile = M.nj + 8 + 1 <= 38 && pts <= 1024;      // room for a few contacts beside the joints, and a 64x64 tile's cloud
	// Up to four frames per CU (1024 on the 256 CUs) a launch gains nothing from the small build's footprint, so the build that keeps 49 contacts and
	// 126 angular rows in LDS runs.  Not when other kernels share the GPU with this launch (the reset path): they need LDS on every CU too.
	int build = a.force_build;
	if (build == 7) { solve_args b = a; b.force_build = 0; b.two_body_levels = 1; ht_launch_solve(M, ph, b, B, s); return; }      // tests only: the two-body rows by the level schedule for every frame
	if (!build) build = tile ? (B <= 1024 && !a.shared_gpu ? 2 : 1) : 3;
	// the angular rows a frame of this launch can have at most (13 CNN-driven + 6 per joint + what the caller states on top: slowfit's relative rows, caller-built rows):
	// beyond the 126 of the ordinary builds the build with four row slots per lane runs (252)
	if (build >= 1 && build <= 3 && 13 + 6 * M.nj + a.ang_extra_bound > MAXA_CAP_OF(2)) build = 6;
	if (build == 2) hipLaunchKernelGGL((k_solve<66, 1024, 126, 1024, false, 2, SOLVE_ONLY_NCG>), dim3(B), dim3(64), 0, s, M, ph, a);
	else if (build == 1) hipLaunchKernelGGL((k_solve<34, 584, 84, 0>), dim3(B), dim3(64), 0, s, M, ph, a);
	else if (build == 3) hipLaunchKernelGGL((k_solve<71, 1520, 126, 1520>), dim3(B), dim3(64), 0, s, M, ph, a);
	else if (build == 5) hipLaunchKernelGGL((k_solve<71, 1520, 126, 1520, true, 4>), dim3(B), dim3(64), 0, s, M, ph, a);      // tests only: the reference's own sweeps (ht_debug_exact_solver)
	else if (build == 6) hipLaunchKernelGGL((k_solve<71, 1520, 126, 1520, false, 4>), dim3(B), dim3(64), 0, s, M, ph, a);     // models / calls with up to 252 angular rows (records beyond 126 in HBM)
	else hipLaunchKernelGGL((k_solve<2, 64, 4, 0>), dim3(B), dim3(64), 0, s, M, ph, a);      // tests only: nothing fits, every frame keeps its groups, sums and angular records in HBM
}
