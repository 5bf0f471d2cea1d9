// ht_prep.hip -- k_solve_prep: the tables of one constraint solve (ht_solve_shared.hpp), four waves per frame, beside the contact kernel.
//
// Reference computations (the same as k_solve's prologue, csrc/ht_solver.hip, whose expressions these are, statement by statement):
//   HandModelEnhancements                        include/handtrack.h:406-441
//   CNNOutputAnalysis::ApplyAngles, landmark rays include/handtrack.h:203-216, 666-676
//   GetAngularConstraints / ConstrainAngularRange include/physmodel.h:321-327, third_party/physics.h:351-399
//   GetLinearConstraints / ConstrainPositionNailed include/physmodel.h:328-334, physics.h:342-346
//   cloud_chamber's rows (ConstrainUnderPlane)    include/physmodel.h:486-496, physics.h:347-350
//   row order of FitPointCloud / PhysicsUpdate    include/physmodel.h:345-351, physics.h:556-562
//
// Until round 5 k_solve's single wave did all of this before its first sweep: 0.9 M of a frame's 4.1 M cycles per step, eight times a step, on the critical path, from
// data that is known as soon as the pose is (the joints' and the angular rows, the blocks' couplings) or as soon as k_cloud_rows is through (the chains).  Here a block of
// four waves makes it per frame, each wave a part, on a side stream while the contact kernel runs:
//   wave 0   HandModelEnhancements, every angular row -> its record, the angular blocks' couplings and sorted edges
//   wave 1   the joints' linear groups, their blocks' couplings and sorted edges
//   wave 2   the landmark-ray rows, the per-body chains of the single-body rows (stable partition by body), their dealing to the four DPP rows of k_solve's walk, the
//            couplings of their blocks of four
//   wave 3   the boundary planes' rows (one per plane and body: the support vertex scan), when the frame has them
// A frame the blocked form of the two-body rows does not hold (an angular row that RemoveBias switches on, more than 126 angular rows) is marked and keeps k_solve's own
// prologue.  Nothing here is new arithmetic: every expression is the one k_solve evaluates, so the results are the same bits (tests/test_gpu_same_bits.py).
#include "ht_solve_shared.hpp"

#define PREP_THREADS 256
#define PREP_NIDX 1664      // chain entries whose list is also kept in LDS for the coupling pass (a 64x64 tile's rows; a larger frame reads its list back from HBM)
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)      // LDS written by some lanes of THIS wave, read by others (a wave's LDS instructions execute in order)

struct prep_lds
{
	// the bodies as k_solve holds them (rbinitvelocity physics.h:500-519)
	float4 lin4[HT_MAXNB], ang4[HT_MAXNB];
	float pos[HT_MAXNB][3], q[HT_MAXNB][4];
	float4 I4[HT_MAXNB][3];
	// wave 0
	float jr[HT_MAXNJ][6];
	int aprefix[HT_MAXNJ + 1];
	unsigned char rowj[128];
	float arec[(128 + 4) * AROW] __attribute__((aligned(16)));
	unsigned short abody[128];
	unsigned etmpA[64];
	// wave 1
	float pool[(HT_MAXNJ + 1) * LGRP] __attribute__((aligned(16)));
	unsigned etmpL[64];
	// wave 2
	float ray[36][HT_ROW];
	int nray;
	int corder[HT_MAXNB];
	unsigned short cidx[PREP_NIDX];
	// wave 3
	float planes[5][4];
	int voff[HT_MAXNB + 1];
};

// a block's (row, side) pairs sorted by body, one per lane (ht_block.hpp): k_solve's edge_word, the wave's own table in LDS
__device__ __forceinline__ unsigned prep_edge_word(unsigned *etmp, int lane, int nb, bool valid, int ba, int bb)
{
	const int ka = (valid && ba < nb) ? ba : 255, kb = (valid && bb < nb) ? bb : 255;
	const unsigned long long lt = (1ull << lane) - 1ull;
	int r0 = -1, r1 = -1, base = 0;
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
	WSYNC();
	etmp[lane] = 0u;
	WSYNC();
	if (r0 >= 0) etmp[r0] = (unsigned)lane | BLK_E_VALID | ((unsigned)ka << 16);
	if (r1 >= 0) etmp[r1] = (unsigned)lane | BLK_E_SIDE | BLK_E_VALID | ((unsigned)kb << 16);
	WSYNC();
	const unsigned e = etmp[lane];
	const bool v = (e & BLK_E_VALID) != 0;
	auto same = [&](int other) -> bool { const unsigned o = etmp[other & 63]; return v && other >= 0 && other < 64 && (o & BLK_E_VALID) != 0 && (o >> 16) == (e >> 16); };
	const int rs = lane & ~15;
	unsigned w = e;
	if (v && !same(lane + 1)) w |= BLK_E_TAIL;
	if (lane - 1 >= rs && same(lane - 1)) w |= 1u << 9;
	if (lane - 2 >= rs && same(lane - 2)) w |= 1u << 10;
	if (lane - 4 >= rs && same(lane - 4)) w |= 1u << 11;
	if (lane - 8 >= rs && same(lane - 8)) w |= 1u << 12;
	if ((lane & 16) && same(rs - 1)) w |= 1u << 13;
	if (lane >= 32 && same(31)) w |= 1u << 14;
	return w;
}
__device__ __forceinline__ void prep_store_regs(float *dst, int lane, const float (&G)[32])
{
#pragma unroll
	for (int k = 0; k < 8; k++) reinterpret_cast<float4 *>(dst)[k * 64 + lane] = make_float4(G[4 * k], G[4 * k + 1], G[4 * k + 2], G[4 * k + 3]);
}

// ---- wave 0: the angular rows (k_solve's prologue: HandModelEnhancements, the row list [ApplyAngles 12][arm cone 1][joint ranges], records, blocks) ----
__device__ __forceinline__ void prep_angular(const ht_model_dev &M, const ht_physics_dev &ph, const prep_args &a, prep_lds &S, float *T, int b, int lane, int &ok, int &na_out)
{
	const int nb = M.nb, nj = M.nj;
	const float dt = ph.deltaT;
	if (lane < nj) for (int i = 0; i < 6; i++) S.jr[lane][i] = M.jointc[lane * HT_JC + HT_JC_RMIN + i];
	WSYNC();
	// HandModelEnhancements (handtrack.h:417-420, 434-440); acos()/cos() are the C double overloads there
	if (nb >= 17)
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
	WSYNC();
	const int na_fix = (a.apply_angles ? 12 : 0) + (a.arm_cone ? 1 : 0);
	int na;
	{
		const int acnt = lane < nj ? angular_range_count(L3(S.jr[lane]), L3(S.jr[lane] + 3)) : 0;
		int apre = 0, atot = 0;
		for (int j = 0; j < nj; j++) { const int ac = __builtin_amdgcn_readlane(acnt, j); apre += j < lane ? ac : 0; atot += ac; }
		na = na_fix + atot;
		if (lane <= nj) S.aprefix[lane] = na_fix + apre;
		for (int k = 0; k < acnt; k++) if (na_fix + apre + k < 128) S.rowj[na_fix + apre + k] = (unsigned char)lane;
	}
	WSYNC();
	na_out = na;
	if (na > MAXA_CAP_OF(2)) { ok = 0; return; }      // more rows than the ordinary builds of k_solve keep: its own prologue counts and reports them
	arow AR[2];
	bool sw = false;
#pragma unroll
	for (int s = 0; s < 2; s++)
	{
		const int r = lane + 64 * s;
		arow &R = AR[s];
		R.rb0 = -1; R.rb1 = -1; R.axis = V3(0, 0, 1); R.targetspin = -FLT_MAX; R.mn = 0; R.mx = 0; R.s2t = 0; R.torque = 0; R.mintorque = 0; R.lev = 0;
		if (r < na)
		{
			float row[8];
			if (r < na_fix)
			{
				const float *cam = a.cams + (size_t)b * HT_CAM;
				const v4 camq = V4(cam[8], cam[9], cam[10], cam[11]);
				const int ra = a.apply_angles ? r : 12;          // index into the ApplyAngles list, 12 = the arm cone
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
			R.s2t = 1.0f / (((R.rb0 >= 0) ? dot(R.axis, mul(body_I(S, R.rb0), R.axis)) : 0.0f) + ((R.rb1 >= 0) ? dot(R.axis, mul(body_I(S, R.rb1), R.axis)) : 0.0f));
			R.mn = mintorque * dt; R.mx = maxtorque * dt; R.mintorque = mintorque; R.torque = 0.0f;
		}
		sw = sw || (r < na && R.targetspin == -FLT_MAX && R.mintorque < 0);
	}
	if (__ballot(sw) != 0ull) { ok = 0; return; }      // a row RemoveBias switches on: the frame keeps the level schedule
	// records (k_solve: "angular rows move from their builder lanes into their records")
#pragma unroll
	for (int s = 0; s < 2; s++)
	{
		const int r = lane + 64 * s;
		const arow &R = AR[s];
		if (r < na)
		{
			const v3 ba0 = R.rb0 >= 0 ? -mul(body_I(S, R.rb0), R.axis) : V3(0, 0, 0);
			const v3 ba1 = R.rb1 >= 0 ? mul(body_I(S, R.rb1), R.axis) : V3(0, 0, 0);
			float *o = S.arec + r * AROW;
			const float ts_post = (R.mintorque < 0) ? 0 : fmin_std(R.targetspin, 0.0f);                      // RemoveBias physics.h:250
			o[AR_S] = R.targetspin; o[AR_S + 1] = ts_post; o[AR_S + 2] = R.mn; o[AR_S + 3] = R.mx;
			o[AR_GAIN] = R.targetspin == -FLT_MAX ? 0.0f : R.s2t; o[AR_TORQUE] = 0.0f;
			o[AR_AXIS] = R.axis.x; o[AR_AXIS + 1] = R.axis.y; o[AR_AXIS + 2] = R.axis.z; o[AR_AXIS + 3] = ts_post == -FLT_MAX ? 0.0f : R.s2t;
			o[AR_BA] = ba0.x; o[AR_BA + 1] = ba0.y; o[AR_BA + 2] = ba0.z; o[AR_BA + 3] = ba1.x; o[AR_BA + 4] = ba1.y; o[AR_BA + 5] = ba1.z;
		}
		S.abody[r] = (unsigned short)(r < na ? ((R.rb0 >= 0 ? R.rb0 : 255) | ((R.rb1 >= 0 ? R.rb1 : 255) << 8)) : 0xFFFF);
	}
	S.arec[na * AROW + lane] = 0.0f;      // idle record + read-ahead slack (4 records)
	WSYNC();
	// the table's copy of the records and body pairs
	for (int i = lane; i < (na + 4) * (AROW / 4); i += 64) reinterpret_cast<float4 *>(T + TB_AREC)[i] = reinterpret_cast<const float4 *>(S.arec)[i];
	reinterpret_cast<unsigned *>(T + TB_ABODY)[lane] = (unsigned)S.abody[2 * lane] | ((unsigned)S.abody[2 * lane + 1] << 16);
	// couplings (ht_block.hpp; k_solve's ang_couplings): lane m of a half holds its forward row (block 2h, row m) and its backward row (block 2h + 1, row 31 - m)
	float GA[32];
#pragma unroll
	for (int i = 0; i < 32; i++) GA[i] = 0.0f;
	const int m = lane & 31, hh = lane >> 5;
	const int nba = (na + 31) >> 5;
	auto ang_couplings = [&](const bool bwd) {
		const float *arec_ = S.arec;
		const int r = bwd ? 64 * hh + 63 - m : 64 * hh + m;
		const float *R = arec_ + (r < na ? r : na) * AROW;
		const float ng = -R[AR_GAIN];
		const v3 b0 = L3(R + AR_BA) * ng, b1 = L3(R + AR_BA + 3) * ng;
		const int bo = (int)S.abody[r], a0 = bo & 255, a1 = bo >> 8;
#pragma unroll
		for (int c8 = 0; c8 < 32; c8 += 8)
		{
			float ax[8], ay[8], az[8]; int pb[8];
#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const int rp = bwd ? 64 * hh + 63 - (c8 + u) : 64 * hh + c8 + u;
				const float *P = arec_ + (rp < na ? rp : na) * AROW + AR_AXIS;
				ax[u] = P[0]; ay[u] = P[1]; az[u] = P[2]; pb[u] = (int)S.abody[rp];
			}
#pragma unroll
			for (int u = 0; u < 8; u++)
			{
				const int rho = c8 + u, p0 = pb[u] & 255, p1 = pb[u] >> 8;
				const float t0 = (b0.x * ax[u] + b0.y * ay[u]) + b0.z * az[u], t1 = (b1.x * ax[u] + b1.y * ay[u]) + b1.z * az[u];
				float g = (a0 == p1 ? t0 : a0 == p0 ? -t0 : 0.0f) + (a1 == p1 ? t1 : a1 == p0 ? -t1 : 0.0f);
				asm volatile("" : "+v"(g));
				GA[rho] = (bwd ? rho > m : rho < m) ? g : GA[rho];
			}
		}
	};
	ang_couplings(false);
	if (nba > 1) ang_couplings(true);
	prep_store_regs(T + TB_GA, lane, GA);
	// sorted edges of every block
	auto abodies = [&](int Q) -> unsigned { const int row = 32 * Q + ((Q & 1) ? 31 - m : m); return row < na ? (unsigned)S.abody[row] : 0xFFFFu; };
	const unsigned abod = abodies(2 * hh) | (abodies(2 * hh + 1) << 16);
	for (int Q = 0; Q < 4; Q++)
	{
		unsigned w = 0u;
		if (Q < nba)
		{
			const int p = (Q & 1) ? 31 - m : m, row = 32 * Q + p;
			const bool on = hh == (Q >> 1) && row < na;
			const unsigned bo = (Q & 1) ? abod >> 16 : abod & 0xFFFFu;
			w = prep_edge_word(S.etmpA, lane, nb, on, (int)(bo & 255u), (int)(bo >> 8));
		}
		reinterpret_cast<unsigned *>(T + TB_EMA)[Q * 64 + lane] = w;
	}
}

// ---- wave 1: the joints' linear groups (physmodel.h:328-334), their blocks' couplings and edges ----
__device__ __forceinline__ void prep_joints(const ht_model_dev &M, const ht_physics_dev &ph, prep_lds &S, float *T, int lane)
{
	const int nb = M.nb, nj = M.nj;
	const float dt = ph.deltaT;
	const int n2 = 3 * nj, ng2 = nj;
	float *const pool = S.pool;
	for (int r = lane; r < n2; r += 64)
	{
		int rb0, rb1, meta = 0, g, kk;
		v3 p0, p1, n; float targetdist, tsnb, fmn, fmx;
		{
			const int j = r / 3, ax = r % 3;
			g = j; kk = ax;
			const float *jc = M.jointc + j * HT_JC;
			rb0 = (int)jc[HT_JC_RB0]; rb1 = (int)jc[HT_JC_RB1];
			p0 = L3(jc + HT_JC_P0) - L3(M.bodyc + rb0 * HT_BC + HT_BC_COM); p1 = L3(jc + HT_JC_P1) - L3(M.bodyc + rb1 * HT_BC + HT_BC_COM);
			const v3 d = anchor_world(S, rb1, p1) - anchor_world(S, rb0, p0);                       // ConstrainPositionNailed physics.h:342-346
			n = V3(ax == 0 ? 1.0f : 0.0f, ax == 1 ? 1.0f : 0.0f, ax == 2 ? 1.0f : 0.0f);
			targetdist = ax == 0 ? d.x : ax == 1 ? d.y : d.z; tsnb = 0.0f; fmn = -FLT_MAX; fmx = FLT_MAX;
		}
		const m3 Z = { V3(0, 0, 0), V3(0, 0, 0), V3(0, 0, 0) };
		const v3 r0 = rb0 >= 0 ? qrot(L4(S.q[rb0]), p0) : p0, r1 = rb1 >= 0 ? qrot(L4(S.q[rb1]), p1) : p1;
		const m3 I0 = rb0 >= 0 ? body_I(S, rb0) : Z, I1 = rb1 >= 0 ? body_I(S, rb1) : Z;
		const float impulsed = (rb0 >= 0 ? S.lin4[rb0].w + dot(cross(mul(I0, cross(r0, n)), r0), n) : 0.0f) + (rb1 >= 0 ? S.lin4[rb1].w + dot(cross(mul(I1, cross(r1, n)), r1), n) : 0.0f);      // physics.h:299-300
		const float ts = targetdist / dt;
		const v3 g0 = -cross(r0, n), g1 = cross(r1, n), b0 = mul(I0, g0), b1 = mul(I1, g1);
		float *o = pool + g * LGRP;
		float *os = o + LG_S + 4 * kk;
		os[0] = ts; os[1] = fmin_std(ts, tsnb); os[2] = fmin_std(fmn, fmx) * dt; os[3] = fmax_std(fmn, fmx) * dt;
		o[LG_RINV + kk] = 1.0f / impulsed; o[LG_SUM + kk] = 0.0f;
		if (kk == 0) o[LG_META] = __int_as_float(meta | (rb0 & 255) | ((rb1 & 255) << 8));
		o[LG_N + 3 * kk] = n.x; o[LG_N + 3 * kk + 1] = n.y; o[LG_N + 3 * kk + 2] = n.z;
		float *og = o + LG_GB + 12 * kk;
		og[0] = g0.x; og[1] = b0.x; og[2] = g0.y; og[3] = b0.y; og[4] = g0.z; og[5] = b0.z;
		og[6] = g1.x; og[7] = b1.x; og[8] = g1.y; og[9] = b1.y; og[10] = g1.z; og[11] = b1.z;
	}
	pool[ng2 * LGRP + lane] = (lane >= LG_RINV && lane < LG_RINV + 3) ? 1.0f : lane == LG_META ? __int_as_float(IDLE_BODY | (IDLE_BODY << 8)) : 0.0f;      // the idle group behind them, as in k_solve (the couplings' partner of a missing row)
	WSYNC();
	for (int i = lane; i < ng2 * (LGRP / 4); i += 64) reinterpret_cast<float4 *>(T + TB_POOL)[i] = reinterpret_cast<const float4 *>(pool)[i];
	// couplings among the joints' rows (k_solve's lin_couplings with n2 = the joints' rows: a contact row couples to the rows before it, never the other way round)
	float GL[32];
#pragma unroll
	for (int i = 0; i < 32; i++) GL[i] = 0.0f;
	const int m = lane & 31, hh = lane >> 5;
	const int nbl = (n2 + BLK_LROWS - 1) / BLK_LROWS;
	auto lbodies = [&](int Q) -> unsigned {
		const int p = (Q & 1) ? 31 - m : m, row = BLK_LROWS * Q + p;
		const bool on = p < BLK_LROWS && row < n2;
		return on ? ((unsigned)__float_as_int(pool[(row / 3) * LGRP + LG_META]) & 0xFFFFu) : 0xFFFFu;
	};
	const unsigned lbod = lbodies(2 * hh) | (lbodies(2 * hh + 1) << 16);
	auto lin_couplings = [&](const bool bwd) {
		const float *pool_ = pool;
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
				const int ppos = bwd ? 31 - (c4 + u) : c4 + u, prow = BLK_LROWS * blk + ppos;
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
	if (nbl > 0) lin_couplings(false);
	if (nbl > 1) lin_couplings(true);
	prep_store_regs(T + TB_GL, lane, GL);
	for (int Q = 0; Q < 4; Q++)
	{
		unsigned w = 0u;
		if (Q < nbl)
		{
			const int p = (Q & 1) ? 31 - m : m, row = BLK_LROWS * Q + p;
			const bool on = hh == (Q >> 1) && p < BLK_LROWS && row < n2;
			const unsigned bo = (Q & 1) ? lbod >> 16 : lbod & 0xFFFFu;
			w = prep_edge_word(S.etmpL, lane, nb, on, (int)(bo & 255u), (int)(bo >> 8));
		}
		reinterpret_cast<unsigned *>(T + TB_EML)[Q * 64 + lane] = w;
	}
}

// the solver's record (ht_quad.hpp) of a single-body row in the reference's layout, as k_solve's prologue forms it
__device__ __forceinline__ void prep_row_record(const prep_lds &S, const float *r, int body, float dt, float *rec)
{
	const v3 p1 = L3(r + 5), n = L3(r + 8);
	const v3 r1 = qrot(L4(S.q[body]), p1);
	const m3 Ib = body_I(S, body);
	const float impulsed = S.lin4[body].w + dot(cross(mul(Ib, cross(r1, n)), r1), n);       // 0 + (...) for the NULL side
	const float ts = r[11] / dt;
	quad_write_record(rec, r1, n, Ib, S.lin4[body].w, ts, fmin_std(ts, r[12]), impulsed, r[13] * dt, r[14] * dt);
}

// ---- wave 2, first half: the landmark-ray rows of MultiStepSim (handtrack.h:666-676) ----
__device__ __forceinline__ void prep_rays(const prep_args &a, prep_lds &S, int b, int lane)
{
	if (lane == 8)
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
	WSYNC();
}

// ---- wave 3: cloud_chamber's rows (physmodel.h:486-496) from the frame's five planes: k_chamber's second half (csrc/ht_cloud.hip), and the rows' records ----
__device__ __forceinline__ void prep_chamber_rows(const ht_model_dev &M, const ht_physics_dev &ph, const prep_args &a, prep_lds &S, int b, int lane, float *scr, int pre_base, int noop_idx)
{
	if (lane < 20) S.planes[lane >> 2][lane & 3] = a.ch_planes[(size_t)b * 20 + lane];
	if (lane <= M.nb) S.voff[lane] = M.vert_off[lane];
	WSYNC();
	const int g = lane & 15;
	for (int item0 = 0; item0 < 5 * M.nb; item0 += 4)
	{
		const int item = item0 + (lane >> 4);
		const bool on = item < 5 * M.nb;
		const int d = on ? item / M.nb : 0, rb = on ? item % M.nb : 0;
		v4 plane = V4(S.planes[d][0], S.planes[d][1], S.planes[d][2], S.planes[d][3]);
		const v4 bq = L4(S.q[rb]);
		v3 dirl = qrot(qconj(bq), xyz(plane));
		const float4 *vs = M.verts + S.voff[rb];
		const int nv = on ? S.voff[rb + 1] - S.voff[rb] : 0;
		int bi = 0x7fffffff; float bd = 0.0f;
		for (int k = g; k < nv; k += 64)
		{
			const int k1 = k + 16, k2 = k + 32, k3 = k + 48;
			const float4 q0 = vs[k], q1 = vs[k1 < nv ? k1 : k], q2 = vs[k2 < nv ? k2 : k], q3 = vs[k3 < nv ? k3 : k];
			const float d0 = dot(V3(q0.x, q0.y, q0.z), dirl), d1 = dot(V3(q1.x, q1.y, q1.z), dirl), d2 = dot(V3(q2.x, q2.y, q2.z), dirl), d3 = dot(V3(q3.x, q3.y, q3.z), dirl);
			if (bi == 0x7fffffff || bd < d0) { bd = d0; bi = k; }
			if (k1 < nv && bd < d1) { bd = d1; bi = k1; }
			if (k2 < nv && bd < d2) { bd = d2; bi = k2; }
			if (k3 < nv && bd < d3) { bd = d3; bi = k3; }
		}
#pragma unroll
		for (int st = 0; st < 4; st++)
		{
			float ob; int oi;
			if (st == 0) { ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bd), 0xB1, 0xF, 0xF, false)); oi = __builtin_amdgcn_update_dpp(0, bi, 0xB1, 0xF, 0xF, false); }
			else if (st == 1) { ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bd), 0x4E, 0xF, 0xF, false)); oi = __builtin_amdgcn_update_dpp(0, bi, 0x4E, 0xF, 0xF, false); }
			else if (st == 2) { ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bd), 0x141, 0xF, 0xF, false)); oi = __builtin_amdgcn_update_dpp(0, bi, 0x141, 0xF, 0xF, false); }
			else { ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bd), 0x140, 0xF, 0xF, false)); oi = __builtin_amdgcn_update_dpp(0, bi, 0x140, 0xF, 0xF, false); }
			const bool take = oi != 0x7fffffff && (bi == 0x7fffffff || bd < ob || (ob == bd && oi < bi));
			bd = take ? ob : bd; bi = take ? oi : bi;
		}
		if (on && g == 0)
		{
			float4 q = vs[bi];
			v3 sv = V3(q.x, q.y, q.z);
			v3 p0 = xyz(plane) * -plane.w, axis = -xyz(plane);
			// tab_to_world of k_chamber: pos + ((X * v.x + Y * v.y) + Z * v.z) with X, Y, Z the columns of qmat(q)
			const m3 rf = qmat(bq);
			const v3 world = L3(S.pos[rb]) + ((rf.x * sv.x + rf.y * sv.y) + rf.z * sv.z);
			float targetdist = dot(world - p0, axis);
			float row[HT_ROW];
			row[0] = -1.0f; row[1] = (float)rb; row[2] = p0.x; row[3] = p0.y; row[4] = p0.z; row[5] = sv.x; row[6] = sv.y; row[7] = sv.z;
			row[8] = axis.x; row[9] = axis.y; row[10] = axis.z; row[11] = targetdist; row[12] = 0.0f; row[13] = fmin_std(0.0f, a.ch_maxforce); row[14] = fmax_std(0.0f, a.ch_maxforce); row[15] = 0.0f;
			float4 *out = reinterpret_cast<float4 *>(a.rows_pre + ((size_t)b * 5 * M.nb + item) * HT_ROW);
			out[0] = make_float4(row[0], row[1], row[2], row[3]); out[1] = make_float4(row[4], row[5], row[6], row[7]);
			out[2] = make_float4(row[8], row[9], row[10], row[11]); out[3] = make_float4(row[12], row[13], row[14], row[15]);
			const int src = pre_base + item;
			if (src < noop_idx) prep_row_record(S, row, rb, ph.deltaT, scr + (size_t)src * CREC);
		}
	}
}

__global__ __launch_bounds__(PREP_THREADS) void k_solve_prep(ht_model_dev M, ht_physics_dev ph, prep_args a)
{
	__shared__ prep_lds S;
	const int b = blockIdx.x, t = threadIdx.x, lane = t & 63;
	const int role = __builtin_amdgcn_readfirstlane(t >> 6);
	if (a.active_flag && !a.active_flag[b]) return;
	const int nb = M.nb;
	float *T = a.tables + (size_t)b * TB_WORDS;
	// ---- the bodies, as k_solve's prologue loads them (rbinitvelocity physics.h:500-519) ----
	if (t < nb)
	{
		const float *s = a.state + ((size_t)b * nb + t) * HT_STATE_STRIDE;
		const float *bc = M.bodyc + t * HT_BC;
		v3 lin = V3(s[7], s[8], s[9]), ang = V3(s[10], s[11], s[12]);
		const float damp = bc[HT_BC_DAMPLEFT];
		lin = lin * damp; ang = ang * damp;
		lin = lin + V3(0, 0, 0); ang = ang + V3(0, 0, 0);
		for (int i = 0; i < 3; i++) S.pos[t][i] = s[i];
		for (int i = 0; i < 4; i++) S.q[t][i] = s[3 + i];
		S.lin4[t] = make_float4(lin.x, lin.y, lin.z, bc[HT_BC_MASSINV]);
		S.ang4[t] = make_float4(ang.x, ang.y, ang.z, bc[HT_BC_FRICTION]);
		m3 I = world_inertia(V4(s[3], s[4], s[5], s[6]), LM(bc + HT_BC_TINV), bc[HT_BC_MASSINV]);
		S.I4[t][0] = make_float4(I.x.x, I.x.y, I.x.z, 0.0f); S.I4[t][1] = make_float4(I.y.x, I.y.y, I.y.z, 0.0f); S.I4[t][2] = make_float4(I.z.x, I.z.y, I.z.z, 0.0f);
	}
	if (t == 64) { S.lin4[IDLE_BODY] = make_float4(0, 0, 0, 0); S.ang4[IDLE_BODY] = make_float4(0, 0, 0, 0); S.I4[IDLE_BODY][0] = S.I4[IDLE_BODY][1] = S.I4[IDLE_BODY][2] = make_float4(0, 0, 0, 0); S.nray = 0; }
	__syncthreads();

	// the single-body prefix of the row list: [landmark rays | boundary planes] then the cloud rows (physmodel.h:348, handtrack.h:672-683)
	const int rec_cap = a.scratch_stride - HT_SCRATCH_TAIL;
	float *scr = a.scratch + (size_t)b * a.scratch_stride * CREC;
	const int pre_base = M.pts_cap, noop_idx = rec_cap - 1;
	const bool chamber = a.ch_planes && a.ch_on[b];
	unsigned *const gidx = reinterpret_cast<unsigned *>(a.scratch + (size_t)a.batch * a.scratch_stride * (CREC + 1)) + (size_t)b * a.scratch_stride;
	float *const gG = a.scratch + (size_t)a.batch * a.scratch_stride * (CREC + 2) + (size_t)b * a.scratch_stride * 4;
	int ok = 1, na = 0;
	const long long t_p0 = HT_DBG(a.dbg, 2048) ? clock64() : 0;      // tuning builds: what every wave's part takes (header words 20-27 of the frame's table)
	// chain bookkeeping of wave 2, alive across the block's barrier
	int npre = 0, ncl = 0, n1 = 0, nlist = 0, c4_total = 0;
	bool idx_lds = false;

	const bool pose = (a.parts & 1) != 0, chains = (a.parts & 2) != 0;
	if (role == 0) { if (pose) prep_angular(M, ph, a, S, T, b, lane, ok, na); else ok = 0; }
	else if (role == 1) { if (pose) prep_joints(M, ph, S, T, lane); }
	else if (!chains) {}
	else if (role == 3)
	{
		if (a.n_pre && lane == 0) a.n_pre[b] = chamber ? 5 * nb : 0;
		if (chamber) prep_chamber_rows(M, ph, a, S, b, lane, scr, pre_base, noop_idx);
	}
	else
	{
		if (a.ray_rows) prep_rays(a, S, b, lane);
		npre = a.ray_rows ? S.nray : (chamber ? 5 * nb : 0);
		ncl = a.cloud_body ? a.n_cloud[b] : 0;
		n1 = npre + ncl;
		const int npad_max = 7 * (nb > 16 ? nb - 16 : 0);
		nlist = n1 + (npad_max > 3 * nb ? npad_max : 3 * nb) + (QUAD_CHAIN_SLACK > QUAD_BLOCK_SLACK ? QUAD_CHAIN_SLACK : QUAD_BLOCK_SLACK);
		idx_lds = nlist <= PREP_NIDX;
		for (int i = lane; i < nlist && i < a.scratch_stride; i += 64) gidx[i] = (unsigned)noop_idx;
		if (idx_lds) for (int i = lane; i < nlist; i += 64) S.cidx[i] = (unsigned short)noop_idx;
		if (lane == 0) quad_write_noop(scr + (size_t)noop_idx * CREC);
		auto body_of = [&](int i) -> int {
			if (i < npre) return a.ray_rows ? (int)S.ray[i][1] : i % nb;      // boundary-plane row i = plane i / nb on body i % nb
			return (int)a.cloud_body[(size_t)b * M.pts_cap + (i - npre)];
		};
		WSYNC();
		int mycnt = 0;                                     // lane bb counts the rows of body bb
		for (int base = 0; base < n1; base += 64)          // pass A: rows per body
		{
			const int i = base + lane;
			int body = (i < n1) ? body_of(i) : -1;
			unsigned long long todo = __ballot(body >= 0);
			while (todo)
			{
				const int leader = __ffsll((long long)todo) - 1;
				const int bb = __builtin_amdgcn_readlane(body, leader);
				const unsigned long long m = __ballot(body == bb);
				if (lane == bb) mycnt += __popcll(m);
				todo &= ~m;
			}
		}
		// the bodies' chains in blocks of four rows, dealt longest first to the four DPP rows of k_solve's walk (each body to the row with the fewest blocks so far)
		int c4_e0 = 0, c4_nblk = 0, c4_head = 0, c4_start = 0, c4_next = -1;
		const int myblk = lane < nb ? (mycnt + 3) >> 2 : 0;
		{
			int myrank = 0;
			for (int k = 0; k < nb; k++) { const int o = __builtin_amdgcn_readlane(myblk, k); myrank += (o > myblk || (o == myblk && k < lane)) ? 1 : 0; }
			if (lane < nb) S.corder[myrank] = lane;
			WSYNC();
			const int sorted = lane < nb ? S.corder[lane] : 0;
			int ld0 = 0, ld1 = 0, ld2 = 0, ld3 = 0, la0 = -1, la1 = -1, la2 = -1, la3 = -1, hd0 = 0, hd1 = 0, hd2 = 0, hd3 = 0, myrow = 0, mypos = 0;
			for (int r = 0; r < nb; r++)
			{
				const int bb = __builtin_amdgcn_readlane(sorted, r);
				const int blk = __builtin_amdgcn_readlane(myblk, bb);
				if (blk == 0) break;
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
			const int R = lane & 3;                        // header words: DPP row R of k_solve's walk
			c4_e0 = R == 0 ? 0 : R == 1 ? sg1 : R == 2 ? sg2 : sg3; c4_nblk = R == 0 ? ld0 : R == 1 ? ld1 : R == 2 ? ld2 : ld3; c4_head = R == 0 ? hd0 : R == 1 ? hd1 : R == 2 ? hd2 : hd3;
		}
		if (lane < HT_MAXNB) { reinterpret_cast<int *>(T + TB_CCNT)[lane] = myblk; reinterpret_cast<int *>(T + TB_CNEXT)[lane] = c4_next; }
		{
			int *H = reinterpret_cast<int *>(T + TB_HDR);
			if (lane < 4) { H[TH_E0 + lane] = c4_e0; H[TH_NBLK + lane] = c4_nblk; H[TH_HEAD + lane] = c4_head; }
			if (lane == 0) { H[TH_NPRE] = npre; H[TH_TOTAL] = c4_total; }
		}
		const int mystart = c4_start;
		int myrun = 0;
		for (int base = 0; base < n1; base += 64)          // pass B: placement in stable order; records of the landmark-ray rows
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
				const int src = i < npre ? pre_base + i : i - npre;
				if (src < noop_idx && dst < nlist - QUAD_CHAIN_SLACK)
				{
					if (dst < a.scratch_stride) gidx[dst] = (unsigned)src;
					if (idx_lds) S.cidx[dst] = (unsigned short)src;
					if (i < npre && a.ray_rows) prep_row_record(S, S.ray[i], body, ph.deltaT, scr + (size_t)src * CREC);      // (a boundary-plane row's record is wave 3's)
				}
			}
		}
	}
	if (HT_DBG(a.dbg, 2048) && lane == 0) reinterpret_cast<int *>(T + TB_HDR)[20 + role] = (int)(clock64() - t_p0);
	__threadfence_block();      // the records and lists are read back by other lanes
	__syncthreads();
	const long long t_p1 = HT_DBG(a.dbg, 2048) ? clock64() : 0;
	if (HT_DBG(a.dbg, 2048) && lane == 0 && role == 3) reinterpret_cast<int *>(T + TB_HDR)[24] = (int)(t_p1 - t_p0);
	if (role == 0)
	{
		int *H = reinterpret_cast<int *>(T + TB_HDR);
		if (lane == 0) { H[TH_OK] = ok; H[TH_NA] = na; H[TH_CHAIN_OK] = chains ? 1 : 0; }
	}
	if (role != 2 || !chains) return;
	// ---- wave 2, second half: the couplings of every block of four rows with the rows before them (k_solve's prologue: a quad per block, lane c takes slot c of the block's
	//      four records; -G(j,i) = -(c_j . d_i), the three lanes' shares summed (p0 + p1) + p2 on lane 2, which writes the rows' entries) ----
	{
		const int quad_ = lane >> 2, c_ = lane & 3;
		for (int blk0 = quad_; blk0 < c4_total; blk0 += 64)
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
					const float t_ = dpp<QP_PREV>(p) + p;
					return -(dpp<QP_PREV>(t_) + p);
				};
				const float g10 = coup(1, 0), g20 = coup(2, 0), g21 = coup(2, 1), g30 = coup(3, 0), g31 = coup(3, 1), g32 = coup(3, 2);
				if (c_ == 2 && blk < c4_total && 4 * blk + 3 < a.scratch_stride)
				{
					float2 *o = reinterpret_cast<float2 *>(gG + QUAD_G_BLOCK * blk);
					o[0] = make_float2(g30, g31); o[1] = make_float2(g32, g20); o[2] = make_float2(g21, 0.0f); o[3] = make_float2(g10, 0.0f); o[4] = make_float2(0.0f, 0.0f);
				}
			}
		}
	}
	if (HT_DBG(a.dbg, 2048) && lane == 0) reinterpret_cast<int *>(T + TB_HDR)[25] = (int)(clock64() - t_p1);
}
void ht_launch_solve_prep(const ht_model_dev &M, const ht_physics_dev &ph, const prep_args &a, int B, hipStream_t s)
{
	hipLaunchKernelGGL(k_solve_prep, dim3(B), dim3((a.parts & 2) ? PREP_THREADS : 128), 0, s, M, ph, a);      // the pose-only tables take waves 0 and 1
}
