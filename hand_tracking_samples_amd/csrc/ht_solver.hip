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
// Exact-order parallelism.  The reference applies rows strictly in vector order.  Rows with rb0 == NULL touch one body only, and
// such rows on different bodies commute exactly; they form a prefix of the row vector (chamber / landmark-ray rows, then cloud rows).
// That prefix is stably partitioned by body, pre-computed (lever arm r1 = R*position1, effective mass, limits*dt: all invariant during
// one PhysicsUpdate) and streamed from HBM/L2; lane b then walks the chain of body b with the body's momenta in registers.  The
// two-body tail (joint rows, contact triples, all angular rows) runs in reference order, wave-uniform, on LDS-resident body state.
// Everything is compiled -ffp-contract=off, so the result is the reference's bit for bit except for acos/sin/cos in a few row builders.
#include "ht_device.hpp"
#include "ht_launch.hpp"

#define SROW 12            // floats per pre-computed single-body row: r1[3] n[3] targetspeed tsnobias fmin*dt fmax*dt impulsed impulsesum
#define MAX2 (3 * HT_MAXNJ + 3 * 48)     // two-body linear rows kept in LDS (joints + 48 contacts)
#define L2W 20             // words per two-body linear row in LDS
#define MAXA 160           // angular rows kept in LDS
#define AW 12              // words per angular row in LDS


struct lds_t
{
	float pos[HT_MAXNB][3], q[HT_MAXNB][4], lin[HT_MAXNB][3], ang[HT_MAXNB][3], Iinv[HT_MAXNB][9], massinv[HT_MAXNB], friction[HT_MAXNB];
	float pos_next[HT_MAXNB][3], q_next[HT_MAXNB][4];
	float jr[HT_MAXNJ][6];                 // joint ranges after HandModelEnhancements
	float l2[MAX2][L2W];                   // rb0 rb1 r0[3] r1[3] n[3] ts tsnb fmin fmax impulsed impulsesum fm | pad
	float an[MAXA][AW];                    // rb0 rb1 axis[3] targetspin min*dt max*dt spintotorque torque mintorque | pad
	int cnt[HT_MAXNB], start[HT_MAXNB];
	int acount[HT_MAXNJ + 1];
	float ray[20][HT_ROW];
	int nray;
};

__device__ __forceinline__ v3 L3(const float *p) { return V3(p[0], p[1], p[2]); }
__device__ __forceinline__ v4 L4(const float *p) { return V4(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ void S3(float *p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ m3 LM(const float *p) { m3 m; m.x = V3(p[0], p[1], p[2]); m.y = V3(p[3], p[4], p[5]); m.z = V3(p[6], p[7], p[8]); return m; }
__device__ __forceinline__ xf body_xf(const lds_t &S, int b) { return XF(L3(S.pos[b]), L4(S.q[b])); }
__device__ __forceinline__ v3 anchor_world(const lds_t &S, int rb, v3 p) { return rb >= 0 ? apply(body_xf(S, rb), p) : p; }

// ---- angular row builders ----------------------------------------------------------------------
__device__ __forceinline__ void put_ang(float *o, int rb0, int rb1, v3 axis, float targetspin, float mintorque, float maxtorque)
{
	o[0] = __int_as_float(rb0); o[1] = __int_as_float(rb1); o[2] = axis.x; o[3] = axis.y; o[4] = axis.z; o[5] = targetspin; o[6] = mintorque; o[7] = maxtorque;
}
// ConstrainAngularRangeW physics.h:351-393; sin() there is the C double overload, the sums are formed in double and rounded once
__device__ int angular_range_w(const ht_physics_dev &ph, int rb0, v4 jb0, int rb1, v4 jf1, v3 lmin, v3 lmax, float (*out)[8])
{
	int n = 0;
	const float dt = ph.deltaT;
	v3 jmin = (lmin * 3.14f) / 180.0f, jmax = (lmax * 3.14f) / 180.0f;
	if (jmin.x == 0 && jmax.x == 0 && jmin.z < jmax.z)
	{
		v4 cb = normalize(V4(0, -1, 0, 1));
		jb0 = qmul(jb0, cb); jf1 = qmul(jf1, cb);
		v3 nmin = V3(lmin.z, lmin.y, 0), nmax = V3(lmax.z, lmax.y, 0);
		lmin = nmin; lmax = nmax;
		jmin = (lmin * 3.14f) / 180.0f; jmax = (lmax * 3.14f) / 180.0f;
		// (the recursion of the reference can fire at most once more only if the swapped x-range is again 0 with z<..., z is now 0: impossible)
	}
	v4 r = qmul(qconj(jb0), jf1);
	v4 s = quat_from_to(V3(0, 0, 1.0f), qzdir(r));
	v4 t = qmul(qconj(s), r);
	if (jmax.x == jmin.x)
		put_ang(out[n++], rb0, rb1, qxdir(jf1), (float)(2 * ((double)(-s.x) + sin((double)(jmin.x / 2.0f))) / (double)dt), -FLT_MAX, FLT_MAX);
	else if (jmax.x - jmin.x < 360.0f * 3.14f / 180.0f)
	{
		put_ang(out[n++], rb0, rb1, qxdir(jf1), (float)(2 * ((double)(-s.x) + sin((double)(jmin.x / 2.0f))) / (double)dt), 0, FLT_MAX);
		put_ang(out[n++], rb0, rb1, -qxdir(jf1), (float)(2 * ((double)(s.x) - sin((double)(jmax.x / 2.0f))) / (double)dt), 0, FLT_MAX);
	}
	if (jmax.y == jmin.y)
		put_ang(out[n++], rb0, rb1, qydir(jf1), ph.biasfactorjoint * 2 * (-s.y + jmin.y) / dt, -FLT_MAX, FLT_MAX);
	else
	{
		put_ang(out[n++], rb0, rb1, qydir(jf1), (float)(2 * ((double)(-s.y) + sin((double)(jmin.y / 2.0f))) / (double)dt), 0, FLT_MAX);
		put_ang(out[n++], rb0, rb1, -qydir(jf1), (float)(2 * ((double)(s.y) - sin((double)(jmax.y / 2.0f))) / (double)dt), 0, FLT_MAX);
	}
	if (jmin.z == jmax.z)
		put_ang(out[n++], rb0, rb1, qzdir(jf1), ph.biasfactorjoint * 2 * -t.z / dt, -FLT_MAX, FLT_MAX);
	else
	{
		put_ang(out[n++], rb0, rb1, qzdir(jf1), (float)(2 * ((double)(-t.z) + sin((double)(jmin.z / 2.0f))) / (double)dt), 0, FLT_MAX);
		put_ang(out[n++], rb0, rb1, -qzdir(jf1), (float)(2 * ((double)(t.z) - sin((double)(jmax.z / 2.0f))) / (double)dt), 0, FLT_MAX);
	}
	return n;
}
// ConstrainConeAngle physics.h:402-414
__device__ void cone_angle(const ht_physics_dev &ph, const lds_t &S, int rb0, v3 n0, int rb1, v3 n1, float limitangle_degrees, float *out)
{
	int equality = (limitangle_degrees == 0);
	v3 a0 = rb0 >= 0 ? qrot(L4(S.q[rb0]), n0) : n0;
	v3 a1 = rb1 >= 0 ? qrot(L4(S.q[rb1]), n1) : n1;
	v3 axis = safenormalize(cross(a1, a0));
	float rbangle = acos_f(clamp_std(dot(a0, a1), 0.0f, 1.0f));
	float dangle = rbangle - (limitangle_degrees) * 3.14f / 180.0f;
	float targetspin = ((equality) ? ph.biasfactorjoint : 1.0f) * dangle / ph.deltaT;
	put_ang(out, rb0, rb1, axis, targetspin, (limitangle_degrees > 0.0f) ? 0 : -FLT_MAX, FLT_MAX);
}
// ConstrainAngularDrive physics.h:313-326
__device__ void angular_drive(const ht_physics_dev &ph, const lds_t &S, int rb0, int rb1, v4 target, float maxtorque, float (*out)[8])
{
	v4 q0 = rb0 >= 0 ? L4(S.q[rb0]) : V4(0, 0, 0, 1), q1 = rb1 >= 0 ? L4(S.q[rb1]) : V4(0, 0, 0, 1);
	v4 dq = qmul(q1, qconj(qmul(q0, target)));
	if (dq.w < 0) dq = -dq;
	v3 axis = safenormalize(xyz(dq));
	v3 binormal = orth(axis);
	v3 normal = cross(axis, binormal);
	put_ang(out[0], rb0, rb1, axis, -ph.biasfactorjoint * (acos_f(clamp_std(dq.w, -1.0f, 1.0f)) * 2.0f) / ph.deltaT, -maxtorque, maxtorque);
	put_ang(out[1], rb0, rb1, binormal, 0, -maxtorque, maxtorque);
	put_ang(out[2], rb0, rb1, normal, 0, -maxtorque, maxtorque);
}

// landmark feature points, handtrack.h:77-81
__constant__ int FEATURE_BONE[8] = { 1, 1, 1, 4, 7, 10, 13, 16 };
__constant__ float FEATURE_OFF[8][3] = { { 0, 0, 0 }, { -0.03f, 0, -0.03f }, { 0.03f, 0, -0.03f }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };

// ---- two-body row maths ------------------------------------------------------------------------
__device__ __forceinline__ v3 spin_of(const lds_t &S, int b) { return mul(LM(S.Iinv[b]), L3(S.ang[b])); }       // physics.h:126
__device__ __forceinline__ void linear_precompute(const ht_physics_dev &ph, const lds_t &S, float *w, int rb0, int rb1, v3 p0, v3 p1, v3 n, float targetdist, float tsnb, float fmin, float fmax, int fm)
{
	v3 r0 = rb0 >= 0 ? qrot(L4(S.q[rb0]), p0) : p0;
	v3 r1 = rb1 >= 0 ? qrot(L4(S.q[rb1]), p1) : p1;
	float impulsed = ((rb0 >= 0) ? S.massinv[rb0] + dot(cross(mul(LM(S.Iinv[rb0]), cross(r0, n)), r0), n) : 0)
	               + ((rb1 >= 0) ? S.massinv[rb1] + dot(cross(mul(LM(S.Iinv[rb1]), cross(r1, n)), r1), n) : 0);
	w[0] = __int_as_float(rb0); w[1] = __int_as_float(rb1); S3(w + 2, r0); S3(w + 5, r1); S3(w + 8, n);
	w[11] = targetdist / ph.deltaT; w[12] = tsnb; w[13] = fmin * ph.deltaT; w[14] = fmax * ph.deltaT; w[15] = impulsed; w[16] = 0.0f; w[17] = __int_as_float(fm);
}

// ------------------------------------------------------------------------------------------------- k_solve
__global__ __launch_bounds__(64) void k_solve(ht_model_dev M, ht_physics_dev ph, solve_args a)
{
	__shared__ lds_t S;
	const int b = blockIdx.x, lane = threadIdx.x;
	if (a.active_flag && !a.active_flag[b]) return;
	const int nb = M.nb, nj = M.nj;
	float *st = a.state + (size_t)b * nb * HT_STATE_STRIDE;
	const float dt = ph.deltaT;

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
		S3(S.lin[lane], lin); S3(S.ang[lane], ang);
		S.massinv[lane] = bc[HT_BC_MASSINV]; S.friction[lane] = bc[HT_BC_FRICTION];
		m3 I = world_inertia(V4(s[3], s[4], s[5], s[6]), LM(bc + HT_BC_TINV), bc[HT_BC_MASSINV]);
		S.Iinv[lane][0] = I.x.x; S.Iinv[lane][1] = I.x.y; S.Iinv[lane][2] = I.x.z; S.Iinv[lane][3] = I.y.x; S.Iinv[lane][4] = I.y.y; S.Iinv[lane][5] = I.y.z; S.Iinv[lane][6] = I.z.x; S.Iinv[lane][7] = I.z.y; S.Iinv[lane][8] = I.z.z;
	}
	if (lane < nj) for (int i = 0; i < 6; i++) S.jr[lane][i] = M.jointc[lane * HT_JC + HT_JC_RMIN + i];
	if (lane == 0) S.nray = 0;
	__syncthreads();

	// ---- HandModelEnhancements (handtrack.h:417-420, 434-440); acos()/cos() are the C double overloads there ----
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
			const int kb[4] = { 14, 11, 8, 5 }; const float r0[4] = { -30.0f, -10.0f, -10.0f, -10.0f }, r1[4] = { 10.0f, 10.0f, 10.0f, 20.0f };
			const int k = lane - 4;
			bool up = (double)dot(qydir(L4(S.q[1])), qydir(L4(S.q[kb[k]]))) > ph.cos40d;
			S.jr[kb[k] - 1][1] = up ? r0[k] : -0.0f;
			S.jr[kb[k] - 1][4] = up ? r1[k] : 0.0f;
		}
	}
	__syncthreads();

	// ---- angular rows: [ApplyAngles 12] [arm cone 1] [joint ranges] ----
	int na_pre = 0;
	if (a.apply_angles || a.arm_cone)
	{
		if (lane == 0)
		{
			float tmp[13][8];
			int k = 0;
			const float *cam = a.cams + (size_t)b * HT_CAM;
			const v4 camq = V4(cam[8], cam[9], cam[10], cam[11]);
			if (a.apply_angles)
			{
				const float *an = a.analysis + (size_t)b * HT_ANALYSIS;
				const v4 palmq = V4(an[HT_AN_PALMQ], an[HT_AN_PALMQ + 1], an[HT_AN_PALMQ + 2], an[HT_AN_PALMQ + 3]);
				const float *fc = an + HT_AN_CLENCH;
				angular_drive(ph, S, -1, 1, qmul(camq, palmq), a.drive_force, tmp); k = 3;                              // handtrack.h:206
				float th = fc[0];
				cone_angle(ph, S, 1, V3((float)cos((double)th), 0, (float)sin((double)th)), 4, V3(0, 0, 1), 10.0f, tmp[k++]);
				for (int finger = 1; finger <= 4; finger++)
				{
					float aa = fc[finger];
					cone_angle(ph, S, 1, V3(0, (float)(-sin((double)aa)), (float)cos((double)aa)), 3 + finger * 3, V3(0, 0, 1), 10.0f, tmp[k++]);
					v4 jf = L4(M.jointc + (1 + finger * 3) * HT_JC + HT_JC_FRAME);
					v3 inner = V3(0, (float)(-sin((double)(aa / 2.0f))), (float)cos((double)(aa / 2.0f)));
					cone_angle(ph, S, 1, qrot(jf, qrot(jf, inner)), 2 + finger * 3, V3(0, 0, 1), 10.0f, tmp[k++]);
				}
			}
			if (a.arm_cone) cone_angle(ph, S, -1, qrot(camq, V3(0, -1, 0)), 0, V3(0, 0, 1), 70.0f, tmp[k++]);             // handtrack.h:426, 684
			for (int i = 0; i < k; i++) for (int j = 0; j < 8; j++) S.an[i][j] = tmp[i][j];
			S.acount[HT_MAXNJ] = k;
		}
		__syncthreads();
		na_pre = S.acount[HT_MAXNJ];
	}
	{
		float jrows[6][8];
		int n = 0;
		if (lane < nj)
		{
			const float *jc = M.jointc + lane * HT_JC;
			const int rb0 = (int)jc[HT_JC_RB0], rb1 = (int)jc[HT_JC_RB1];
			const v4 jf = L4(jc + HT_JC_FRAME);
			n = angular_range_w(ph, rb0, rb0 >= 0 ? qmul(L4(S.q[rb0]), jf) : jf, rb1, rb1 >= 0 ? L4(S.q[rb1]) : V4(0, 0, 0, 1), L3(S.jr[lane]), L3(S.jr[lane] + 3), jrows);
			S.acount[lane] = n;
		}
		__syncthreads();
		int off = na_pre;
		for (int j = 0; j < lane && j < nj; j++) off += S.acount[j];
		if (lane < nj) for (int i = 0; i < n && off + i < MAXA; i++) for (int k = 0; k < 8; k++) S.an[off + i][k] = jrows[i][k];
	}
	__syncthreads();
	int na = na_pre;
	for (int j = 0; j < nj; j++) na += S.acount[j];
	if (na > MAXA) na = MAXA;
	// pre-compute per angular row: min*dt, max*dt, spintotorque (physics.h:256-259); Iinv is invariant during the update
	for (int i = lane; i < na; i += 64)
	{
		float *w = S.an[i];
		const int rb0 = __float_as_int(w[0]), rb1 = __float_as_int(w[1]);
		const v3 axis = L3(w + 2);
		const float mintorque = w[6], maxtorque = w[7];
		float spintotorque = 1.0f / (((rb0 >= 0) ? dot(axis, mul(LM(S.Iinv[rb0]), axis)) : 0.0f) + ((rb1 >= 0) ? dot(axis, mul(LM(S.Iinv[rb1]), axis)) : 0.0f));
		w[6] = mintorque * dt; w[7] = maxtorque * dt; w[8] = spintotorque; w[9] = 0.0f; w[10] = mintorque;
	}

	// ---- two-body linear rows: joints (physmodel.h:328-334) then contacts (physics.h:463-489) ----
	if (lane < nj)
	{
		const float *jc = M.jointc + lane * HT_JC;
		const int rb0 = (int)jc[HT_JC_RB0], rb1 = (int)jc[HT_JC_RB1];
		const v3 p0 = L3(jc + HT_JC_P0) - L3(M.bodyc + rb0 * HT_BC + HT_BC_COM), p1 = L3(jc + HT_JC_P1) - L3(M.bodyc + rb1 * HT_BC + HT_BC_COM);
		const v3 d = anchor_world(S, rb1, p1) - anchor_world(S, rb0, p0);
		linear_precompute(ph, S, S.l2[3 * lane + 0], rb0, rb1, p0, p1, V3(1, 0, 0), d.x, 0.0f, -FLT_MAX, FLT_MAX, 0);
		linear_precompute(ph, S, S.l2[3 * lane + 1], rb0, rb1, p0, p1, V3(0, 1, 0), d.y, 0.0f, -FLT_MAX, FLT_MAX, 0);
		linear_precompute(ph, S, S.l2[3 * lane + 2], rb0, rb1, p0, p1, V3(0, 0, 1), d.z, 0.0f, -FLT_MAX, FLT_MAX, 0);
	}
	int nc = (a.contacts && ph.use_collision) ? a.ncontacts[b] : 0;
	if (nc > 48) nc = 48;
	if (lane < nc)
	{
		const float *c = a.contacts + ((size_t)b * HT_MAXCONTACT + lane) * HT_CONTACT;
		const int rb0 = (int)c[0], rb1 = (int)c[1];
		const v3 normal = L3(c + 2), p0w = L3(c + 5), p1w = L3(c + 8);
		const float separation = c[11];
		const v3 p0 = apply(inverse(body_xf(S, rb0)), p0w), p1 = apply(inverse(body_xf(S, rb1)), p1w);        // PhysContact physics.h:431-432
		const v3 r0 = p0w - L3(S.pos[rb0]), r1 = p1w - L3(S.pos[rb1]);
		const v3 v0 = cross(spin_of(S, rb0), r0) + L3(S.lin[rb0]) * S.massinv[rb0];
		const v3 v1 = cross(spin_of(S, rb1), r1) + L3(S.lin[rb1]) * S.massinv[rb1];
		const v3 v = v0 - v1;
		const float minsep = ph.driftmax * 0.25f;
		const float bouncevel = fmax_std(0.0f, (-dot(normal, v) - ph.gravity_len * ph.falltime_to_ballistic) * ph.restitution);
		float *w = S.l2[3 * nj + 3 * lane];
		linear_precompute(ph, S, w, rb0, rb1, p0, p1, -normal, fmin_std((separation - minsep) * ph.biasfactorpositive, separation), -bouncevel, 0, FLT_MAX, 0);
		v4 q = quat_from_to(V3(0, 0, 1), -normal);
		v3 tangent = qxdir(q), binormal = qydir(q);
		linear_precompute(ph, S, w + L2W, rb0, rb1, p0, p1, binormal, 0, 0, 0, 0, -1);
		linear_precompute(ph, S, w + 2 * L2W, rb0, rb1, p0, p1, tangent, 0, 0, 0, 0, -2);
	}
	const int n2 = 3 * nj + 3 * nc;

	// ---- landmark-ray rows of MultiStepSim (handtrack.h:666-676): 2 dead-zone pairs per open finger ----
	if (a.ray_rows && lane == 0)
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

	// ---- single-body prefix: [ray rows | chamber rows] then cloud rows; stable partition by body + pre-compute -> scratch ----
	const int npre_g = a.rows_pre ? a.n_pre[b] : 0;
	const int npre = a.ray_rows ? S.nray : npre_g;
	const int ncl = a.rows_cloud ? a.n_cloud[b] : 0;
	const int n1 = npre + ncl;
	float *scr = a.scratch + (size_t)b * a.scratch_stride * SROW;
	auto row_ptr = [&](int i) -> const float * {
		if (i < npre) return a.ray_rows ? S.ray[i] : a.rows_pre + ((size_t)b * a.pre_stride + i) * HT_ROW;
		return a.rows_cloud + ((size_t)b * HT_MAXPTS + (i - npre)) * HT_ROW;
	};
	int mycnt = 0;                                     // lane bb counts the rows of body bb
	for (int base = 0; base < n1; base += 64)          // pass A: rows per body
	{
		const int i = base + lane;
		int body = (i < n1) ? (int)row_ptr(i)[1] : -1;
		unsigned long long todo = __ballot(body >= 0);
		while (todo)
		{
			const int leader = __ffsll((long long)todo) - 1;
			const int bb = __shfl(body, leader);
			const unsigned long long m = __ballot(body == bb);
			if (lane == bb) mycnt += __popcll(m);
			todo &= ~m;
		}
	}
	int mystart = mycnt;                               // exclusive prefix over lanes = segment start of body `lane`
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(mystart, o); if (lane >= o) mystart += v; }
	mystart -= mycnt;
	if (lane < nb) { S.cnt[lane] = mycnt; S.start[lane] = mystart; }
	int myrun = 0;
	for (int base = 0; base < n1; base += 64)          // pass B: placement in stable order + pre-compute
	{
		const int i = base + lane;
		const float *r = (i < n1) ? row_ptr(i) : nullptr;
		int body = r ? (int)r[1] : -1;
		int dst = -1;
		unsigned long long todo = __ballot(body >= 0);
		while (todo)
		{
			const int leader = __ffsll((long long)todo) - 1;
			const int bb = __shfl(body, leader);
			const unsigned long long m = __ballot(body == bb);
			const int segbase = __shfl(mystart + myrun, bb);
			if (body == bb) dst = segbase + __popcll(m & ((1ull << lane) - 1ull));
			if (lane == bb) myrun += __popcll(m);
			todo &= ~m;
		}
		if (r && dst < a.scratch_stride)
		{
			const v3 p1 = L3(r + 5), n = L3(r + 8);
			const v3 r1 = qrot(L4(S.q[body]), p1);
			const float impulsed = S.massinv[body] + dot(cross(mul(LM(S.Iinv[body]), cross(r1, n)), r1), n);       // 0 + (...) for the NULL side
			float4 *o = reinterpret_cast<float4 *>(scr + (size_t)dst * SROW);
			o[0] = make_float4(r1.x, r1.y, r1.z, n.x);
			o[1] = make_float4(n.y, n.z, r[11] / dt, r[12]);
			o[2] = make_float4(r[13] * dt, r[14] * dt, impulsed, 0.0f);
		}
	}
	__syncthreads();

	// ---- Gauss-Seidel sweeps ----
	const int total_sweeps = ph.iterations + ph.iterations_post;
	for (int sweep = 0; sweep < total_sweeps; sweep++)
	{
		const bool post = sweep >= ph.iterations;
		// chains: lane b applies the rows of body b in order
		if (lane < nb && mycnt > 0)
		{
			v3 lin = L3(S.lin[lane]), ang = L3(S.ang[lane]);
			const m3 I = LM(S.Iinv[lane]);
			const float minv = S.massinv[lane];
			float *rp = scr + (size_t)mystart * SROW;
			const int cnt = mycnt;
			float4 c0 = reinterpret_cast<float4 *>(rp)[0], c1 = reinterpret_cast<float4 *>(rp)[1], c2 = reinterpret_cast<float4 *>(rp)[2];
			for (int k = 0; k < cnt; k++)
			{
				float4 n0 = c0, n1 = c1, n2 = c2;
				if (k + 1 < cnt) { const float4 *nx = reinterpret_cast<const float4 *>(rp + (size_t)(k + 1) * SROW); n0 = nx[0]; n1 = nx[1]; n2 = nx[2]; }      // prefetch
				const v3 r1 = V3(c0.x, c0.y, c0.z), n = V3(c0.w, c1.x, c1.y);
				const float ts = post ? fmin_std(c1.z, c1.w) : c1.z;                      // RemoveBias physics.h:288
				const v3 v1 = cross(mul(I, ang), r1) + lin * minv;
				const float vn = dot(v1, n);
				const float impulsen = -ts - vn;
				float impulse = impulsen / c2.z;
				impulse = fmin_std(c2.y - c2.w, impulse);
				impulse = fmax_std(c2.x - c2.w, impulse);
				const v3 imp = n * impulse;
				lin = lin + imp; ang = ang + cross(r1, imp);
				rp[(size_t)k * SROW + 11] = c2.w + impulse;
				c0 = n0; c1 = n1; c2 = n2;
			}
			S3(S.lin[lane], lin); S3(S.ang[lane], ang);
		}
		__syncthreads();
		// two-body linear rows, reference order, wave-uniform
		for (int i = 0; i < n2; i++)
		{
			float *w = S.l2[i];
			const int rb0 = __float_as_int(w[0]), rb1 = __float_as_int(w[1]), fm = __float_as_int(w[17]);
			float fmn = w[13], fmx = w[14];
			if (fm)
			{
				const float master = S.l2[i + fm][16];
				const float lim = fmax_std(((rb0 >= 0) ? S.friction[rb0] : 0), ((rb1 >= 0) ? S.friction[rb1] : 0)) * master / dt;       // physics.h:292
				fmx = lim * dt; fmn = (-lim) * dt;
			}
			const v3 r0 = L3(w + 2), r1 = L3(w + 5), n = L3(w + 8);
			const float ts = post ? fmin_std(w[11], w[12]) : w[11];
			const v3 v0 = (rb0 >= 0) ? cross(spin_of(S, rb0), r0) + L3(S.lin[rb0]) * S.massinv[rb0] : V3(0, 0, 0);
			const v3 v1 = (rb1 >= 0) ? cross(spin_of(S, rb1), r1) + L3(S.lin[rb1]) * S.massinv[rb1] : V3(0, 0, 0);
			const float vn = dot(v1 - v0, n);
			const float impulsen = -ts - vn;
			float impulse = impulsen / w[15];
			impulse = fmin_std(fmx - w[16], impulse);
			impulse = fmax_std(fmn - w[16], impulse);
			if (rb0 >= 0) { const v3 imp = n * -impulse; S3(S.lin[rb0], L3(S.lin[rb0]) + imp); S3(S.ang[rb0], L3(S.ang[rb0]) + cross(r0, imp)); }
			if (rb1 >= 0) { const v3 imp = n * impulse; S3(S.lin[rb1], L3(S.lin[rb1]) + imp); S3(S.ang[rb1], L3(S.ang[rb1]) + cross(r1, imp)); }
			w[16] = w[16] + impulse;
		}
		// angular rows
		for (int i = 0; i < na; i++)
		{
			float *w = S.an[i];
			float targetspin = w[5];
			if (post) targetspin = (w[10] < 0) ? 0 : fmin_std(targetspin, 0.0f);            // RemoveBias physics.h:250
			if (targetspin == -FLT_MAX) continue;
			const int rb0 = __float_as_int(w[0]), rb1 = __float_as_int(w[1]);
			const v3 axis = L3(w + 2);
			const float currentspin = ((rb1 >= 0) ? dot(spin_of(S, rb1), axis) : 0.0f) - ((rb0 >= 0) ? dot(spin_of(S, rb0), axis) : 0.0f);
			const float dspin = targetspin - currentspin;
			float dtorque = dspin * w[8];
			dtorque = fmin_std(dtorque, w[7] - w[9]);
			dtorque = fmax_std(dtorque, w[6] - w[9]);
			if (rb0 >= 0) S3(S.ang[rb0], L3(S.ang[rb0]) - axis * dtorque);
			if (rb1 >= 0) S3(S.ang[rb1], L3(S.ang[rb1]) + axis * dtorque);
			w[9] = w[9] + dtorque;
		}
		__syncthreads();
		if (sweep + 1 == ph.iterations && lane < nb)
		{
			// rbcalcnextpose physics.h:522-531 with rkupdateq :211-218 (momentum-preserving RK4 on the quaternion)
			const float *bc = M.bodyc + lane * HT_BC;
			const float minv = S.massinv[lane];
			const v3 pn = L3(S.pos[lane]) + (L3(S.lin[lane]) * minv) * dt;
			const m3 tinv = LM(bc + HT_BC_TINV) * minv;
			const v3 angm = L3(S.ang[lane]);
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
			S3(S.pos_next[lane], pn);
			S.q_next[lane][0] = o.x; S.q_next[lane][1] = o.y; S.q_next[lane][2] = o.z; S.q_next[lane][3] = o.w;
		}
		__syncthreads();
	}

	// ---- rbupdatepose (physics.h:533-541), SanityCheck (physmodel.h:437-442), optional momentum reset (handtrack.h:686-687) ----
	if (lane < nb)
	{
		float *s = st + lane * HT_STATE_STRIDE;
		const float *bc = M.bodyc + lane * HT_BC;
		v3 pos = L3(S.pos_next[lane]); v4 q = L4(S.q_next[lane]);
		v3 lin = L3(S.lin[lane]), ang = L3(S.ang[lane]);
		const bool bad = isnan(lin.x) || isnan(lin.y) || isnan(lin.z) || isnan(pos.x) || isnan(pos.y) || isnan(pos.z) || isnan(ang.x) || isnan(ang.y) || isnan(ang.z)
		              || isnan(q.x) || isnan(q.y) || isnan(q.z) || isnan(q.w);
		if (bad) { pos = L3(bc + HT_BC_POS0); q = L4(bc + HT_BC_Q0); lin = V3(0, 0, 0); ang = V3(0, 0, 0); }
		if (a.zero_momenta) { lin = V3(0, 0, 0); ang = V3(0, 0, 0); }
		s[0] = pos.x; s[1] = pos.y; s[2] = pos.z; s[3] = q.x; s[4] = q.y; s[5] = q.z; s[6] = q.w;
		s[7] = lin.x; s[8] = lin.y; s[9] = lin.z; s[10] = ang.x; s[11] = ang.y; s[12] = ang.z;
	}
}

void ht_launch_solve(const ht_model_dev &M, const ht_physics_dev &ph, const solve_args &a, int B, hipStream_t s)
{
	hipLaunchKernelGGL(k_solve, dim3(B), dim3(64), 0, s, M, ph, a);
}
