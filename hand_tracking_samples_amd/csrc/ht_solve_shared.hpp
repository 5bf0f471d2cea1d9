// ht_solve_shared.hpp -- what k_solve (ht_solver.hip) and k_solve_prep (ht_prep.hip) share: the layouts of a two-body linear group and of an angular record, the row builders
// (ConstrainAngularRange(W) / ConeAngle / AngularDrive, physics.h:313-414), and the per-frame SOLVE TABLES k_solve_prep hands to k_solve (round 6).
#pragma once
#include "ht_device.hpp"
#include "ht_launch.hpp"

// A two-body linear GROUP = the 3 consecutive rows of a joint (x, y, z) or of a contact (normal + 2 friction rows): same two bodies.  64 floats:
#define LGRP 64
#define LG_S 0             // 3 x float4: targetspeed, targetspeed after RemoveBias, fmin*dt, fmax*dt (friction rows: fmax slot = mu)
#define LG_RINV 12         // 3: 1 / effective mass
#define LG_META 15         // flags | rb0 | rb1 << 8
#define LG_SUM 16          // 3: impulse sums (the only words a sweep writes)
#define LG_N 19            // 9: row direction n_k[c] at LG_N + 3k + c
#define LG_GB 28           // 36: (g, b) of row k, side s, component c at LG_GB + ((2k + s)*3 + c)*2; side 0 (rb0) carries its minus sign
// An angular row, 16 floats:
#define AROW 16
#define AR_S 0             // float4: targetspin, targetspin after RemoveBias, mintorque*dt, maxtorque*dt
#define AR_GAIN 4          // 1 / (axis.Iinv0.axis + axis.Iinv1.axis); 0 for a disabled row (physics.h:252)
#define AR_TORQUE 5        // accumulated torque (the only word a sweep writes)
#define AR_AXIS 6          // 3, then the gain of the sweeps after RemoveBias
#define AR_BA 10           // 6: -(Iinv0*axis), then Iinv1*axis
// Angular rows are built in registers: row r by lane r % 64, slot r / 64.  AS = slots per lane is a build parameter: 2 (up to 126 rows: the stock hand has 13 CNN-driven +
// 71 of its joints) for the tile builds, 4 (up to 252 rows) for the build that serves any model ht_create accepts -- 13 + 6 rows per joint + slowfit's 3 relative rows per
// joint = 13 + 9 * 26 joints = 247 (ht_launch_solve picks it from the bound the caller states; ht_create refuses more than 26 joints).
#define MAXA2_OF(AS) (64 * (AS))
#define MAXA_CAP_OF(AS) ((AS) == 2 ? 126 : 252)
#define MAXA_RUNS 128      // runs of consecutive angular rows on one body pair a solve schedules (a joint's rows are one run: 13 + 2 per joint; beyond: counted, dropped)
#define MAXG (HT_MAXNJ + HT_MAXCONTACT_LDS + 1)      // groups the level schedule has LDS tables for: every joint, 96 contacts, the idle group
#define MAXG_CAP (32 + HT_MAXNJ + HT_MAXCONTACT + 1)      // groups a frame can have (their records in the tail of its scratch slot when they exceed the build's LDS pool): a caller's 32, every joint,
                                                          // every contact the contact kernel keeps, the idle group.  Beyond MAXG - 1 groups there is no level schedule: one group per step, in row order
// LDS per frame: ~5 KB of body state and schedule tables, a 5 KB union of prologue scratch and the angular records, and three arrays whose size is
// the build's choice -- the two-body linear groups (256 B each), the impulse sums of the single-body rows (4 B each), the angular records (64 B each).
// A frame whose rows do not fit an array keeps THAT array in its slot of the solver scratch in HBM instead (same code through a generic pointer): slower
// for that frame, correct for every frame, one launch.  Builds (ht_launch_solve):
//   small   34 groups (16 joints + 17 contacts), 584 sums, 84 angular rows (13 CNN-driven + 71 of the hand's joints), chain lists in HBM: 20 KB = 40 LDS
//           allocation units of 512 B, EIGHT frames per CU (2048 frames fill the GPU in one round, 8192 in four).  Batches above 1024 frames of 64x64 tiles.
//   only    66 groups (16 joints + 49 contacts), 1024 sums and chain entries, 126 angular rows: 36 KB, four frames per CU (a 1024-frame batch in one round).
//   mid     71 groups, 1520 sums and chain entries, 126 angular rows: 40 KB, four frames per CU.  Larger models, full-size frames.
#define IDLE_BODY (HT_MAXNB - 1)      // lane pairs without a row in a step work on this all-zero body and on an all-zero record
#define LM_FRIC 0x10000    // meta bits of a group: contact (friction rows limited by the normal row's impulse sum, physics.h:292)
#define LM_NORMAL 0x20000

#include "ht_quad.hpp"
#include "ht_block.hpp"

__device__ __forceinline__ v3 L3(const float *p) { return V3(p[0], p[1], p[2]); }
__device__ __forceinline__ v4 L4(const float *p) { return V4(p[0], p[1], p[2], p[3]); }
__device__ __forceinline__ void S3(float *p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
__device__ __forceinline__ m3 LM(const float *p) { m3 m; m.x = V3(p[0], p[1], p[2]); m.y = V3(p[3], p[4], p[5]); m.z = V3(p[6], p[7], p[8]); return m; }
template <class LDS> __device__ __forceinline__ xf body_xf(const LDS &S, int b) { return XF(L3(S.pos[b]), L4(S.q[b])); }
__device__ __forceinline__ v3 F3(float4 f) { return V3(f.x, f.y, f.z); }
template <class LDS> __device__ __forceinline__ m3 body_I(const LDS &S, int b) { m3 m; m.x = F3(S.I4[b][0]); m.y = F3(S.I4[b][1]); m.z = F3(S.I4[b][2]); return m; }
template <class LDS> __device__ __forceinline__ v3 anchor_world(const LDS &S, int rb, v3 p) { return rb >= 0 ? apply(body_xf(S, rb), p) : p; }

// ---- angular row builders ----------------------------------------------------------------------
__device__ __forceinline__ void put_ang(float *o, int rb0, int rb1, v3 axis, float targetspin, float mintorque, float maxtorque)
{
	o[0] = __int_as_float(rb0); o[1] = __int_as_float(rb1); o[2] = axis.x; o[3] = axis.y; o[4] = axis.z; o[5] = targetspin; o[6] = mintorque; o[7] = maxtorque;
}
// Row builders that can emit several rows take a sink that keeps only the row its lane owns: nothing is indexed dynamically, so the
// rows stay in registers (a private array indexed with a run-time value would live in scratch memory).
struct ang_sink { int want, n; float row[8]; };
__device__ __forceinline__ void emit_ang(ang_sink &k, int rb0, int rb1, v3 axis, float targetspin, float mintorque, float maxtorque)
{
	if (k.n == k.want) put_ang(k.row, rb0, rb1, axis, targetspin, mintorque, maxtorque);
	k.n++;
}
// ConstrainAngularRangeW physics.h:351-393; sin() there is the C double overload, the sums are formed in double and rounded once
__device__ __forceinline__ void angular_range_w(const ht_physics_dev &ph, int rb0, v4 jb0, int rb1, v4 jf1, v3 lmin, v3 lmax, ang_sink &out)
{
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
		emit_ang(out, rb0, rb1, qxdir(jf1), (float)(2 * ((double)(-s.x) + sin((double)(jmin.x / 2.0f))) / (double)dt), -FLT_MAX, FLT_MAX);
	else if (jmax.x - jmin.x < 360.0f * 3.14f / 180.0f)
	{
		emit_ang(out, rb0, rb1, qxdir(jf1), (float)(2 * ((double)(-s.x) + sin((double)(jmin.x / 2.0f))) / (double)dt), 0, FLT_MAX);
		emit_ang(out, rb0, rb1, -qxdir(jf1), (float)(2 * ((double)(s.x) - sin((double)(jmax.x / 2.0f))) / (double)dt), 0, FLT_MAX);
	}
	if (jmax.y == jmin.y)
		emit_ang(out, rb0, rb1, qydir(jf1), ph.biasfactorjoint * 2 * (-s.y + jmin.y) / dt, -FLT_MAX, FLT_MAX);
	else
	{
		emit_ang(out, rb0, rb1, qydir(jf1), (float)(2 * ((double)(-s.y) + sin((double)(jmin.y / 2.0f))) / (double)dt), 0, FLT_MAX);
		emit_ang(out, rb0, rb1, -qydir(jf1), (float)(2 * ((double)(s.y) - sin((double)(jmax.y / 2.0f))) / (double)dt), 0, FLT_MAX);
	}
	if (jmin.z == jmax.z)
		emit_ang(out, rb0, rb1, qzdir(jf1), ph.biasfactorjoint * 2 * -t.z / dt, -FLT_MAX, FLT_MAX);
	else
	{
		emit_ang(out, rb0, rb1, qzdir(jf1), (float)(2 * ((double)(-t.z) + sin((double)(jmin.z / 2.0f))) / (double)dt), 0, FLT_MAX);
		emit_ang(out, rb0, rb1, -qzdir(jf1), (float)(2 * ((double)(t.z) - sin((double)(jmax.z / 2.0f))) / (double)dt), 0, FLT_MAX);
	}
}
// Row `want` of the rows ConstrainAngularRangeW emits for a joint (same order, same expressions as angular_range_w above, which builds them all): a lane
// that owns one row pays for one double-precision sine instead of up to six.  The two-sided rows' target spins are 2*((-c) + sin(min/2))/dt and
// 2*(c - sin(max/2))/dt; c - s is evaluated as c + (-s), which is the same IEEE operation.
__device__ __forceinline__ void angular_range_row(const ht_physics_dev &ph, int rb0, v4 jb0, int rb1, v4 jf1, v3 lmin, v3 lmax, int want, float *row)
{
	const float dt = ph.deltaT;
	v3 jmin = (lmin * 3.14f) / 180.0f, jmax = (lmax * 3.14f) / 180.0f;
	if (jmin.x == 0 && jmax.x == 0 && jmin.z < jmax.z)
	{
		v4 cb = normalize(V4(0, -1, 0, 1));
		jb0 = qmul(jb0, cb); jf1 = qmul(jf1, cb);
		v3 nmin = V3(lmin.z, lmin.y, 0), nmax = V3(lmax.z, lmax.y, 0);
		lmin = nmin; lmax = nmax;
		jmin = (lmin * 3.14f) / 180.0f; jmax = (lmax * 3.14f) / 180.0f;
	}
	const v4 r = qmul(qconj(jb0), jf1);
	const v4 s = quat_from_to(V3(0, 0, 1.0f), qzdir(r));
	const v4 t = qmul(qconj(s), r);
	// which axis and which of its rows
	const int nx = (jmax.x == jmin.x) ? 1 : ((jmax.x - jmin.x < 360.0f * 3.14f / 180.0f) ? 2 : 0), ny = (jmax.y == jmin.y) ? 1 : 2;
	int k = want, axis = 0;
	if (k >= nx) { k -= nx; axis = 1; if (k >= ny) { k -= ny; axis = 2; } }
	const float lo = axis == 0 ? jmin.x : axis == 1 ? jmin.y : jmin.z, hi = axis == 0 ? jmax.x : axis == 1 ? jmax.y : jmax.z;
	const float comp = axis == 0 ? s.x : axis == 1 ? s.y : t.z;
	const v3 dir = axis == 0 ? qxdir(jf1) : axis == 1 ? qydir(jf1) : qzdir(jf1);
	const bool equal = hi == lo, upper = !equal && k == 1;
	const double sn = sin((double)((upper ? hi : lo) / 2.0f));
	const float two_sided = (float)(2 * ((double)(upper ? comp : -comp) + (upper ? -sn : sn)) / (double)dt);
	float ts = two_sided;
	if (equal && axis == 1) ts = ph.biasfactorjoint * 2 * (-s.y + jmin.y) / dt;
	if (equal && axis == 2) ts = ph.biasfactorjoint * 2 * -t.z / dt;
	put_ang(row, rb0, rb1, upper ? -dir : dir, ts, equal ? -FLT_MAX : 0, FLT_MAX);
}
// ConstrainConeAngle physics.h:402-414
template <class LDS> __device__ __forceinline__ void cone_angle(const ht_physics_dev &ph, const LDS &S, int rb0, v3 n0, int rb1, v3 n1, float limitangle_degrees, float *out)
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
template <class LDS> __device__ __forceinline__ void angular_drive(const ht_physics_dev &ph, const LDS &S, int rb0, int rb1, v4 target, float maxtorque, float (*out)[8])
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
static __constant__ int FEATURE_BONE[8] = { 1, 1, 1, 4, 7, 10, 13, 16 };
static __constant__ float FEATURE_OFF[8][3] = { { 0, 0, 0 }, { -0.03f, 0, -0.03f }, { 0.03f, 0, -0.03f }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };

// ---- two-body row maths ------------------------------------------------------------------------
template <class LDS> __device__ __forceinline__ v3 spin_of(const LDS &S, int b) { return mul(body_I(S, b), F3(S.ang4[b])); }       // physics.h:126
struct ht_true { static constexpr bool value = true; };
struct ht_false { static constexpr bool value = false; };
struct arow { int rb0, rb1; v3 axis; float targetspin, mn, mx, s2t, torque, mintorque; int lev; };

// row counts of ConstrainAngularRangeW (physics.h:351-393) for given limits, without building the rows
__device__ __forceinline__ int angular_range_count(v3 lmin, v3 lmax)
{
	v3 jmin = (lmin * 3.14f) / 180.0f, jmax = (lmax * 3.14f) / 180.0f;
	if (jmin.x == 0 && jmax.x == 0 && jmin.z < jmax.z)
	{
		v3 nmin = V3(lmin.z, lmin.y, 0), nmax = V3(lmax.z, lmax.y, 0);
		jmin = (nmin * 3.14f) / 180.0f; jmax = (nmax * 3.14f) / 180.0f;
	}
	int n = 0;
	if (jmax.x == jmin.x) n += 1; else if (jmax.x - jmin.x < 360.0f * 3.14f / 180.0f) n += 2;
	n += (jmax.y == jmin.y) ? 1 : 2;
	n += (jmin.z == jmax.z) ? 1 : 2;
	return n;
}

// ---- the SOLVE TABLES (round 6) ------------------------------------------------------------------------------------------------------------------------------------
// Everything of a solve that follows from the pose, the CNN's decode and the single-body rows alone -- the joints' linear groups, every angular record, the couplings and
// sorted edges of the two-body blocks (ht_block.hpp), the per-body chain lists of the single-body rows, their dealing to the four DPP rows and the couplings of their blocks
// of four (ht_quad.hpp) -- is made by k_solve_prep (ht_prep.hip: four waves per frame, on a side stream beside the contact kernel) and handed over in the frame's table.
// k_solve then starts at "load the tables": what is left of its one-wave prologue is the state, the contacts' groups and their couplings.  Until round 5 all of it ran on
// k_solve's single wave, on the critical path of every solve (0.9 M of a frame's 4.1 M cycles per step).  Same expressions in the same order: results are bit-identical
// (tests/test_gpu_same_bits.py).  Words (floats or ints) per frame:
#define TB_HDR 0           // 32 ints, below
#define TB_CCNT 32         // 32 ints: blocks of four rows of every body's chain
#define TB_CNEXT 64        // 32 ints: the body that follows on the same DPP row (-1: none)
#define TB_AREC 96         // (128 + 4) angular records of AROW floats: the rows, the idle record, read-ahead slack
#define TB_ABODY (TB_AREC + 132 * AROW)      // 128 body pairs of the angular rows (rb0 | rb1 << 8, 0xFFFF: no row), two per word
#define TB_GA (TB_ABODY + 64)                // the angular blocks' coupling registers: register 4k + c of lane l at ((k * 64 + l) * 4 + c)
#define TB_EMA (TB_GA + 2048)                // edge words of angular block Q, lane l at Q * 64 + l
#define TB_POOL (TB_EMA + 256)               // the joints' linear groups, LGRP floats each
#define TB_GL (TB_POOL + HT_MAXNJ * LGRP)    // the linear blocks' coupling registers among joint rows (layout of TB_GA)
#define TB_EML (TB_GL + 2048)                // edge words of linear block Q as they are while the frame has no contact
static_assert(TB_EML + 256 == TB_WORDS, "ht_launch.hpp states the size of a frame's tables");
// header words
#define TH_OK 0            // 1: the pose-only tables (joints' groups, angular records, block couplings and edges) hold the frame (0: a frame the blocked form does not hold, or not made -- k_solve runs its own prologue)
#define TH_CHAIN_OK 16     // 1: the chain tables (lists, dealing, four-row couplings, the landmark rays' and boundary planes' records) were made too
#define TH_NA 1            // angular rows
#define TH_NPRE 2          // single-body rows ahead of the cloud rows (landmark rays / boundary planes)
#define TH_TOTAL 3         // blocks of four single-body rows over all four DPP rows
#define TH_E0 4            // 4: first chain entry of DPP row R
#define TH_NBLK 8          // 4: its blocks
#define TH_HEAD 12         // 4: its first body
