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


// One LimitLinear::Iter (physics.h:289-307) on a body held in quad layout.  Lane c < 3 passes rv = r1[c] (lever arm, world frame) and
// n = normal[c]; lane 3 passes the row's target speed in rv.  t = (fmin*dt, fmax*dt, effective mass, impulse sum).  Returns the new sum.
struct quad_body { float l, av, minv, Ix, Iy, Iz; };      // this lane's component of the linear / angular momentum, 1/mass, row c of Iinv
__device__ __forceinline__ float quad_row_step(quad_body &B, float rv, float n, float4 t)
{
	const float w = (B.Ix * dpp<QP_BC0>(B.av) + B.Iy * dpp<QP_BC1>(B.av)) + B.Iz * dpp<QP_BC2>(B.av);      // (Iinv * angular_momentum)[c]
	const float m1 = w * dpp<QP_ROT1>(rv), m2 = w * dpp<QP_ROT2>(rv);                                      // w[c]*r1[c+1], w[c]*r1[c+2]
	const float v1 = (dpp<QP_ROT1>(m1) - dpp<QP_ROT2>(m2)) + B.l * B.minv;                                 // (cross(spin, r1) + lin*massinv)[c]
	const float p = v1 * n;
	const float vn = (dpp<QP_BC0>(p) + dpp<QP_BC1>(p)) + dpp<QP_BC2>(p);
	const float impulsen = -dpp<QP_BC3>(rv) - vn;
	float impulse = div_ieee(impulsen, t.z);
	impulse = clamp_med3(impulse, t.x - t.w, t.y - t.w);
	const float imp = n * impulse;
	B.l = B.l + imp;
	const float k1 = rv * dpp<QP_ROT1>(imp), k2 = rv * dpp<QP_ROT2>(imp);                                  // r1[c]*imp[c+1], r1[c]*imp[c+2]
	B.av = B.av + (dpp<QP_ROT1>(k1) - dpp<QP_ROT2>(k2));                                                   // cross(r1, imp)[c]
	return t.w + impulse;
}
