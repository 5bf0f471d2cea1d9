// ht_math.hpp -- small fp32 vector / quaternion / rigid-transform algebra for the MI355X hand-tracking kernels.
//
// Product code (device + host).  Semantics follow the reference's linalg.h / geometric.h so that, compiled with
// -ffp-contract=off, the device evaluates the same sequence of IEEE operations as the reference CPU path:
//   dot      = ((ax*bx + ay*by) + az*bz) [+ aw*bw]                  third_party/linalg.h:262
//   qrot     = qxdir(q)*v.x + qydir(q)*v.y + qzdir(q)*v.z           third_party/linalg.h:284-288
//   normalize= v / sqrt(dot(v,v)) (true divisions)                  third_party/linalg.h:264-265
// Divisions and square roots are the correctly rounded HIP defaults.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>

#define HT_HD __host__ __device__ __forceinline__

struct v2 { float x, y; };
struct v3 { float x, y, z; };
struct v4 { float x, y, z, w; };
struct m3 { v3 x, y, z; };          // columns, like linalg::mat<float,3,3>
struct xf { v3 p; v4 q; };          // Pose (geometric.h:111-125)

HT_HD v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
HT_HD v4 V4(float x, float y, float z, float w) { v4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
HT_HD v4 V4(v3 v, float w) { return V4(v.x, v.y, v.z, w); }
HT_HD v3 xyz(v4 v) { return V3(v.x, v.y, v.z); }

// std::min / std::max as the reference spells them (argument order matters for NaN)
HT_HD float fmax_std(float a, float b) { return (a < b) ? b : a; }
HT_HD float fmin_std(float a, float b) { return (b < a) ? b : a; }
HT_HD float clamp_std(float a, float mn, float mx) { return fmin_std(fmax_std(a, mn), mx); }

HT_HD v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
HT_HD v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
HT_HD v3 operator-(v3 a) { return V3(-a.x, -a.y, -a.z); }
HT_HD v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
HT_HD v3 operator/(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
HT_HD v4 operator+(v4 a, v4 b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
HT_HD v4 operator-(v4 a) { return V4(-a.x, -a.y, -a.z, -a.w); }
HT_HD v4 operator*(v4 a, float s) { return V4(a.x * s, a.y * s, a.z * s, a.w * s); }
HT_HD v4 operator/(v4 a, float s) { return V4(a.x / s, a.y / s, a.z / s, a.w / s); }
HT_HD bool is_zero(v3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
HT_HD bool same(v3 a, v3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

HT_HD float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
HT_HD float dot(v4 a, v4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
HT_HD float dot_plane(v4 p, v3 v) { return ((p.x * v.x + p.y * v.y) + p.z * v.z) + p.w; }     // dot(p, float4(v,1)): p.w*1 is exact
HT_HD v3 cross(v3 a, v3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HT_HD float length(v3 a) { return sqrtf(dot(a, a)); }
HT_HD float length(v4 a) { return sqrtf(dot(a, a)); }
HT_HD v3 normalize(v3 a) { return a / length(a); }
HT_HD v4 normalize(v4 a) { return a / length(a); }
HT_HD v3 safenormalize(v3 a) { return is_zero(a) ? V3(0, 0, 1) : normalize(a); }              // geometric.h:58

HT_HD v4 qconj(v4 q) { return V4(-q.x, -q.y, -q.z, q.w); }
HT_HD v4 qmul(v4 a, v4 b)
{
	return V4(a.x * b.w + a.w * b.x + a.y * b.z - a.z * b.y, a.y * b.w + a.w * b.y + a.z * b.x - a.x * b.z,
	          a.z * b.w + a.w * b.z + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
HT_HD v3 qxdir(v4 q) { return V3(q.w * q.w + q.x * q.x - q.y * q.y - q.z * q.z, (q.x * q.y + q.z * q.w) * 2, (q.z * q.x - q.y * q.w) * 2); }
HT_HD v3 qydir(v4 q) { return V3((q.x * q.y - q.z * q.w) * 2, q.w * q.w - q.x * q.x + q.y * q.y - q.z * q.z, (q.y * q.z + q.x * q.w) * 2); }
HT_HD v3 qzdir(v4 q) { return V3((q.z * q.x + q.y * q.w) * 2, (q.y * q.z - q.x * q.w) * 2, q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z); }
HT_HD m3 qmat(v4 q) { m3 m; m.x = qxdir(q); m.y = qydir(q); m.z = qzdir(q); return m; }
HT_HD v3 mul(m3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
HT_HD v3 qrot(v4 q, v3 v) { return mul(qmat(q), v); }
HT_HD m3 mul(m3 a, m3 b) { m3 m; m.x = mul(a, b.x); m.y = mul(a, b.y); m.z = mul(a, b.z); return m; }
HT_HD m3 transpose(m3 m) { m3 t; t.x = V3(m.x.x, m.y.x, m.z.x); t.y = V3(m.x.y, m.y.y, m.z.y); t.z = V3(m.x.z, m.y.z, m.z.z); return t; }
HT_HD m3 operator*(m3 m, float s) { m3 r; r.x = m.x * s; r.y = m.y * s; r.z = m.z * s; return r; }
HT_HD float determinant(m3 a) { return a.x.x * (a.y.y * a.z.z - a.z.y * a.y.z) + a.x.y * (a.y.z * a.z.x - a.z.z * a.y.x) + a.x.z * (a.y.x * a.z.y - a.z.x * a.y.y); }
HT_HD m3 inverse(m3 a)
{
	m3 j;
	j.x = V3(a.y.y * a.z.z - a.z.y * a.y.z, a.z.y * a.x.z - a.x.y * a.z.z, a.x.y * a.y.z - a.y.y * a.x.z);
	j.y = V3(a.y.z * a.z.x - a.z.z * a.y.x, a.z.z * a.x.x - a.x.z * a.z.x, a.x.z * a.y.x - a.y.z * a.x.x);
	j.z = V3(a.y.x * a.z.y - a.z.x * a.y.y, a.z.x * a.x.y - a.x.x * a.z.y, a.x.x * a.y.y - a.y.x * a.x.y);
	float d = determinant(a);
	m3 r; r.x = j.x / d; r.y = j.y / d; r.z = j.z / d; return r;
}
// world inverse inertia R (I^-1 / m) R^T, physics.h:518,540
HT_HD m3 world_inertia(v4 q, m3 tensorinv_massless, float massinv) { m3 M = qmat(q); return mul(M, mul(tensorinv_massless * massinv, transpose(M))); }

HT_HD xf XF(v3 p, v4 q) { xf r; r.p = p; r.q = q; return r; }
HT_HD xf inverse(xf a) { v4 q = qconj(a.q); return XF(qrot(q, -a.p), q); }
HT_HD v3 apply(xf a, v3 v) { return a.p + qrot(a.q, v); }
HT_HD xf mul(xf a, xf b) { return XF(apply(a, b.p), qmul(a.q, b.q)); }
HT_HD v4 transform_plane(xf a, v4 pl) { v3 n = qrot(a.q, xyz(pl)); return V4(n, pl.w - dot(a.p, n)); }

// Transcendentals.  The reference calls glibc; on the device the value is formed in double and rounded once, which
// reproduces a correctly rounded float result (and the reference's own double evaluations) except in rare half-ulp ties.
HT_HD float sin_f(float x) { return (float)sin((double)x); }
HT_HD float cos_f(float x) { return (float)cos((double)x); }
HT_HD float acos_f(float x) { return (float)acos((double)x); }
HT_HD v4 quat_axis_angle(v3 axis, float angle) { return V4(axis * sin_f(angle / 2), cos_f(angle / 2)); }    // geometric.h:102

HT_HD v3 orth(v3 v)      // geometric.h:312-318
{
	float a0 = fabsf(v.x), a1 = fabsf(v.y), a2 = fabsf(v.z);
	int k = 0; float m = a0;
	if (m < a1) { k = 1; m = a1; }
	if (m < a2) { k = 2; }
	v3 u = V3(k == 0 ? 0.0f : 1.0f, k == 1 ? 0.0f : 1.0f, k == 2 ? 0.0f : 1.0f);
	return normalize(cross(u, v));
}
HT_HD v4 quat_from_to(v3 a, v3 b)      // geometric.h:319-328
{
	v3 v0 = normalize(a), v1 = normalize(b);
	v3 c = cross(v0, v1);
	float d = dot(v0, v1);
	if (d <= -1.0f) { v3 o = orth(v0); return V4(o.x, o.y, o.z, 0); }
	float s = sqrtf((1 + d) * 2);
	return V4(c.x / s, c.y / s, c.z / s, s / 2.0f);
}
HT_HD float line_project_time(v3 p0, v3 p1, v3 a) { v3 d = p1 - p0; return dot(d, a - p0) / dot(d, d); }           // geometric.h:153-159
HT_HD v3 line_project(v3 p0, v3 p1, v3 a) { return p0 + (p1 - p0) * line_project_time(p0, p1, a); }
HT_HD v3 tri_normal(v3 v0, v3 v1, v3 v2)       // geometric.h:234-240
{
	v3 cp = cross(v1 - v0, v2 - v1);
	float m = length(cp);
	if (m == 0) return V3(0, 0, 0);
	return cp * (1.0f / m);
}
HT_HD v3 plane_project_of(v3 v0, v3 v1, v3 v2, v3 point)     // geometric.h:204-214
{
	v3 cp = cross(v2 - v0, v2 - v1);
	float dtcpm = -dot(cp, v0);
	float cpm2 = dot(cp, cp);
	if (cpm2 == 0.0f) return line_project(v0, (length(v1 - v0) > length(v2 - v0)) ? v1 : v2, point);
	return point - (cp * (dot(cp, point) + dtcpm)) / cpm2;
}
HT_HD v3 barycentric(v3 v0, v3 v1, v3 v2, v3 s)      // geometric.h:185-195
{
	m3 m; m.x = v0; m.y = v1; m.z = v2;
	if (determinant(m) == 0)
	{
		int k = (length(v1 - v2) > length(v0 - v2)) ? 1 : 0;
		float t = line_project_time(v2, k ? v1 : v0, s);
		return V3((1 - k) * t, k * t, 1 - t);
	}
	return mul(inverse(m), s);
}
