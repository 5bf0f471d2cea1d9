/*
 * ho_math.h -- scalar restatement of the vector/quaternion/matrix algebra the reference hot path uses.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/): the CPU checker for the HIP product; never linked by the product.
 *
 * Every function states the reference expression it follows (third_party/linalg.h, geometric.h) and keeps the
 * same association order, so that built with -ffp-contract=off it reproduces the reference bit for bit.
 * Matrices are column major like linalg::mat (m.x, m.y, m.z are columns; m[i][j] = column i, row j).
 */
#ifndef HO_MATH_H
#define HO_MATH_H
#include <math.h>
#include <float.h>

typedef struct { float x, y; } f2;
typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;
typedef struct { f3 x, y, z; } m33;
typedef struct { f4 x, y, z, w; } m44;
typedef struct { f3 position; f4 orientation; } pose_t;

/* std::min / std::max / clamp exactly as written (geometric.h:61-62); NaN behaviour follows the comparisons */
static inline float ho_maxf(float a, float b) { return (a < b) ? b : a; }            /* std::max(a,b) */
static inline float ho_minf(float a, float b) { return (b < a) ? b : a; }            /* std::min(a,b) */
static inline float ho_clampf(float a, float mn, float mx) { return ho_minf(ho_maxf(a, mn), mx); }
static inline int ho_maxi(int a, int b) { return (a < b) ? b : a; }
static inline int ho_mini(int a, int b) { return (b < a) ? b : a; }

static inline f3 F3(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f4 F4(float x, float y, float z, float w) { f4 r = { x, y, z, w }; return r; }
static inline f4 F4v(f3 v, float w) { f4 r = { v.x, v.y, v.z, w }; return r; }
static inline f3 xyz(f4 v) { return F3(v.x, v.y, v.z); }
static inline float f3_get(f3 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : v.z; }
static inline float f4_get(f4 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

/* linalg.h:220-223 elementwise operators */
static inline f3 add3(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 scale3(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
static inline f3 div3(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
static inline f3 neg3(f3 a) { return F3(-a.x, -a.y, -a.z); }
static inline f4 add4(f4 a, f4 b) { return F4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline f4 scale4(f4 a, float s) { return F4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline f4 div4(f4 a, float s) { return F4(a.x / s, a.y / s, a.z / s, a.w / s); }
static inline f4 neg4(f4 a) { return F4(-a.x, -a.y, -a.z, -a.w); }
static inline int eq3(f3 a, f3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static inline int eq4(f4 a, f4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }

/* linalg.h:261-265 */
static inline f3 cross3(f3 a, f3 b) { return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float dot4(f4 a, f4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
static inline float dot2(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
static inline float length3(f3 a) { return sqrtf(dot3(a, a)); }
static inline float length4(f4 a) { return sqrtf(dot4(a, a)); }
static inline f3 normalize3(f3 a) { return div3(a, length3(a)); }
static inline f4 normalize4(f4 a) { return div4(a, length4(a)); }
static inline f3 safenormalize3(f3 v) { return (v.x == 0 && v.y == 0 && v.z == 0) ? F3(0, 0, 1) : normalize3(v); }   /* geometric.h:58 */

/* linalg.h:277-288 quaternion algebra */
static inline f4 qconj(f4 q) { return F4(-q.x, -q.y, -q.z, q.w); }
static inline f4 qmul(f4 a, f4 b)
{
	return F4(a.x * b.w + a.w * b.x + a.y * b.z - a.z * b.y, a.y * b.w + a.w * b.y + a.z * b.x - a.x * b.z,
	          a.z * b.w + a.w * b.z + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
static inline f3 qxdir(f4 q) { return F3(q.w * q.w + q.x * q.x - q.y * q.y - q.z * q.z, (q.x * q.y + q.z * q.w) * 2, (q.z * q.x - q.y * q.w) * 2); }
static inline f3 qydir(f4 q) { return F3((q.x * q.y - q.z * q.w) * 2, q.w * q.w - q.x * q.x + q.y * q.y - q.z * q.z, (q.y * q.z + q.x * q.w) * 2); }
static inline f3 qzdir(f4 q) { return F3((q.z * q.x + q.y * q.w) * 2, (q.y * q.z - q.x * q.w) * 2, q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z); }
static inline m33 qmat(f4 q) { m33 m = { qxdir(q), qydir(q), qzdir(q) }; return m; }
static inline f3 qrot(f4 q, f3 v) { return add3(add3(scale3(qxdir(q), v.x), scale3(qydir(q), v.y)), scale3(qzdir(q), v.z)); }

/* linalg.h:296-331 matrix algebra */
static inline f3 m33_mulv(m33 a, f3 b) { return add3(add3(scale3(a.x, b.x), scale3(a.y, b.y)), scale3(a.z, b.z)); }
static inline m33 m33_mul(m33 a, m33 b) { m33 m = { m33_mulv(a, b.x), m33_mulv(a, b.y), m33_mulv(a, b.z) }; return m; }
static inline m33 m33_transpose(m33 m) { m33 t = { F3(m.x.x, m.y.x, m.z.x), F3(m.x.y, m.y.y, m.z.y), F3(m.x.z, m.y.z, m.z.z) }; return t; }
static inline m33 m33_scale(m33 m, float s) { m33 r = { scale3(m.x, s), scale3(m.y, s), scale3(m.z, s) }; return r; }
static inline float m33_det(m33 a) { return a.x.x * (a.y.y * a.z.z - a.z.y * a.y.z) + a.x.y * (a.y.z * a.z.x - a.z.z * a.y.x) + a.x.z * (a.y.x * a.z.y - a.z.x * a.y.y); }
static inline m33 m33_adjugate(m33 a)
{
	m33 r = { F3(a.y.y * a.z.z - a.z.y * a.y.z, a.z.y * a.x.z - a.x.y * a.z.z, a.x.y * a.y.z - a.y.y * a.x.z),
	          F3(a.y.z * a.z.x - a.z.z * a.y.x, a.z.z * a.x.x - a.x.z * a.z.x, a.x.z * a.y.x - a.y.z * a.x.x),
	          F3(a.y.x * a.z.y - a.z.x * a.y.y, a.z.x * a.x.y - a.x.x * a.z.y, a.x.x * a.y.y - a.y.x * a.x.y) };
	return r;
}
static inline m33 m33_inverse(m33 a) { m33 j = m33_adjugate(a); float d = m33_det(a); m33 r = { div3(j.x, d), div3(j.y, d), div3(j.z, d) }; return r; }

/* geometric.h:111-125 Pose */
static inline pose_t POSE(f3 p, f4 q) { pose_t r = { p, q }; return r; }
static inline pose_t pose_identity(void) { return POSE(F3(0, 0, 0), F4(0, 0, 0, 1)); }
static inline pose_t pose_inverse(pose_t p) { f4 q = qconj(p.orientation); return POSE(qrot(q, neg3(p.position)), q); }
static inline f3 pose_apply(pose_t p, f3 v) { return add3(p.position, qrot(p.orientation, v)); }
static inline pose_t pose_mul(pose_t a, pose_t b) { return POSE(pose_apply(a, b.position), qmul(a.orientation, b.orientation)); }
static inline f4 pose_transform_plane(pose_t p, f4 pl) { f3 n = qrot(p.orientation, xyz(pl)); return F4v(n, pl.w - dot3(p.position, n)); }

/* geometric.h:102 / linalg.h:344 */
/* The reference's float sine / cosine / arc cosine are glibc's (std::sin(float) in geometric.h:102, acos(float) in physics.h:319,408), which are within one ulp but
 * not always correctly rounded; the device forms them in double and rounds once.  ho_round_once = 1 (tests only: tests/test_gpu_exact_solver.py) makes the
 * restatement do the same, so that a device result can be compared with it bit for bit; 0 (the default, and the setting every fixture is pinned with) = glibc's. */
extern int ho_round_once;
static inline float ho_sinf(float x) { return ho_round_once ? (float)sin((double)x) : sinf(x); }
static inline float ho_cosf(float x) { return ho_round_once ? (float)cos((double)x) : cosf(x); }
static inline float ho_acosf(float x) { return ho_round_once ? (float)acos((double)x) : acosf(x); }
static inline f4 quat_axis_angle(f3 axis, float angle) { return F4v(scale3(axis, ho_sinf(angle / 2)), ho_cosf(angle / 2)); }

/* geometric.h:312-318 Orth: zero the largest-magnitude component of (1,1,1) (first maximum wins), cross, normalise */
static inline f3 ho_orth(f3 v)
{
	float a[3] = { fabsf(v.x), fabsf(v.y), fabsf(v.z) };
	int k = 0;
	if (a[k] < a[1]) k = 1;
	if (a[k] < a[2]) k = 2;
	f3 u = F3(k == 0 ? 0.0f : 1.0f, k == 1 ? 0.0f : 1.0f, k == 2 ? 0.0f : 1.0f);
	return normalize3(cross3(u, v));
}
/* geometric.h:319-328 quat_from_to */
static inline f4 quat_from_to(f3 v0_, f3 v1_)
{
	f3 v0 = normalize3(v0_), v1 = normalize3(v1_);
	f3 c = cross3(v0, v1);
	float d = dot3(v0, v1);
	if (d <= -1.0f) { f3 a = ho_orth(v0); return F4(a.x, a.y, a.z, 0); }
	float s = sqrtf((1 + d) * 2);
	return F4(c.x / s, c.y / s, c.z / s, s / 2.0f);
}
/* geometric.h:153-160 */
static inline float line_project_time(f3 p0, f3 p1, f3 a) { f3 d = sub3(p1, p0); return dot3(d, sub3(a, p0)) / dot3(d, d); }
static inline f3 line_project(f3 p0, f3 p1, f3 a) { return add3(p0, scale3(sub3(p1, p0), line_project_time(p0, p1, a))); }
/* geometric.h:234-240 */
static inline f3 tri_normal(f3 v0, f3 v1, f3 v2)
{
	f3 cp = cross3(sub3(v1, v0), sub3(v2, v1));
	float m = length3(cp);
	if (m == 0) return F3(0, 0, 0);
	return scale3(cp, 1.0f / m);
}
/* geometric.h:204-214 */
static inline f3 plane_project_of(f3 v0, f3 v1, f3 v2, f3 point)
{
	f3 cp = cross3(sub3(v2, v0), sub3(v2, v1));
	float dtcpm = -dot3(cp, v0);
	float cpm2 = dot3(cp, cp);
	if (cpm2 == 0.0f)
		return line_project(v0, (length3(sub3(v1, v0)) > length3(sub3(v2, v0))) ? v1 : v2, point);
	return sub3(point, div3(scale3(cp, dot3(cp, point) + dtcpm), cpm2));
}
/* geometric.h:185-195 */
static inline f3 barycentric(f3 v0, f3 v1, f3 v2, f3 s)
{
	m33 m = { v0, v1, v2 };
	if (m33_det(m) == 0)
	{
		int k = (length3(sub3(v1, v2)) > length3(sub3(v0, v2))) ? 1 : 0;
		float t = line_project_time(v2, k ? v1 : v0, s);
		return F3((1 - k) * t, k * t, 1 - t);
	}
	return m33_mulv(m33_inverse(m), s);
}
#endif
