// ht_gjk.hip -- bone-bone narrow phase on CDNA4: broad-phase cull, GJK closest features, expanding-polytope fallback
// for penetration, and the 5-sample contact patch.
//
// Reference computations:
//   FindShapeShapeContacts            third_party/physics.h:451-462
//   ContactPatch / Separated          third_party/gjk.h:607-643, 367-437  (NextMinkSimplex0..3 :82-275, calcpoints :337-363)
//   ExpandingPolytopeAlgorithm        third_party/hull.h:233-310 (Tri bookkeeping :79-186)
//   SupportFunc / SupportFuncTrans    third_party/gjk.h:568-582, maxdir third_party/geometric.h:218-224
//
// Mapping: one 256-thread block per frame; each of its 4 waves takes candidate pairs c = wave, wave+4, ... and runs GJK
// wave-cooperatively: the support map (arg-max of a dot product over all 162/258 collision vertices of a bone, first
// maximum wins) is a strided scan + butterfly reduction over the 64 lanes; the simplex logic is wave-uniform.
// Contacts are staged per wave in LDS and written out in pair order, so the solver sees the reference's row order.
//
// The 4 extra "jiggle" GJK runs of the contact patch are skipped when they provably cannot add a contact: an extra
// sample is rejected if it lies within 0.05 m of an accepted one on either shape (gjk.h:637) and every sample lies in the
// convex hull of its shape, so a shape whose diameter is below 0.05 m can never contribute a second sample.
#include <stdlib.h>
#include "ht_device.hpp"
#include "ht_launch.hpp"

struct support_t { const float4 *verts; int n; v3 pos; v4 q; int outer; v3 opos; v4 oq; };
struct mkpoint { v3 a, b, p; float t; };
struct simplex { v3 v; mkpoint W[4]; int count; };
struct gjk_hit { v3 normal, p0w, p1w; float separation; };

__device__ __forceinline__ v3 support_inner(const support_t &s, v3 dir, int lane)
{
	const v3 dl = qrot(qconj(s.q), dir);
	float best = -INFINITY; int bi = 0x7fffffff;
	for (int i = lane; i < s.n; i += 64)
	{
		float4 q = s.verts[i];
		float d = dot(V3(q.x, q.y, q.z), dl);
		if (bi == 0x7fffffff || best < d) { best = d; bi = i; }
	}
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1)
	{
		float ob = __shfl_xor(best, o); int oi = __shfl_xor(bi, o);
		if (oi != 0x7fffffff && (bi == 0x7fffffff || best < ob || (ob == best && oi < bi))) { best = ob; bi = oi; }
	}
	float4 q = s.verts[bi];
	return s.pos + qrot(s.q, V3(q.x, q.y, q.z));
}
__device__ __forceinline__ v3 support(const support_t &s, v3 dir, int lane)
{
	if (s.outer) return s.opos + qrot(s.oq, support_inner(s, qrot(qconj(s.oq), dir), lane));
	return support_inner(s, dir, lane);
}
__device__ __forceinline__ mkpoint point_on_minkowski(const support_t &A, const support_t &B, v3 n, int lane)
{
	mkpoint m; m.a = support(A, n, lane); m.b = support(B, -n, lane); m.p = m.a - m.b; m.t = 0; return m;
}

__device__ void next1(simplex &dst, const simplex &src, const mkpoint &w)
{
	const v3 O = V3(0, 0, 0);
	float t = line_project_time(w.p, src.W[0].p, O);
	if (t < 0.0f) { dst.W[0] = w; dst.W[0].t = 1.0f; dst.v = w.p; dst.count = 1; return; }
	dst.W[0] = src.W[0]; dst.W[0].t = t;
	dst.W[1] = w; dst.W[1].t = 1.0f - t;
	dst.v = w.p + (src.W[0].p - w.p) * t;
	dst.count = 2;
}
__device__ __forceinline__ void keep_edge(simplex &dst, const mkpoint &keep, const mkpoint &w, float t, v3 v)
{
	mkpoint k = keep;
	dst.W[0] = k; dst.W[0].t = t; dst.W[1] = w; dst.W[1].t = 1.0f - t; dst.v = v; dst.count = 2;
}
__device__ void next2(simplex &dst, const simplex &src, const mkpoint &w)
{
	const v3 O = V3(0, 0, 0);
	const v3 w0 = src.W[0].p, w1 = src.W[1].p;
	float t0 = line_project_time(w.p, w0, O), t1 = line_project_time(w.p, w1, O);
	v3 v0 = w.p + (w0 - w.p) * t0, v1 = w.p + (w1 - w.p) * t1;
	int ine0 = (dot(-v0, w1 - v0) > 0.0f), ine1 = (dot(-v1, w0 - v1) > 0.0f);
	if (ine0 && ine1)
	{
		mkpoint a = src.W[0], b = src.W[1];
		dst.count = 3; dst.v = plane_project_of(w0, w1, w.p, O); dst.W[0] = a; dst.W[1] = b; dst.W[2] = w; return;
	}
	if (!ine0 && (t0 > 0.0f)) { keep_edge(dst, src.W[0], w, t0, v0); return; }
	if (!ine1 && (t1 > 0.0f)) { keep_edge(dst, src.W[1], w, t1, v1); return; }
	dst.W[0] = w; dst.W[0].t = 1.0f; dst.v = w.p; dst.count = 1;
}
__device__ void next3(simplex &dst, const simplex &src, const mkpoint &w)
{
	const v3 O = V3(0, 0, 0);
	const v3 w0 = src.W[0].p, w1 = src.W[1].p, w2 = src.W[2].p;
	float t0 = line_project_time(w.p, w0, O), t1 = line_project_time(w.p, w1, O), t2 = line_project_time(w.p, w2, O);
	v3 v0 = w.p + (w0 - w.p) * t0, v1 = w.p + (w1 - w.p) * t1, v2 = w.p + (w2 - w.p) * t2;
	v3 vc0 = plane_project_of(w.p, w1, w2, O), vc1 = plane_project_of(w.p, w2, w0, O), vc2 = plane_project_of(w.p, w0, w1, O);
	int inp0 = (dot(-vc0, w0 - vc0) > 0.0f), inp1 = (dot(-vc1, w1 - vc1) > 0.0f), inp2 = (dot(-vc2, w2 - vc2) > 0.0f);
	const mkpoint s0 = src.W[0], s1 = src.W[1], s2 = src.W[2];
	if (inp0 && inp1 && inp2) { dst.W[0] = s0; dst.W[1] = s1; dst.W[2] = s2; dst.count = 4; dst.v = O; dst.W[3] = w; return; }
	int inp2e0 = (dot(-v0, w1 - v0) > 0.0f), inp2e1 = (dot(-v1, w0 - v1) > 0.0f);
	if (!inp2 && inp2e0 && inp2e1) { dst.count = 3; dst.v = plane_project_of(w0, w1, w.p, O); dst.W[0] = s0; dst.W[1] = s1; dst.W[2] = w; return; }
	int inp0e1 = (dot(-v1, w2 - v1) > 0.0f), inp0e2 = (dot(-v2, w1 - v2) > 0.0f);
	if (!inp0 && inp0e1 && inp0e2) { dst.count = 3; dst.v = plane_project_of(w1, w2, w.p, O); dst.W[0] = s1; dst.W[1] = s2; dst.W[2] = w; return; }
	int inp1e2 = (dot(-v2, w0 - v2) > 0.0f), inp1e0 = (dot(-v0, w2 - v0) > 0.0f);
	if (!inp1 && inp1e2 && inp1e0) { dst.count = 3; dst.v = plane_project_of(w2, w0, w.p, O); dst.W[0] = s2; dst.W[1] = s0; dst.W[2] = w; return; }
	if (!inp1e0 && !inp2e0 && t0 > 0.0f) { keep_edge(dst, s0, w, t0, v0); return; }
	if (!inp2e1 && !inp0e1 && t1 > 0.0f) { keep_edge(dst, s1, w, t1, v1); return; }
	if (!inp0e2 && !inp1e2 && t2 > 0.0f) { keep_edge(dst, s2, w, t2, v2); return; }
	dst.W[0] = w; dst.W[0].t = 1.0f; dst.v = w.p; dst.count = 1;
}

__device__ gjk_hit calcpoints(simplex &src)
{
	if (src.count == 3)
	{
		v3 b = barycentric(src.W[0].p, src.W[1].p, src.W[2].p, src.v);
		src.W[0].t = b.x; src.W[1].t = b.y; src.W[2].t = b.z;
	}
	// pa = sum t_i a_i in index order, starting from zero (gjk.h:348-353); static indices keep the simplex in registers
	v3 pa = V3(0, 0, 0), pb = V3(0, 0, 0);
	if (src.count > 0) { pa = pa + src.W[0].a * src.W[0].t; pb = pb + src.W[0].b * src.W[0].t; }
	if (src.count > 1) { pa = pa + src.W[1].a * src.W[1].t; pb = pb + src.W[1].b * src.W[1].t; }
	if (src.count > 2) { pa = pa + src.W[2].a * src.W[2].t; pb = pb + src.W[2].b * src.W[2].t; }
	gjk_hit h;
	h.p0w = pa; h.p1w = pb;
	h.separation = length(pa - pb) + FLT_MIN;
	h.normal = normalize(src.v);
	return h;
}

// ---- expanding polytope; its triangle / vertex arrays live in a per-(frame, wave) HBM workspace (cold path) ------
#define EPA_MAXT 192
#define EPA_MAXV 96
struct epa_mem { int tv[EPA_MAXT][3]; int tn[EPA_MAXT][3]; int tid[EPA_MAXT]; float vx[EPA_MAXV], vy[EPA_MAXV], vz[EPA_MAXV]; };
// every lane executes the same statements on the same values, so each lane's own program order keeps the arrays coherent
__device__ __forceinline__ v3 ev(const epa_mem &m, int i) { return V3(m.vx[i], m.vy[i], m.vz[i]); }
__device__ __forceinline__ bool tri_dead(const epa_mem &m, int t) { return m.tn[t][0] == -1; }
__device__ __forceinline__ bool hasvert(const epa_mem &m, int t, int x) { return m.tv[t][0] == x || m.tv[t][1] == x || m.tv[t][2] == x; }
__device__ int *neib(epa_mem &m, int t, int va, int vb)
{
	for (int i = 0; i < 3; i++)
	{
		int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
		if (m.tv[t][i] == va && m.tv[t][i1] == vb) return &m.tn[t][i2];
		if (m.tv[t][i] == vb && m.tv[t][i1] == va) return &m.tn[t][i2];
	}
	return &m.tn[t][0];      // unreachable for a consistent mesh (the reference asserts)
}
__device__ void tri_set(epa_mem &m, int t, int a, int b, int c, int id, int n0, int n1, int n2)
{
	m.tv[t][0] = a; m.tv[t][1] = b; m.tv[t][2] = c; m.tid[t] = id; m.tn[t][0] = n0; m.tn[t][1] = n1; m.tn[t][2] = n2;
}
__device__ void nnfix(epa_mem &m, int k)
{
	if (m.tid[k] == -1) return;
	for (int i = 0; i < 3; i++)
	{
		int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
		if (m.tn[k][i] != -1) *neib(m, m.tn[k][i], m.tv[k][i2], m.tv[k][i1]) = k;
	}
}
__device__ void swapn(epa_mem &m, int a, int b)
{
	for (int i = 0; i < 3; i++) { int t = m.tv[a][i]; m.tv[a][i] = m.tv[b][i]; m.tv[b][i] = t; t = m.tn[a][i]; m.tn[a][i] = m.tn[b][i]; m.tn[b][i] = t; }
	// ids are swapped twice by the reference (std::swap of the Tri, then of the ids) => they stay in place
	nnfix(m, a); nnfix(m, b);
}
__device__ void b2bfix(epa_mem &m, int s, int t)
{
	for (int i = 0; i < 3; i++)
	{
		int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
		int va = m.tv[s][i1], vb = m.tv[s][i2];
		*neib(m, *neib(m, s, va, vb), vb, va) = *neib(m, t, vb, va);
		*neib(m, *neib(m, t, vb, va), va, vb) = *neib(m, s, va, vb);
	}
	for (int i = 0; i < 3; i++) { m.tn[s][i] = -1; m.tn[t][i] = -1; }
}
__device__ bool extrude(epa_mem &m, int &nt, int t0, int v)
{
	if (nt + 3 > EPA_MAXT) return false;
	int t[3] = { m.tv[t0][0], m.tv[t0][1], m.tv[t0][2] };
	int b = nt;
	int n[3] = { m.tn[t0][0], m.tn[t0][1], m.tn[t0][2] };
	tri_set(m, nt++, v, t[1], t[2], b + 0, n[0], b + 1, b + 2); *neib(m, n[0], t[1], t[2]) = b + 0;
	tri_set(m, nt++, v, t[2], t[0], b + 1, n[1], b + 2, b + 0); *neib(m, n[1], t[2], t[0]) = b + 1;
	tri_set(m, nt++, v, t[0], t[1], b + 2, n[2], b + 0, b + 1); *neib(m, n[2], t[0], t[1]) = b + 2;
	m.tn[t0][0] = m.tn[t0][1] = m.tn[t0][2] = -1;
	if (hasvert(m, n[0], v)) b2bfix(m, b + 0, n[0]);
	if (hasvert(m, n[1], v)) b2bfix(m, b + 1, n[1]);
	if (hasvert(m, n[2], v)) b2bfix(m, b + 2, n[2]);
	return true;
}
__device__ __forceinline__ bool above(const epa_mem &m, int t, v3 p, float epsilon)
{
	v3 n = tri_normal(ev(m, m.tv[t][0]), ev(m, m.tv[t][1]), ev(m, m.tv[t][2]));
	return dot(n, p - ev(m, m.tv[t][0])) > epsilon;
}
__device__ v4 expanding_polytope(epa_mem &m, const v3 start[4], const support_t &A, const support_t &B, int lane)
{
	v4 plane = V4(0, 0, 0, -FLT_MAX);
	const float epsilon = 0.001f;
	int nv = 4, nt = 0;
	for (int i = 0; i < 4; i++) { m.vx[i] = start[i].x; m.vy[i] = start[i].y; m.vz[i] = start[i].z; }
	v3 center = (((ev(m, 0) + ev(m, 1)) + ev(m, 2)) + ev(m, 3)) / 4.0f;
	if (dot(cross(ev(m, 2) - ev(m, 0), ev(m, 1) - ev(m, 0)), ev(m, 3) - ev(m, 0)) > 0.0f)
	{
		v3 a = ev(m, 2), b = ev(m, 3);
		m.vx[2] = b.x; m.vy[2] = b.y; m.vz[2] = b.z; m.vx[3] = a.x; m.vy[3] = a.y; m.vz[3] = a.z;
	}
	tri_set(m, nt++, 2, 3, 1, 0, 2, 3, 1); tri_set(m, nt++, 3, 2, 0, 1, 3, 2, 0); tri_set(m, nt++, 0, 1, 3, 2, 0, 1, 3); tri_set(m, nt++, 1, 0, 2, 3, 1, 0, 2);
	for (int guard = 0; guard < 128; guard++)
	{
		v4 face = V4(0, 0, 0, -FLT_MAX);
		for (int i = 0; i < nt; i++)
		{
			v3 n = tri_normal(ev(m, m.tv[i][0]), ev(m, m.tv[i][1]), ev(m, m.tv[i][2]));
			float d = -dot(n, ev(m, m.tv[i][0]));
			if (d > face.w) face = V4(n, d);
		}
		v3 v = support(A, xyz(face), lane) - support(B, -xyz(face), lane);
		v4 p = V4(xyz(face), -dot(xyz(face), v));
		if (p.w > plane.w) plane = p;
		bool dup = false;
		for (int i = 0; i < nv; i++) if (same(v, ev(m, i))) { dup = true; break; }
		if (dup) break;
		if (plane.w >= face.w - epsilon) break;
		if (nv >= EPA_MAXV) break;
		const int vid = nv;
		m.vx[nv] = v.x; m.vy[nv] = v.y; m.vz[nv] = v.z; nv++;
		bool okk = true;
		int j = nt;
		while (j--)
		{
			if (tri_dead(m, j)) continue;
			if (above(m, j, ev(m, vid), 0.01f * epsilon)) okk = okk && extrude(m, nt, j, vid);
		}
		j = nt;
		while (okk && j--)
		{
			if (tri_dead(m, j)) continue;
			if (!hasvert(m, j, vid)) break;
			v3 a = ev(m, m.tv[j][0]), b = ev(m, m.tv[j][1]), c = ev(m, m.tv[j][2]);
			if (above(m, j, center, 0.01f * epsilon) || length(cross(b - a, c - b)) < epsilon * epsilon * 0.1f)
			{
				int nb = m.tn[j][0];
				okk = extrude(m, nt, nb, vid);
				j = nt;
			}
		}
		if (!okk) break;
		j = nt;
		while (j--)
		{
			if (!tri_dead(m, j)) continue;
			swapn(m, j, nt - 1);
			nt--;
		}
	}
	return plane;
}
// last column of inverse(float4x4({c0,1},{c1,1},{c2,1},{c3,1})) with the cofactor expressions of linalg.h:321-331
__device__ v4 inverse_w(v3 c0, v3 c1, v3 c2, v3 c3)
{
	struct { v4 x, y, z, w; } a = { V4(c0, 1), V4(c1, 1), V4(c2, 1), V4(c3, 1) };
	v4 adjw = V4(
		a.y.x * a.w.y * a.z.z + a.z.x * a.y.y * a.w.z + a.w.x * a.z.y * a.y.z - a.y.x * a.z.y * a.w.z - a.w.x * a.y.y * a.z.z - a.z.x * a.w.y * a.y.z,
		a.x.x * a.z.y * a.w.z + a.w.x * a.x.y * a.z.z + a.z.x * a.w.y * a.x.z - a.x.x * a.w.y * a.z.z - a.z.x * a.x.y * a.w.z - a.w.x * a.z.y * a.x.z,
		a.x.x * a.w.y * a.y.z + a.y.x * a.x.y * a.w.z + a.w.x * a.y.y * a.x.z - a.x.x * a.y.y * a.w.z - a.w.x * a.x.y * a.y.z - a.y.x * a.w.y * a.x.z,
		a.x.x * a.y.y * a.z.z + a.z.x * a.x.y * a.y.z + a.y.x * a.z.y * a.x.z - a.x.x * a.z.y * a.y.z - a.y.x * a.x.y * a.z.z - a.z.x * a.y.y * a.x.z);
	float det = a.x.x * (a.y.y * a.z.z * a.w.w + a.w.y * a.y.z * a.z.w + a.z.y * a.w.z * a.y.w - a.y.y * a.w.z * a.z.w - a.z.y * a.y.z * a.w.w - a.w.y * a.z.z * a.y.w)
	          + a.x.y * (a.y.z * a.w.w * a.z.x + a.z.z * a.y.w * a.w.x + a.w.z * a.z.w * a.y.x - a.y.z * a.z.w * a.w.x - a.w.z * a.y.w * a.z.x - a.z.z * a.w.w * a.y.x)
	          + a.x.z * (a.y.w * a.z.x * a.w.y + a.w.w * a.y.x * a.z.y + a.z.w * a.w.x * a.y.y - a.y.w * a.w.x * a.z.y - a.z.w * a.y.x * a.w.y - a.w.w * a.z.x * a.y.y)
	          + a.x.w * (a.y.x * a.w.y * a.z.z + a.z.x * a.y.y * a.w.z + a.w.x * a.z.y * a.y.z - a.y.x * a.z.y * a.w.z - a.w.x * a.y.y * a.z.z - a.z.x * a.w.y * a.y.z);
	return adjw / det;
}

// Separated(A, B, findclosest = 1), gjk.h:367-437
// `cutoff` > 0 enables an exact early-out: dot(w,v)/|v| is a lower bound of the distance between the shapes, and the separation the
// reference finally reports is never below it; once the bound exceeds the contact cut-off the pair cannot produce a contact.
__device__ gjk_hit separated(const support_t &A, const support_t &B, epa_mem &em, int lane, float cutoff, bool &far_apart)
{
	simplex last, next;
	last.count = 0; next.count = 0;
	for (int i = 0; i < 4; i++) { last.W[i].a = last.W[i].b = last.W[i].p = V3(0, 0, 0); last.W[i].t = 0; next.W[i] = last.W[i]; }
	int iter = 0;
	v3 v = point_on_minkowski(A, B, V3(0, 0, 1), lane).p;
	last.v = v;
	mkpoint w = point_on_minkowski(A, B, -v, lane);
	next.W[0] = w; next.W[0].t = 1.0f; next.v = w.p; next.count = 1;                        // NextMinkSimplex0
	for (;;)
	{
		bool go;
		if (iter == 0) { iter++; go = true; }
		else { iter++; go = (dot(w.p, v) < dot(v, v) - 0.00001f); if (go) { go = (iter < 100); iter++; } }      // while(!iter++ || (... && iter++<100))
		if (!go) break;
		last = next;
		v = last.v;
		w = point_on_minkowski(A, B, -v, lane);
		if (cutoff > 0.0f) { const float wv = dot(w.p, v); if (wv > 0.0f && wv > (cutoff * 1.01f + 1e-6f) * length(v)) { far_apart = true; break; } }
		if (dot(w.p, v) >= dot(v, v) - 0.00001f - 0.00001f * dot(v, v)) break;
		if (last.count == 1) next1(next, last, w); else if (last.count == 2) next2(next, last, w); else next3(next, last, w);
		if (is_zero(next.v))
		{
			if (next.count == 2) { last = next; v3 n = orth(next.W[0].p - next.W[1].p); next.W[2] = point_on_minkowski(A, B, n, lane); next.count = 3; }
			if (next.count == 3) { last = next; v3 n = tri_normal(next.W[0].p, next.W[1].p, next.W[2].p); next.W[3] = point_on_minkowski(A, B, n, lane); next.count = 4; }
			v3 start[4] = { next.W[0].p, next.W[1].p, next.W[2].p, next.W[3].p };
			v4 mpp = expanding_polytope(em, start, A, B, lane);
			gjk_hit h;
			h.normal = -xyz(mpp);
			h.separation = fmin_std(0.0f, mpp.w);
			v4 bw = inverse_w(next.W[0].p, next.W[1].p, next.W[2].p, next.W[3].p);
			h.p0w = ((next.W[0].a * bw.x + next.W[1].a * bw.y) + next.W[2].a * bw.z) + next.W[3].a * bw.w;
			h.p1w = ((next.W[0].b * bw.x + next.W[1].b * bw.y) + next.W[2].b * bw.z) + next.W[3].b * bw.w;
			return h;
		}
		if (dot(next.v, next.v) >= dot(last.v, last.v)) break;
	}
	return calcpoints(last);
}

// ------------------------------------------------------------------------------------------------- k_contacts
#define GJK_WAVES 4
#define GJK_WCAP 48         // contacts staged per wave
__global__ __launch_bounds__(64 * GJK_WAVES) void k_contacts(ht_model_dev M, const float *__restrict__ state, float driftmax, float jiggle_sin, const int *__restrict__ active_flag,
                                                             epa_mem *__restrict__ epa_ws, float *__restrict__ contacts, int *__restrict__ ncontacts, int dbg)
{
	__shared__ float spos[HT_MAXNB][8];             // pos3 q4 radius
	__shared__ unsigned char cand[HT_MAXNB * HT_MAXNB / 2][2];
	__shared__ int ncand;
	__shared__ unsigned char ccount[HT_MAXNB * HT_MAXNB / 2];
	__shared__ int cprefix[HT_MAXNB * HT_MAXNB / 2 + 1];
	__shared__ float wlist[GJK_WAVES][GJK_WCAP][HT_CONTACT];
	__shared__ unsigned short wcand[GJK_WAVES][GJK_WCAP];
	const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
	if (active_flag && !active_flag[b]) { if (t == 0) ncontacts[b] = 0; return; }
	epa_mem &em = epa_ws[(size_t)b * GJK_WAVES + wave];
	if (t < M.nb)
	{
		const float *s = state + ((size_t)b * M.nb + t) * HT_STATE_STRIDE;
		for (int i = 0; i < 7; i++) spos[t][i] = s[i];
		spos[t][7] = M.bodyc[t * HT_BC + HT_BC_RADIUS];
	}
	if (t == 0) ncand = 0;
	__syncthreads();
	// broad phase in the reference's pair order (physics.h:453-457): wave 0 tests 64 pairs at a time and compacts with a ballot
	if (wave == 0)
	{
		const int npairs = M.nb * (M.nb - 1) / 2;
		int k = 0;
		for (int base = 0; base < npairs; base += 64)
		{
			const int pidx = base + lane;
			// pair index -> (i, j), i < j, row-major over the upper triangle
			int i = 0, rem = pidx;
			while (i < M.nb - 1 && rem >= M.nb - 1 - i) { rem -= M.nb - 1 - i; i++; }
			const int j = i + 1 + rem;
			bool keep = false;
			if (pidx < npairs)
			{
				keep = (M.collide[i] & M.collide[j] & 2) != 0;
				v3 d = V3(spos[j][0], spos[j][1], spos[j][2]) - V3(spos[i][0], spos[i][1], spos[i][2]);
				if (length(d) > spos[i][7] + spos[j][7]) keep = false;
				if (M.ignore[i] & (1u << j)) keep = false;
			}
			const unsigned long long m = __ballot(keep);
			if (keep) { const int dst = k + __popcll(m & ((1ull << lane) - 1ull)); cand[dst][0] = (unsigned char)i; cand[dst][1] = (unsigned char)j; }
			k += __popcll(m);
		}
		if (lane == 0) ncand = k;
	}
	__syncthreads();
	const int nc = (dbg & 8) ? 0 : ncand;
	for (int c = t; c < nc; c += 64 * GJK_WAVES) ccount[c] = 0;
	__syncthreads();
	int wn = 0;
	for (int c = wave; c < nc; c += GJK_WAVES)
	{
		const int i = cand[c][0], j = cand[c][1];
		support_t A, Bs;
		A.verts = M.verts + M.vert_off[i]; A.n = M.vert_off[i + 1] - M.vert_off[i]; A.pos = V3(spos[i][0], spos[i][1], spos[i][2]); A.q = V4(spos[i][3], spos[i][4], spos[i][5], spos[i][6]); A.outer = 0; A.opos = V3(0, 0, 0); A.oq = V4(0, 0, 0, 1);
		Bs.verts = M.verts + M.vert_off[j]; Bs.n = M.vert_off[j + 1] - M.vert_off[j]; Bs.pos = V3(spos[j][0], spos[j][1], spos[j][2]); Bs.q = V4(spos[j][3], spos[j][4], spos[j][5], spos[j][6]); Bs.outer = 0; Bs.opos = V3(0, 0, 0); Bs.oq = V4(0, 0, 0, 1);
		bool far_apart = false;
		const gjk_hit h0 = separated(A, Bs, em, lane, (dbg & 16) ? 0.0f : driftmax, far_apart);
		int kept = 0;
		auto stage = [&](const gjk_hit &h) {
			if (wn < GJK_WCAP)
			{
				if (lane == 0)
				{
					float *o = wlist[wave][wn];
					o[0] = (float)i; o[1] = (float)j; o[2] = h.normal.x; o[3] = h.normal.y; o[4] = h.normal.z;
					o[5] = h.p0w.x; o[6] = h.p0w.y; o[7] = h.p0w.z; o[8] = h.p1w.x; o[9] = h.p1w.y; o[10] = h.p1w.z; o[11] = h.separation;
					wcand[wave][wn] = (unsigned short)c;
				}
				wn++; kept++;
			}
		};
		if (!far_apart && !(h0.separation > driftmax))
		{
			const int first = wn;
			stage(h0);
			const float dmin = fminf(M.bodyc[i * HT_BC + HT_BC_DIAM], M.bodyc[j * HT_BC + HT_BC_DIAM]);
			if (!(dmin < 0.049f))         // otherwise every jiggle sample is rejected by the 0.05 m proximity test (see header)
			{
				const v3 n = h0.normal;
				v4 qs = quat_from_to(n, V3(0, 0, 1));
				v3 tangent = qxdir(qs), bitangent = qydir(qs);
				for (int r = 0; r < 4; r++)
				{
					const v3 raxis = r == 0 ? tangent : r == 1 ? bitangent : r == 2 ? -tangent : -bitangent;
					v4 jiggle = normalize(V4(raxis * jiggle_sin, 1));
					v3 pivot = h0.p0w;
					v4 id = V4(0, 0, 0, 1);
					xf ar = mul(mul(mul(XF(n * 0.2f, id), XF(-pivot, id)), XF(V3(0, 0, 0), jiggle)), XF(pivot, id));
					support_t AJ = A; AJ.outer = 1; AJ.opos = ar.p; AJ.oq = ar.q;
					bool dummy = false;
					gjk_hit hj = separated(AJ, Bs, em, lane, 0.0f, dummy);
					hj.normal = n;
					hj.p0w = apply(inverse(ar), hj.p0w);
					hj.separation = dot(n, hj.p0w - hj.p1w);
					bool match = false;
					for (int q = first; !match && q < wn; q++)
					{
						const float *o = wlist[wave][q];
						match = length(hj.p0w - V3(o[5], o[6], o[7])) < 0.05f || length(hj.p1w - V3(o[8], o[9], o[10])) < 0.05f;
					}
					if (match) continue;
					stage(hj);
				}
			}
		}
		if (lane == 0) ccount[c] = (unsigned char)kept;
	}
	__syncthreads();
	if (t == 0)
	{
		int acc = 0;
		for (int c = 0; c < nc; c++) { cprefix[c] = acc; acc += ccount[c]; }
		cprefix[nc] = acc;
		ncontacts[b] = acc < HT_MAXCONTACT ? acc : HT_MAXCONTACT;
	}
	__syncthreads();
	// each wave writes its staged contacts at their rank in pair order
	{
		int prevc = -1, k = 0;
		for (int e = 0; e < wn; e++)
		{
			int c = wcand[wave][e];
			k = (c == prevc) ? k + 1 : 0; prevc = c;
			int dst = cprefix[c] + k;
			if (dst < HT_MAXCONTACT && lane < HT_CONTACT) contacts[((size_t)b * HT_MAXCONTACT + dst) * HT_CONTACT + lane] = wlist[wave][e][lane];
		}
	}
}

size_t ht_contacts_workspace_bytes(int B) { return (size_t)B * GJK_WAVES * sizeof(epa_mem); }
void ht_launch_contacts(const ht_model_dev &M, const float *state, float driftmax, float jiggle_sin, const int *active_flag, void *epa_ws, float *contacts, int *ncontacts, int B, hipStream_t s)
{
	hipLaunchKernelGGL(k_contacts, dim3(B), dim3(64 * GJK_WAVES), 0, s, M, state, driftmax, jiggle_sin, active_flag, (epa_mem *)epa_ws, contacts, ncontacts, []{ const char *e = getenv("HT_DEBUG_SKIP"); return e ? atoi(e) : 0; }());
}
