/*
 * ho_gjk.c -- CPU restatement of the reference narrow phase: GJK closest features (third_party/gjk.h:82-437),
 * the expanding-polytope fallback for penetration (third_party/hull.h:79-186, 233-310) and the 5-sample contact
 * patch (gjk.h:607-643), specialised to posed convex vertex sets (SupportFunc / SupportFuncTrans gjk.h:568-582).
 *
 * TEST INFRASTRUCTURE ONLY (oracle/).  Contact::type / v[] (fillhitv, gjk.h:277-336) are not produced: nothing on the
 * hot path reads them (PhysContact uses normal, p0w, p1w, separation only, physics.h:425-434).
 */
#include <stdlib.h>
#include <string.h>
#include "ht_oracle.h"

typedef struct { const f3 *verts; int n; f3 pos; f4 q; int outer; f3 opos; f4 oq; } support_t;
int ho_maxdir(const f3 *p, int count, f3 dir);

static f3 support_inner(const support_t *s, f3 dir)
{
	f3 dl = qrot(qconj(s->q), dir);
	return add3(s->pos, qrot(s->q, s->verts[ho_maxdir(s->verts, s->n, dl)]));
}
static f3 support(const support_t *s, f3 dir)
{
	if (s->outer) return add3(s->opos, qrot(s->oq, support_inner(s, qrot(qconj(s->oq), dir))));
	return support_inner(s, dir);
}

typedef struct { f3 a, b, p; float t; } mkpoint;
typedef struct { f3 v; mkpoint W[4]; int count; f3 pa, pb; } simplex;

static mkpoint point_on_minkowski(const support_t *A, const support_t *B, f3 n)   /* gjk.h:68-73 */
{
	mkpoint m; m.a = support(A, n); m.b = support(B, neg3(n)); m.p = sub3(m.a, m.b); m.t = 0; return m;
}
static const f3 ORG = { 0, 0, 0 };

static void next0(simplex *dst, const simplex *src, const mkpoint *w)   /* gjk.h:82-91 */
{
	(void)src; dst->W[0] = *w; dst->W[0].t = 1.0f; dst->v = w->p; dst->count = 1;
}
static void next1(simplex *dst, const simplex *src, const mkpoint *w)   /* gjk.h:93-112 */
{
	float t = line_project_time(w->p, src->W[0].p, ORG);
	if (t < 0.0f) { dst->W[0] = *w; dst->W[0].t = 1.0f; dst->v = w->p; dst->count = 1; return; }
	dst->W[0] = src->W[0]; dst->W[0].t = t;
	dst->W[1] = *w; dst->W[1].t = 1.0f - t;
	dst->v = add3(w->p, scale3(sub3(src->W[0].p, w->p), t));
	dst->count = 2;
}
static void keep_edge(simplex *dst, const mkpoint *keep, const mkpoint *w, float t, f3 v)
{
	mkpoint k = *keep;
	dst->W[0] = k; dst->W[0].t = t; dst->W[1] = *w; dst->W[1].t = 1.0f - t; dst->v = v; dst->count = 2;
}
static void next2(simplex *dst, const simplex *src, const mkpoint *w)   /* gjk.h:114-164 */
{
	f3 w0 = src->W[0].p, w1 = src->W[1].p;
	float t0 = line_project_time(w->p, w0, ORG);
	float t1 = line_project_time(w->p, w1, ORG);
	f3 v0 = add3(w->p, scale3(sub3(w0, w->p), t0));
	f3 v1 = add3(w->p, scale3(sub3(w1, w->p), t1));
	int ine0 = (dot3(neg3(v0), sub3(w1, v0)) > 0.0f);
	int ine1 = (dot3(neg3(v1), sub3(w0, v1)) > 0.0f);
	if (ine0 && ine1)
	{
		dst->count = 3; dst->v = plane_project_of(w0, w1, w->p, ORG);
		mkpoint a = src->W[0], b = src->W[1];
		dst->W[0] = a; dst->W[1] = b; dst->W[2] = *w;
		return;
	}
	if (!ine0 && (t0 > 0.0f)) { keep_edge(dst, &src->W[0], w, t0, v0); return; }
	if (!ine1 && (t1 > 0.0f)) { keep_edge(dst, &src->W[1], w, t1, v1); return; }
	dst->W[0] = *w; dst->W[0].t = 1.0f; dst->v = w->p; dst->count = 1;
}
static int next3(simplex *dst, const simplex *src, const mkpoint *w)   /* gjk.h:166-275; returns 0 when the origin is enclosed */
{
	f3 w0 = src->W[0].p, w1 = src->W[1].p, w2 = src->W[2].p;
	float t[3]; f3 v[3], vc[3];
	t[0] = line_project_time(w->p, w0, ORG); t[1] = line_project_time(w->p, w1, ORG); t[2] = line_project_time(w->p, w2, ORG);
	v[0] = add3(w->p, scale3(sub3(w0, w->p), t[0]));
	v[1] = add3(w->p, scale3(sub3(w1, w->p), t[1]));
	v[2] = add3(w->p, scale3(sub3(w2, w->p), t[2]));
	vc[0] = plane_project_of(w->p, w1, w2, ORG);
	vc[1] = plane_project_of(w->p, w2, w0, ORG);
	vc[2] = plane_project_of(w->p, w0, w1, ORG);
	int inp0 = (dot3(neg3(vc[0]), sub3(w0, vc[0])) > 0.0f);
	int inp1 = (dot3(neg3(vc[1]), sub3(w1, vc[1])) > 0.0f);
	int inp2 = (dot3(neg3(vc[2]), sub3(w2, vc[2])) > 0.0f);
	mkpoint s0 = src->W[0], s1 = src->W[1], s2 = src->W[2];
	if (inp0 && inp1 && inp2)
	{
		simplex tmp = *src; *dst = tmp; dst->count = 4; dst->v = F3(0, 0, 0); dst->W[3] = *w; return 0;
	}
	int inp2e0 = (dot3(neg3(v[0]), sub3(w1, v[0])) > 0.0f);
	int inp2e1 = (dot3(neg3(v[1]), sub3(w0, v[1])) > 0.0f);
	if (!inp2 && inp2e0 && inp2e1) { dst->count = 3; dst->v = plane_project_of(w0, w1, w->p, ORG); dst->W[0] = s0; dst->W[1] = s1; dst->W[2] = *w; return 1; }
	int inp0e1 = (dot3(neg3(v[1]), sub3(w2, v[1])) > 0.0f);
	int inp0e2 = (dot3(neg3(v[2]), sub3(w1, v[2])) > 0.0f);
	if (!inp0 && inp0e1 && inp0e2) { dst->count = 3; dst->v = plane_project_of(w1, w2, w->p, ORG); dst->W[0] = s1; dst->W[1] = s2; dst->W[2] = *w; return 1; }
	int inp1e2 = (dot3(neg3(v[2]), sub3(w0, v[2])) > 0.0f);
	int inp1e0 = (dot3(neg3(v[0]), sub3(w2, v[0])) > 0.0f);
	if (!inp1 && inp1e2 && inp1e0) { dst->count = 3; dst->v = plane_project_of(w2, w0, w->p, ORG); dst->W[0] = s2; dst->W[1] = s0; dst->W[2] = *w; return 1; }
	if (!inp1e0 && !inp2e0 && t[0] > 0.0f) { keep_edge(dst, &s0, w, t[0], v[0]); return 1; }
	if (!inp2e1 && !inp0e1 && t[1] > 0.0f) { keep_edge(dst, &s1, w, t[1], v[1]); return 1; }
	if (!inp0e2 && !inp1e2 && t[2] > 0.0f) { keep_edge(dst, &s2, w, t[2], v[2]); return 1; }
	dst->W[0] = *w; dst->W[0].t = 1.0f; dst->v = w->p; dst->count = 1;
	return 1;
}

static ho_gjk_contact calcpoints(simplex *src)   /* gjk.h:337-363 */
{
	if (src->count == 3)
	{
		f3 b = barycentric(src->W[0].p, src->W[1].p, src->W[2].p, src->v);
		src->W[0].t = b.x; src->W[1].t = b.y; src->W[2].t = b.z;
	}
	src->pa = src->pb = F3(0, 0, 0);
	for (int i = 0; i < src->count; i++)
	{
		src->pa = add3(src->pa, scale3(src->W[i].a, src->W[i].t));
		src->pb = add3(src->pb, scale3(src->W[i].b, src->W[i].t));
	}
	ho_gjk_contact h;
	h.type = -1;
	h.p0w = src->pa; h.p1w = src->pb;
	h.impact = scale3(add3(src->pa, src->pb), 0.5f);
	h.separation = length3(sub3(src->pa, src->pb)) + FLT_MIN;
	h.normal = normalize3(src->v);
	h.dist = -dot3(h.normal, h.impact);
	return h;
}

/* ---- expanding polytope (hull.h) ---- */
typedef struct { int v[3]; int n[3]; int id; } tri_t;
typedef struct { tri_t *t; int n, cap; } trivec;
static void tv_push(trivec *tv, int a, int b, int c, int id, int n0, int n1, int n2)
{
	if (tv->n == tv->cap) { tv->cap = tv->cap ? tv->cap * 2 : 64; tv->t = realloc(tv->t, sizeof(tri_t) * tv->cap); }
	tri_t *t = &tv->t[tv->n++]; t->v[0] = a; t->v[1] = b; t->v[2] = c; t->id = id; t->n[0] = n0; t->n[1] = n1; t->n[2] = n2;
}
static int tri_dead(const tri_t *t) { return t->n[0] == -1; }
static int hasvert(const int v[3], int x) { return v[0] == x || v[1] == x || v[2] == x; }
static int *neib(tri_t *t, int va, int vb)   /* hull.h:97-109 */
{
	for (int i = 0; i < 3; i++)
	{
		int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
		if (t->v[i] == va && t->v[i1] == vb) return &t->n[i2];
		if (t->v[i] == vb && t->v[i1] == va) return &t->n[i2];
	}
	abort();
}
static void nnfix(trivec *tv, int k)   /* hull.h:112-127 */
{
	if (tv->t[k].id == -1) return;
	for (int i = 0; i < 3; i++)
	{
		int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
		if (tv->t[k].n[i] != -1) *neib(&tv->t[tv->t[k].n[i]], tv->t[k].v[i2], tv->t[k].v[i1]) = k;
	}
}
static void swapn(trivec *tv, int a, int b)   /* hull.h:128-134 */
{
	tri_t tmp = tv->t[a]; tv->t[a] = tv->t[b]; tv->t[b] = tmp;
	int id = tv->t[a].id; tv->t[a].id = tv->t[b].id; tv->t[b].id = id;
	nnfix(tv, a); nnfix(tv, b);
}
static void b2bfix(trivec *tv, int s, int t)   /* hull.h:136-150 */
{
	for (int i = 0; i < 3; i++)
	{
		int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
		int va = tv->t[s].v[i1], vb = tv->t[s].v[i2];
		*neib(&tv->t[*neib(&tv->t[s], va, vb)], vb, va) = *neib(&tv->t[t], vb, va);
		*neib(&tv->t[*neib(&tv->t[t], vb, va)], va, vb) = *neib(&tv->t[s], va, vb);
	}
	for (int i = 0; i < 3; i++) tv->t[s].n[i] = tv->t[t].n[i] = -1;
}
static void extrude(trivec *tv, int t0, int v)   /* hull.h:167-186 */
{
	int t[3] = { tv->t[t0].v[0], tv->t[t0].v[1], tv->t[t0].v[2] };
	int b = tv->n;
	int n[3] = { tv->t[t0].n[0], tv->t[t0].n[1], tv->t[t0].n[2] };
	tv_push(tv, v, t[1], t[2], b + 0, n[0], b + 1, b + 2); *neib(&tv->t[n[0]], t[1], t[2]) = b + 0;
	tv_push(tv, v, t[2], t[0], b + 1, n[1], b + 2, b + 0); *neib(&tv->t[n[1]], t[2], t[0]) = b + 1;
	tv_push(tv, v, t[0], t[1], b + 2, n[2], b + 0, b + 1); *neib(&tv->t[n[2]], t[0], t[1]) = b + 2;
	tv->t[t0].n[0] = tv->t[t0].n[1] = tv->t[t0].n[2] = -1;
	if (hasvert(tv->t[n[0]].v, v)) b2bfix(tv, b + 0, n[0]);
	if (hasvert(tv->t[n[1]].v, v)) b2bfix(tv, b + 1, n[1]);
	if (hasvert(tv->t[n[2]].v, v)) b2bfix(tv, b + 2, n[2]);
}
static int above(const f3 *verts, const int t[3], f3 p, float epsilon)   /* hull.h:50-54 */
{
	f3 n = tri_normal(verts[t[0]], verts[t[1]], verts[t[2]]);
	return dot3(n, sub3(p, verts[t[0]])) > epsilon;
}
static f4 expanding_polytope(const f3 start[4], const support_t *A, const support_t *B)   /* hull.h:233-310 */
{
	f4 plane = F4(0, 0, 0, -FLT_MAX);
	float epsilon = 0.001f;
	int nv = 4, vcap = 64;
	f3 *verts = malloc(sizeof(f3) * vcap);
	memcpy(verts, start, sizeof(f3) * 4);
	trivec tv = { NULL, 0, 0 };
	f3 center = div3(add3(add3(add3(verts[0], verts[1]), verts[2]), verts[3]), 4.0f);
	if (dot3(cross3(sub3(verts[2], verts[0]), sub3(verts[1], verts[0])), sub3(verts[3], verts[0])) > 0.0f) { f3 tmp = verts[2]; verts[2] = verts[3]; verts[3] = tmp; }
	tv_push(&tv, 2, 3, 1, 0, 2, 3, 1); tv_push(&tv, 3, 2, 0, 1, 3, 2, 0); tv_push(&tv, 0, 1, 3, 2, 0, 1, 3); tv_push(&tv, 1, 0, 2, 3, 1, 0, 2);
	for (int guard = 0; guard < 256; guard++)
	{
		f4 face = F4(0, 0, 0, -FLT_MAX);
		for (int i = 0; i < tv.n; i++)
		{
			tri_t *t = &tv.t[i];
			f3 n = tri_normal(verts[t->v[0]], verts[t->v[1]], verts[t->v[2]]);
			float d = -dot3(n, verts[t->v[0]]);
			if (d > face.w) face = F4v(n, d);
		}
		f3 v = sub3(support(A, xyz(face)), support(B, neg3(xyz(face))));
		f4 p = F4v(xyz(face), -dot3(xyz(face), v));
		if (p.w > plane.w) plane = p;
		int dup = 0;
		for (int i = 0; i < nv; i++) if (eq3(v, verts[i])) { dup = 1; break; }
		if (dup) break;
		if (plane.w >= face.w - epsilon) break;
		int vid = nv;
		if (nv == vcap) { vcap *= 2; verts = realloc(verts, sizeof(f3) * vcap); }
		verts[nv++] = v;
		int j = tv.n;
		while (j--)
		{
			if (tri_dead(&tv.t[j])) continue;
			int t[3] = { tv.t[j].v[0], tv.t[j].v[1], tv.t[j].v[2] };
			if (above(verts, t, verts[vid], 0.01f * epsilon)) extrude(&tv, j, vid);
		}
		j = tv.n;
		while (j--)
		{
			if (tri_dead(&tv.t[j])) continue;
			if (!hasvert(tv.t[j].v, vid)) break;
			int nt[3] = { tv.t[j].v[0], tv.t[j].v[1], tv.t[j].v[2] };
			if (above(verts, nt, center, 0.01f * epsilon) || length3(cross3(sub3(verts[nt[1]], verts[nt[0]]), sub3(verts[nt[2]], verts[nt[1]]))) < epsilon * epsilon * 0.1f)
			{
				int nb = tv.t[j].n[0];
				extrude(&tv, nb, vid);
				j = tv.n;
			}
		}
		j = tv.n;
		while (j--)
		{
			if (!tri_dead(&tv.t[j])) continue;
			swapn(&tv, j, tv.n - 1);
			tv.n--;
		}
	}
	free(verts); free(tv.t);
	return plane;
}

/* last column of inverse(float4x4(c0,c1,c2,c3)) : linalg.h:321-331 (adjugate().w / determinant) */
static f4 m44_inverse_w(f4 cx, f4 cy, f4 cz, f4 cw)
{
	m44 a = { cx, cy, cz, cw };
	f4 adjw = F4(
		a.y.x * a.w.y * a.z.z + a.z.x * a.y.y * a.w.z + a.w.x * a.z.y * a.y.z - a.y.x * a.z.y * a.w.z - a.w.x * a.y.y * a.z.z - a.z.x * a.w.y * a.y.z,
		a.x.x * a.z.y * a.w.z + a.w.x * a.x.y * a.z.z + a.z.x * a.w.y * a.x.z - a.x.x * a.w.y * a.z.z - a.z.x * a.x.y * a.w.z - a.w.x * a.z.y * a.x.z,
		a.x.x * a.w.y * a.y.z + a.y.x * a.x.y * a.w.z + a.w.x * a.y.y * a.x.z - a.x.x * a.y.y * a.w.z - a.w.x * a.x.y * a.y.z - a.y.x * a.w.y * a.x.z,
		a.x.x * a.y.y * a.z.z + a.z.x * a.x.y * a.y.z + a.y.x * a.z.y * a.x.z - a.x.x * a.z.y * a.y.z - a.y.x * a.x.y * a.z.z - a.z.x * a.y.y * a.x.z);
	float det = a.x.x * (a.y.y * a.z.z * a.w.w + a.w.y * a.y.z * a.z.w + a.z.y * a.w.z * a.y.w - a.y.y * a.w.z * a.z.w - a.z.y * a.y.z * a.w.w - a.w.y * a.z.z * a.y.w)
	          + a.x.y * (a.y.z * a.w.w * a.z.x + a.z.z * a.y.w * a.w.x + a.w.z * a.z.w * a.y.x - a.y.z * a.z.w * a.w.x - a.w.z * a.y.w * a.z.x - a.z.z * a.w.w * a.y.x)
	          + a.x.z * (a.y.w * a.z.x * a.w.y + a.w.w * a.y.x * a.z.y + a.z.w * a.w.x * a.y.y - a.y.w * a.w.x * a.z.y - a.z.w * a.y.x * a.w.y - a.w.w * a.z.x * a.y.y)
	          + a.x.w * (a.y.x * a.w.y * a.z.z + a.z.x * a.y.y * a.w.z + a.w.x * a.z.y * a.y.z - a.y.x * a.z.y * a.w.z - a.w.x * a.y.y * a.z.z - a.z.x * a.w.y * a.y.z);
	return div4(adjw, det);
}
static f3 m44_mulv_xyz(f3 c0, f3 c1, f3 c2, f3 c3, f4 b)   /* mul(float4x4({c0,1},...), b).xyz(), linalg.h:297 */
{
	return add3(add3(add3(scale3(c0, b.x), scale3(c1, b.y)), scale3(c2, b.z)), scale3(c3, b.w));
}

int ho_dbg_calls, ho_dbg_iters, ho_dbg_maxiter;      /* iteration statistics for sizing the device kernels */
/* Separated(A,B,findclosest=1), gjk.h:367-437 */
static ho_gjk_contact separated(const support_t *A, const support_t *B)
{
	simplex last, next;
	memset(&last, 0, sizeof last); memset(&next, 0, sizeof next);
	int iter = 0;
	f3 v = point_on_minkowski(A, B, F3(0, 0, 1)).p;
	last.count = 0; last.v = v;
	mkpoint w = point_on_minkowski(A, B, neg3(v));
	next0(&next, &last, &w);
	int loops = 0;
	ho_dbg_calls++;
	for (;;)
	{
		int go;
		loops++; ho_dbg_iters++; if (loops > ho_dbg_maxiter) ho_dbg_maxiter = loops;
		if (iter == 0) { iter++; go = 1; }
		else
		{
			iter++;
			go = (dot3(w.p, v) < dot3(v, v) - 0.00001f);
			if (go) { go = (iter < 100); iter++; }
		}
		if (!go) break;
		last = next;
		v = last.v;
		w = point_on_minkowski(A, B, neg3(v));
		if (dot3(w.p, v) >= dot3(v, v) - 0.00001f - 0.00001f * dot3(v, v)) break;
		switch (last.count)
		{
		case 0: next0(&next, &last, &w); break;
		case 1: next1(&next, &last, &w); break;
		case 2: next2(&next, &last, &w); break;
		default: next3(&next, &last, &w); break;
		}
		if (next.v.x == 0 && next.v.y == 0 && next.v.z == 0)
		{
			if (next.count == 2)
			{
				last = next;
				f3 n = ho_orth(sub3(next.W[0].p, next.W[1].p));
				next.W[next.count++] = point_on_minkowski(A, B, n);
			}
			if (next.count == 3)
			{
				last = next;
				f3 n = tri_normal(next.W[0].p, next.W[1].p, next.W[2].p);
				next.W[next.count++] = point_on_minkowski(A, B, n);
			}
			f3 start[4] = { next.W[0].p, next.W[1].p, next.W[2].p, next.W[3].p };
			f4 mpp = expanding_polytope(start, A, B);
			ho_gjk_contact h;
			h.type = -1;
			h.normal = neg3(xyz(mpp));
			h.dist = -mpp.w;
			h.separation = ho_minf(0.0f, mpp.w);
			f4 b = m44_inverse_w(F4v(next.W[0].p, 1), F4v(next.W[1].p, 1), F4v(next.W[2].p, 1), F4v(next.W[3].p, 1));
			h.p0w = m44_mulv_xyz(next.W[0].a, next.W[1].a, next.W[2].a, next.W[3].a, b);
			h.p1w = m44_mulv_xyz(next.W[0].b, next.W[1].b, next.W[2].b, next.W[3].b, b);
			h.impact = scale3(add3(h.p0w, h.p1w), 0.5f);
			return h;
		}
		if (dot3(next.v, next.v) >= dot3(last.v, last.v)) break;
	}
	return calcpoints(&last);
}

static support_t body_support(const ho_body *b)
{
	support_t s; memset(&s, 0, sizeof s);
	s.verts = b->shape.verts; s.n = b->shape.nverts; s.pos = b->position; s.q = b->orientation; s.outer = 0;
	return s;
}
ho_gjk_contact ho_separated_bodies(const ho_body *a, const ho_body *b)
{
	support_t A = body_support(a), B = body_support(b);
	return separated(&A, &B);
}

/* ContactPatch, gjk.h:607-643 */
int ho_contact_patch_bodies(const ho_body *a, const ho_body *b, float max_separation, ho_gjk_contact *hit)
{
	support_t A = body_support(a), B = body_support(b);
	int count = 0;
	hit[0] = separated(&A, &B);
	if (hit[0].separation > max_separation) return 0;
	f3 n = hit[0].normal;
	int hc = ++count;
	f4 qs = quat_from_to(n, F3(0, 0, 1));
	f3 tangent = qxdir(qs), bitangent = qydir(qs);
	f3 rollaxes[4] = { tangent, bitangent, neg3(tangent), neg3(bitangent) };
	for (int r = 0; r < 4; r++)
	{
		const float contactpatchjiggle = 4.0f;
		f4 jiggle = normalize4(F4v(scale3(rollaxes[r], sinf(3.14f / 180.0f * (contactpatchjiggle) / 2.0f)), 1));
		f3 pivot = hit[0].p0w;
		f4 id = F4(0, 0, 0, 1);
		pose_t ar = pose_mul(pose_mul(pose_mul(POSE(scale3(n, 0.2f), id), POSE(neg3(pivot), id)), POSE(F3(0, 0, 0), jiggle)), POSE(pivot, id));
		support_t AJ = A; AJ.outer = 1; AJ.opos = ar.position; AJ.oq = ar.orientation;
		hit[hc] = separated(&AJ, &B);
		hit[hc].normal = n;
		hit[hc].p0w = pose_apply(pose_inverse(ar), hit[hc].p0w);
		hit[hc].separation = dot3(n, sub3(hit[hc].p0w, hit[hc].p1w));
		int match = 0;
		for (int j = 0; !match && j < hc; j++)
			match = length3(sub3(hit[hc].p0w, hit[j].p0w)) < 0.05f || length3(sub3(hit[hc].p1w, hit[j].p1w)) < 0.05f;
		if (match) continue;
		hc++;
	}
	return hc;
}

/* FindShapeShapeContacts physics.h:451-462 + PhysContact physics.h:425-434 */
int ho_find_contacts(ho_tracker *t, ho_model *m, ho_contact *out, int cap)
{
	int n = 0;
	for (int i = 0; i < m->nb; i++) for (int j = 0; j < m->nb; j++) if (i < j)
	{
		ho_body *rb0 = &m->bodies[i], *rb1 = &m->bodies[j];
		if (!(rb0->collide & rb1->collide & 2)) continue;
		if (length3(sub3(rb1->position, rb0->position)) > rb0->radius + rb1->radius) continue;
		if (m->ignore[i][j]) continue;
		ho_gjk_contact hit[5];
		int cnt = ho_contact_patch_bodies(rb0, rb1, t->phys.driftmax, hit);
		for (int k = 0; k < cnt && n < cap; k++)
		{
			ho_contact c; c.rb0 = i; c.rb1 = j; c.normal = hit[k].normal; c.p0w = hit[k].p0w; c.p1w = hit[k].p1w; c.separation = hit[k].separation;
			c.p0 = pose_apply(pose_inverse(POSE(rb0->position, rb0->orientation)), c.p0w);
			c.p1 = pose_apply(pose_inverse(POSE(rb1->position, rb1->orientation)), c.p1w);
			out[n++] = c;
		}
	}
	return n;
}
