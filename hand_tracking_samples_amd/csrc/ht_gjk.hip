// ht_gjk.hip -- bone-bone narrow phase on CDNA4: broad-phase cull, GJK closest features, expanding-polytope fallback
// for penetration, and the 5-sample contact patch.
//
// Reference computations:
//   FindShapeShapeContacts            third_party/physics.h:451-462
//   ContactPatch / Separated          third_party/gjk.h:607-643, 367-437  (NextMinkSimplex0..3 :82-275, calcpoints :337-363)
//   ExpandingPolytopeAlgorithm        third_party/hull.h:233-310 (Tri bookkeeping :79-186)
//   SupportFunc / SupportFuncTrans    third_party/gjk.h:568-582, maxdir third_party/geometric.h:218-224
//
// Two organisations of the same arithmetic (ht_launch_contacts):
//   k_contacts_coop (every batch size, when a frame of the model fits its LDS): lane-per-run simplex logic in owner waves, support scans worked off cooperatively by all waves of a
//     block, one scan pair per DPP row, polytope jobs taken by any wave -- described at the kernel below.  Owns the CU (156 KB of LDS, 8 waves of 242 VGPRs).
//   k_contacts (the fall-back for models whose padded vertices leave no room for a frame in the other's LDS; until round 3 also the faster one
//     above ~1100 frames): a block serves two frames (one wave each) and stages all collision vertices (3042 x float4 = 48 KB, w = vertex
//     index) in LDS once.  GJK runs on lane groups: a candidate pair is served by 1, 2 or 4 neighbouring lanes (4 when a frame has <= 16
//     candidates, 2 up to 32) that hold identical simplex state, scan interleaved 6-vertex blocks of the support map (128-bit LDS reads issued
//     together, packed x/y multiply, compare in index order: first maximum wins as std::max_element does) and agree through DPP quad permutes; the
//     simplex logic is the reference's branchy code executed per lane.  Measured on the animation bank: 3.5 iterations per pair on average, ~23
//     pairs per frame.  Contacts are compacted in pair order with a prefix sum, so the solver sees the reference's row order.  61 KB of LDS: two
//     blocks per CU, which leaves room for the cloud-row kernel of the same fit step beside it -- with many frames per CU the whole step is faster
//     this way although the kernel alone is not (cross-over measured a little above 1024 frames).
// Both hand a pair whose simplex encloses the origin to the expanding polytope; that part is rare but long, so it runs wave-cooperatively, one
// pair at a time per wave: triangles are scored one per lane, both shapes' support scans are walked together strided over the 64 lanes with a
// butterfly arg-max, and the mesh surgery (extrude / back-to-back fix / compaction, hull.h:136-186) visits only the triangles a ballot marks.
// (Tried and dropped: the triangle list in registers with v_readlane look-ups instead of LDS reads -- 2.1 against 1.5 ms per step.)
//
// The 4 extra "jiggle" GJK runs of the contact patch are skipped when they provably cannot add a contact: an extra
// sample is rejected if it lies within 0.05 m of an accepted one on either shape (gjk.h:637) and every sample lies in the
// convex hull of its shape, so a shape whose diameter is below 0.05 m can never contribute a second sample.
// A second exact shortcut: dot(w,v)/|v| is a lower bound of the distance between the shapes and the separation the reference
// finally reports is never below it, so once the bound exceeds the contact cut-off the pair cannot produce a contact.
#include <stdlib.h>
#include <mutex>
#include "ht_device.hpp"
#include "ht_launch.hpp"

extern __shared__ __attribute__((aligned(16))) float4 g_sm[];      // [0, nvert): collision vertices of all bodies; then per-wave areas

typedef float f32x2 __attribute__((ext_vector_type(2)));
struct support_t { int voff, n; v3 pos; v4 q; int outer; v3 opos; v4 oq; int sub, grp; };      // sub/grp: this lane is member `sub` of a group of `grp` (1, 2 or 4) lanes sharing the pair
struct mkpoint { v3 a, b; float t; __device__ __forceinline__ v3 p() const { return a - b; } };      // p = a - b is formed where it is used (the same IEEE subtraction as PointOnMinkowski's, gjk.h:68-73): a run's two simplices hold 24 registers less
struct simplex { v3 v; mkpoint W[4]; int count; };
struct gjk_hit { v3 normal, p0w, p1w; float separation; };

// ---- support maps ----------------------------------------------------------------------------------------------------
// per-lane scan (every lane its own shape)
// Per-pair scan.  A pair is served by a group of 1, 2 or 4 neighbouring lanes (same quad) that hold identical state; the members scan
// interleaved 6-vertex blocks and then agree on the first maximum (largest value, lowest index) through DPP quad permutes.
__device__ __forceinline__ v3 support_inner(const support_t &s, v3 dir)
{
	const v3 dl = qrot(qconj(s.q), dir);
	const float4 *vs = g_sm + s.voff;
	float4 q0 = vs[0];
	float best = dot(V3(q0.x, q0.y, q0.z), dl); int bi = 0;
	const f32x2 dlxy = { dl.x, dl.y };
	// six vertices per trip (162 and 258 are multiples of 6): the LDS reads of a trip are issued together, then compared in index order.
	// Vertex 0 is simply tested again (it cannot displace itself); a tail shorter than 6 is handled one by one by member 0.
	int i = 6 * s.sub;
	for (; i + 6 <= s.n; i += 6 * s.grp)
	{
		const float4 *g = vs + i;
		float d[6]; int id[6];
#pragma unroll
		for (int k = 0; k < 6; k++)
		{
			const float4 q = g[k];
			const f32x2 xy = f32x2{ q.x, q.y } * dlxy;       // packed multiply on the register pair the 128-bit read delivers
			d[k] = (xy.x + xy.y) + q.z * dl.z;                // = dot(vertex, dl) in the reference's order
			id[k] = __float_as_int(q.w);                       // w carries the vertex index
		}
#pragma unroll
		for (int k = 0; k < 6; k++) if (best < d[k]) { best = d[k]; bi = id[k]; }
	}
	if (s.sub == 0) for (i = (s.n / 6) * 6; i < s.n; i++)
	{
		const float4 q = vs[i];
		const float d = dot(V3(q.x, q.y, q.z), dl);
		if (best < d) { best = d; bi = i; }
	}
	if (s.grp > 1)
	{
		float ob = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(best), 0xB1, 0xF, 0xF, true)); int oi = __builtin_amdgcn_mov_dpp(bi, 0xB1, 0xF, 0xF, true);      // quad_perm:[1,0,3,2]
		if (best < ob || (ob == best && oi < bi)) { best = ob; bi = oi; }
		if (s.grp > 2)
		{
			ob = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(best), 0x4E, 0xF, 0xF, true)); oi = __builtin_amdgcn_mov_dpp(bi, 0x4E, 0xF, 0xF, true);              // quad_perm:[2,3,0,1]
			if (best < ob || (ob == best && oi < bi)) { best = ob; bi = oi; }
		}
	}
	float4 q = vs[bi];
	return s.pos + qrot(s.q, V3(q.x, q.y, q.z));
}
__device__ __forceinline__ v3 support(const support_t &s, v3 dir)
{
	if (s.outer) return s.opos + qrot(s.oq, support_inner(s, qrot(qconj(s.oq), dir)));
	return support_inner(s, dir);
}
// Wave-wide arg-max of (value, index): the larger value wins, equal values go to the lower index, index 0x7fffffff marks "nothing".  That
// order is total, so any reduction tree gives the same winner: four DPP exchanges inside each row of 16 lanes (no LDS round trips as with
// ds_bpermute shuffles), then the four row results are folded through v_readlane.  Every lane returns the winner.
__device__ __forceinline__ void amax_take(float &b, int &i, float ob, int oi)
{
	const bool take = oi != 0x7fffffff && (i == 0x7fffffff || b < ob || (ob == b && oi < i));
	b = take ? ob : b; i = take ? oi : i;
}
template <int CTRL> __device__ __forceinline__ void amax_dpp(float &b, int &i)
{
	const float ob = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), CTRL, 0xF, 0xF, false));
	const int oi = __builtin_amdgcn_update_dpp(0, i, CTRL, 0xF, 0xF, false);
	amax_take(b, i, ob, oi);
}
__device__ __forceinline__ void wave_argmax(float &b, int &i)
{
	amax_dpp<0xB1>(b, i);       // quad_perm [1,0,3,2]
	amax_dpp<0x4E>(b, i);       // quad_perm [2,3,0,1]
	amax_dpp<0x141>(b, i);      // row_half_mirror
	amax_dpp<0x140>(b, i);      // row_mirror
	float rb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 0)); int ri = __builtin_amdgcn_readlane(i, 0);
	amax_take(rb, ri, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 16)), __builtin_amdgcn_readlane(i, 16));
	amax_take(rb, ri, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 32)), __builtin_amdgcn_readlane(i, 32));
	amax_take(rb, ri, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 48)), __builtin_amdgcn_readlane(i, 48));
	b = rb; i = ri;
}
__device__ __forceinline__ mkpoint point_on_minkowski(const support_t &A, const support_t &B, v3 n)      // gjk.h:68-73
{
	mkpoint m; m.a = support(A, n); m.b = support(B, -n); m.t = 0; return m;
}

// ---- simplex updates (NextMinkSimplex1..3, gjk.h:93-275) -------------------------------------------------------------
__device__ void next1(simplex &dst, const simplex &src, const mkpoint &w)
{
	const v3 O = V3(0, 0, 0);
	float t = line_project_time(w.p(), src.W[0].p(), O);
	if (t < 0.0f) { dst.W[0] = w; dst.W[0].t = 1.0f; dst.v = w.p(); dst.count = 1; return; }
	dst.W[0] = src.W[0]; dst.W[0].t = t;
	dst.W[1] = w; dst.W[1].t = 1.0f - t;
	dst.v = w.p() + (src.W[0].p() - w.p()) * t;
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
	const v3 w0 = src.W[0].p(), w1 = src.W[1].p();
	float t0 = line_project_time(w.p(), w0, O), t1 = line_project_time(w.p(), w1, O);
	v3 v0 = w.p() + (w0 - w.p()) * t0, v1 = w.p() + (w1 - w.p()) * t1;
	int ine0 = (dot(-v0, w1 - v0) > 0.0f), ine1 = (dot(-v1, w0 - v1) > 0.0f);
	if (ine0 && ine1)
	{
		mkpoint a = src.W[0], b = src.W[1];
		dst.count = 3; dst.v = plane_project_of(w0, w1, w.p(), O); dst.W[0] = a; dst.W[1] = b; dst.W[2] = w; return;
	}
	if (!ine0 && (t0 > 0.0f)) { keep_edge(dst, src.W[0], w, t0, v0); return; }
	if (!ine1 && (t1 > 0.0f)) { keep_edge(dst, src.W[1], w, t1, v1); return; }
	dst.W[0] = w; dst.W[0].t = 1.0f; dst.v = w.p(); dst.count = 1;
}
__device__ void next3(simplex &dst, const simplex &src, const mkpoint &w)
{
	const v3 O = V3(0, 0, 0);
	const v3 w0 = src.W[0].p(), w1 = src.W[1].p(), w2 = src.W[2].p();
	float t0 = line_project_time(w.p(), w0, O), t1 = line_project_time(w.p(), w1, O), t2 = line_project_time(w.p(), w2, O);
	v3 v0 = w.p() + (w0 - w.p()) * t0, v1 = w.p() + (w1 - w.p()) * t1, v2 = w.p() + (w2 - w.p()) * t2;
	v3 vc0 = plane_project_of(w.p(), w1, w2, O), vc1 = plane_project_of(w.p(), w2, w0, O), vc2 = plane_project_of(w.p(), w0, w1, O);
	int inp0 = (dot(-vc0, w0 - vc0) > 0.0f), inp1 = (dot(-vc1, w1 - vc1) > 0.0f), inp2 = (dot(-vc2, w2 - vc2) > 0.0f);
	const mkpoint s0 = src.W[0], s1 = src.W[1], s2 = src.W[2];
	if (inp0 && inp1 && inp2) { dst.W[0] = s0; dst.W[1] = s1; dst.W[2] = s2; dst.count = 4; dst.v = O; dst.W[3] = w; return; }
	int inp2e0 = (dot(-v0, w1 - v0) > 0.0f), inp2e1 = (dot(-v1, w0 - v1) > 0.0f);
	if (!inp2 && inp2e0 && inp2e1) { dst.count = 3; dst.v = plane_project_of(w0, w1, w.p(), O); dst.W[0] = s0; dst.W[1] = s1; dst.W[2] = w; return; }
	int inp0e1 = (dot(-v1, w2 - v1) > 0.0f), inp0e2 = (dot(-v2, w1 - v2) > 0.0f);
	if (!inp0 && inp0e1 && inp0e2) { dst.count = 3; dst.v = plane_project_of(w1, w2, w.p(), O); dst.W[0] = s1; dst.W[1] = s2; dst.W[2] = w; return; }
	int inp1e2 = (dot(-v2, w0 - v2) > 0.0f), inp1e0 = (dot(-v0, w2 - v0) > 0.0f);
	if (!inp1 && inp1e2 && inp1e0) { dst.count = 3; dst.v = plane_project_of(w2, w0, w.p(), O); dst.W[0] = s2; dst.W[1] = s0; dst.W[2] = w; return; }
	if (!inp1e0 && !inp2e0 && t0 > 0.0f) { keep_edge(dst, s0, w, t0, v0); return; }
	if (!inp2e1 && !inp0e1 && t1 > 0.0f) { keep_edge(dst, s1, w, t1, v1); return; }
	if (!inp0e2 && !inp1e2 && t2 > 0.0f) { keep_edge(dst, s2, w, t2, v2); return; }
	dst.W[0] = w; dst.W[0].t = 1.0f; dst.v = w.p(); dst.count = 1;
}
__device__ gjk_hit calcpoints(simplex &src)      // gjk.h:337-363
{
	if (src.count == 3)
	{
		v3 b = barycentric(src.W[0].p(), src.W[1].p(), src.W[2].p(), src.v);
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

// Separated(A, B, findclosest = 1), gjk.h:367-437, per lane.  Returns 0: separated (hit valid), 1: far apart (no contact possible),
// 2: the simplex `tet` encloses the origin and the expanding polytope has to run (gjk.h:400-428).
__device__ int gjk_run(const support_t &A, const support_t &B, float cutoff, gjk_hit &hit, simplex &tet)
{
	simplex last, next;
	last.count = 0; next.count = 0;
	for (int i = 0; i < 4; i++) { last.W[i].a = last.W[i].b = V3(0, 0, 0); last.W[i].t = 0; next.W[i] = last.W[i]; }
	int iter = 0;
	v3 v = point_on_minkowski(A, B, V3(0, 0, 1)).p();
	last.v = v;
	mkpoint w = point_on_minkowski(A, B, -v);
	next.W[0] = w; next.W[0].t = 1.0f; next.v = w.p(); next.count = 1;                        // NextMinkSimplex0
	for (;;)
	{
		bool go;
		if (iter == 0) { iter++; go = true; }
		else { iter++; go = (dot(w.p(), v) < dot(v, v) - 0.00001f); if (go) { go = (iter < 100); iter++; } }      // while(!iter++ || (... && iter++<100))
		if (!go) break;
		last = next;
		v = last.v;
		w = point_on_minkowski(A, B, -v);
		if (cutoff > 0.0f) { const float wv = dot(w.p(), v); if (wv > 0.0f && wv > (cutoff * 1.01f + 1e-6f) * length(v)) return 1; }
		if (dot(w.p(), v) >= dot(v, v) - 0.00001f - 0.00001f * dot(v, v)) break;
		if (last.count == 1) next1(next, last, w); else if (last.count == 2) next2(next, last, w); else next3(next, last, w);
		if (is_zero(next.v))
		{
			if (next.count == 2) { v3 n = orth(next.W[0].p() - next.W[1].p()); next.W[2] = point_on_minkowski(A, B, n); next.count = 3; }
			if (next.count == 3) { v3 n = tri_normal(next.W[0].p(), next.W[1].p(), next.W[2].p()); next.W[3] = point_on_minkowski(A, B, n); next.count = 4; }
			tet = next;
			return 2;
		}
		if (dot(next.v, next.v) >= dot(last.v, last.v)) break;
	}
	hit = calcpoints(last);
	return 0;
}

// ---- expanding polytope, wave-cooperative on a per-wave LDS mesh (hull.h:233-310) ---------------------------------------
#ifndef HT_EPA_INLINE
#define HT_EPA_INLINE __forceinline__      // inlined into its two callers: no stack frame for the shapes, no register save around a call (round 4; see epa_handover)
#endif
#define EPA_MAXT 192
#define EPA_MAXV 96
// A triangle is one 16-byte record (vertex ids, pad, neighbour ids, pad) so that the mesh surgery, which is a chain of dependent look-ups,
// fetches a whole triangle with one 128-bit LDS read and decides in registers; neighbour slots are patched with 16-bit stores.
struct __attribute__((aligned(16))) epa_tri { short v[3], pad0, n[3], pad1; };
struct epa_mem { epa_tri t[EPA_MAXT]; float4 v[EPA_MAXV]; unsigned char who[EPA_MAXT], map[EPA_MAXT]; };      // a vertex is one 128-bit read (the per-lane phases gather three per triangle); who / map: the compaction's slot tables
struct tri_r { int v0, v1, v2, n0, n1, n2; };
// the mesh surgery is executed by every lane on the same values (same stores from all lanes), so each lane's own program order keeps it coherent
__device__ __forceinline__ v3 ev(const epa_mem &m, int i) { const float4 q = m.v[i]; return V3(q.x, q.y, q.z); }
__device__ __forceinline__ tri_r tri_ld(const epa_mem &m, int t)
{
	const int4 q = *reinterpret_cast<const int4 *>(&m.t[t]);
	tri_r r = { (int)(short)(q.x & 0xffff), q.x >> 16, (int)(short)(q.y & 0xffff), (int)(short)(q.z & 0xffff), q.z >> 16, (int)(short)(q.w & 0xffff) };
	return r;
}
__device__ __forceinline__ void tri_set(epa_mem &m, int t, int a, int b, int c, int n0, int n1, int n2)
{
	*reinterpret_cast<int4 *>(&m.t[t]) = make_int4((a & 0xffff) | (b << 16), c & 0xffff, (n0 & 0xffff) | (n1 << 16), n2 & 0xffff);
}
__device__ __forceinline__ void tri_kill(epa_mem &m, int t) { *reinterpret_cast<int2 *>(&m.t[t].n[0]) = make_int2(-1, 0xffff); }      // n = -1, -1, -1
__device__ __forceinline__ bool tri_dead(const epa_mem &m, int t) { return m.t[t].n[0] == -1; }
__device__ __forceinline__ bool hasvert(const tri_r &T, int x) { return T.v0 == x || T.v1 == x || T.v2 == x; }
// a field picked by a run-time index comes out of a packed word by a shift: a select chain over the struct's fields is turned into an indexed array on
// the STACK by the compiler (scratch memory: a round trip to the memory system inside the surgery's dependent chain)
__device__ __forceinline__ int tri_n(const tri_r &T, int slot)
{
	const unsigned p = ((unsigned)T.n0 & 255u) | (((unsigned)T.n1 & 255u) << 8) | (((unsigned)T.n2 & 255u) << 16);
	const int r = (int)((p >> (8 * slot)) & 255u);
	return r == 255 ? -1 : r;
}
__device__ __forceinline__ int tri_v(const tri_r &T, int k) { return (int)((((unsigned)T.v0 & 255u) | (((unsigned)T.v1 & 255u) << 8) | (((unsigned)T.v2 & 255u) << 16)) >> (8 * k)) & 255; }
// which neighbour slot of T lies across the edge (va, vb), either direction: hull.h:97-109 (edge i -> slot (i+2)%3, first match wins)
__device__ __forceinline__ int nslot(const tri_r &T, int va, int vb)
{
	if ((T.v0 == va && T.v1 == vb) || (T.v0 == vb && T.v1 == va)) return 2;
	if ((T.v1 == va && T.v2 == vb) || (T.v1 == vb && T.v2 == va)) return 0;
	if ((T.v2 == va && T.v0 == vb) || (T.v2 == vb && T.v0 == va)) return 1;
	return 0;      // unreachable for a consistent mesh (the reference asserts)
}
__device__ __forceinline__ void set_n(epa_mem &m, int t, int slot, int val) { m.t[t].n[slot] = (short)val; }
__device__ void b2bfix(epa_mem &m, int s, int t)      // hull.h:136-150 (its three rounds on three lanes at once, with this loop as the fall-back for the
{                                                      // irregular cases, were measured: no faster)
	for (int i = 0; i < 3; i++)
	{
		tri_r S = tri_ld(m, s), T = tri_ld(m, t);      // fresh neighbour ids: the previous round may have patched s or t
		const int va = tri_v(S, i == 2 ? 0 : i + 1), vb = tri_v(S, i == 0 ? 2 : i - 1);      // tv[s][(i+1)%3], tv[s][(i+2)%3]
		const int ss = nslot(S, va, vb), ts = nslot(T, vb, va);
		const int X = tri_n(S, ss), Y = tri_n(T, ts);
		const tri_r RX = tri_ld(m, X), RY = tri_ld(m, Y);
		set_n(m, X, nslot(RX, vb, va), Y);                       // *neib(*neib(s, va, vb), vb, va) = *neib(t, vb, va)
		int sval = X, yy = Y;
		if (X == s || X == t) { S = tri_ld(m, s); T = tri_ld(m, t); sval = tri_n(S, ss); yy = tri_n(T, ts); }      // the store above hit s or t itself
		set_n(m, yy, nslot(yy == Y ? RY : tri_ld(m, yy), va, vb), sval);      // *neib(*neib(t, vb, va), va, vb) = *neib(s, va, vb)
	}
	tri_kill(m, s); tri_kill(m, t);
}
__device__ bool extrude(epa_mem &m, int &nt, int t0, int v)      // hull.h:167-186
{
	if (nt + 3 > EPA_MAXT) return false;
	const tri_r T0 = tri_ld(m, t0);
	const tri_r R0 = tri_ld(m, T0.n0), R1 = tri_ld(m, T0.n1), R2 = tri_ld(m, T0.n2);
	const int b = nt;
	tri_set(m, nt++, v, T0.v1, T0.v2, T0.n0, b + 1, b + 2); set_n(m, T0.n0, nslot(R0, T0.v1, T0.v2), b + 0);
	tri_set(m, nt++, v, T0.v2, T0.v0, T0.n1, b + 2, b + 0); set_n(m, T0.n1, nslot(R1, T0.v2, T0.v0), b + 1);
	tri_set(m, nt++, v, T0.v0, T0.v1, T0.n2, b + 0, b + 1); set_n(m, T0.n2, nslot(R2, T0.v0, T0.v1), b + 2);
	tri_kill(m, t0);
	if (hasvert(R0, v)) b2bfix(m, b + 0, T0.n0);
	if (hasvert(R1, v)) b2bfix(m, b + 1, T0.n1);
	if (hasvert(R2, v)) b2bfix(m, b + 2, T0.n2);
	return true;
}
__device__ __forceinline__ bool above(const epa_mem &m, int t, v3 p, float epsilon)      // hull.h:50-54
{
	const tri_r T = tri_ld(m, t);
	v3 n = tri_normal(ev(m, T.v0), ev(m, T.v1), ev(m, T.v2));
	return dot(n, p - ev(m, T.v0)) > epsilon;
}
// w = support(A, n) - support(B, -n) for the polytope, on the whole wave: both shapes are walked together, lane l takes vertices l, l+64, ..., four
// reads of each shape in flight (an index past a shape's end repeats its last vertex, which ties with itself).  The rotations are the matrices
// the caller built once per run: qrot(q, v) = qmat(q) * v and qrot(qconj(q), v) = transpose(qmat(q)) * v, rounding for rounding (linalg.h:284-288).
__device__ __forceinline__ v3 epa_minkowski(const support_t &A, const support_t &B, const m3 &RA, const m3 &RB, const m3 &RO, v3 n, int lane)
{
	const v3 da = mul(transpose(RA), A.outer ? mul(transpose(RO), n) : n), db = mul(transpose(RB), -n);
	const float4 *va = g_sm + A.voff, *vb = g_sm + B.voff;
	const int nmax = A.n > B.n ? A.n : B.n;
	float ba = 0.0f, bb = 0.0f; int ia = 0x7fffffff, ib = 0x7fffffff;
	for (int base = 0; base < nmax; base += 256)
	{
		float4 qa[4], qb[4]; int ka[4], kb[4];
#pragma unroll
		for (int k = 0; k < 4; k++) { const int i = base + 64 * k + lane; ka[k] = i < A.n ? i : A.n - 1; kb[k] = i < B.n ? i : B.n - 1; qa[k] = va[ka[k]]; qb[k] = vb[kb[k]]; }
#pragma unroll
		for (int k = 0; k < 4; k++)
		{
			const float d = dot(V3(qa[k].x, qa[k].y, qa[k].z), da), e = dot(V3(qb[k].x, qb[k].y, qb[k].z), db);
			if (ia == 0x7fffffff || ba < d) { ba = d; ia = ka[k]; }
			if (ib == 0x7fffffff || bb < e) { bb = e; ib = kb[k]; }
		}
	}
	wave_argmax(ba, ia); wave_argmax(bb, ib);
	const float4 pa = va[ia], pb = vb[ib];
	const v3 sa = A.pos + mul(RA, V3(pa.x, pa.y, pa.z));
	return (A.outer ? A.opos + mul(RO, sa) : sa) - (B.pos + mul(RB, V3(pb.x, pb.y, pb.z)));
}
// The mesh lives in the wave's own LDS area and is touched in two ways: UNIFORMLY (every lane executes the same stores with the same values: the surgery) and
// ONE ELEMENT PER LANE (scoring, the "who sees the new vertex" test, the verification walk, the compaction's moves).  Hardware executes a wave's LDS instructions in
// order, so a hand-over between the two needs no waiting -- but the COMPILER only reasons per thread: between a store one lane makes and a load another lane makes
// of it there is, for the compiler, no dependence to respect.  Every such hand-over therefore carries a wavefront-scope fence + wave barrier (nothing but a
// scheduling fence and an s_waitcnt in the code), which is what keeps the routine correct when the optimiser sees it together with its caller (inlined).
__device__ __forceinline__ void epa_handover() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
// A value every lane of the wave agrees on, SAID so: whatever decides a loop's exit in this routine (and in the job loop around it) is wave-uniform at run time,
// but as the result of a vector compare or an LDS read the compiler has to treat it as per-lane, i.e. the loop as one its lanes may leave at different trips --
// and it restructures such loops around the convergent operations inside them (ballots, v_readfirstlane, DPP exchanges).  As a function of its own the routine's
// loops were structurised by themselves; inline, nested in a job loop whose own exit test (job number against a count read from LDS) looked divergent too, the
// restructured code re-read the job number without the atomic that hands it out, and a wave never left the loop (round 3's hang; found in the ISA: a loop level
// between the job loop and the polytope's whose header was just the v_readfirstlane).  With the exits scalar there is nothing to restructure.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool uni(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
// all 64 lanes call this with identical arguments
// `capped` is set when the run ends on one of this implementation's capacities (the reference's loop is unbounded, hull.h:246)
__device__ __forceinline__ v4 expanding_polytope_wave_body(epa_mem &m, v3 s0, v3 s1, v3 s2, v3 s3, const support_t &A, const support_t &B, int lane, long long *ec, bool &capped)
{
	capped = false;
	long long tm = ec ? clock64() : 0;
	const m3 RA = qmat(A.q), RB = qmat(B.q), RO = qmat(A.oq);
	v4 plane = V4(0, 0, 0, -FLT_MAX);
	const float epsilon = 0.001f;
	int nv = 4, nt = 0;
	const bool flip = dot(cross(s2 - s0, s1 - s0), s3 - s0) > 0.0f;
	m.v[0] = make_float4(s0.x, s0.y, s0.z, 0.0f); m.v[1] = make_float4(s1.x, s1.y, s1.z, 0.0f);
	m.v[2] = flip ? make_float4(s3.x, s3.y, s3.z, 0.0f) : make_float4(s2.x, s2.y, s2.z, 0.0f);
	m.v[3] = flip ? make_float4(s2.x, s2.y, s2.z, 0.0f) : make_float4(s3.x, s3.y, s3.z, 0.0f);
	v3 center = (((s0 + s1) + s2) + s3) / 4.0f;
	tri_set(m, nt++, 2, 3, 1, 2, 3, 1); tri_set(m, nt++, 3, 2, 0, 3, 2, 0); tri_set(m, nt++, 0, 1, 3, 0, 1, 3); tri_set(m, nt++, 1, 0, 2, 1, 0, 2);
	int guard = 0;
	for (; guard < 128; guard++)
	{
		epa_handover();      // uniform stores (the start simplex, the previous iteration's compaction) -> one triangle per lane
		// face with the largest plane offset; the sequential scan keeps the first maximum (strict >), hull.h:248-261
		float bd = 0.0f; int bi = 0x7fffffff; v3 bn = V3(0, 0, 0);
		for (int i = lane; i < nt; i += 64)
		{
			const tri_r T = tri_ld(m, i);
			v3 n = tri_normal(ev(m, T.v0), ev(m, T.v1), ev(m, T.v2));
			float d = -dot(n, ev(m, T.v0));
			if (d > -FLT_MAX && (bi == 0x7fffffff || d > bd)) { bd = d; bi = i; bn = n; }
		}
		wave_argmax(bd, bi);
		if (bi != 0x7fffffff)      // the winner's normal sits in the lane that scored it (triangle i was scored by lane i % 64)
		{
			const int src = __builtin_amdgcn_readfirstlane(bi) & 63;
			bn = V3(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(bn.x), src)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bn.y), src)),
			        __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bn.z), src)));
		}
		v4 face = (bi == 0x7fffffff) ? V4(0, 0, 0, -FLT_MAX) : V4(bn, bd);
		if (ec) { const long long t = clock64(); ec[3] += t - tm; tm = t; ec[6] += 1; }
		v3 v = epa_minkowski(A, B, RA, RB, RO, xyz(face), lane);
		if (ec) { const long long t = clock64(); ec[4] += t - tm; tm = t; }
		v4 p = V4(xyz(face), -dot(xyz(face), v));
		if (p.w > plane.w) plane = p;
		bool dup = false;
		for (int i = lane; i < nv; i += 64) dup = dup || same(v, ev(m, i));
		if (__any(dup)) break;
		if (uni(plane.w >= face.w - epsilon)) break;
		if (nv >= EPA_MAXV) { capped = true; break; }
		const int vid = nv;
		m.v[nv] = make_float4(v.x, v.y, v.z, 0.0f); nv++;
		epa_handover();      // the new vertex (uniform store) -> read per lane below
		// Which triangles see the new vertex is tested one triangle per lane (vertices of existing triangles never change, and a triangle
		// that died during the surgery is skipped when its turn comes); the reference's descending scan then only visits the set bits.
		bool okk = true;
		const int nt0 = nt;
		for (int base = ((nt0 - 1) >> 6) << 6; base >= 0; base -= 64)
		{
			const int i = base + lane;
			unsigned long long mask = __ballot(i < nt0 && !tri_dead(m, i) && above(m, i, v, 0.01f * epsilon));
			while (mask)
			{
				const int bit = 63 - __clzll((long long)mask);
				mask &= ~(1ull << bit);
				if (uni(!tri_dead(m, base + bit))) okk = okk && extrude(m, nt, base + bit, vid);
			}
			okk = uni(okk); nt = uni(nt);
			epa_handover();      // the surgery's uniform stores -> the next block of triangles, one per lane
		}
		// The reference then walks down from the newest triangle while triangles carry the new vertex (dead ones skipped) and extrudes from the
		// neighbour of the first one that faces the centre or is degenerate, starting over after each extrusion (hull.h:283-297).  Every lane
		// judges one triangle; the first event from the top -- a live triangle without the vertex ends the walk, a bad one extrudes -- is read
		// off the ballots.
		while (okk)
		{
			int from = -1; bool done = false;
			for (int base = ((nt - 1) >> 6) << 6; base >= 0 && !done; base -= 64)
			{
				const int i = base + lane;
				bool live = false, hv = false, bad = false;
				if (i < nt)
				{
					const tri_r J = tri_ld(m, i);
					live = J.n0 != -1; hv = hasvert(J, vid);
					if (live && hv)
					{
						const v3 a = ev(m, J.v0), b = ev(m, J.v1), c = ev(m, J.v2);
						bad = dot(tri_normal(a, b, c), center - a) > 0.01f * epsilon || length(cross(b - a, c - b)) < epsilon * epsilon * 0.1f;
					}
				}
				const unsigned long long stop = __ballot(live && !hv), badm = __ballot(bad), evm = stop | badm;
				if (evm)
				{
					const int bit = 63 - __clzll((long long)evm);
					done = true;
					if ((badm >> bit) & 1ull) from = tri_ld(m, base + bit).n0;
				}
			}
			from = uni(from);
			if (from < 0) break;
			okk = uni(extrude(m, nt, from, vid)); nt = uni(nt);
			epa_handover();      // uniform stores -> the walk starts over, one triangle per lane
		}
		if (!okk) { capped = true; break; }
		// Compaction (hull.h:300-306): the reference fills every dead slot, highest first, with the then-last triangle (swapn: swap the records, patch the
		// neighbours' back references).  What that amounts to is a permutation plus a renaming, so it is done as such: (1) the dead slots by ballot;
		// (2) WHICH triangle ends in which slot by replaying the reference's loop on one byte per slot (who[a] = who[last]; no record is touched: a round
		// trip per dead triangle instead of five); (3) the moves, one lane per filled slot, noting old -> new in a table; (4) every surviving triangle
		// renames its neighbours through the table.  A consistent mesh (which the reference asserts) refers to a moved triangle exactly where the
		// reference's back-reference patch would have written the new id.
		{
			const int nt0 = nt;
			unsigned long long dead[3]; int ndead = 0;
#pragma unroll
			for (int k = 0; k < 3; k++) { const int i = 64 * k + lane; dead[k] = __ballot(i < nt0 && tri_dead(m, i)); ndead += __popcll(dead[k]); }
			if (ndead)
			{
				const int ntf = nt0 - ndead;
#pragma unroll
				for (int k = 0; k < 3; k++) { const int i = 64 * k + lane; if (i >= ntf && i < nt0) m.who[i] = (unsigned char)i; }
				epa_handover();      // one byte per lane -> the replay below reads other lanes' bytes
				int last = nt0;
#pragma unroll
				for (int k = 2; k >= 0; k--)
				{
					unsigned long long mk = dead[k];
					while (mk)
					{
						const int bit = 63 - __clzll((long long)mk);
						mk &= ~(1ull << bit);
						const int a = 64 * k + bit;
						last--;
						if (a != last) m.who[a] = m.who[last];
					}
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
				__builtin_amdgcn_wave_barrier();
#pragma unroll
				for (int k = 0; k < 3; k++)
				{
					const int i = 64 * k + lane;
					if (((dead[k] >> lane) & 1ull) && i < ntf)
					{
						const int src = m.who[i];
						*reinterpret_cast<int4 *>(&m.t[i]) = *reinterpret_cast<const int4 *>(&m.t[src]);
						m.map[src] = (unsigned char)i;
					}
				}
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
				__builtin_amdgcn_wave_barrier();
#pragma unroll
				for (int k = 0; k < 3; k++)
				{
					const int i = 64 * k + lane;
					if (64 * k < ntf && i < ntf)
					{
						const tri_r T = tri_ld(m, i);
						const bool m0 = T.n0 >= ntf && T.n0 < nt0, m1 = T.n1 >= ntf && T.n1 < nt0, m2 = T.n2 >= ntf && T.n2 < nt0;
						if (m0) m.t[i].n[0] = (short)m.map[T.n0];
						if (m1) m.t[i].n[1] = (short)m.map[T.n1];
						if (m2) m.t[i].n[2] = (short)m.map[T.n2];
					}
				}
				nt = ntf;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		__builtin_amdgcn_wave_barrier();
		if (ec) { const long long t = clock64(); ec[5] += t - tm; tm = t; }
	}
	if (guard == 128) capped = true;
	return plane;
}
// the cooperative kernel takes the routine inline (HT_EPA_INLINE); the lane-per-pair kernel, whose own register budget is spent on its GJK groups, calls it
__device__ HT_EPA_INLINE v4 expanding_polytope_wave(epa_mem &m, v3 s0, v3 s1, v3 s2, v3 s3, const support_t &A, const support_t &B, int lane, long long *ec, bool &capped)
{
	return expanding_polytope_wave_body(m, s0, s1, s2, s3, A, B, lane, ec, capped);
}
__device__ __noinline__ v4 expanding_polytope_wave_call(epa_mem &m, v3 s0, v3 s1, v3 s2, v3 s3, const support_t &A, const support_t &B, int lane, long long *ec, bool &capped)
{
	return expanding_polytope_wave_body(m, s0, s1, s2, s3, A, B, lane, ec, capped);
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
__device__ __forceinline__ support_t bcast(const support_t &s, int src)
{
	support_t r;
	r.voff = __shfl(s.voff, src); r.n = __shfl(s.n, src); r.outer = __shfl(s.outer, src); r.sub = 0; r.grp = 1;
	r.pos = V3(__shfl(s.pos.x, src), __shfl(s.pos.y, src), __shfl(s.pos.z, src));
	r.q = V4(__shfl(s.q.x, src), __shfl(s.q.y, src), __shfl(s.q.z, src), __shfl(s.q.w, src));
	r.opos = V3(__shfl(s.opos.x, src), __shfl(s.opos.y, src), __shfl(s.opos.z, src));
	r.oq = V4(__shfl(s.oq.x, src), __shfl(s.oq.y, src), __shfl(s.oq.z, src), __shfl(s.oq.w, src));
	return r;
}
// One GJK run per lane (lanes with run == false idle), then the expanding polytope for the lanes that need it, one pair at a time on the
// whole wave.  On return status is 0 (hit valid) or 1 (far apart).
__device__ void separated_wave(bool run, const support_t &A, const support_t &B, float cutoff, epa_mem &em, int lane, int &status, gjk_hit &hit, int dbg, long long *cyc = nullptr, int *caps = nullptr)
{
	const long long t0 = cyc ? clock64() : 0;
	simplex tet;
	tet.count = 0;
	for (int i = 0; i < 4; i++) { tet.W[i].a = tet.W[i].b = V3(0, 0, 0); tet.W[i].t = 0; }
	tet.v = V3(0, 0, 0);
	status = 1;
	if (run) status = gjk_run(A, B, cutoff, hit, tet);
	unsigned long long need = __ballot(run && status == 2 && A.sub == 0);
	const long long t1 = cyc ? clock64() : 0;
	if (cyc) { cyc[0] += t1 - t0; cyc[2] += __popcll(need); }
	while (need)
	{
		const int src = __ffsll((long long)need) - 1;
		need &= need - 1;
		const support_t Ab = bcast(A, src), Bb = bcast(B, src);
		v3 s[4];
#pragma unroll
		for (int k = 0; k < 4; k++) s[k] = V3(__shfl(tet.W[k].p().x, src), __shfl(tet.W[k].p().y, src), __shfl(tet.W[k].p().z, src));
		bool capped = false;
		v4 mpp = HT_DBG(dbg, 32) ? V4(0, 0, 1, -0.001f) : expanding_polytope_wave_call(em, s[0], s[1], s[2], s[3], Ab, Bb, lane, cyc, capped);
		if (capped && caps && lane == 0) atomicAdd(caps, 1);
		if (lane == src)
		{
			hit.normal = -xyz(mpp);                                  // gjk.h:417-423
			hit.separation = fmin_std(0.0f, mpp.w);
			v4 bw = inverse_w(tet.W[0].p(), tet.W[1].p(), tet.W[2].p(), tet.W[3].p());
			hit.p0w = ((tet.W[0].a * bw.x + tet.W[1].a * bw.y) + tet.W[2].a * bw.z) + tet.W[3].a * bw.w;
			hit.p1w = ((tet.W[0].b * bw.x + tet.W[1].b * bw.y) + tet.W[2].b * bw.z) + tet.W[3].b * bw.w;
			status = 0;
		}
	}
	// the polytope ran for member 0 of a group only: hand its result to the other members so that a group stays in lock step
	if (A.grp > 1 && __any(run && status == 2))
	{
		auto m0 = [&](float v) -> float { return A.grp == 4 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x00, 0xF, 0xF, true)) : __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xA0, 0xF, 0xF, true)); };
		const int st0 = A.grp == 4 ? __builtin_amdgcn_mov_dpp(status, 0x00, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(status, 0xA0, 0xF, 0xF, true);
		gjk_hit h0;
		h0.normal = V3(m0(hit.normal.x), m0(hit.normal.y), m0(hit.normal.z)); h0.p0w = V3(m0(hit.p0w.x), m0(hit.p0w.y), m0(hit.p0w.z));
		h0.p1w = V3(m0(hit.p1w.x), m0(hit.p1w.y), m0(hit.p1w.z)); h0.separation = m0(hit.separation);
		if (run && status == 2) { hit = h0; status = st0; }
	}
	if (cyc) cyc[1] += clock64() - t1;
}

// ------------------------------------------------------------------------------------------------- k_contacts
#define GJK_FRAMES 2        // frames per block sharing the LDS vertex copy
// Waves per frame (template parameter GJK_WPF): the per-lane support scans are bound by the CU's LDS throughput, so for the 17-bone hand (about
// 23 candidate pairs per frame: two or four lanes per pair on one wave) a second wave is slower (contacts 2.10 -> 2.39 ms per step); with 26
// bones (325 pairs, about 58 candidates per frame: one lane per pair) the second wave wins (9.2 -> 6.8 ms).  The launcher picks by pair count.
#define GJK_MAXWPF 2
struct gjk_frame_mem { float P[HT_MAXNB][8]; unsigned char cand[HT_MAXNB * (HT_MAXNB - 1) / 2][2]; int ncand, nchunk, cnt[GJK_MAXWPF]; };      // poses (pos3 q4 radius), candidate pairs
__host__ __device__ inline size_t gjk_frame_stride() { return (sizeof(gjk_frame_mem) + 15) & ~(size_t)15; }
__host__ __device__ inline size_t gjk_wave_stride() { return (sizeof(epa_mem) + 15) & ~(size_t)15; }
template <int GJK_WPF> __global__ __launch_bounds__(64 * GJK_FRAMES * GJK_WPF) void k_contacts(ht_model_dev M, const float *__restrict__ state, float driftmax, float jiggle_sin, const int *__restrict__ active_flag,
                                                                        float *__restrict__ contacts, int *__restrict__ ncontacts, int B, int dbg, int *__restrict__ caps)
{
	constexpr int GJK_LANES = 64 * GJK_WPF;
	const int nvert = M.vert_off[M.nb];
	float4 *sverts = g_sm;
	unsigned char *fbase = reinterpret_cast<unsigned char *>(g_sm + nvert);
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = wave / GJK_WPF, half = wave % GJK_WPF;
	gjk_frame_mem &F = *reinterpret_cast<gjk_frame_mem *>(fbase + fr * gjk_frame_stride());
	epa_mem &em = *reinterpret_cast<epa_mem *>(fbase + GJK_FRAMES * gjk_frame_stride() + wave * gjk_wave_stride());
	float (*P)[8] = F.P;
	const int b = blockIdx.x * GJK_FRAMES + fr;
	for (int i = t; i < nvert; i += 64 * GJK_FRAMES * GJK_WPF) sverts[i] = M.verts[i];
	const bool live = b < B && !(active_flag && !active_flag[b]);      // frames outside the active set keep whatever another launch produced for them
	if (live && half == 0 && lane < M.nb)
	{
		const float *s = state + ((size_t)b * M.nb + lane) * HT_STATE_STRIDE;
		for (int i = 0; i < 7; i++) P[lane][i] = s[i];
		P[lane][7] = M.bodyc[lane * HT_BC + HT_BC_RADIUS];
	}
	__syncthreads();
	// broad phase in the reference's pair order (physics.h:453-457), compacted with a ballot by the frame's first wave; pair index -> (i, j), i < j, row-major
	if (half == 0)
	{
		const int npairs = M.nb * (M.nb - 1) / 2;
		int nc = 0;
		for (int base = 0; live && base < npairs; base += 64)
		{
			const int pidx = base + lane;
			int i = 0, rem = pidx;
			while (i < M.nb - 1 && rem >= M.nb - 1 - i) { rem -= M.nb - 1 - i; i++; }
			const int j = i + 1 + rem;
			bool keep = false;
			if (pidx < npairs)
			{
				keep = (M.collide[i] & M.collide[j] & 2) != 0;
				v3 d = V3(P[j][0], P[j][1], P[j][2]) - V3(P[i][0], P[i][1], P[i][2]);
				if (length(d) > P[i][7] + P[j][7]) keep = false;
				if (M.ignore[i] & (1u << j)) keep = false;
			}
			if (HT_DBG(dbg, 8)) keep = false;
			const unsigned long long m = __ballot(keep);
			if (keep) { const int dst = nc + __popcll(m & ((1ull << lane) - 1ull)); F.cand[dst][0] = (unsigned char)i; F.cand[dst][1] = (unsigned char)j; }
			nc += __popcll(m);
		}
		// lanes per pair: spare lanes share the support scans
		const int gshf = nc <= GJK_LANES / 4 ? 2 : nc <= GJK_LANES / 2 ? 1 : 0;
		if (lane == 0) { F.ncand = nc; F.nchunk = (nc + (GJK_LANES >> gshf) - 1) / (GJK_LANES >> gshf); }
	}
	__syncthreads();
	const int ncand = F.ncand;
	int nchunk = 0;       // the block's frames step through the same number of chunks so that they can share barriers
	for (int f = 0; f < GJK_FRAMES; f++) { const int n = reinterpret_cast<gjk_frame_mem *>(fbase + f * gjk_frame_stride())->nchunk; nchunk = n > nchunk ? n : nchunk; }
	const int gsh = ncand <= GJK_LANES / 4 ? 2 : ncand <= GJK_LANES / 2 ? 1 : 0, grp = 1 << gsh, sub = lane & (grp - 1);
	long long cyc[7] = { 0, 0, 0, 0, 0, 0, 0 }, cycj[7] = { 0, 0, 0, 0, 0, 0, 0 }; const bool stats = HT_DBG(dbg, 2048) != 0; const long long t_begin = stats ? clock64() : 0; int njig = 0;
	int nout = 0;                       // contacts written so far for this frame (frame-uniform)
	for (int ch = 0; ch < nchunk; ch++)
	{
		const int cidx = ch * (GJK_LANES >> gsh) + ((half * 64 + lane) >> gsh);
		const bool keep = live && cidx < ncand;
		const int i = keep ? F.cand[cidx][0] : 0, j = keep ? F.cand[cidx][1] : 1;
		// narrow phase, one lane group per surviving pair (ContactPatch gjk.h:607-643)
		support_t A, Bs;
		A.voff = M.vert_off[i]; A.n = M.vert_off[i + 1] - M.vert_off[i]; A.pos = V3(P[i][0], P[i][1], P[i][2]); A.q = V4(P[i][3], P[i][4], P[i][5], P[i][6]); A.outer = 0; A.opos = V3(0, 0, 0); A.oq = V4(0, 0, 0, 1); A.sub = sub; A.grp = grp;
		Bs.voff = M.vert_off[j]; Bs.n = M.vert_off[j + 1] - M.vert_off[j]; Bs.pos = V3(P[j][0], P[j][1], P[j][2]); Bs.q = V4(P[j][3], P[j][4], P[j][5], P[j][6]); Bs.outer = 0; Bs.opos = V3(0, 0, 0); Bs.oq = V4(0, 0, 0, 1); Bs.sub = sub; Bs.grp = grp;
		gjk_hit hits[5];
		int hc = 0, status;
		separated_wave(keep, A, Bs, HT_DBG(dbg, 16) ? 0.0f : driftmax, em, lane, status, hits[0], dbg, stats ? cyc : nullptr, caps);
		const bool touching = keep && status == 0 && !(hits[0].separation > driftmax);      // identical in all members of a group
		if (touching) hc = 1;
		const float dmin = fminf(M.bodyc[i * HT_BC + HT_BC_DIAM], M.bodyc[j * HT_BC + HT_BC_DIAM]);
		const bool jig = touching && !(dmin < 0.049f);      // otherwise every jiggle sample is rejected by the 0.05 m proximity test (see header)
		if (stats) njig += __popcll(__ballot(jig));
		if (__any(jig))
		{
			const v3 n = jig ? hits[0].normal : V3(0, 0, 1);
			v4 qs = quat_from_to(n, V3(0, 0, 1));
			v3 tangent = qxdir(qs), bitangent = qydir(qs);
#pragma unroll 1
			for (int r = 0; r < 4; r++)
			{
				const v3 raxis = r == 0 ? tangent : r == 1 ? bitangent : r == 2 ? -tangent : -bitangent;
				v4 jiggle = normalize(V4(raxis * jiggle_sin, 1));
				v3 pivot = jig ? hits[0].p0w : V3(0, 0, 0);
				v4 id = V4(0, 0, 0, 1);
				xf ar = mul(mul(mul(XF(n * 0.2f, id), XF(-pivot, id)), XF(V3(0, 0, 0), jiggle)), XF(pivot, id));
				support_t AJ = A; AJ.outer = 1; AJ.opos = ar.p; AJ.oq = ar.q;
				gjk_hit hj; int st;
				separated_wave(jig, AJ, Bs, 0.0f, em, lane, st, hj, dbg, stats ? cycj : nullptr, caps);
				if (jig)
				{
					hj.normal = n;
					hj.p0w = apply(inverse(ar), hj.p0w);
					hj.separation = dot(n, hj.p0w - hj.p1w);
					bool match = false;
					for (int q = 0; q < 4; q++) if (q < hc && !match) match = length(hj.p0w - hits[q].p0w) < 0.05f || length(hj.p1w - hits[q].p1w) < 0.05f;
					if (!match)
					{
						if (hc == 1) hits[1] = hj; else if (hc == 2) hits[2] = hj; else if (hc == 3) hits[3] = hj; else hits[4] = hj;
						hc++;
					}
				}
			}
		}
		if (sub != 0) hc = 0;               // member 0 of a group reports the pair
		// compaction in pair order: exclusive prefix of the per-lane contact counts across the frame's waves
		int incl = hc;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
		if (lane == 63) F.cnt[half] = incl;
		__syncthreads();
		int before = 0, total = 0;
		for (int h = 0; h < GJK_WPF; h++) { const int cw = F.cnt[h]; if (h < half) before += cw; total += cw; }
		const int dst = nout + before + incl - hc;
		for (int k = 0; k < 5; k++) if (k < hc && dst + k < HT_MAXCONTACT)
		{
			const gjk_hit &h = k == 0 ? hits[0] : k == 1 ? hits[1] : k == 2 ? hits[2] : k == 3 ? hits[3] : hits[4];
			float4 *o = reinterpret_cast<float4 *>(contacts + ((size_t)b * HT_MAXCONTACT + dst + k) * HT_CONTACT);
			o[0] = make_float4((float)i, (float)j, h.normal.x, h.normal.y);
			o[1] = make_float4(h.normal.z, h.p0w.x, h.p0w.y, h.p0w.z);
			o[2] = make_float4(h.p1w.x, h.p1w.y, h.p1w.z, h.separation);
		}
		nout += total;
		__syncthreads();
	}
	if (live && half == 0 && lane == 0) { ncontacts[b] = nout < HT_MAXCONTACT ? nout : HT_MAXCONTACT; if (nout > HT_MAXCONTACT && caps) atomicAdd(caps + 1, nout - HT_MAXCONTACT); }
	if (stats && live && half == 0 && lane == 0 && nout < HT_MAXCONTACT - 1)      // timing experiments: statistics of the frame's first wave accumulate in the last contact slot
	{
		float *o = contacts + ((size_t)b * HT_MAXCONTACT + HT_MAXCONTACT - 1) * HT_CONTACT;
		o[0] += 1.0f; o[1] += (float)cyc[0]; o[2] += (float)cyc[1]; o[3] += (float)cyc[2]; o[4] += (float)cyc[3]; o[5] += (float)cyc[4]; o[6] += (float)cyc[5]; o[11] += (float)cyc[6];
		o[7] += (float)(clock64() - t_begin); o[8] += (float)ncand; o[9] += (float)njig; o[10] += (float)nout;
	}
}


// ================================================================================================= k_contacts_coop
// Second organisation of the narrow phase.  Two kinds of work with opposite shapes are separated:
//   * the simplex logic of a GJK run (Separated, gjk.h:367-437) is branchy scalar code: it runs one run per lane, the runs of all the block's
//     frames packed into as few "owner" waves as they need (so that every path through NextMinkSimplex1..3 is paid once per 64 runs);
//   * the support scans are streaming arg-max reductions: an owner lane posts its two scans of the step (shape, direction in the shape's frame)
//     to a list in LDS and ALL waves of the block work the list off, one scan per DPP row of 16 lanes: 16 consecutive vertices per read (the
//     four rows of a wave read four contiguous 256-byte segments -- no bank conflicts, where one-lane-per-pair scans hit random banks), first
//     maximum by four DPP exchanges.  The scans of a step are balanced over the whole block whatever run or frame they belong to, so a frame
//     with many candidates or long runs costs the block its share of scans, not a lone wave its whole latency.
// The body rotations are matrices built once per frame (qrot = qmat(q) * v, linalg.h:284-288; qmat(qconj(q)) is the transpose, entry by entry the
// same roundings), the vertices lie padded to whole rows (pads repeat vertex 0, which cannot win against itself).  Runs whose simplex encloses the
// origin queue for the expanding polytope, which every wave of the block takes jobs from.  Touching samples go to a per-frame pool in LDS keyed
// by (pair, sample) and leave in key order = the reference's contact order.  The four extra samples of a contact patch (gjk.h:626-641) are a
// second pass over the patches the first pass lists.
#define GJK_POOL 192        // touching samples a frame can hold before they are ordered (all of them are kept: HT_MAXCONTACT)
static_assert(HT_MAXCONTACT >= GJK_POOL, "every touching sample the pool holds is kept as a contact");
#define GJK_JMAX 40         // pairs per frame whose contact patch takes the four extra samples
#define CO_NW 8             // waves per block
#define CO_MAXF 4           // frames per block (as many as the LDS holds)
#define CO_WORK_PATCH 2     // weight of a five-sample patch (four more runs on one pair) in a frame's work estimate, in candidate pairs
#define CO_OWN 4            // owner waves per round: 256 runs in flight
#define CO_EPAQ 32          // polytope jobs per queue round
struct gjk_sample { float n[3], p0[3], p1[3], sep; int key, flag; };      // key = candidate * 8 + sample number; flag 1 = counts as a contact
struct co_body { float pos[3], radius, q[4], R[9], pad[3]; };
struct __attribute__((aligned(16))) co_req { int shape; float dx, dy, dz; };      // shape = padded vertex offset | rows of 16 vertices << 16
struct __attribute__((aligned(16))) co_job { float p[4][3], opos[3], oq[4]; int f, bi, bj, outer; float res[4]; int capped, pad0, pad1, pad2; };
struct __attribute__((aligned(16))) co_frame
{
	co_body body[HT_MAXNB];
	unsigned char cand[HT_MAXNB * (HT_MAXNB - 1) / 2][2];
	gjk_sample pool[GJK_POOL];
	unsigned short jig[GJK_JMAX];      // pool slot of the first sample of each patch that takes extra samples
	unsigned short jslot[GJK_JMAX];    // its four extra samples: slots jslot .. jslot+3
	int ncand, off, npool, njig, joff, nepa;
};
// (the model's per-body tables sit in the block's LDS: indexing the kernel-argument copy of ht_model_dev with a run-time body number makes the compiler either
// fetch from the argument segment with dependent global loads or -- with the polytope routine inline -- copy the whole struct to scratch memory, per lane)
struct __attribute__((aligned(16))) co_block { int nreq[2], total, jtotal, nepa, enext, pad[2]; int cvoff[HT_MAXNB + 1], vn[HT_MAXNB], collide[HT_MAXNB]; unsigned ignore[HT_MAXNB]; float diam[HT_MAXNB]; };
struct co_lds { co_block *H; co_frame *F; co_req *req; int *resp; co_job *jobs; };
enum { RS_IDLE = 0, RS_INIT0, RS_INIT1, RS_ITER, RS_TET2, RS_TET3, RS_EPA, RS_WAIT, RS_HIT, RS_FAR };
__device__ __forceinline__ void row_argmax(float &b, int &i) { amax_dpp<0xB1>(b, i); amax_dpp<0x4E>(b, i); amax_dpp<0x141>(b, i); amax_dpp<0x140>(b, i); }      // every lane of the row ends with the row's winner
__device__ __forceinline__ m3 body_R(const co_body &b) { m3 m; m.x = V3(b.R[0], b.R[1], b.R[2]); m.y = V3(b.R[3], b.R[4], b.R[5]); m.z = V3(b.R[6], b.R[7], b.R[8]); return m; }

// every wave: scans of the current list.  A row takes the two scans of a run (entries 2k and 2k+1: shape A along n, shape B along -n) and walks both
// shapes together, two vertex rows of each per trip, so that four independent reads are in flight; a row index past a shape's end repeats its last
// row, which cannot displace itself.  Wave w takes runs 4w.., 4(w+8).. (the scans are alike enough for a fixed split).
__device__ __forceinline__ void co_scan(co_block &H, int buf, const co_req *req, int *resp, int lane, int wave)
{
	const int sub = lane & 15, row = lane >> 4;
	const int nrun = H.nreq[buf];
	for (int k = 4 * wave + row; k < nrun; k += 4 * CO_NW)
	{
		const co_req ra = req[2 * k], rb = req[2 * k + 1];
		const float4 *va = g_sm + (ra.shape & 0xffff) + sub, *vb = g_sm + (rb.shape & 0xffff) + sub;
		const int na = ra.shape >> 16, nb = rb.shape >> 16, nmax = na > nb ? na : nb;
		const f32x2 axy = { ra.dx, ra.dy }, bxy = { rb.dx, rb.dy };
		float4 qa = va[0], qb = vb[0];
		f32x2 pa = f32x2{ qa.x, qa.y } * axy, pb = f32x2{ qb.x, qb.y } * bxy;
		float ba = (pa.x + pa.y) + qa.z * ra.dz, bb = (pb.x + pb.y) + qb.z * rb.dz;      // = dot(vertex, direction) in the reference's order
		int ia = __float_as_int(qa.w), ib = __float_as_int(qb.w);
		for (int r = 1; r < nmax; r += 2)
		{
			const int a0 = r < na ? r : na - 1, a1 = r + 1 < na ? r + 1 : na - 1, b0 = r < nb ? r : nb - 1, b1 = r + 1 < nb ? r + 1 : nb - 1;
			const float4 x0 = va[16 * a0], x1 = va[16 * a1], y0 = vb[16 * b0], y1 = vb[16 * b1];
			const f32x2 p0 = f32x2{ x0.x, x0.y } * axy, p1 = f32x2{ x1.x, x1.y } * axy, s0 = f32x2{ y0.x, y0.y } * bxy, s1 = f32x2{ y1.x, y1.y } * bxy;
			const float d0 = (p0.x + p0.y) + x0.z * ra.dz, d1 = (p1.x + p1.y) + x1.z * ra.dz, e0 = (s0.x + s0.y) + y0.z * rb.dz, e1 = (s1.x + s1.y) + y1.z * rb.dz;
			if (ba < d0) { ba = d0; ia = __float_as_int(x0.w); }
			if (bb < e0) { bb = e0; ib = __float_as_int(y0.w); }
			if (ba < d1) { ba = d1; ia = __float_as_int(x1.w); }
			if (bb < e1) { bb = e1; ib = __float_as_int(y1.w); }
		}
		row_argmax(ba, ia); row_argmax(bb, ib);
		if (sub == 0) { resp[2 * k] = ia; resp[2 * k + 1] = ib; }
	}
}

// One pass over a work list: JIG = false: the candidate pairs (Separated with the contact cut-off); JIG = true: the extra samples of the patches.
template <bool JIG> __device__ __forceinline__ void co_pass(const co_lds &L, int nfr, epa_mem &em, float driftmax, float jiggle_sin, int t, int dbg, int *caps, int &parity, long long *cyc)
{
	co_block &H = *L.H;
	const int lane = t & 63, wave = t >> 6;
	const int total = JIG ? H.jtotal : H.total;
	const float cutoff = JIG ? 0.0f : (HT_DBG(dbg, 16) ? 0.0f : driftmax);
	for (int x0 = 0; x0 < total; x0 += 64 * CO_OWN)
	{
		// ---- this lane's run ----
		const int x = x0 + t;
		int st = RS_IDLE, fsel = 0, cidx = 0, bi = 0, bj = 1, slot = 0, iter = 0, myreq = 0, myjob = 0, shA = 0, shB = 0;
		v3 opos = V3(0, 0, 0), jn = V3(0, 0, 1); v4 oq = V4(0, 0, 0, 1); xf ar = XF(V3(0, 0, 0), V4(0, 0, 0, 1));
		simplex last, next;
		last.count = 0; next.count = 0; last.v = next.v = V3(0, 0, 0);
		for (int i = 0; i < 4; i++) { last.W[i].a = last.W[i].b = V3(0, 0, 0); last.W[i].t = 0; next.W[i] = last.W[i]; }
		mkpoint w = last.W[0]; v3 v = V3(0, 0, 0);
		gjk_hit hit; hit.normal = V3(0, 0, 1); hit.p0w = hit.p1w = V3(0, 0, 0); hit.separation = 0;
		if (wave < CO_OWN && x < total)
		{
			int f = 0;
			for (int g = 1; g < nfr; g++) if (x >= (JIG ? L.F[g].joff : L.F[g].off)) f = g;
			co_frame &F = L.F[f];
			fsel = f;
			if (JIG)
			{
				const int y = x - F.joff, r = y & 3;
				slot = F.jslot[y >> 2] + r;
				const gjk_sample &S0 = F.pool[F.jig[y >> 2]];
				cidx = S0.key >> 3; jn = V3(S0.n[0], S0.n[1], S0.n[2]);
				const v3 pivot = V3(S0.p0[0], S0.p0[1], S0.p0[2]);
				const v4 qs = quat_from_to(jn, V3(0, 0, 1)); const v3 tangent = qxdir(qs), bitangent = qydir(qs);
				const v3 raxis = r == 0 ? tangent : r == 1 ? bitangent : r == 2 ? -tangent : -bitangent;
				const v4 jiggle = normalize(V4(raxis * jiggle_sin, 1)); const v4 id = V4(0, 0, 0, 1);
				ar = mul(mul(mul(XF(jn * 0.2f, id), XF(-pivot, id)), XF(V3(0, 0, 0), jiggle)), XF(pivot, id)); opos = ar.p; oq = ar.q;
			}
			else cidx = x - F.off;
			bi = F.cand[cidx][0]; bj = F.cand[cidx][1];
			shA = H.cvoff[bi] | (((H.cvoff[bi + 1] - H.cvoff[bi]) >> 4) << 16); shB = H.cvoff[bj] | (((H.cvoff[bj + 1] - H.cvoff[bj]) >> 4) << 16);
			st = RS_INIT0;
		}
		const co_frame &FR = L.F[fsel];
		// ---- Separated (gjk.h:367-437) unrolled into states around the one place where the two supports of a step are taken ----
		for (;;)
		{
			const long long tc0 = cyc ? clock64() : 0;
			const bool sup = st >= RS_INIT0 && st <= RS_TET3;
			{
				const unsigned long long m = __ballot(sup);
				if (m)
				{
					// every lane takes part in the atomic (lane 0 adds the wave's count, the others 0) and lane 0's return value is the wave's base: no lane-0-only
					// block whose bypass the compiler could thread the other lanes through (see the polytope's job loop below)
					int base = atomicAdd(&H.nreq[parity], lane == 0 ? __popcll(m) : 0);
					base = __builtin_amdgcn_readfirstlane(base);
					myreq = base + __popcll(m & ((1ull << lane) - 1ull));
				}
			}
			if (sup)
			{
				v3 dir = V3(0, 0, 1);
				if (st == RS_INIT1 || st == RS_ITER) dir = -v;
				else if (st == RS_TET2) dir = orth(next.W[0].p() - next.W[1].p());
				else if (st == RS_TET3) dir = tri_normal(next.W[0].p(), next.W[1].p(), next.W[2].p());
				const v3 da = mul(transpose(body_R(FR.body[bi])), JIG ? qrot(qconj(oq), dir) : dir);      // qrot(qconj(q), .)
				const v3 db = mul(transpose(body_R(FR.body[bj])), -dir);
				co_req ra; ra.shape = shA; ra.dx = da.x; ra.dy = da.y; ra.dz = da.z;
				co_req rb; rb.shape = shB; rb.dx = db.x; rb.dy = db.y; rb.dz = db.z;
				L.req[2 * myreq] = ra; L.req[2 * myreq + 1] = rb;
			}
			if (cyc) cyc[0] += clock64() - tc0;
			__syncthreads();
			if (H.nreq[parity] == 0) break;
			if (t == 0) H.nreq[parity ^ 1] = 0;
			const long long ts0 = cyc ? clock64() : 0;
			co_scan(H, parity, L.req, L.resp, lane, wave);
			if (cyc) { cyc[1] += clock64() - ts0; cyc[3] += 1; }
			__syncthreads();
			const long long tc1 = cyc ? clock64() : 0;
			if (cyc) cyc[4] += tc1 - ts0;
			parity ^= 1;
			if (sup)
			{
				const co_body &BA = FR.body[bi], &BB = FR.body[bj];
				const float4 qa = g_sm[(shA & 0xffff) + L.resp[2 * myreq]], qb = g_sm[(shB & 0xffff) + L.resp[2 * myreq + 1]];
				mkpoint m;
				const v3 sa = V3(BA.pos[0], BA.pos[1], BA.pos[2]) + mul(body_R(BA), V3(qa.x, qa.y, qa.z));
				m.a = JIG ? opos + qrot(oq, sa) : sa;
				m.b = V3(BB.pos[0], BB.pos[1], BB.pos[2]) + mul(body_R(BB), V3(qb.x, qb.y, qb.z));
				m.t = 0;
				if (st == RS_INIT0) { v = m.p(); last.v = v; st = RS_INIT1; }
				else if (st == RS_TET2) { next.W[2] = m; next.count = 3; st = RS_TET3; }
				else if (st == RS_TET3) { next.W[3] = m; next.count = 4; st = RS_EPA; }
				else
				{
					bool fin = false;
					if (st == RS_INIT1)
					{
						w = m; next.W[0] = w; next.W[0].t = 1.0f; next.v = w.p(); next.count = 1;      // NextMinkSimplex0
						iter = 1;                                                                    // first trip of the while: !iter++
						last = next; v = last.v; st = RS_ITER;
					}
					else
					{
						w = m;
						bool far = false;
						if (cutoff > 0.0f) { const float wv = dot(w.p(), v); far = wv > 0.0f && wv > (cutoff * 1.01f + 1e-6f) * length(v); }
						if (far) st = RS_FAR;
						else if (dot(w.p(), v) >= dot(v, v) - 0.00001f - 0.00001f * dot(v, v)) fin = true;
						else
						{
							if (last.count == 1) next1(next, last, w); else if (last.count == 2) next2(next, last, w); else next3(next, last, w);
							if (is_zero(next.v)) st = next.count == 2 ? RS_TET2 : next.count == 3 ? RS_TET3 : RS_EPA;
							else if (dot(next.v, next.v) >= dot(last.v, last.v)) fin = true;
							else
							{
								iter++;                                                              // while(!iter++ || (cond && iter++ < 100))
								bool go = dot(w.p(), v) < dot(v, v) - 0.00001f;
								if (go) { go = iter < 100; iter++; }
								if (!go) fin = true; else { last = next; v = last.v; }
							}
						}
					}
					if (fin) { hit = calcpoints(last); st = RS_HIT; }
				}
			}
			if (cyc) cyc[0] += clock64() - tc1;
		}
		// ---- expanding polytope for the runs whose simplex encloses the origin: a queue every wave takes jobs from ----
		const long long te0 = cyc ? clock64() : 0;
		for (;;)
		{
			if (st == RS_EPA)
			{
				const int j = atomicAdd(&H.nepa, 1);
				if (j < CO_EPAQ)
				{
					co_job &J = L.jobs[j];
					for (int k = 0; k < 4; k++) { J.p[k][0] = next.W[k].p().x; J.p[k][1] = next.W[k].p().y; J.p[k][2] = next.W[k].p().z; }
					J.opos[0] = opos.x; J.opos[1] = opos.y; J.opos[2] = opos.z; J.oq[0] = oq.x; J.oq[1] = oq.y; J.oq[2] = oq.z; J.oq[3] = oq.w;
					J.f = fsel; J.bi = bi; J.bj = bj; J.outer = JIG ? 1 : 0;
					myjob = j; st = RS_WAIT;
				}
			}
			__syncthreads();
			const int nj = uni(H.nepa < CO_EPAQ ? H.nepa : CO_EPAQ);      // every lane reads the same count; a scalar for the job loop's exit test
			if (nj == 0) break;
			// Jobs go round the waves (wave w takes jobs w, w + 8, ...: a block has a handful).  They used to be handed out by an atomic counter -- lane 0 fetched a
			// number, v_readfirstlane spread it -- and THAT was round 3's hang once the polytope was inline: the compiler threaded the lanes that are not lane 0
			// (whose copy of the number is the initial 0) from the store of the result, which only lane 0 makes, straight back to the v_readfirstlane, past the
			// atomic; the 63 lanes then read job 0 from their own first lane, for ever (seen in the ISA: a loop level between the job loop and the polytope's
			// with the v_readfirstlane as its header and lane 0 masked out).  No number travels between lanes now.
			for (int j = wave; j < nj; j += CO_NW)
			{
				co_job &J = L.jobs[j];
				const co_frame &F = L.F[J.f];
				support_t Ab, Bb;
				const co_body &BA = F.body[J.bi], &BB = F.body[J.bj];
				Ab.voff = H.cvoff[J.bi]; Ab.n = H.vn[J.bi]; Ab.pos = V3(BA.pos[0], BA.pos[1], BA.pos[2]); Ab.q = V4(BA.q[0], BA.q[1], BA.q[2], BA.q[3]);
				Ab.outer = J.outer; Ab.opos = V3(J.opos[0], J.opos[1], J.opos[2]); Ab.oq = V4(J.oq[0], J.oq[1], J.oq[2], J.oq[3]); Ab.sub = 0; Ab.grp = 1;
				Bb.voff = H.cvoff[J.bj]; Bb.n = H.vn[J.bj]; Bb.pos = V3(BB.pos[0], BB.pos[1], BB.pos[2]); Bb.q = V4(BB.q[0], BB.q[1], BB.q[2], BB.q[3]);
				Bb.outer = 0; Bb.opos = V3(0, 0, 0); Bb.oq = V4(0, 0, 0, 1); Bb.sub = 0; Bb.grp = 1;
				bool capped = false;
				const v4 mpp = HT_DBG(dbg, 32) ? V4(0, 0, 1, -0.001f) : expanding_polytope_wave(em, V3(J.p[0][0], J.p[0][1], J.p[0][2]), V3(J.p[1][0], J.p[1][1], J.p[1][2]), V3(J.p[2][0], J.p[2][1], J.p[2][2]),
				                                                                           V3(J.p[3][0], J.p[3][1], J.p[3][2]), Ab, Bb, lane, cyc ? cyc + 4 : nullptr, capped);
				if (cyc) cyc[11] += 1;
				if (lane == 0) { J.res[0] = mpp.x; J.res[1] = mpp.y; J.res[2] = mpp.z; J.res[3] = mpp.w; if (capped && caps) atomicAdd(caps, 1); atomicAdd(&L.F[J.f].nepa, 1); }
			}
			__syncthreads();
			if (st == RS_WAIT)
			{
				const co_job &J = L.jobs[myjob];
				hit.normal = -V3(J.res[0], J.res[1], J.res[2]);                                  // gjk.h:417-423
				hit.separation = fmin_std(0.0f, J.res[3]);
				const v4 bw = inverse_w(next.W[0].p(), next.W[1].p(), next.W[2].p(), next.W[3].p());
				hit.p0w = ((next.W[0].a * bw.x + next.W[1].a * bw.y) + next.W[2].a * bw.z) + next.W[3].a * bw.w;
				hit.p1w = ((next.W[0].b * bw.x + next.W[1].b * bw.y) + next.W[2].b * bw.z) + next.W[3].b * bw.w;
				st = RS_HIT;
			}
			__syncthreads();
			if (t == 0) { H.nepa = 0; H.enext = 0; }
			__syncthreads();
		}
		if (cyc) cyc[2] += clock64() - te0;
		// ---- finished runs: the sample goes to the frame's pool ----
		if (st == RS_HIT || st == RS_FAR)
		{
			co_frame &F = L.F[fsel];
			if (JIG)
			{
				// gjk.h:632-636: the sample is reported on the undisturbed shape, along the patch normal
				const v3 p0 = apply(inverse(ar), hit.p0w);
				gjk_sample &S = F.pool[slot];
				S.n[0] = jn.x; S.n[1] = jn.y; S.n[2] = jn.z; S.p0[0] = p0.x; S.p0[1] = p0.y; S.p0[2] = p0.z; S.p1[0] = hit.p1w.x; S.p1[1] = hit.p1w.y; S.p1[2] = hit.p1w.z; S.sep = dot(jn, p0 - hit.p1w);
			}
			else if (st == RS_HIT && !(hit.separation > driftmax))
			{
				const float dmin = fminf(H.diam[bi], H.diam[bj]);
				const bool jig = !(dmin < 0.049f);      // otherwise every extra sample is rejected by the 0.05 m proximity test (see the header)
				const int need = jig ? 5 : 1;
				const int s0 = atomicAdd(&F.npool, need);
				if (s0 + need <= GJK_POOL)
				{
					gjk_sample &S = F.pool[s0];
					S.n[0] = hit.normal.x; S.n[1] = hit.normal.y; S.n[2] = hit.normal.z; S.p0[0] = hit.p0w.x; S.p0[1] = hit.p0w.y; S.p0[2] = hit.p0w.z;
					S.p1[0] = hit.p1w.x; S.p1[1] = hit.p1w.y; S.p1[2] = hit.p1w.z; S.sep = hit.separation; S.key = cidx * 8; S.flag = 1;
					if (jig)
					{
						for (int r = 0; r < 4; r++) { F.pool[s0 + 1 + r].key = cidx * 8 + 1 + r; F.pool[s0 + 1 + r].flag = 0; }
						const int jx = atomicAdd(&F.njig, 1);
						if (jx < GJK_JMAX) { F.jig[jx] = (unsigned short)s0; F.jslot[jx] = (unsigned short)(s0 + 1); }
						else if (caps) atomicAdd(caps + 1, 1);
					}
				}
				else if (caps) atomicAdd(caps + 1, 1);
			}
		}
		__syncthreads();
	}
}

// Which frame is slot f of a block: frame f * blocks + (block + 61 f) mod blocks, a bijection of (block, slot) onto the frames.  Neighbouring frames of a
// batch are often alike (consecutive frames of one stream, a tiled test set) and a block's time goes with the runs of its heaviest frames, so a block takes
// its frames from four far-apart places instead of four neighbours: the slowest block of a launch, which is the launch's time, comes closer to the mean.
__device__ __forceinline__ int co_frame_of(int block, int f, int blocks) { return f * blocks + (block + 61 * f) % blocks; }
// Round 5: when the frames' work of the same launch of the PREVIOUS update is known (k_contact_order below), slot f of a block is the table's entry instead: the
// frames dealt heaviest first, back and forth over the blocks, so that the slowest block of a launch -- which is the launch's time -- comes down to the mean.
// The assignment decides where a frame is computed, never what: contacts are the same bit for bit.
__device__ __forceinline__ int co_frame_at(const int *__restrict__ order, int block, int f, int blocks) { return order ? order[f * blocks + block] : co_frame_of(block, f, blocks); }
__global__ __launch_bounds__(64 * CO_NW) void k_contacts_coop(ht_model_dev M, const float *__restrict__ state, float driftmax, float jiggle_sin, const int *__restrict__ active_flag,
                                                              float *__restrict__ contacts, int *__restrict__ ncontacts, int B, int nfr, int nvp, int dbg, int *__restrict__ caps, const int *__restrict__ order, int *__restrict__ work_out)
{
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	// LDS: padded vertices | block header | frames | scan list | answers | polytope jobs | polytope meshes
	unsigned char *base = reinterpret_cast<unsigned char *>(g_sm + nvp);
	co_lds L;
	L.H = reinterpret_cast<co_block *>(base); base += sizeof(co_block);
	L.F = reinterpret_cast<co_frame *>(base); base += (size_t)nfr * sizeof(co_frame);
	L.req = reinterpret_cast<co_req *>(base); base += (size_t)CO_OWN * 64 * 2 * sizeof(co_req);
	L.resp = reinterpret_cast<int *>(base); base += (size_t)CO_OWN * 64 * 2 * sizeof(int);
	L.jobs = reinterpret_cast<co_job *>(base); base += (size_t)CO_EPAQ * sizeof(co_job);
	epa_mem &em = *reinterpret_cast<epa_mem *>(base + wave * gjk_wave_stride());
	co_block &H = *L.H;
	const long long t_begin = HT_DBG(dbg, 2048) ? clock64() : 0;
	if (active_flag)      // a masked launch: blocks without a live frame leave at once (a handful of frames of a large batch take this kernel on their own)
	{
		bool any = false;
		for (int f = 0; f < nfr; f++) { const int b = co_frame_at(order, blockIdx.x, f, gridDim.x); any = any || (b < B && active_flag[b] != 0); }
		if (!any) return;
	}
	for (int k = t; k < nvp; k += 64 * CO_NW) g_sm[k] = M.cverts[k];      // the padded vertex image (ht_model_dev::cverts), 16 vertices per row
	{
		// the per-body tables, picked out of the kernel arguments with compile-time indices (thread k takes body k)
#pragma unroll
		for (int k = 0; k < HT_MAXNB; k++) if (t == k) { H.cvoff[k] = M.cvert_off[k]; H.vn[k] = M.vert_off[k + 1] - M.vert_off[k]; H.collide[k] = M.collide[k]; H.ignore[k] = M.ignore[k]; }
		if (t == HT_MAXNB) H.cvoff[HT_MAXNB] = M.cvert_off[HT_MAXNB];
		if (t < M.nb) H.diam[t] = M.bodyc[t * HT_BC + HT_BC_DIAM];
	}
	__syncthreads();
	// broad phase in the reference's pair order (physics.h:453-457), one wave per frame, compacted with a ballot; pair index -> (i, j), i < j, row-major
	if (wave < nfr)
	{
		co_frame &F = L.F[wave];
		const int b = co_frame_at(order, blockIdx.x, wave, gridDim.x);
		const bool live = b < B && !(active_flag && !active_flag[b]);      // frames outside the active set keep whatever another launch produced for them
		if (live && lane < M.nb)
		{
			const float *s = state + ((size_t)b * M.nb + lane) * HT_STATE_STRIDE;
			co_body &Y = F.body[lane];
			for (int i = 0; i < 3; i++) Y.pos[i] = s[i];
			for (int i = 0; i < 4; i++) Y.q[i] = s[3 + i];
			Y.radius = M.bodyc[lane * HT_BC + HT_BC_RADIUS];
			const m3 R = qmat(V4(s[3], s[4], s[5], s[6]));
			Y.R[0] = R.x.x; Y.R[1] = R.x.y; Y.R[2] = R.x.z; Y.R[3] = R.y.x; Y.R[4] = R.y.y; Y.R[5] = R.y.z; Y.R[6] = R.z.x; Y.R[7] = R.z.y; Y.R[8] = R.z.z;
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		__builtin_amdgcn_wave_barrier();
		const int npairs = M.nb * (M.nb - 1) / 2;
		int nc = 0;
		for (int base2 = 0; live && base2 < npairs; base2 += 64)
		{
			const int pidx = base2 + lane;
			int i = 0, rem = pidx;
			while (i < M.nb - 1 && rem >= M.nb - 1 - i) { rem -= M.nb - 1 - i; i++; }
			const int j = i + 1 + rem;
			bool keep = false;
			if (pidx < npairs)
			{
				keep = (H.collide[i] & H.collide[j] & 2) != 0;
				v3 d = V3(F.body[j].pos[0], F.body[j].pos[1], F.body[j].pos[2]) - V3(F.body[i].pos[0], F.body[i].pos[1], F.body[i].pos[2]);
				if (length(d) > F.body[i].radius + F.body[j].radius) keep = false;
				if (H.ignore[i] & (1u << j)) keep = false;
			}
			if (HT_DBG(dbg, 8)) keep = false;
			const unsigned long long m = __ballot(keep);
			if (keep) { const int dst = nc + __popcll(m & ((1ull << lane) - 1ull)); F.cand[dst][0] = (unsigned char)i; F.cand[dst][1] = (unsigned char)j; }
			nc += __popcll(m);
		}
		for (int e = lane; e < GJK_POOL; e += 64) { F.pool[e].flag = 0; F.pool[e].key = 0x7fffffff; }
		if (lane == 0) { F.ncand = nc; F.npool = 0; F.njig = 0; F.nepa = 0; }
	}
	__syncthreads();
	if (t == 0)
	{
		int tot = 0;
		for (int f = 0; f < nfr; f++) { L.F[f].off = tot; tot += L.F[f].ncand; }
		H.total = tot; H.jtotal = 0; H.nreq[0] = H.nreq[1] = 0; H.nepa = 0; H.enext = 0;
	}
	__syncthreads();
	long long cyc[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };      // [7..10]: this wave's polytope runs (face search, support, surgery, iterations), [11] their number
	const long long t_pro = HT_DBG(dbg, 2048) ? clock64() : 0;
	int parity = 0;
	co_pass<false>(L, nfr, em, driftmax, jiggle_sin, t, dbg, caps, parity, HT_DBG(dbg, 2048) ? cyc : nullptr);
	const long long t_post = HT_DBG(dbg, 2048) ? clock64() : 0;
	if (t == 0)
	{
		int tot = 0;
		for (int f = 0; f < nfr; f++) { co_frame &F = L.F[f]; if (F.njig > GJK_JMAX) F.njig = GJK_JMAX; if (F.npool > GJK_POOL) F.npool = GJK_POOL; F.joff = tot; tot += 4 * F.njig; }
		H.jtotal = tot;
	}
	__syncthreads();
	if (H.jtotal > 0)
	{
		co_pass<true>(L, nfr, em, driftmax, jiggle_sin, t, dbg, caps, parity, nullptr);
		// which of the extra samples count (gjk.h:637-640): one lane per patch, samples in order, each against the ones accepted before it
		for (int f = 0; f < nfr; f++)
		{
			co_frame &F = L.F[f];
			for (int jx = t; jx < F.njig; jx += 64 * CO_NW)
			{
				int acc[5]; int hc = 1; acc[0] = F.jig[jx];
				for (int r = 0; r < 4; r++)
				{
					gjk_sample &S = F.pool[F.jslot[jx] + r];
					const v3 p0 = V3(S.p0[0], S.p0[1], S.p0[2]), p1 = V3(S.p1[0], S.p1[1], S.p1[2]);
					bool match = false;
					for (int q = 0; q < 4; q++) if (q < hc && !match) { const gjk_sample &Q = F.pool[acc[q]]; match = length(p0 - V3(Q.p0[0], Q.p0[1], Q.p0[2])) < 0.05f || length(p1 - V3(Q.p1[0], Q.p1[1], Q.p1[2])) < 0.05f; }
					if (!match) { S.flag = 1; acc[hc] = F.jslot[jx] + r; hc++; }
				}
			}
		}
		__syncthreads();
	}
	// ---- the samples of a frame in key order = the reference's contact order; one wave per frame ----
	if (wave < nfr)
	{
		co_frame &F = L.F[wave];
		const int b = co_frame_at(order, blockIdx.x, wave, gridDim.x);
		const bool live = b < B && !(active_flag && !active_flag[b]);
		const int np = F.npool;
		// a sample's place = how many counting samples have a smaller key.  The keys are laid out once as a dense array (a sample that does not count: the
		// largest int) in the scan list's area, which nothing uses any more, so that a lane reads four per LDS access instead of key and flag of one sample
		static_assert(CO_MAXF * GJK_POOL * sizeof(int) <= (size_t)CO_OWN * 64 * 2 * sizeof(co_req), "the key arrays of a block's frames fit the scan list's area");
		int *const kk = reinterpret_cast<int *>(L.req) + wave * GJK_POOL;
		for (int e = lane; e < ((np + 3) & ~3); e += 64) kk[e] = (e < np && F.pool[e].flag != 0) ? F.pool[e].key : 0x7fffffff;
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		__builtin_amdgcn_wave_barrier();
		int total = 0;
		for (int e0 = 0; e0 < np; e0 += 64)
		{
			const int e = e0 + lane;
			const bool valid = e < np && F.pool[e].flag != 0;
			total += __popcll(__ballot(valid));
			if (valid)
			{
				const gjk_sample S = F.pool[e];
				int rank = 0;
				for (int o = 0; o < np; o += 4) { const int4 k4 = *reinterpret_cast<const int4 *>(kk + o); rank += (k4.x < S.key ? 1 : 0) + (k4.y < S.key ? 1 : 0) + (k4.z < S.key ? 1 : 0) + (k4.w < S.key ? 1 : 0); }
				if (live && rank < HT_MAXCONTACT)
				{
					const int c = S.key >> 3;
					float4 *o = reinterpret_cast<float4 *>(contacts + ((size_t)b * HT_MAXCONTACT + rank) * HT_CONTACT);
					o[0] = make_float4((float)F.cand[c][0], (float)F.cand[c][1], S.n[0], S.n[1]);
					o[1] = make_float4(S.n[2], S.p0[0], S.p0[1], S.p0[2]);
					o[2] = make_float4(S.p1[0], S.p1[1], S.p1[2], S.sep);
				}
			}
		}
		if (live && lane == 0) { ncontacts[b] = total < HT_MAXCONTACT ? total : HT_MAXCONTACT; if (total > HT_MAXCONTACT && caps) atomicAdd(caps + 1, total - HT_MAXCONTACT); }
		if (live && lane == 0 && work_out) work_out[b] = (F.ncand + CO_WORK_PATCH * F.njig) | (F.nepa << 16);      // what the frame took this launch, for the next update's assignment (k_contact_order)
		if (HT_DBG(dbg, 2048) && live && lane == 0 && total < HT_MAXCONTACT - 2)      // timing experiments: per-frame statistics accumulate in the last two contact slots
		{
			float *o = contacts + ((size_t)b * HT_MAXCONTACT + HT_MAXCONTACT - 1) * HT_CONTACT;
			o[0] += 1.0f; o[1] += (float)cyc[0]; o[2] += (float)cyc[2]; o[3] += (float)F.nepa; o[4] += (float)cyc[3]; o[5] += (float)cyc[1]; o[6] += (float)cyc[4]; o[7] += (float)(clock64() - t_begin);
			o[8] += (float)F.ncand; o[9] += (float)(clock64() - t_post); o[10] += (float)(t_post - t_pro); o[11] += (float)(t_pro - t_begin);
			o -= HT_CONTACT;
			for (int k = 0; k < 5; k++) o[k] += (float)cyc[7 + k];
		}
	}
}

size_t ht_contacts_workspace_bytes(int B) { (void)B; return 16; }      // the polytope mesh lives in LDS; the workspace holds the capacity counters (polytope runs cut short, contacts dropped, k_solve's angular overflow)

// frames per block of the cooperative kernel for a batch of B frames (as many as the LDS holds beside the padded vertex copy, the scan list and the waves' polytope areas:
// 4 for the 17-bone hand); 0 = the model does not fit it
int ht_contacts_frames_per_block(const ht_model_dev &M, int B)
{
	const size_t fixed = (size_t)M.cvert_off[M.nb] * sizeof(float4) + sizeof(co_block) + (size_t)CO_OWN * 64 * 2 * (sizeof(co_req) + sizeof(int)) + CO_EPAQ * sizeof(co_job) + CO_NW * gjk_wave_stride();
	if (fixed + sizeof(co_frame) > 160 * 1024) return 0;
	int nfr = CO_MAXF;
	while (nfr > 1 && fixed + nfr * sizeof(co_frame) > 160 * 1024) nfr--;
	return B < nfr ? B : nfr;
}
// The frames of a launch dealt to the blocks by what they took in the same launch of the PREVIOUS update (poses move little between updates).  A block's time is its
// scan rounds (latency: about the same for one frame's runs as for four frames') plus, when any of its frames has a run whose simplex encloses the origin, its polytope
// phases, which cost about one run's time whether one wave or all eight have a job.  So the frames WITH polytope runs are gathered -- sorted by their number of runs and
// dealt back and forth, epb to a block, over as few blocks as that takes, which go first in the grid; the lightest of the other frames fill those blocks up -- and the
// rest, sorted by their candidate pairs, are dealt back and forth over the remaining blocks: fewer blocks pay a polytope phase at all.  epb: all of a block's frames
// when the batch takes several rounds per CU (the sum of the blocks' times counts, and the long blocks start first), fewer when every block has a CU of its own (the
// slowest block is the launch's time, and a block of four such frames would be it).
// work = candidates + 2 x patches | polytope runs << 16 (k_contacts_coop); order[slot * blocks + block] = frame, B = no frame.  One block per launch slot of an
// update and segment of 4096 frames (work / order: [slots][stride]); ranks by counting, the keys in LDS.  Runs beside the CNN at the head of an update: off every critical path.
#define CO_ORDER_SEG 4096      // frames a block of k_contact_order ranks (by counting: quadratic); a larger batch is dealt segment by segment, each over its own blocks
__global__ __launch_bounds__(1024) void k_contact_order(const int *__restrict__ work, int *__restrict__ order, int Ball, int nfr, int stride, unsigned slots, int epb)
{
	extern __shared__ int ko_w[];
	__shared__ int ko_ne;
	if (!((slots >> blockIdx.x) & 1u)) return;
	const int seg = (CO_ORDER_SEG / nfr) * nfr, f0 = blockIdx.y * seg, B = Ball - f0 < seg ? Ball - f0 : seg;      // this block's frames: [f0, f0 + B)
	const int blocks_all = (Ball + nfr - 1) / nfr, b0 = f0 / nfr;
	const int *w = work + (size_t)blockIdx.x * stride + f0;
	int *o = order + (size_t)blockIdx.x * stride;
	const int blocks = (B + nfr - 1) / nfr, B4 = (B + 3) & ~3;
	if (threadIdx.x == 0) ko_ne = 0;
	__syncthreads();
	int mine = 0;
	for (int i = threadIdx.x; i < B4; i += 1024)
	{
		const int v = i < B ? w[i] : -1;
		const int runs = v < 0 ? 0 : (v >> 16), pairs = v < 0 ? 0 : (v & 0xffff);
		ko_w[i] = v < 0 ? -1 : ((runs > 1023 ? 1023 : runs) << 20) | pairs;
		mine += runs > 0 ? 1 : 0;
	}
	if (mine) atomicAdd(&ko_ne, mine);
	for (int i = threadIdx.x; i < blocks * nfr; i += 1024) o[(i / blocks) * blocks_all + b0 + i % blocks] = Ball;
	__syncthreads();
	const int ne = ko_ne, ng = B - ne;                      // frames with polytope runs (the first ne places of the sorted order) and without
	if (epb > nfr) epb = nfr;
	if (epb * blocks < ne) epb = (ne + blocks - 1) / blocks;
	const int neb = (ne + epb - 1) / epb, ngb = blocks - neb;      // blocks that get frames with polytope runs, epb each, and the others
	const int nfree = neb * nfr - ne;                       // places left in the first kind: the LIGHTEST frames without runs fill them
	const int nheavy = ng - nfree > 0 ? ng - nfree : 0;     // frames without runs that go to the second kind
	for (int i = threadIdx.x; i < B; i += 1024)
	{
		const int wi = ko_w[i];
		int rank = 0;
		for (int j = 0; j < B4; j += 4)
		{
			const int4 k = *reinterpret_cast<const int4 *>(ko_w + j);
			rank += ((k.x > wi || (k.x == wi && j < i)) ? 1 : 0) + ((k.y > wi || (k.y == wi && j + 1 < i)) ? 1 : 0) + ((k.z > wi || (k.z == wi && j + 2 < i)) ? 1 : 0) + ((k.w > wi || (k.w == wi && j + 3 < i)) ? 1 : 0);
		}
		int nb_, base, r;
		if (rank < ne) { nb_ = neb; base = 0; r = rank; }                                        // place r of the first kind's grid (neb wide, round by round)
		else if (rank - ne < nheavy) { nb_ = ngb; base = neb; r = rank - ne; }                   // place r of the second kind's grid
		else { nb_ = neb; base = 0; r = ne + (rank - ne - nheavy); }                             // the lightest: the places the first kind has left
		const int round = r / nb_, pos = r - round * nb_;
		o[round * blocks_all + b0 + base + ((round & 1) ? nb_ - 1 - pos : pos)] = f0 + i;
	}
}
void ht_launch_contact_order(const int *work, int *order, int B, int nfr, int stride, unsigned slots, int nslots, int epb, hipStream_t s)
{
	const int seg = (CO_ORDER_SEG / nfr) * nfr, nseg = (B + seg - 1) / seg;
	hipLaunchKernelGGL(k_contact_order, dim3(nslots, nseg), dim3(1024), (size_t)(((B < seg ? B : seg) + 3) & ~3) * sizeof(int), s, work, order, B, nfr, stride, slots, epb);
}
void ht_launch_contacts(const ht_model_dev &M, const float *state, float driftmax, float jiggle_sin, const int *active_flag, void *epa_ws, float *contacts, int *ncontacts, int B, hipStream_t s, bool beside_cloud_rows, int force_kernel, int few_frames,
                        const int *order, int *work_out)
{
	int *caps = reinterpret_cast<int *>(epa_ws);
	const int dbg = ht_tuning_flags();
	static bool attr_set[64];                 // per device: the attribute belongs to the device's copy of the code object
	static std::mutex attr_lock;              // contexts of several host threads (one per GPU) may launch at the same time
	int dev = 0; (void)hipGetDevice(&dev); dev &= 63;
	std::unique_lock<std::mutex> lk(attr_lock);
	if (!attr_set[dev])
	{
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_contacts<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_contacts<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_contacts_coop), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		attr_set[dev] = true;
	}
	lk.unlock();
	// Which organisation: the cooperative kernel (a CU per block: 156 KB of LDS, 242 VGPRs) whenever a frame of the model fits its LDS beside the padded vertex
	// copy.  Whole steps on one device, cooperative against lane-per-pair: 1536 frames 7.57 against 7.64 ms, 2048 frames 8.46 / 8.56, 4096 frames 14.74 / 14.96,
	// 8192 frames 27.00 / 27.57 (tools/exp_coop_max.sh; until the polytope and the frame-to-block assignment were reworked in round 3 the lane-per-pair kernel won
	// above ~1100 frames because its smaller blocks run beside the cloud-row kernel).  force_kernel (ht_debug_contact_kernel): 1 cooperative, 2 lane-per-pair.
	static const int coop_max = [] {      // read once, when the first launch gets here (a C++11 static: thread-safe)
#ifdef HT_TUNING
		if (const char *e = getenv("HT_CONTACTS_COOP_MAX")) return atoi(e);
#endif
		return 1 << 30;
	}();
	static const int main_lanes = ht_tuning_int("HT_CONTACTS_MAIN_LANES", 0);
	const size_t coop_fixed = (size_t)M.cvert_off[M.nb] * sizeof(float4) + sizeof(co_block) + (size_t)CO_OWN * 64 * 2 * (sizeof(co_req) + sizeof(int)) + CO_EPAQ * sizeof(co_job) + CO_NW * gjk_wave_stride();
	const bool coop_fits = coop_fixed + sizeof(co_frame) <= 160 * 1024;
	if (force_kernel == 1 ? coop_fits : (force_kernel != 2 && coop_fits && B <= coop_max && !(beside_cloud_rows && main_lanes)))
	{
		// as many frames per block as the LDS holds beside the padded vertex copy, the scan list and the waves' polytope areas (4 for the 17-bone hand)
		const int nvp = M.cvert_off[M.nb];
		const size_t fixed = (size_t)nvp * sizeof(float4) + sizeof(co_block) + (size_t)CO_OWN * 64 * 2 * (sizeof(co_req) + sizeof(int)) + CO_EPAQ * sizeof(co_job) + CO_NW * gjk_wave_stride();
		// few_frames: a masked launch that only a handful of frames take (the reset frames' own first step): a block per frame, all waves on it -- while the
		// batch is small enough that its empty blocks, each of which still asks for a whole CU's LDS, cost less than that gains (1024 frames: four rounds of them)
		int nfr = few_frames && B <= 2048 ? 1 : CO_MAXF;
		while (nfr > 1 && fixed + nfr * sizeof(co_frame) > 160 * 1024) nfr--;
		if (B < nfr) nfr = B;
		const size_t smem = fixed + nfr * sizeof(co_frame);
		hipLaunchKernelGGL(k_contacts_coop, dim3((B + nfr - 1) / nfr), dim3(64 * CO_NW), smem, s, M, state, driftmax, jiggle_sin, active_flag, contacts, ncontacts, B, nfr, nvp, dbg, caps, nfr == ht_contacts_frames_per_block(M, B) ? order : nullptr, work_out);
		return;
	}
	const int wpf = M.nb * (M.nb - 1) / 2 > 200 ? 2 : 1;
	const size_t smem = (size_t)M.vert_off[M.nb] * sizeof(float4) + GJK_FRAMES * gjk_frame_stride() + GJK_FRAMES * wpf * gjk_wave_stride();
	const dim3 grid((B + GJK_FRAMES - 1) / GJK_FRAMES);
	if (wpf == 2) hipLaunchKernelGGL(k_contacts<2>, grid, dim3(64 * GJK_FRAMES * 2), smem, s, M, state, driftmax, jiggle_sin, active_flag, contacts, ncontacts, B, dbg, caps);
	else hipLaunchKernelGGL(k_contacts<1>, grid, dim3(64 * GJK_FRAMES), smem, s, M, state, driftmax, jiggle_sin, active_flag, contacts, ncontacts, B, dbg, caps);
}
