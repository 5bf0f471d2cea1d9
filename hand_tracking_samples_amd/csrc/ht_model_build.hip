// ht_model_build.hip -- init-time model build on the host.  Host code only, no kernels.
//
// Turns a PhysModel JSON (control cages + joints) into the arrays the context loader consumes, evaluating the same
// sequence of fp32 operations as the reference's model construction so the result is bit-identical to what the
// reference holds after `LoadHandModel()`:
//   PhysModel(const char*)            include/physmodel.h:444-475     (cages -> 2x subdivision -> hull -> rigid bodies)
//   WingMeshCreate / LinkMesh / ...   third_party/wingmesh.h:170-278,946-966
//   WingMeshSubDiv                    third_party/wingmesh.h:730-786  (Catmull-Clark step on a half-edge mesh)
//   calchull                          third_party/hull.h:311-424      (greedy volume-maximising hull, vertex limit 48)
//   Volume/CenterOfMass/Inertia       third_party/geometric.h:372-428, third_party/physics.h:58-100
//   RigidBody ctor, rbscalemass       third_party/physics.h:147-183
//   Planes / PolyPlane                include/physmodel.h:44-53, third_party/geometric.h:247-260
//   build_ignore_lists                include/physmodel.h:260-277
//   LoadHandModel                     include/handtrack.h:347-366
//   UnibodyFit proxy cube             include/handtrack.h:454-455, third_party/wingmesh.h:879-881
// The file is compiled with -ffp-contract=off like the rest of the library.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <utility>
#include "ht_math.hpp"
#include "ht_model_build.hpp"
#include "../../include/ht_json.hpp"

// ------------------------------------------------------------------------------------------------- HTFX container
bool fx_load(const char *path, fx_map &out)
{
	FILE *fp = fopen(path, "rb");
	if (!fp) return false;
	char magic[8]; uint32_t count;
	if (fread(magic, 1, 8, fp) != 8 || memcmp(magic, "HTFX0001", 8) || fread(&count, 4, 1, fp) != 1) { fclose(fp); return false; }
	for (uint32_t i = 0; i < count; i++)
	{
		char nm[48]; fx_arr a; uint64_t n;
		if (fread(nm, 1, 48, fp) != 48 || fread(&a.dtype, 4, 1, fp) != 1 || fread(&a.ndim, 4, 1, fp) != 1 || fread(a.dims, 4, 4, fp) != 4 || fread(&n, 8, 1, fp) != 1) { fclose(fp); return false; }
		a.data.resize(n);
		if (n && fread(a.data.data(), 1, n, fp) != n) { fclose(fp); return false; }
		fseek(fp, (long)((8 - (n & 7)) & 7), SEEK_CUR);
		nm[47] = 0;
		out[nm] = std::move(a);
	}
	fclose(fp);
	return true;
}
bool fx_save(const char *path, const fx_map &in)
{
	FILE *fp = fopen(path, "wb");
	if (!fp) return false;
	uint32_t count = (uint32_t)in.size();
	fwrite("HTFX0001", 1, 8, fp); fwrite(&count, 4, 1, fp);
	for (auto &kv : in)
	{
		char nm[48]; memset(nm, 0, sizeof nm); strncpy(nm, kv.first.c_str(), 47);
		const fx_arr &a = kv.second; uint64_t n = a.data.size(), zero = 0;
		uint32_t d[4]; for (int i = 0; i < 4; i++) d[i] = (uint32_t)i < a.ndim ? a.dims[i] : 1;
		fwrite(nm, 1, 48, fp); fwrite(&a.dtype, 4, 1, fp); fwrite(&a.ndim, 4, 1, fp); fwrite(d, 4, 4, fp); fwrite(&n, 8, 1, fp);
		if (n) fwrite(a.data.data(), 1, n, fp);
		fwrite(&zero, 1, (8 - (n & 7)) & 7, fp);
	}
	return fclose(fp) == 0;
}
static fx_arr fx_f32(const std::vector<float> &v, std::initializer_list<uint32_t> dims)
{
	fx_arr a; a.dtype = 0; a.ndim = (uint32_t)dims.size(); int k = 0; for (uint32_t d : dims) a.dims[k++] = d;
	a.data.resize(v.size() * 4); if (v.size()) memcpy(a.data.data(), v.data(), v.size() * 4); return a;
}
static fx_arr fx_i32(const std::vector<int> &v, std::initializer_list<uint32_t> dims)
{
	fx_arr a; a.dtype = 1; a.ndim = (uint32_t)dims.size(); int k = 0; for (uint32_t d : dims) a.dims[k++] = d;
	a.data.resize(v.size() * 4); if (v.size()) memcpy(a.data.data(), v.data(), v.size() * 4); return a;
}

// ------------------------------------------------------------------------------------------------- JSON: include/ht_json.hpp (shared with the data-set reader of include/ht_formats.hpp)
namespace {
using ht_json::jnode; using ht_json::jparser;
bool read_v3(const jnode *n, v3 &o) { if (!n || n->kind != jnode::ARR || n->items.size() < 3) return false; o = V3(n->items[0].as_float(), n->items[1].as_float(), n->items[2].as_float()); return true; }
bool read_v4(const jnode *n, v4 &o) { if (!n || n->kind != jnode::ARR || n->items.size() < 4) return false; o = V4(n->items[0].as_float(), n->items[1].as_float(), n->items[2].as_float(), n->items[3].as_float()); return true; }

// ------------------------------------------------------------------------------------------------- half-edge mesh
// Same connectivity conventions as the reference's WingMesh (edge i starts at vertex v, runs to edges[next].v, `adj` is the
// opposite half-edge) because vertex positions after subdivision depend on the traversal order of the one-ring.
struct hedge { int id, v, adj, next, prev, face; };
struct hemesh
{
	std::vector<hedge> e;
	std::vector<v3> verts;
	int nfaces = 0;
	std::vector<int> vback, fback;       // first half-edge leaving a vertex / belonging to a face

	bool link()      // wingmesh.h:170-203: pair every half-edge (a->b) with the (b->a) one
	{
		std::vector<std::vector<int>> from(verts.size());
		for (auto &h : e) { if (h.v < 0 || h.v >= (int)verts.size()) return false; from[h.v].push_back(h.id); }
		for (auto &h : e)
		{
			if (h.adj != -1) continue;
			int a = h.v, b = e[h.next].v;
			for (int k : from[b]) if (e[e[k].next].v == a) { h.adj = k; e[k].adj = h.id; break; }
			if (h.adj == -1) return false;      // open mesh
		}
		return true;
	}
	void init_back_lists()      // wingmesh.h:205-216
	{
		vback.assign(verts.size(), -1); fback.assign(nfaces, -1);
		for (int i = (int)e.size(); i--;) { vback[e[i].v] = i; fback[e[i].face] = i; }
	}
	bool create(const std::vector<v3> &vs, const std::vector<std::vector<int>> &polys)      // wingmesh.h:946-966
	{
		verts = vs; e.clear(); nfaces = 0;
		for (auto &poly : polys)
		{
			int base = (int)e.size(), n = (int)poly.size();
			if (n < 3) return false;
			for (int i = 0; i < n; i++)
			{
				if (poly[i] < 0 || poly[i] >= (int)verts.size()) return false;
				hedge h = { base + i, poly[i], -1, base + (i + 1) % n, base + (i + n - 1) % n, nfaces };
				e.push_back(h);
			}
			nfaces++;
		}
		if (!link()) return false;
		init_back_lists();
		return true;
	}
	std::vector<int> vert_edges(int v) const      // wingmesh.h:86-89
	{
		std::vector<int> r; int e0 = vback[v], k = e0;
		if (k != -1) do { r.push_back(k); k = e[e[k].adj].next; } while (k != e0 && r.size() <= e.size());
		return r;
	}
	std::vector<int> face_edges(int f) const      // FaceView, wingmesh.h:118-146
	{
		std::vector<int> r; int e0 = fback[f], k = e0;
		do { r.push_back(k); k = e[k].next; } while (k != e0 && r.size() <= e.size());
		return r;
	}
	int build_edge(int ea, int eb)      // wingmesh.h:217-247: new edge ea.v -> eb.v splits the face, returns the half-edge on the new face
	{
		int newface = nfaces, n = (int)e.size();
		hedge sa = { n + 0, e[ea].v, n + 1, eb, e[ea].prev, newface };
		hedge sb = { n + 1, e[eb].v, n + 0, ea, e[eb].prev, e[ea].face };
		e[sa.prev].next = e[sa.next].prev = sa.id;
		e[sb.prev].next = e[sb.next].prev = sb.id;
		e.push_back(sa); e.push_back(sb);
		nfaces++;
		if (fback.size()) { fback.push_back(sa.id); fback[sb.face] = sb.id; }
		for (int k = e[sa.id].next; k != sa.id; k = e[k].next) e[k].face = newface;
		return sa.id;
	}
	void split_edge(int ei, v3 vpos)      // wingmesh.h:250-270: inserts vertex vpos in the middle of edge ei
	{
		int ea = e[ei].adj, v = (int)verts.size(), n = (int)e.size();
		hedge s0 = { n + 0, v, n + 1, e[ei].next, ei, e[ei].face };
		hedge sa = { n + 1, e[ea].v, n + 0, ea, e[ea].prev, e[ea].face };
		e[s0.prev].next = e[s0.next].prev = s0.id;
		e[sa.prev].next = e[sa.next].prev = sa.id;
		e[ea].v = v;
		e.push_back(s0); e.push_back(sa);
		verts.push_back(vpos);
		if (vback.size()) { vback.push_back(s0.id); vback[sa.v] = sa.id; }
	}
	std::vector<int> tris() const      // GenerateTris, wingmesh.h:563-575: fan per face from its first half-edge
	{
		std::vector<int> t;
		for (int e0 : fback)
		{
			if (e0 == -1) continue;
			int ea = e0, eb = e[ea].next;
			while ((eb = e[ea = eb].next) != e0) { t.push_back(e[e0].v); t.push_back(e[ea].v); t.push_back(e[eb].v); }
		}
		return t;
	}
};

bool subdivide(hemesh &m)      // wingmesh.h:730-786
{
	const int nv = (int)m.verts.size(), ne = (int)m.e.size(), nf = m.nfaces;
	std::vector<v3> fpoint;
	for (int f = 0; f < nf; f++)
	{
		v3 c = V3(0, 0, 0); auto fe = m.face_edges(f);
		for (int k : fe) c = c + m.verts[m.e[k].v];
		fpoint.push_back(c / (float)fe.size());
	}
	for (int i = 0; i < ne; i++)
	{
		if (m.e[i].v >= nv || m.e[m.e[i].adj].v >= nv) continue;      // already split from the other side
		const hedge &h = m.e[i], &a = m.e[h.adj];
		v3 mid = (((m.verts[h.v] + m.verts[a.v]) + fpoint[h.face]) + fpoint[a.face]) / 4.0f;
		m.split_edge(i, mid);
	}
	for (int v = 0; v < nv; v++)
	{
		int k = 0; v3 fsum = V3(0, 0, 0), esum = V3(0, 0, 0);
		for (int ei : m.vert_edges(v)) { esum = esum + m.verts[m.e[m.e[ei].adj].v]; fsum = fsum + fpoint[m.e[ei].face]; k++; }
		if (!k) return false;
		m.verts[v] = (m.verts[v] * ((k - 2.0f) / k) + esum * (1.0f / k / k)) + fsum * (1.0f / k / k);
	}
	for (int f = 0; f < nf; f++)
	{
		std::vector<int> mids;
		for (int k : m.face_edges(f)) if (m.e[k].v >= nv) mids.push_back(k);
		if (mids.size() < 3) return false;
		m.build_edge(mids[1], mids[0]);
		m.split_edge((int)m.e.size() - 1, fpoint[f]);
		int spoke = (int)m.e.size() - 2;
		for (int k = (int)mids.size() - 1; k >= 2; k--) m.build_edge(mids[k], spoke);
	}
	return true;
}

// ------------------------------------------------------------------------------------------------- greedy convex hull (hull.h)
int maxdir(const v3 *p, int n, v3 dir)      // geometric.h:218-224 (std::max_element: first maximum)
{
	int best = 0;
	for (int i = 1; i < n; i++) if (dot(p[best], dir) < dot(p[i], dir)) best = i;
	return best;
}
struct htri { int v[3], n[3], id, vmax; float rise; bool dead() const { return n[0] == -1; } };
htri make_tri(int a, int b, int c, int id, int n0, int n1, int n2) { htri t; t.v[0] = a; t.v[1] = b; t.v[2] = c; t.n[0] = n0; t.n[1] = n1; t.n[2] = n2; t.id = id; t.vmax = -1; t.rise = 0.0f; return t; }
struct hull_builder
{
	std::vector<htri> T;
	bool ok = true;
	int &neib(int t, int va, int vb)      // hull.h:93-105
	{
		for (int i = 0; i < 3; i++)
		{
			int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
			if ((T[t].v[i] == va && T[t].v[i1] == vb) || (T[t].v[i] == vb && T[t].v[i1] == va)) return T[t].n[i2];
		}
		ok = false; return T[t].n[0];
	}
	static bool hasvert(const htri &t, int v) { return t.v[0] == v || t.v[1] == v || t.v[2] == v; }
	void nnfix(int k)      // hull.h:108-122
	{
		if (T[k].id == -1) return;
		for (int i = 0; i < 3; i++)
		{
			int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
			if (T[k].n[i] != -1) neib(T[k].n[i], T[k].v[i2], T[k].v[i1]) = k;
		}
	}
	void swapn(int a, int b) { std::swap(T[a], T[b]); std::swap(T[a].id, T[b].id); nnfix(a); nnfix(b); }      // hull.h:123-129
	void b2bfix(int s, int t)      // hull.h:131-144
	{
		for (int i = 0; i < 3; i++)
		{
			int va = T[s].v[(i + 1) % 3], vb = T[s].v[(i + 2) % 3];
			neib(neib(s, va, vb), vb, va) = neib(t, vb, va);
			neib(neib(t, vb, va), va, vb) = neib(s, va, vb);
		}
		for (int i = 0; i < 3; i++) T[s].n[i] = T[t].n[i] = -1;
	}
	void extrude(int t0, int v)      // hull.h:162-183
	{
		int t[3] = { T[t0].v[0], T[t0].v[1], T[t0].v[2] }, n[3] = { T[t0].n[0], T[t0].n[1], T[t0].n[2] };
		int b = (int)T.size();
		T.push_back(make_tri(v, t[1], t[2], b + 0, n[0], b + 1, b + 2)); neib(n[0], t[1], t[2]) = b + 0;
		T.push_back(make_tri(v, t[2], t[0], b + 1, n[1], b + 2, b + 0)); neib(n[1], t[2], t[0]) = b + 1;
		T.push_back(make_tri(v, t[0], t[1], b + 2, n[2], b + 0, b + 1)); neib(n[2], t[0], t[1]) = b + 2;
		T[t0].n[0] = T[t0].n[1] = T[t0].n[2] = -1;
		if (hasvert(T[n[0]], v)) b2bfix(b + 0, n[0]);
		if (hasvert(T[n[1]], v)) b2bfix(b + 1, n[1]);
		if (hasvert(T[n[2]], v)) b2bfix(b + 2, n[2]);
	}
	int extrudable(float epsilon) const      // hull.h:185-199
	{
		int t = -1;
		for (int i = 0; i < (int)T.size(); i++) if (t < 0 || T[t].rise < T[i].rise) t = i;
		return (T[t].rise > epsilon) ? t : -1;
	}
};
bool above(const v3 *verts, const int *t, v3 p, float epsilon) { v3 n = tri_normal(verts[t[0]], verts[t[1]], verts[t[2]]); return dot(n, p - verts[t[0]]) > epsilon; }      // hull.h:48-52
bool find_simplex(const v3 *verts, int count, int p[4])      // hull.h:201-231
{
	v3 b0 = V3(0.01f, 0.02f, 1.0f);
	int p0 = maxdir(verts, count, b0), p1 = maxdir(verts, count, -b0);
	b0 = verts[p0] - verts[p1];
	if (p0 == p1 || is_zero(b0)) return false;
	v3 b1 = cross(V3(1, 0, 0), b0), b2 = cross(V3(0, 1, 0), b0);
	b1 = normalize((length(b1) > length(b2)) ? b1 : b2);
	int p2 = maxdir(verts, count, b1);
	if (p2 == p0 || p2 == p1) p2 = maxdir(verts, count, -b1);
	if (p2 == p0 || p2 == p1) return false;
	b1 = verts[p2] - verts[p0];
	b2 = cross(b1, b0);
	int p3 = maxdir(verts, count, b2);
	if (p3 == p0 || p3 == p1 || p3 == p2) p3 = maxdir(verts, count, -b2);
	if (p3 == p0 || p3 == p1 || p3 == p2) return false;
	if (dot(verts[p3] - verts[p0], cross(verts[p1] - verts[p0], verts[p2] - verts[p0])) < 0) std::swap(p2, p3);
	p[0] = p0; p[1] = p1; p[2] = p2; p[3] = p3;
	return true;
}
// Reorders `verts` so the hull's vertices come first (like the reference) and returns indexed triangles.
bool greedy_hull(std::vector<v3> &vs, int vlimit, std::vector<int> &tris_out)      // hull.h:311-424
{
	v3 *verts = vs.data(); const int count = (int)vs.size();
	tris_out.clear();
	if (count < 4) return false;
	if (vlimit == 0) vlimit = 1000000000;
	v3 bmin = verts[0], bmax = verts[0];
	std::vector<int> isextreme(count, 0);
	for (int j = 0; j < count; j++)
	{
		bmin = V3(fminf(bmin.x, verts[j].x), fminf(bmin.y, verts[j].y), fminf(bmin.z, verts[j].z));
		bmax = V3(fmaxf(bmax.x, verts[j].x), fmaxf(bmax.y, verts[j].y), fmaxf(bmax.z, verts[j].z));
	}
	const float epsilon = length(bmax - bmin) * 0.001f;
	int p[4];
	if (!find_simplex(verts, count, p)) return false;
	hull_builder H; auto &T = H.T;
	const v3 center = (((verts[p[0]] + verts[p[1]]) + verts[p[2]]) + verts[p[3]]) / 4.0f;
	T.push_back(make_tri(p[2], p[3], p[1], 0, 2, 3, 1));
	T.push_back(make_tri(p[3], p[2], p[0], 1, 3, 2, 0));
	T.push_back(make_tri(p[0], p[1], p[3], 2, 0, 1, 3));
	T.push_back(make_tri(p[1], p[0], p[2], 3, 1, 0, 2));
	isextreme[p[0]] = isextreme[p[1]] = isextreme[p[2]] = isextreme[p[3]] = 1;
	for (auto &t : T)
	{
		v3 n = tri_normal(verts[t.v[0]], verts[t.v[1]], verts[t.v[2]]);
		t.vmax = maxdir(verts, count, n);
		t.rise = dot(n, verts[t.vmax] - verts[t.v[0]]);
	}
	int te;
	vlimit -= 4;
	while (vlimit > 0 && (te = H.extrudable(epsilon)) >= 0)
	{
		const int v = T[te].vmax;
		isextreme[v] = 1;
		for (int j = (int)T.size(); j--;)
		{
			if (T[j].dead()) continue;
			int t[3] = { T[j].v[0], T[j].v[1], T[j].v[2] };
			if (above(verts, t, verts[v], 0.01f * epsilon)) H.extrude(j, v);
		}
		for (int j = (int)T.size(); j--;)      // flipped or sliver triangles next to the new vertex
		{
			if (T[j].dead()) continue;
			if (!hull_builder::hasvert(T[j], v)) break;
			int nt[3] = { T[j].v[0], T[j].v[1], T[j].v[2] };
			if (above(verts, nt, center, 0.01f * epsilon) || length(cross(verts[nt[1]] - verts[nt[0]], verts[nt[2]] - verts[nt[1]])) < epsilon * epsilon * 0.1f)
			{
				int nb = T[j].n[0];
				if (nb < 0 || T[nb].dead()) return false;
				H.extrude(nb, v);
				j = (int)T.size();
			}
		}
		for (int j = (int)T.size(); j--;)
		{
			htri &t = T[j];
			if (t.dead()) continue;
			if (t.vmax >= 0) break;
			v3 n = tri_normal(verts[t.v[0]], verts[t.v[1]], verts[t.v[2]]);
			t.vmax = maxdir(verts, count, n);
			if (isextreme[t.vmax]) t.vmax = -1;
			else t.rise = dot(n, verts[t.vmax] - verts[t.v[0]]);
		}
		for (int j = (int)T.size(); j--;)
		{
			if (!T[j].dead()) continue;
			H.swapn(j, (int)T.size() - 1);
			T.pop_back();
		}
		if (!H.ok) return false;
		vlimit--;
	}
	for (auto &t : T) for (int k = 0; k < 3; k++) tris_out.push_back(t.v[k]);
	std::vector<int> used(count, 0), map(count, 0);
	for (int idx : tris_out) used[idx]++;
	for (int i = 0, n = 0; i < count; i++) { if (used[i]) { map[i] = n++; std::swap(verts[map[i]], verts[i]); } else map[i] = -1; }
	for (int &idx : tris_out) idx = map[idx];
	return H.ok;
}

// ------------------------------------------------------------------------------------------------- mass properties
m3 M3(v3 a, v3 b, v3 c) { m3 m; m.x = a; m.y = b; m.z = c; return m; }
float mesh_volume(const v3 *v, const int *t, int count)      // geometric.h:372-381
{
	float vol = 0;
	for (int i = 0; i < count; i++) vol += determinant(M3(v[t[3 * i]], v[t[3 * i + 1]], v[t[3 * i + 2]]));
	return vol / 6.0f;
}
v3 mesh_com(const v3 *v, const int *t, int count)      // geometric.h:383-397
{
	v3 com = V3(0, 0, 0); float volume = 0;
	for (int i = 0; i < count; i++)
	{
		m3 A = M3(v[t[3 * i]], v[t[3 * i + 1]], v[t[3 * i + 2]]);
		float vol = determinant(A);
		com = com + ((A.x + A.y) + A.z) * vol;
		volume += vol;
	}
	return com / (volume * 4.0f);
}
m3 mesh_inertia(const v3 *v, const int *t, int count, v3 com)      // geometric.h:398-428
{
	float volume = 0; float diag[3] = { 0, 0, 0 }, offd[3] = { 0, 0, 0 };
	for (int i = 0; i < count; i++)
	{
		v3 c0 = v[t[3 * i]] - com, c1 = v[t[3 * i + 1]] - com, c2 = v[t[3 * i + 2]] - com;
		float A[3][3] = { { c0.x, c0.y, c0.z }, { c1.x, c1.y, c1.z }, { c2.x, c2.y, c2.z } };
		float d = determinant(M3(c0, c1, c2));
		volume += d;
		for (int j = 0; j < 3; j++)
		{
			int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
			diag[j] += (A[0][j] * A[1][j] + A[1][j] * A[2][j] + A[2][j] * A[0][j] +
			            A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]) * d;
			offd[j] += (A[0][j1] * A[1][j2] + A[1][j1] * A[2][j2] + A[2][j1] * A[0][j2] +
			            A[0][j1] * A[2][j2] + A[1][j1] * A[0][j2] + A[2][j1] * A[1][j2] +
			            A[0][j1] * A[0][j2] * 2 + A[1][j1] * A[1][j2] * 2 + A[2][j1] * A[2][j2] * 2) * d;
		}
	}
	float dd = volume * (60.0f / 6.0f), od = volume * (120.0f / 6.0f);
	for (int j = 0; j < 3; j++) { diag[j] /= dd; offd[j] /= od; }
	return M3(V3(diag[1] + diag[2], -offd[2], -offd[1]), V3(-offd[2], diag[0] + diag[2], -offd[0]), V3(-offd[1], -offd[0], diag[0] + diag[1]));
}
v4 poly_plane3(v3 a, v3 b, v3 c)      // PolyPlane on a triangle, geometric.h:247-260
{
	v3 p[3] = { a, b, c }; v3 ctr = V3(0, 0, 0), n = V3(0, 0, 0);
	for (int i = 0; i < 3; i++) ctr = ctr + p[i] * (1.0f / 3);
	for (int i = 0; i < 3; i++) n = n + cross(p[i] - ctr, p[(i + 1) % 3] - ctr);
	if (is_zero(n)) return V4(0, 0, 0, 0);
	n = normalize(n);
	return V4(n, -dot(ctr, n));
}

struct body
{
	std::vector<v3> verts; std::vector<int> tris; std::vector<v4> planes;
	float mass, massinv, radius, radius_inner, damping, friction, gravscale;
	v3 com, position; v4 orientation; m3 tensorinv; int collide;
	std::vector<int> ignore;
};
bool make_body(body &b, std::vector<v3> verts, std::vector<int> tris, v3 position, float friction)      // RigidBody ctor, physics.h:147-172
{
	const int nt = (int)tris.size() / 3;
	if (!nt) return false;
	b.verts = std::move(verts); b.tris = std::move(tris);
	b.orientation = V4(0, 0, 0, 1); b.collide = 3; b.mass = 1; b.gravscale = 1.0f; b.damping = 0.0f; b.friction = friction; b.radius_inner = 0;
	{      // CenterOfMass(shapes), physics.h:66-82
		v3 cg = mesh_com(b.verts.data(), b.tris.data(), nt);
		float v = mesh_volume(b.verts.data(), b.tris.data(), nt), vol = 0;
		vol += v;
		b.com = (V3(0, 0, 0) + cg * v) / vol;
	}
	b.position = position + b.com;
	for (auto &v : b.verts) v = v - b.com;
	m3 tensor;
	{      // Inertia(shapes, 0), physics.h:85-100
		float v = mesh_volume(b.verts.data(), b.tris.data(), nt), vol = 0;
		m3 I = mesh_inertia(b.verts.data(), b.tris.data(), nt, V3(0, 0, 0)) * v;
		tensor = M3(V3(0, 0, 0) + I.x, V3(0, 0, 0) + I.y, V3(0, 0, 0) + I.z);
		vol += v;
		tensor = tensor * (1.0f / vol);
	}
	b.massinv = 1.0f / b.mass;
	b.tensorinv = inverse(tensor);
	int far = 0;
	for (int i = 1; i < (int)b.verts.size(); i++) if (dot(b.verts[far], b.verts[far]) < dot(b.verts[i], b.verts[i])) far = i;
	b.radius = length(b.verts[far]);
	return true;
}
v3 position_user(const body &b) { return apply(XF(b.position, b.orientation), -b.com); }      // physics.h:142

void put_body(fx_map &out, int i, const body &b, std::vector<float> &bf)
{
	std::vector<float> v, p;
	for (auto &x : b.verts) { v.push_back(x.x); v.push_back(x.y); v.push_back(x.z); }
	for (auto &x : b.planes) { p.push_back(x.x); p.push_back(x.y); p.push_back(x.z); p.push_back(x.w); }
	std::string k = "b" + std::to_string(i);
	out[k + "/verts"] = fx_f32(v, { (uint32_t)b.verts.size(), 3 });
	out[k + "/planes"] = fx_f32(p, { (uint32_t)b.planes.size(), 4 });
	out[k + "/tris"] = fx_i32(b.tris, { (uint32_t)b.tris.size() / 3, 3 });
	float row[26] = { b.mass, b.massinv, b.radius, b.radius_inner, b.damping, b.friction, b.gravscale, b.com.x, b.com.y, b.com.z, b.position.x, b.position.y, b.position.z,
		b.orientation.x, b.orientation.y, b.orientation.z, b.orientation.w,
		b.tensorinv.x.x, b.tensorinv.x.y, b.tensorinv.x.z, b.tensorinv.y.x, b.tensorinv.y.y, b.tensorinv.y.z, b.tensorinv.z.x, b.tensorinv.z.y, b.tensorinv.z.z };
	bf.insert(bf.end(), row, row + 26);
}
} // namespace

bool ht_build_model(const char *json_path, int flags, fx_map &out, std::string &err)
{
	std::string text;
	{
		FILE *fp = fopen(json_path, "rb");
		if (!fp) { err = std::string("cannot open model file ") + json_path; return false; }
		char buf[65536]; size_t n;
		while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
		fclose(fp);
	}
	jnode root; jparser jp = { text.data(), text.data() + text.size(), "" };
	if (!jp.value(root, 0) || root.kind != jnode::OBJ) { err = std::string("model json: ") + (jp.err.empty() ? "object expected" : jp.err); return false; }
	const jnode *cages = root.get("controlcages"), *jjoints = root.get("joints");
	if (!cages || cages->kind != jnode::ARR || !jjoints || jjoints->kind != jnode::ARR) { err = "model json: \"controlcages\" and \"joints\" arrays required"; return false; }

	const float physics_coloumb = 0.6f;      // physics.h:37
	struct joint { int rb0, rb1; v3 p0, p1, rmin, rmax; v4 frame; };
	std::vector<joint> joints;
	for (auto &j : jjoints->items)
	{
		joint J; const jnode *a = j.get("rbi0"), *b = j.get("rbi1");
		if (!a || !b || !read_v3(j.get("p0"), J.p0) || !read_v3(j.get("p1"), J.p1) || !read_v3(j.get("rangemin"), J.rmin) || !read_v3(j.get("rangemax"), J.rmax) || !read_v4(j.get("jointframe"), J.frame))
		{ err = "model json: malformed joint"; return false; }
		J.rb0 = a->as_int(); J.rb1 = b->as_int();
		joints.push_back(J);
	}
	const int nb = (int)cages->items.size(), nj = (int)joints.size();
	if (nb < 1 || nj != nb - 1) { err = "model json: expected one joint per body after the root"; return false; }
	for (int j = 0; j < nj; j++) if (joints[j].rb0 < 0 || joints[j].rb0 > j || joints[j].rb1 < 0 || joints[j].rb1 >= nb) { err = "model json: joint " + std::to_string(j) + " must attach to an earlier body"; return false; }

	std::vector<body> bodies(nb);
	std::vector<std::vector<v3>> sdverts(nb);      // the subdivision surfaces GetMeshes(true) hands to a renderer (physmodel.h:295-303)
	for (int i = 0; i < nb; i++)
	{
		const jnode &c = cages->items[i]; const jnode *jv = c.get("verts"), *jf = c.get("faces");
		if (!jv || !jf || jv->kind != jnode::ARR || jf->kind != jnode::ARR) { err = "model json: cage needs \"verts\" and \"faces\""; return false; }
		std::vector<v3> cv; std::vector<std::vector<int>> cf;
		for (auto &v : jv->items) { v3 p; if (!read_v3(&v, p)) { err = "model json: malformed vertex"; return false; } cv.push_back(p); }
		for (auto &f : jf->items) { std::vector<int> poly; for (auto &k : f.items) poly.push_back(k.as_int()); cf.push_back(poly); }
		hemesh m;
		if (!m.create(cv, cf)) { err = "model json: cage " + std::to_string(i) + " is not a closed manifold"; return false; }
		if (!subdivide(m) || !subdivide(m)) { err = "subdivision failed on cage " + std::to_string(i); return false; }      // physmodel.h:256
		{      // sdmeshes (physmodel.h:258): MeshFlatShadeTex(subdiv.verts, subdiv.GenerateTris()) -- three fresh vertices per triangle, in the bone's rig frame
			const std::vector<int> st = m.tris();
			for (int k : st) sdverts[i].push_back(m.verts[k]);
		}
		std::vector<v3> verts = m.verts; std::vector<int> tris;
		if (!greedy_hull(verts, 48, tris)) { err = "convex hull failed on cage " + std::to_string(i); return false; }      // physmodel.h:454
		v3 position = V3(0, 0, 0);
		if (i) { const joint &J = joints[i - 1]; position = (position_user(bodies[J.rb0]) + J.p0) - J.p1; }      // physmodel.h:455
		if (!make_body(bodies[i], verts, tris, position, physics_coloumb)) { err = "degenerate body " + std::to_string(i); return false; }
	}
	// build_ignore_lists, physmodel.h:260-277 (duplicates are harmless: only membership is used)
	auto ignore = [&](int a, int b) { bodies[a].ignore.push_back(b); bodies[b].ignore.push_back(a); };
	for (auto &j : joints) ignore(j.rb0, j.rb1);
	for (auto &ja : joints) for (auto &jb : joints) if (ja.rb0 == jb.rb0 && ja.rb1 != jb.rb1) ignore(ja.rb1, jb.rb1);
	for (auto &ja : joints) for (auto &jb : joints) if (ja.rb1 == jb.rb0) ignore(ja.rb0, jb.rb1);
	// rbscalemass(wrist, 3), (palm, 5): physmodel.h:460-461, physics.h:176-183
	const float scale[2] = { 3.0f, 5.0f };
	for (int i = 0; i < 2 && i < nb; i++) { bodies[i].mass *= scale[i]; bodies[i].massinv *= 1.0f / scale[i]; }
	for (auto &b : bodies) { b.damping = 0.8f; b.gravscale = 0; }
	for (auto &b : bodies)      // physmodel.h:468-473
	{
		for (size_t t = 0; t + 2 < b.tris.size(); t += 3)
		{
			v4 p = poly_plane3(b.verts[b.tris[t]], b.verts[b.tris[t + 1]], b.verts[b.tris[t + 2]]);
			if (p.x != 0.0f || p.y != 0.0f || p.z != 0.0f || p.w != 0.0f) b.planes.push_back(p);
		}
		if (b.planes.empty()) { err = "body without planes"; return false; }
		float wmax = b.planes[0].w; for (auto &p : b.planes) if (wmax < p.w) wmax = p.w;
		b.radius_inner = -wmax;
	}
	if (flags & HT_BUILD_HAND_TWEAKS)      // handtrack.h:350-358
	{
		for (int i = 2; i < nb; i++) for (auto &v : bodies[i].verts) v = V3(v.x * 0.7f, v.y * 0.7f, v.z * 0.9f);
		for (int i : { 7, 10, 13, 16 }) if (i < nb) ignore(i, 2);
	}

	out.clear();
	out["nb"] = fx_i32({ nb }, { 1 }); out["nj"] = fx_i32({ nj }, { 1 });
	std::vector<float> bf, rest; std::vector<int> collide, ign((size_t)nb * nb, 0), nverts, nplanes, igncount;
	for (int i = 0; i < nb; i++)
	{
		const body &b = bodies[i];
		put_body(out, i, b, bf);
		{ std::vector<float> sv; for (auto &x : sdverts[i]) { sv.push_back(x.x); sv.push_back(x.y); sv.push_back(x.z); } out["b" + std::to_string(i) + "/sdverts"] = fx_f32(sv, { (uint32_t)sdverts[i].size(), 3 }); }
		collide.push_back(b.collide); nverts.push_back((int)b.verts.size()); nplanes.push_back((int)b.planes.size());
		for (int k : b.ignore) ign[(size_t)i * nb + k] = 1;
		igncount.push_back((int)b.ignore.size());      // entries incl. duplicates: what `ignore.size()` reads in handtrack.h:408
		float st[13] = { b.position.x, b.position.y, b.position.z, b.orientation.x, b.orientation.y, b.orientation.z, b.orientation.w, 0, 0, 0, 0, 0, 0 };
		rest.insert(rest.end(), st, st + 13);
	}
	out["body_f"] = fx_f32(bf, { (uint32_t)nb, 26 }); out["body_collide"] = fx_i32(collide, { (uint32_t)nb }); out["ignore"] = fx_i32(ign, { (uint32_t)nb, (uint32_t)nb });
	out["ignore_count"] = fx_i32(igncount, { (uint32_t)nb });
	out["nverts"] = fx_i32(nverts, { (uint32_t)nb }); out["nplanes"] = fx_i32(nplanes, { (uint32_t)nb });
	out["rest_state"] = fx_f32(rest, { (uint32_t)nb, 13 });
	std::vector<int> ji; std::vector<float> jf;
	for (auto &J : joints)
	{
		ji.push_back(J.rb0); ji.push_back(J.rb1);
		float row[16] = { J.p0.x, J.p0.y, J.p0.z, J.p1.x, J.p1.y, J.p1.z, J.rmin.x, J.rmin.y, J.rmin.z, J.rmax.x, J.rmax.y, J.rmax.z, J.frame.x, J.frame.y, J.frame.z, J.frame.w };
		jf.insert(jf.end(), row, row + 16);
	}
	out["joint_i"] = fx_i32(ji, { (uint32_t)nj, 2 }); out["joint_f"] = fx_f32(jf, { (uint32_t)nj, 16 });
	// physics globals as the HandTracker ctor leaves them (physics.h:34-47, handtrack.h:837-838, physmodel.h:234, handtrack.h:369,450)
	out["physics"] = fx_f32({ (1.0f / 60.0f), 0.4f, 0.0f, 0.0f, 0.0f, physics_coloumb, 0.3f, 0.3f, 0.3f, 0.2f, 0.03f / 8.0f, 0.15f, 16.0f, 4.0f, 1.0f, 0.4f, 4.0f, 0.1f }, { 18 });
	{      // UnibodyFit's 0.1 m cube proxy: WingMeshCube(0.1f) -> RigidBody (handtrack.h:454-455, wingmesh.h:857-881)
		const float r = 0.1f;
		std::vector<v3> cv = { V3(-r, -r, -r), V3(-r, -r, r), V3(-r, r, -r), V3(-r, r, r), V3(r, -r, -r), V3(r, -r, r), V3(r, r, -r), V3(r, r, r) };
		std::vector<std::vector<int>> cf = { { 0, 1, 3, 2 }, { 6, 7, 5, 4 }, { 0, 4, 5, 1 }, { 3, 7, 6, 2 }, { 0, 2, 6, 4 }, { 1, 5, 7, 3 } };
		hemesh box; body ub;
		if (!box.create(cv, cf) || !make_body(ub, box.verts, box.tris(), V3(0, 0, 0), physics_coloumb)) { err = "unibody proxy failed"; return false; }
		std::vector<float> v; for (auto &x : ub.verts) { v.push_back(x.x); v.push_back(x.y); v.push_back(x.z); }
		out["unibody/verts"] = fx_f32(v, { 8, 3 });
		out["unibody/f"] = fx_f32({ ub.mass, ub.massinv, ub.radius, ub.damping, ub.friction, ub.gravscale, ub.com.x, ub.com.y, ub.com.z,
			ub.tensorinv.x.x, ub.tensorinv.x.y, ub.tensorinv.x.z, ub.tensorinv.y.x, ub.tensorinv.y.y, ub.tensorinv.y.z, ub.tensorinv.z.x, ub.tensorinv.z.y, ub.tensorinv.z.z }, { 18 });
	}
	return true;
}

// Top-level members of a JSON object file as raw text (numbers keep their digits); used by ht_config_read.
bool ht_json_top_level(const char *path, std::map<std::string, std::string> &numbers, std::string &err)
{
	std::string text;
	FILE *fp = fopen(path, "rb");
	if (!fp) { err = std::string("cannot open ") + path; return false; }
	char buf[65536]; size_t n;
	while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
	fclose(fp);
	jnode root; jparser jp = { text.data(), text.data() + text.size(), "" };
	if (!jp.value(root, 0) || root.kind != jnode::OBJ) { err = std::string("json: ") + (jp.err.empty() ? "object expected" : jp.err); return false; }
	for (size_t i = 0; i < root.keys.size(); i++) if (root.items[i].kind == jnode::NUM) numbers[root.keys[i]] = root.items[i].text;
	return true;
}
