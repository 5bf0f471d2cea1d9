// ht_model_host.hip -- host-only view of a built hand model behind the C-ABI (ht_model_*): what an application needs from a PhysModel object
// that is NOT being tracked -- the reference's synthetic-tracker keeps such a "fake hand" to pose, draw and ray-cast its input frames
// (synthetic-hand-tracker/synthetic-tracker.cpp:94-96,139,69-76).  No device call is made here.
//
// Reference interfaces: PhysModel::PhysModel(const char*) + LoadHandModel() (include/physmodel.h:444-475, include/handtrack.h:347-366) through the
// same host builder ht_create uses; PhysModel::HitCheck (physmodel.h:287-294) = ConvexHitCheck of every body's hull planes
// (third_party/geometric.h:275-302); RigidBody::PositionUser (third_party/physics.h:142).
#include <string.h>
#include <string>
#include <vector>
#include "ht_device.hpp"
#include "ht_model_build.hpp"

struct ht_model
{
	int nb = 0, nj = 0;
	std::vector<std::vector<float>> verts, planes;      // per body: [n][3] com-centred vertices, [n][4] local half-space planes
	std::vector<std::vector<int>> tris;                 // per body: [n][3] hull triangles over verts
	std::vector<std::vector<float>> sdverts;            // per body: [3 t][3] the subdivision surface's triangles corner by corner, in the bone's rig frame (GetMeshes(true), physmodel.h:258)
	std::vector<float> com, rest;                       // [nb][3], [nb][7]
	std::string err;
};

extern "C" int ht_model_open(const char *path, int hand_tweaks, ht_model **out)
{
	if (!path || !out) return HT_ERR_ARG;
	ht_model *m = new ht_model();
	*out = m;
	fx_map fx;
	char magic[8] = { 0 };
	{ FILE *fp = fopen(path, "rb"); if (!fp) { m->err = std::string("cannot open model ") + path; return HT_ERR_IO; } size_t n = fread(magic, 1, 8, fp); (void)n; fclose(fp); }
	if (!memcmp(magic, "HTFX0001", 8)) { if (!fx_load(path, fx)) { m->err = std::string("cannot read baked model ") + path; return HT_ERR_IO; } }
	else if (!ht_build_model(path, hand_tweaks ? HT_BUILD_HAND_TWEAKS : 0, fx, m->err)) return HT_ERR_IO;
	auto get = [&](const std::string &n) -> const fx_arr * { auto it = fx.find(n); return it == fx.end() ? nullptr : &it->second; };
	const fx_arr *nb = get("nb"), *nj = get("nj"), *bf = get("body_f");
	if (!nb || !nj || !bf) { m->err = "model entries missing"; return HT_ERR_IO; }
	m->nb = nb->i()[0]; m->nj = nj->i()[0];
	for (int b = 0; b < m->nb; b++)
	{
		const std::string k = "b" + std::to_string(b);
		const fx_arr *v = get(k + "/verts"), *p = get(k + "/planes"), *t = get(k + "/tris");
		if (!v || !p || !t) { m->err = "model body arrays missing"; return HT_ERR_IO; }
		m->verts.emplace_back(v->f(), v->f() + (size_t)v->dims[0] * 3);
		m->planes.emplace_back(p->f(), p->f() + (size_t)p->dims[0] * 4);
		m->tris.emplace_back(t->i(), t->i() + (size_t)t->dims[0] * 3);
		const fx_arr *sv = get(k + "/sdverts");
		if (sv) m->sdverts.emplace_back(sv->f(), sv->f() + (size_t)sv->dims[0] * 3); else m->sdverts.emplace_back();
		const float *r = bf->f() + 26 * b;      // mass massinv radius radius_inner damping friction gravscale com3 pos_start3 quat_start4 tensorinv9
		m->com.insert(m->com.end(), r + 7, r + 10);
		m->rest.insert(m->rest.end(), r + 10, r + 17);
	}
	return HT_OK;
}
extern "C" int ht_model_close(ht_model *m) { if (!m) return HT_ERR_ARG; delete m; return HT_OK; }
extern "C" const char *ht_model_error(const ht_model *m) { return m ? m->err.c_str() : "null model"; }
extern "C" int ht_model_counts(const ht_model *m, int *nb, int *nj) { if (!m) return HT_ERR_ARG; if (nb) *nb = m->nb; if (nj) *nj = m->nj; return HT_OK; }
extern "C" int ht_model_body(const ht_model *m, int body, int *nverts, int *ntris, int *nplanes, float *com3, float *rest_pose7)
{
	if (!m || body < 0 || body >= m->nb) return HT_ERR_ARG;
	if (nverts) *nverts = (int)m->verts[body].size() / 3;
	if (ntris) *ntris = (int)m->tris[body].size() / 3;
	if (nplanes) *nplanes = (int)m->planes[body].size() / 4;
	if (com3) memcpy(com3, &m->com[3 * body], 3 * sizeof(float));
	if (rest_pose7) memcpy(rest_pose7, &m->rest[7 * body], 7 * sizeof(float));
	return HT_OK;
}
extern "C" int ht_model_body_mesh(const ht_model *m, int body, float *verts, int *tris)
{
	if (!m || body < 0 || body >= m->nb) return HT_ERR_ARG;
	if (verts) memcpy(verts, m->verts[body].data(), m->verts[body].size() * sizeof(float));
	if (tris) memcpy(tris, m->tris[body].data(), m->tris[body].size() * sizeof(int));
	return HT_OK;
}
// the subdivision surface of a body (PhysModel::sdmeshes, physmodel.h:258: MeshFlatShadeTex of the twice-subdivided control cage): *nverts corner positions, three per
// triangle in order, in the bone's rig frame -- drawn at Pose{PositionUser, orientation} (physmodel.h:297-298).  verts may be NULL to ask for the count.
extern "C" int ht_model_body_sdmesh(const ht_model *m, int body, int *nverts, float *verts)
{
	if (!m || body < 0 || body >= m->nb) return HT_ERR_ARG;
	if (nverts) *nverts = (int)m->sdverts[body].size() / 3;
	if (verts) memcpy(verts, m->sdverts[body].data(), m->sdverts[body].size() * sizeof(float));
	return HT_OK;
}
// PhysModel::HitCheck(v0, v1): the segment is clipped against the hull of every body in turn, each body starting from the impact the previous
// ones left (physmodel.h:291), so the nearest hit along the segment wins.  poses [nb][7] = the bodies' poses (centre-of-mass frames).
extern "C" int ht_model_hitcheck(const ht_model *m, const float *poses, const float *v0_, const float *v1_, float *impact3, float *normal3, int *body_out)
{
	if (!m || !poses || !v0_ || !v1_) return HT_ERR_ARG;
	const v3 v0w = V3(v0_[0], v0_[1], v0_[2]);
	v3 impact = V3(v1_[0], v1_[1], v1_[2]), normal = V3(0, 0, 0);
	int who = -1;
	for (int b = 0; b < m->nb; b++)
	{
		const float *p = poses + 7 * b;
		const xf pose = XF(V3(p[0], p[1], p[2]), V4(p[3], p[4], p[5], p[6])), inv = inverse(pose);
		v3 a = apply(inv, v0w), c = apply(inv, impact), n = V3(0, 0, 0);
		const v3 c_in = c;
		bool hit = true;
		const std::vector<float> &pl = m->planes[b];
		for (size_t k = 0; k + 3 < pl.size(); k += 4)      // ConvexHitCheck geometric.h:275-297
		{
			const v4 plane = V4(pl[k], pl[k + 1], pl[k + 2], pl[k + 3]);
			const float d0 = dot_plane(plane, a), d1 = dot_plane(plane, c);
			if (d0 >= 0 && d1 >= 0) { hit = false; break; }
			if (d0 <= 0 && d1 <= 0) continue;
			const v3 x = a + ((c - a) * d0) / (d0 - d1);
			if (d0 >= 0) { n = xyz(plane); a = x; } else c = x;
		}
		(void)c_in;
		if (hit) { impact = apply(pose, a); normal = qrot(pose.q, n); who = b; }
	}
	if (impact3) { impact3[0] = impact.x; impact3[1] = impact.y; impact3[2] = impact.z; }
	if (normal3) { normal3[0] = normal.x; normal3[1] = normal.y; normal3[2] = normal.z; }
	if (body_out) *body_out = who;
	return HT_OK;
}
