// ht_handtrack.hpp -- C++ wrappers over the C-ABI (ht_mi355x.h) that keep the reference's own names and signatures, so that
// code written against include/handtrack.h / third_party/cnn.h of IntelRealSense/hand_tracking_samples keeps compiling:
//
//     HandTracker htk;                                   // handtrack.h:830 (paths are constructor arguments here)
//     htk.microforce = 3.0f; htk.mainthreadpasses = 3;   // synthetic-tracker.cpp:91-93
//     std::vector<Pose> pose = htk.update(std::move(dimage));      // handtrack.h:748
//     std::vector<float> y = htk.cnn.Eval(x);                        // cnn.h:550
//
// Provided: the members the per-frame path and synthetic-tracker.cpp touch: HandTracker {update, update_cnn_model, kickstart, slowfit,
// scale, load_config, handmodel / othermodel facades, cnn, cnn_input, cnn_output, cnn_output_analysis}, CNN {Eval, Train, loadb, saveb},
// PoseInitializerCNN, PhysModel / LoadHandModel (a host-side model that is posed, drawn and ray-cast, never tracked), HandSegmentVR, camsub,
// GatherHandExpectedCNN, Pose / Image<T> / DCamera / Mesh.  Define HT_MI355X_GLOBAL_NAMES before including to have these names in the global
// namespace, as the reference's headers put them.  Caller-built constraint rows cross the boundary too: LimitLinear / LimitAngular (physics.h:239-308) hold
// RigidBody pointers into a tracked model's `rigidbodies`, handmodel.FitPointCloud(points, linears, angulars, microforce) (physmodel.h:345) and the free
// PhysicsUpdate(Addresses(model.rigidbodies), linears, angulars) (physics.h:543) keep the reference's signatures.  Documented deviation: update()
// runs the CNN job synchronously every frame (the reference polls a background std::async job for 1 ms, which makes its output
// timing dependent, handtrack.h:755-768); this is HandTracker::update_cnn_model followed by the main-thread passes.
// Errors are reported the way the reference's apps expect them: by throwing std::runtime_error (synthetic-tracker.cpp:255-264).
#pragma once
#include <cfloat>
#include <cstdint>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "ht_mi355x.h"

namespace ht_mi355x
{
struct float2 { float x, y; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct Pose { float3 position{ 0, 0, 0 }; float4 orientation{ 0, 0, 0, 1 }; };            // geometric.h:111-125

class DCamera                                                                                // misc_image.h:30-55
{
	int2 dim_{ 64, 64 }; float2 focal_{ 164.f, 164.f }; float2 principal_{ 32.f, 32.f };
public:
	float depth_scale = 0.001f; Pose pose;
	DCamera() {}
	DCamera(int2 dim, float2 focal, float2 principal, float depth_scale, Pose pose = Pose()) : dim_(dim), focal_(focal), principal_(principal), depth_scale(depth_scale), pose(pose) {}
	int2 &dim() { return dim_; } const int2 &dim() const { return dim_; }
	float2 &focal() { return focal_; } const float2 &focal() const { return focal_; }
	float2 &principal() { return principal_; } const float2 &principal() const { return principal_; }
};
template <class T> struct Image                                                              // misc_image.h:109-129
{
	DCamera cam; std::vector<T> raster;
	Image() {}
	Image(DCamera cam) : cam(cam), raster((size_t)cam.dim().x * cam.dim().y, T()) {}
	Image(DCamera cam, std::vector<T> data) : cam(cam), raster(std::move(data)) {}
	const int2 dim() const { return cam.dim(); }
	T &pixel(int2 p) { return raster[(size_t)p.y * dim().x + p.x]; }
};

// ---- constraint rows a caller builds (physics.h:239-308).  A RigidBody here is a handle to body `index` of one of a tracker's two models: rows hold
//      RigidBody pointers exactly like the reference's (NULL = the world) and only live as long as the HandTracker they point into.
struct RigidBody { ht_ctx *ctx = nullptr; int which = 0, index = 0; };
struct LimitAngular
{
	RigidBody *rb0 = nullptr, *rb1 = nullptr; float3 axis{ 0, 0, 1 }; float torque = 0, targetspin = 0, mintorque = -FLT_MAX, maxtorque = FLT_MAX;
	LimitAngular() {}
	LimitAngular(RigidBody *rb0, RigidBody *rb1, const float3 &axis, float targetspin = 0, float mintorque = -FLT_MAX, float maxtorque = FLT_MAX)
		: rb0(rb0), rb1(rb1), axis(axis), targetspin(targetspin), mintorque(mintorque), maxtorque(maxtorque) {}                                 // physics.h:248-249
};
struct LimitLinear
{
	RigidBody *rb0 = nullptr, *rb1 = nullptr; float3 position0{ 0, 0, 0 }, position1{ 0, 0, 0 }, normal{ 0, 0, 1 };
	float targetdist = 0, targetspeednobias = 0; float2 forcelimit{ -FLT_MAX, FLT_MAX }; int friction_master = 0; float targetspeed = 0, impulsesum = 0;
	LimitLinear() {}
	LimitLinear(RigidBody *rb0, RigidBody *rb1, const float3 &position0, const float3 &position1, const float3 &normal = float3{ 0, 0, 1 }, float targetdist = 0.0f,
	            float targetspeednobias = 0.0f, const float2 forcelimit = { -FLT_MAX, FLT_MAX })
		: rb0(rb0), rb1(rb1), position0(position0), position1(position1), normal(normal), targetdist(targetdist), targetspeednobias(targetspeednobias),
		  forcelimit{ forcelimit.x < forcelimit.y ? forcelimit.x : forcelimit.y, forcelimit.x < forcelimit.y ? forcelimit.y : forcelimit.x } {}   // physics.h:283-287
};
template <class T> std::vector<T *> Addresses(std::vector<T> &v) { std::vector<T *> a; for (auto &e : v) a.push_back(&e); return a; }             // misc.h
namespace detail
{
	inline void pack_rows(const std::vector<LimitLinear> &L, const std::vector<LimitAngular> &A, std::vector<float> &l, std::vector<float> &a)
	{
		for (auto &c : L)
		{
			const float r[16] = { c.rb0 ? (float)c.rb0->index : -1.0f, c.rb1 ? (float)c.rb1->index : -1.0f, c.position0.x, c.position0.y, c.position0.z, c.position1.x, c.position1.y, c.position1.z,
			                      c.normal.x, c.normal.y, c.normal.z, c.targetdist, c.targetspeednobias, c.forcelimit.x, c.forcelimit.y, (float)c.friction_master };
			l.insert(l.end(), r, r + 16);
		}
		for (auto &c : A)
		{
			const float r[8] = { c.rb0 ? (float)c.rb0->index : -1.0f, c.rb1 ? (float)c.rb1->index : -1.0f, c.axis.x, c.axis.y, c.axis.z, c.targetspin, c.mintorque, c.maxtorque };
			a.insert(a.end(), r, r + 8);
		}
	}
}
// void PhysicsUpdate(const std::vector<RigidBody*> &rigidbodies, std::vector<LimitLinear> &Linears, std::vector<LimitAngular> &Angulars, wgeom) (physics.h:543-587):
// one solver step of the model the bodies belong to under the caller's rows (plus the collision rows); `rigidbodies` must be all bodies of one tracked model
inline void PhysicsUpdate(const std::vector<RigidBody *> &rigidbodies, std::vector<LimitLinear> &Linears, std::vector<LimitAngular> &Angulars, const std::vector<std::vector<float3> *> &wgeom = {})
{
	if (rigidbodies.empty() || !rigidbodies[0]) throw std::runtime_error("PhysicsUpdate: no bodies");
	if (!wgeom.empty()) throw std::runtime_error("PhysicsUpdate: world geometry is not supported (every call site of the reference passes none)");
	for (size_t i = 0; i < rigidbodies.size(); i++) if (!rigidbodies[i] || rigidbodies[i]->ctx != rigidbodies[0]->ctx || rigidbodies[i]->which != rigidbodies[0]->which || rigidbodies[i]->index != (int)i) throw std::runtime_error("PhysicsUpdate: pass Addresses(model.rigidbodies) of one tracked model");
	std::vector<float> l, a; detail::pack_rows(Linears, Angulars, l, a);
	const int nl = (int)Linears.size(), na = (int)Angulars.size();
	if (ht_physics_update(rigidbodies[0]->ctx, rigidbodies[0]->which, 1, l.data(), nl, &nl, a.data(), na, &na) != HT_OK) throw std::runtime_error(std::string("PhysicsUpdate: ") + ht_last_error(rigidbodies[0]->ctx));
}

// std::vector<float3> PointCloud(const Image<T> &dimage, float2 filter_range) (misc_image.h:409-417): the in-range pixels of a depth image as camera-space
// points, in row-major order (what synthetic-tracker.cpp:233 draws; the tracker's own cloud is made on the device, k_prepare).  Same expressions as the
// reference: d = pixel * depth_scale, kept when filter_range.x <= d < filter_range.y, point = ((x - cx) / fx, (y - cy) / fy, 1) * d.
template <class T> std::vector<float3> PointCloud(const Image<T> &dimage, float2 filter_range)
{
	std::vector<float3> pointcloud;
	const DCamera &c = dimage.cam;
	for (int y = 0; y < dimage.dim().y; y++) for (int x = 0; x < dimage.dim().x; x++)
	{
		const float d = dimage.raster[(size_t)y * dimage.dim().x + x] * c.depth_scale;
		if (d >= filter_range.x && d < filter_range.y) pointcloud.push_back({ ((float)x - c.principal().x) / c.focal().x * d, ((float)y - c.principal().y) / c.focal().y * d, 1.0f * d });
	}
	return pointcloud;
}
inline DCamera camsub(const DCamera &c, int s)                                               // misc_image.h:60
{
	return DCamera({ c.dim().x / s, c.dim().y / s }, { c.focal().x / (float)s, c.focal().y / (float)s }, { c.principal().x / (float)s, c.principal().y / (float)s }, c.depth_scale, c.pose);
}
struct Mesh { std::vector<float3> verts; std::vector<int3> tris; Pose pose; float4 hack{ 1, 1, 1, 1 }; std::string material; };      // mesh.h (what GetMeshes hands to a renderer)
namespace detail
{
// PhysModel::sdmeshes[body] (physmodel.h:258): the twice-subdivided control cage, flat-shaded -- three fresh vertices per triangle, as MeshFlatShadeTex lays them out
// (mesh.h:154-177; positions only: the tangent frames and texture coordinates a renderer derives from them are not part of this surface)
inline Mesh subdivision_mesh(const ht_model *m, int body)
{
	Mesh mesh; int nv = 0;
	ht_model_body_sdmesh(m, body, &nv, nullptr);
	std::vector<float> v((size_t)nv * 3);
	ht_model_body_sdmesh(m, body, nullptr, v.data());
	for (int i = 0; i < nv; i++) mesh.verts.push_back({ v[3 * i], v[3 * i + 1], v[3 * i + 2] });
	for (int i = 0; i + 2 < nv; i += 3) mesh.tris.push_back({ i, i + 1, i + 2 });
	return mesh;
}
}

// ---- the host-side image helpers the applications draw with (synthetic-tracker.cpp:191,206,208-209,218,221-222): plain loops over small rasters, no device work.
//      Same names and results as the reference's templates; written for this Image / DCamera.
struct byte3 { unsigned char x, y, z; byte3() : x(0), y(0), z(0) {} byte3(unsigned char a, unsigned char b, unsigned char c) : x(a), y(b), z(c) {} explicit byte3(int v) : x((unsigned char)v), y((unsigned char)v), z((unsigned char)v) {} };      // linalg.h:355
inline bool operator==(const byte3 &a, const byte3 &b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
// Transform(image, f) (misc_image.h:165): f over every pixel, the camera kept
template <typename F, typename S> auto Transform(const Image<S> &src, F f) -> Image<decltype(f(S()))>
{
	Image<decltype(f(S()))> dst; dst.cam = src.cam; dst.raster.reserve(src.raster.size());
	for (const S &v : src.raster) dst.raster.push_back(f(v));
	return dst;
}
inline unsigned char ToGrayScale(float x) { const float v = x * 255.0f; return (unsigned char)(v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v)); }      // misc_image.h:169 (truncating, clamp first)
inline unsigned char ToGrayScale(unsigned char x) { return x; }
template <class T> Image<unsigned char> ToGrayScale(const Image<T> &src) { return Transform(src, [](T x) { return ToGrayScale(x); }); }                   // misc_image.h:172
inline Image<byte3> ToRGB(const Image<unsigned char> &src) { return Transform(src, [](unsigned char p) { return byte3(p, p, p); }); }                    // misc_image.h:173
inline Image<byte3> ToRGB(const Image<float> &src) { return ToRGB(ToGrayScale(src)); }
// UpSample (misc_image.h:96-102,135): every pixel becomes a 2x2 block; the camera's dimensions, focal lengths and principal point double (operator*, :61)
template <class T> Image<T> UpSample(const Image<T> &src)
{
	const int w = src.dim().x, h = src.dim().y;
	Image<T> dst; dst.cam = DCamera({ w * 2, h * 2 }, { src.cam.focal().x * 2.0f, src.cam.focal().y * 2.0f }, { src.cam.principal().x * 2.0f, src.cam.principal().y * 2.0f }, src.cam.depth_scale, src.cam.pose);
	dst.raster.resize(src.raster.size() * 4);
	for (int y = 0; y < 2 * h; y++) for (int x = 0; x < 2 * w; x++) dst.raster[(size_t)y * 2 * w + x] = src.raster[(size_t)(y / 2) * w + x / 2];
	return dst;
}
// ImageConcat (misc_image.h:225-238): the images stacked top to bottom in a canvas as wide as the widest, each row copied to the left edge
template <class T> Image<T> ImageConcat(const std::vector<Image<T>> &images)
{
	int w = 0, h = 0;
	for (auto &im : images) { w = im.dim().x > w ? im.dim().x : w; h += im.dim().y; }
	Image<T> dst; dst.cam = DCamera({ w, h }, { (float)w, (float)h }, { (float)w / 2.0f, (float)h / 2.0f }, 0.001f); dst.cam.depth_scale = DCamera().depth_scale;      // DCamera(int2 dim), misc_image.h:47
	dst.raster.assign((size_t)w * h, T());
	size_t row = 0;
	for (auto &im : images) for (int y = 0; y < im.dim().y; y++, row++) for (int x = 0; x < im.dim().x; x++) dst.raster[row * w + x] = im.raster[(size_t)y * im.dim().x + x];
	return dst;
}
// ImageOverlayYellowBlue (handtrack.h:249-254): red = green = the heat-map, blue = the background image as grey
template <class T> Image<byte3> ImageOverlayYellowBlue(const Image<unsigned char> &image_rg, const Image<T> &image_b)
{
	Image<byte3> o; o.cam = image_rg.cam; o.raster.resize(image_rg.raster.size());
	for (size_t i = 0; i < image_rg.raster.size(); i++) o.raster[i] = byte3(image_rg.raster[i], image_rg.raster[i], ToGrayScale(image_b.raster[i]));
	return o;
}
// VisualizeHMaps (handtrack.h:256-267): every heat-map blown up to the background's size, laid over it, doubled once more, the eight stacked (synthetic-tracker.cpp:208,221)
template <class T> Image<byte3> VisualizeHMaps(const std::vector<Image<unsigned char>> &hmaps, Image<T> &dmap_d)
{
	std::vector<Image<byte3>> vmaps;
	for (auto hmap : hmaps)
	{
		while ((long)hmap.dim().x * hmap.dim().y < (long)dmap_d.dim().x * dmap_d.dim().y) hmap = UpSample(hmap);
		vmaps.push_back(UpSample(ImageOverlayYellowBlue(hmap, dmap_d)));
	}
	return ImageConcat(vmaps);
}
// DepthMesh (misc_image.h:419-450; synthetic-tracker.cpp:191): a triangle mesh over the in-range pixels, one vertex per skip x skip cell (the first in-range pixel of the
// cell, scanned towards cells that already have a vertex), two triangles per 2x2 block of vertices whose depth steps stay under gaplimit
template <class T> std::pair<std::vector<float3>, std::vector<int3>> DepthMesh(const Image<T> &dimage, float2 filter_range, float gaplimit = FLT_MAX, int skip = 1)
{
	std::vector<float3> verts; std::vector<int3> tris;
	const int W = dimage.dim().x, H = dimage.dim().y, w = W / skip, h = H / skip;
	std::vector<int> vmap((size_t)w * h, -1);
	const DCamera &c = dimage.cam;
	for (int py = 0; py < h; py++) for (int px = 0; px < w; px++)
	{
		const int rvx = (px && vmap[(size_t)py * w + px - 1] != -1) ? 1 : 0, rvy = (py && vmap[(size_t)(py - 1) * w + px] != -1) ? 1 : 0;
		bool placed = false;
		for (int sy = 0; sy < skip && !placed; sy++) for (int sx = 0; sx < skip && !placed; sx++)
		{
			const int x = px * skip + sx + rvx * ((skip - 1) - sx * 2), y = py * skip + sy + rvy * ((skip - 1) - sy * 2);
			const float d = dimage.raster[(size_t)y * W + x] * c.depth_scale;
			if (d >= filter_range.x && d < filter_range.y)
			{
				vmap[(size_t)py * w + px] = (int)verts.size();
				verts.push_back({ ((float)x - c.principal().x) / c.focal().x * d, ((float)y - c.principal().y) / c.focal().y * d, 1.0f * d });
				placed = true;
			}
		}
	}
	auto step = [&](int a, int b) { const float dz = verts[a].z - verts[b].z; return (dz < 0 ? -dz : dz) > gaplimit; };
	auto ok = [&](int a, int b, int cc) { return !(step(a, b) || step(b, cc) || step(cc, a)); };
	for (int py = 0; py + 1 < h; py++) for (int px = 0; px + 1 < w; px++)
	{
		const int a = vmap[(size_t)py * w + px], b = vmap[(size_t)(py + 1) * w + px], cc = vmap[(size_t)(py + 1) * w + px + 1], d = vmap[(size_t)py * w + px + 1];      // counter-clockwise
		if (a >= 0 && cc >= 0 && ((b >= 0 && ok(a, b, cc)) || (d >= 0 && ok(cc, d, a))))
		{
			if (b >= 0 && ok(a, b, cc)) tris.push_back({ a, b, cc });
			if (d >= 0 && ok(cc, d, a)) tris.push_back({ cc, d, a });
		}
		else if (b >= 0 && d >= 0)
		{
			if (a >= 0 && ok(d, a, b)) tris.push_back({ d, a, b });
			if (cc >= 0 && ok(b, cc, d)) tris.push_back({ b, cc, d });
		}
	}
	return { verts, tris };
}
inline float3 qrot_(const float4 &q, const float3 &v)                                       // linalg.h:288
{
	const float3 X{ q.w * q.w + q.x * q.x - q.y * q.y - q.z * q.z, (q.x * q.y + q.z * q.w) * 2, (q.z * q.x - q.y * q.w) * 2 };
	const float3 Y{ (q.x * q.y - q.z * q.w) * 2, q.w * q.w - q.x * q.x + q.y * q.y - q.z * q.z, (q.y * q.z + q.x * q.w) * 2 };
	const float3 Z{ (q.z * q.x + q.y * q.w) * 2, (q.y * q.z - q.x * q.w) * 2, q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z };
	return { (X.x * v.x + Y.x * v.y) + Z.x * v.z, (X.y * v.x + Y.y * v.y) + Z.y * v.z, (X.z * v.x + Y.z * v.y) + Z.z * v.z };
}
inline float3 operator*(const Pose &p, const float3 &v) { const float3 r = qrot_(p.orientation, v); return { p.position.x + r.x, p.position.y + r.y, p.position.z + r.z }; }      // geometric.h:119

namespace detail { inline ht_ctx *&live_ctx() { static ht_ctx *c = nullptr; return c; } }      // a context the free functions below can run on
inline void check(ht_ctx *ctx, int rc) { if (rc != HT_OK) throw std::runtime_error(std::string("ht_mi355x: ") + (ctx ? ht_last_error(ctx) : "no context")); }

class CNN;
CNN PoseInitializerCNN(std::string filename, int device);
class CNN                                                                                    // third_party/cnn.h:100-605 (Eval / Train / loadb / saveb)
{
	ht_ctx *ctx_ = nullptr;
	std::shared_ptr<ht_ctx> own_;      // set when the object owns its (CNN-only) context: PoseInitializerCNN
	friend struct HandTracker;
	friend CNN PoseInitializerCNN(std::string, int);
public:
	std::vector<float> Eval(const std::vector<float> &x)                                     // cnn.h:550
	{
		if (x.size() != HT_CNN_IN) throw std::runtime_error("CNN::Eval expects 64*64 inputs");
		std::vector<float> y(HT_CNN_OUT);
		check(ctx_, ht_cnn_eval(ctx_, x.data(), y.data(), 1));
		return y;
	}
	void loadb(std::istream &s)                                                              // cnn.h:590
	{
		std::vector<float> w(HT_CNNB_COUNT);
		s.read((char *)w.data(), (std::streamsize)(w.size() * sizeof(float)));
		if ((size_t)s.gcount() != w.size() * sizeof(float)) throw std::runtime_error("CNN::loadb: short .cnnb stream");
		check(ctx_, ht_cnn_load_weights(ctx_, w.data(), w.size()));
	}
	// one SGD step; returns the mean squared error of the forward pass, as cnn.h:558 does
	float Train(const std::vector<float> &x, const std::vector<float> &expected, float alpha = 0.01f)
	{
		if (x.size() != HT_CNN_IN || expected.size() != HT_CNN_OUT) throw std::runtime_error("CNN::Train expects 64*64 inputs and 2304 labels");
		float mse = 0;
		check(ctx_, ht_cnn_train(ctx_, x.data(), expected.data(), 1, alpha, &mse));
		return mse;
	}
	void saveb(std::ostream &s)                                                              // cnn.h:591
	{
		std::vector<float> w(HT_CNNB_COUNT);
		check(ctx_, ht_cnn_get_weights(ctx_, w.data(), w.size()));
		s.write((const char *)w.data(), (std::streamsize)(w.size() * sizeof(float)));
	}
	void saveb(std::string fname) { std::ofstream os(fname, std::ios_base::binary | std::ios_base::out); if (!os.is_open()) throw std::runtime_error("cannot open " + fname); saveb(os); }   // cnn.h:593
	void loadb(std::string fname) { std::ifstream is(fname, std::ios_base::binary | std::ios_base::in); if (!is.is_open()) throw std::runtime_error("cannot open " + fname); loadb(is); }   // cnn.h:592
};

struct HandTracker                                                                            // include/handtrack.h:513-846
{
	// tunables with the reference's names and defaults (handtrack.h:523-547); pushed to the device context before every update
	float segment_scale = 0.17f;
	float full_reset_on_error = 0.6f; bool angles_only = false; bool always_take_cnn = false; float drangey = 0.7f; int boundary_planes = 1;
	float microforce = 1.0f; float cloudforce_max_point = 15.0f; float cloudforce_max_sum = 3000.0f; int mainthreadpasses = 1; int subsample_fraction = 4;
	int subsample_voxel = 0; float subsample_size = 0.0f;                     // handtrack.h:535-536: voxel sub-sampling of the main-thread cloud
	size_t min_point_num = 400; float accum_error_threshold = 0.0f; float min_cray_prob = 0.0f;
	int steps = 5, steps_keypoints = 3, steps_keyangles = 2, steps_palmangle = 2, steps_cloudstart = 1, steps_unibody = 3;
	CNN cnn;
	// The reference's update() overlaps the CNN job with the caller's passes (std::async, handtrack.h:755-768) and is therefore timing dependent; the default here is the
	// deterministic synchronous sum (what every parity statement is made on).  overlapped_update = true gives the reference's structure: the job of frame k runs
	// on a second device context beside the caller's passes and is collected by a later call, when it is through (the 1 ms wait_for = one poll); the caller's latency is
	// the passes alone.  overlapped_wait = true (tests) waits for the job before the passes: the same sequence of operations as the synchronous call.
	bool overlapped_update = false, overlapped_wait = false;
	// PhysModel facade of the two tracked models (handtrack.h:517-518): the members the applications use on them (physmodel.h:295-303,345,433-435)
	struct TrackedModel
	{
		std::vector<Pose> GetPose() const { return read(false); }                                                    // body poses (centre-of-mass frames)
		std::vector<Pose> GetPoseUser() const { return read(true); }                                                 // rig space: RigidBody::PositionUser physics.h:142
		TrackedModel &SetPose(const std::vector<Pose> &poses)                                                        // poses only, momenta untouched (physmodel.h:435)
		{
			std::vector<float> st((size_t)nb_ * HT_STATE);
			check(ctx_, ht_get_state(ctx_, which_, 0, 1, st.data()));
			for (size_t b = 0; b < poses.size() && (int)b < nb_; b++) { float *s = &st[b * HT_STATE]; s[0] = poses[b].position.x; s[1] = poses[b].position.y; s[2] = poses[b].position.z; s[3] = poses[b].orientation.x; s[4] = poses[b].orientation.y; s[5] = poses[b].orientation.z; s[6] = poses[b].orientation.w; }
			check(ctx_, ht_set_state(ctx_, which_, 0, 1, st.data()));
			return *this;
		}
		// physmodel.h:295-303: the subdivision meshes live in rig space (pose = PositionUser, orientation), the hull meshes in the centre-of-mass frames
		std::vector<Mesh> &GetMeshes(int show_subdiv = 0)
		{
			const std::vector<Pose> p = GetPose(), u = GetPoseUser();
			for (size_t b = 0; b < sdmeshes_.size() && b < u.size(); b++) sdmeshes_[b].pose = u[b];
			for (size_t b = 0; b < meshes_.size() && b < p.size(); b++) meshes_[b].pose = p[b];
			return show_subdiv ? sdmeshes_ : meshes_;
		}
		// void FitPointCloud(const std::vector<float3> &points, std::vector<LimitLinear> linears = {}, std::vector<LimitAngular> angulars = {}, float microforce = 1.0f)
		// (physmodel.h:345-356): one fit step of this model against a point cloud under the caller's rows, the cloud rows, the joint rows and the collision rows
		std::vector<RigidBody> rigidbodies;                                                                          // handles for the rows (physmodel.h:253)
		void FitPointCloud(const std::vector<float3> &points, std::vector<LimitLinear> linears = {}, std::vector<LimitAngular> angulars = {}, float microforce = 1.0f)
		{
			std::vector<float> l, a; detail::pack_rows(linears, angulars, l, a);
			const int n = (int)points.size(), nl = (int)linears.size(), na = (int)angulars.size();
			check(ctx_, ht_fit_rows(ctx_, which_, 1, n ? &points[0].x : nullptr, n, &n, l.data(), nl, &nl, a.data(), na, &na, microforce));
		}
		void FitPointCloud(const std::vector<float3> &points, float microforce) { FitPointCloud(points, {}, {}, microforce); }      // round-1 shorthand
	private:
		friend struct HandTracker;
		ht_ctx *ctx_ = nullptr; int which_ = 0, nb_ = 0; std::vector<float3> com_; std::vector<Mesh> meshes_, sdmeshes_;
		std::vector<Pose> read(bool user) const
		{
			std::vector<float> st((size_t)nb_ * HT_STATE);
			check(ctx_, ht_get_state(ctx_, which_, 0, 1, st.data()));
			std::vector<Pose> out(nb_);
			for (int b = 0; b < nb_; b++)
			{
				const float *s = &st[(size_t)b * HT_STATE];
				out[b].position = { s[0], s[1], s[2] }; out[b].orientation = { s[3], s[4], s[5], s[6] };
				if (user && (size_t)b < com_.size()) out[b].position = out[b] * float3{ -com_[b].x, -com_[b].y, -com_[b].z };
			}
			return out;
		}
	} handmodel, othermodel;
	Image<float> cnn_input; std::vector<float> cnn_output;
	// the parts of CNNOutputAnalysis that synthetic-tracker.cpp draws (handtrack.h:186,188; synthetic-tracker.cpp:221-222)
	struct { std::vector<Image<unsigned char>> hmaps; Image<float> vmap; } cnn_output_analysis;

	// Defaults are the reference's hard-coded asset paths (handtrack.h:349,831): the model JSON is built on the host at
	// construction like PhysModel + LoadHandModel do; a model baked with ht_model_bake is accepted too.  Like the reference
	// (handtrack.h:123-126) a missing weight file is not an error at construction, but update() then fails loudly instead
	// of running on random weights.
	explicit HandTracker(const std::string &model_path = "../assets/model_hand.json", const std::string &cnnb_path = "../assets/handposedd.cnnb", int device = 0)
	{
		int rc = ht_create(model_path.c_str(), 1, device, &ctx_);
		if (rc != HT_OK) { std::string msg = ctx_ ? ht_last_error(ctx_) : "ht_create failed"; if (ctx_) ht_destroy(ctx_); ctx_ = nullptr; throw std::runtime_error("HandTracker: " + msg); }
		model_path_ = model_path; device_ = device;
		cnn.ctx_ = ctx_;
		if (!detail::live_ctx()) detail::live_ctx() = ctx_;
		ht_model_info(ctx_, &nb_, nullptr, nullptr);
		{
			// geometry for the facades (GetPoseUser needs the centres of mass, GetMeshes the hulls): the same host build, no device involved
			ht_model *m = nullptr;
			if (ht_model_open(model_path.c_str(), 1, &m) == HT_OK)
				for (int b = 0; b < nb_; b++)
				{
					int nv = 0, nt = 0; float com[3] = { 0, 0, 0 };
					ht_model_body(m, b, &nv, &nt, nullptr, com, nullptr);
					Mesh mesh; std::vector<float> v((size_t)nv * 3); std::vector<int> t((size_t)nt * 3);
					ht_model_body_mesh(m, b, v.data(), t.data());
					for (int i = 0; i < nv; i++) mesh.verts.push_back({ v[3 * i], v[3 * i + 1], v[3 * i + 2] });
					for (int i = 0; i < nt; i++) mesh.tris.push_back({ t[3 * i], t[3 * i + 1], t[3 * i + 2] });
					handmodel.com_.push_back({ com[0], com[1], com[2] }); handmodel.meshes_.push_back(mesh); handmodel.sdmeshes_.push_back(detail::subdivision_mesh(m, b));
				}
			if (m) ht_model_close(m);
			othermodel.com_ = handmodel.com_; othermodel.meshes_ = handmodel.meshes_; othermodel.sdmeshes_ = handmodel.sdmeshes_;
			handmodel.ctx_ = othermodel.ctx_ = ctx_; handmodel.nb_ = othermodel.nb_ = nb_; handmodel.which_ = 0; othermodel.which_ = 1;
			for (int b = 0; b < nb_; b++) { handmodel.rigidbodies.push_back(RigidBody{ ctx_, 0, b }); othermodel.rigidbodies.push_back(RigidBody{ ctx_, 1, b }); }
		}
		if (!cnnb_path.empty()) { std::ifstream is(cnnb_path, std::ios_base::in | std::ios_base::binary); if (is.is_open()) cnn.loadb(is); }
		cnn_output.assign(HT_CNN_OUT, 0.01f);
	}
	~HandTracker() { if (job_) { ht_job_wait(job_); ht_destroy(job_); } if (detail::live_ctx() == ctx_) detail::live_ctx() = nullptr; if (ctx_) ht_destroy(ctx_); }      // the reference's destructor joins the job too (handtrack.h:841-844)
	HandTracker(const HandTracker &) = delete; HandTracker &operator=(const HandTracker &) = delete;

	void load_config(const std::string &jsonfile)                                            // handtrack.h:822-828
	{
		ht_params p; pull_params(p);
		float pfe = 0.0f; float seg = segment_scale; int ini = 0;
		check(ctx_, ht_get_tracker_flags(ctx_, 0, 1, &pfe, &ini));
		const float pfe0 = pfe;
		const int rc = ht_config_read(jsonfile.c_str(), &p, &seg, &pfe);
		if (rc != HT_OK) throw std::runtime_error("json parse error - " + jsonfile);
		segment_scale = seg;
		full_reset_on_error = p.full_reset_on_error; angles_only = p.angles_only != 0; always_take_cnn = p.always_take_cnn != 0; drangey = p.drangey; boundary_planes = p.boundary_planes;
		microforce = p.microforce; cloudforce_max_point = p.cloudforce_max_point; cloudforce_max_sum = p.cloudforce_max_sum; mainthreadpasses = p.mainthreadpasses;
		subsample_fraction = p.subsample_fraction; min_point_num = p.min_point_num; accum_error_threshold = p.accum_error_threshold; min_cray_prob = p.min_cray_prob;
		subsample_voxel = p.subsample_voxel; subsample_size = p.subsample_size;
		steps = p.steps; steps_keypoints = p.steps_keypoints; steps_keyangles = p.steps_keyangles; steps_palmangle = p.steps_palmangle; steps_cloudstart = p.steps_cloudstart; steps_unibody = p.steps_unibody;
		check(ctx_, ht_set_params(ctx_, &p));      // also carries physics_iterations(_post), physics_use_collision, physics_weak_force, bone_sum_error_scale, unibody_force
		if (pfe != pfe0) check(ctx_, ht_set_tracker_flags(ctx_, 0, 1, &pfe, &ini));
	}
	void SetPose(const std::vector<Pose> &pose) { check(ctx_, ht_tracker_reset(ctx_, 0, 1, flat(pose).data())); }          // handmodel/othermodel.SetPose

	std::vector<Pose> update(Image<unsigned short> dimage)                                   // handtrack.h:748
	{
		// A frame that is not 64x64 goes through ht_update_frames_sync, which does what update() does with it (handtrack.h:693-785): the CNN
		// sees HandSegmentVR(dimage, 0xF, {0.1, drangey}, segment_scale), the cloud and FitError come from the full frame.  The segment is
		// fetched once more below, only for the visualisation members (cnn_input and the heat-map camera).
		push_params();
		const bool full = dimage.dim().x != 64 || dimage.dim().y != 64;
		std::vector<float> out((size_t)nb_ * HT_POSE);
		const DCamera &fc = dimage.cam;
		const float fcam[HT_CAM] = { fc.focal().x, fc.focal().y, fc.principal().x, fc.principal().y, fc.depth_scale, fc.pose.position.x, fc.pose.position.y, fc.pose.position.z,
		                             fc.pose.orientation.x, fc.pose.orientation.y, fc.pose.orientation.z, fc.pose.orientation.w };
		if (overlapped_update)
		{
			// handtrack.h:755-768 on two device contexts (include/ht_mi355x.h: ht_job_start / ht_job_poll / ht_job_collect / ht_update_passes_sync)
			ensure_job_context();
			if (!job_in_flight_)
			{
				check(job_, ht_job_start(job_, ctx_, dimage.raster.data(), fcam, dimage.dim().x, dimage.dim().y, segment_scale, 1));
				job_in_flight_ = true; job_image_ = dimage;
			}
			int ready = 0;
			if (overlapped_wait) { check(job_, ht_job_wait(job_)); ready = 1; } else check(job_, ht_job_poll(job_, &ready));
			if (ready)
			{
				check(job_, ht_job_collect(job_, ctx_, 1, nullptr));
				check(job_, ht_get_cnn_results(job_, 0, 1, nullptr, cnn_output.data(), nullptr));
				Image<unsigned short> seen = (job_image_.dim().x != 64 || job_image_.dim().y != 64) ? segment(job_image_, 0xF, { 0.1f, drangey }, segment_scale) : job_image_;
				fill_visualisation(seen);      // cnn_input / cnn_output / the heat-maps of the frame the job saw
				job_in_flight_ = false;
			}
			check(ctx_, ht_update_passes_sync(ctx_, dimage.raster.data(), fcam, dimage.dim().x, dimage.dim().y, 1, out.data()));
		}
		else
		{
			check(ctx_, ht_update_frames_sync(ctx_, dimage.raster.data(), fcam, dimage.dim().x, dimage.dim().y, segment_scale, 1, out.data(), cnn_output.data()));
			if (full) dimage = segment(dimage, 0xF, { 0.1f, drangey }, segment_scale);
			fill_visualisation(dimage);
		}
		std::vector<Pose> pose(nb_);
		for (int b = 0; b < nb_; b++) { const float *p = &out[(size_t)b * HT_POSE]; pose[b].position = { p[0], p[1], p[2] }; pose[b].orientation = { p[3], p[4], p[5], p[6] }; }
		return pose;
	}
	// slowfit (handtrack.h:786-821); the selected bone is given by index (-1: none) instead of a RigidBody pointer
	void slowfit(const std::vector<float3> &points, int hold, const std::vector<Pose> &refpose, int steps_ = 6, int selectrb = -1, float3 spoint = { 0, 0, 0 }, float3 rbpoint = { 0, 0, 0 },
	             const std::vector<float4> &crays = std::vector<float4>())
	{
		push_params();
		const int n = (int)points.size();
		check(ctx_, ht_set_points(ctx_, 1, n ? &points[0].x : &spoint.x, n > 0 ? n : 1, &n));
		std::vector<float> ref = flat(refpose), cr(32, 0.0f);
		const int ncray = crays.size() < 8 ? (int)crays.size() : 8;
		for (int i = 0; i < ncray; i++) { cr[4 * i] = crays[i].x; cr[4 * i + 1] = crays[i].y; cr[4 * i + 2] = crays[i].z; cr[4 * i + 3] = crays[i].w; }
		check(ctx_, ht_slowfit(ctx_, 1, refpose.empty() ? 0 : hold, refpose.empty() ? nullptr : ref.data(), steps_, selectrb, &spoint.x, &rbpoint.x, ncray ? cr.data() : nullptr, ncray));
	}
	float scale(float s) { check(ctx_, ht_scale(ctx_, s)); segment_scale *= s; return segment_scale; }                     // handtrack.h:591
	// update_cnn_model (handtrack.h:734-741): the CNN job alone, synchronously -- othermodel is not re-seeded from handmodel, no main-thread pass;
	// returns othermodel.GetPose() when the tracker would take it over (:720-722) and an empty vector otherwise
	std::vector<Pose> update_cnn_model(Image<unsigned short> dimage) { return cnn_job(std::move(dimage), false); }
	// kickstart (handtrack.h:743-746): handmodel.SetPose(update_cnn_model(dimage))
	void kickstart(Image<unsigned short> dimage) { cnn_job(std::move(dimage), true); }
	// HandSegmentVR (handtrack.h:280-344) on this tracker's device
	Image<unsigned short> segment(const Image<unsigned short> &depth, int entry_options = 0xF, float2 wrange = { 0.1f, 0.65f }, float diam = 0.17f) const { return segment_on(ctx_, depth, entry_options, wrange, diam); }
	static Image<unsigned short> segment_on(ht_ctx *ctx, const Image<unsigned short> &depth, int entry_options, float2 wrange, float diam)
	{
		const DCamera &c = depth.cam;
		float cam[HT_CAM] = { c.focal().x, c.focal().y, c.principal().x, c.principal().y, c.depth_scale, c.pose.position.x, c.pose.position.y, c.pose.position.z,
		                      c.pose.orientation.x, c.pose.orientation.y, c.pose.orientation.z, c.pose.orientation.w };
		float co[HT_CAM]; std::vector<unsigned short> tile(4096);
		check(ctx, ht_segment_vr(ctx, depth.raster.data(), cam, depth.dim().x, depth.dim().y, 1, entry_options, wrange.x, wrange.y, diam, tile.data(), co));
		Pose pose; pose.position = { co[5], co[6], co[7] }; pose.orientation = { co[8], co[9], co[10], co[11] };
		return Image<unsigned short>(DCamera({ 64, 64 }, { co[0], co[1] }, { co[2], co[3] }, co[4], pose), std::move(tile));
	}
private:
	ht_ctx *ctx_ = nullptr; int nb_ = 0;
	ht_ctx *job_ = nullptr; bool job_in_flight_ = false; Image<unsigned short> job_image_; std::string model_path_; int device_ = 0;      // the overlapped mode's second context
	void ensure_job_context()
	{
		if (!job_)
		{
			int rc = ht_create(model_path_.c_str(), 1, device_, &job_);
			if (rc != HT_OK) { std::string msg = job_ ? ht_last_error(job_) : "ht_create failed"; if (job_) ht_destroy(job_); job_ = nullptr; throw std::runtime_error("HandTracker (job context): " + msg); }
			std::vector<float> w(HT_CNNB_COUNT);
			check(ctx_, ht_cnn_get_weights(ctx_, w.data(), w.size()));      // the net the tracker carries (fails loudly when none was loaded)
			check(job_, ht_cnn_load_weights(job_, w.data(), w.size()));
		}
		if (!job_in_flight_) { ht_params p; pull_params(p); check(job_, ht_set_params(job_, &p)); }      // the job runs with the tunables as they stood when it was started
	}
	std::vector<Pose> cnn_job(Image<unsigned short> dimage, bool apply)
	{
		push_params();
		const DCamera &fc = dimage.cam;
		const float fcam[HT_CAM] = { fc.focal().x, fc.focal().y, fc.principal().x, fc.principal().y, fc.depth_scale, fc.pose.position.x, fc.pose.position.y, fc.pose.position.z,
		                             fc.pose.orientation.x, fc.pose.orientation.y, fc.pose.orientation.z, fc.pose.orientation.w };
		std::vector<float> out((size_t)nb_ * HT_POSE); int accepted = 0;
		check(ctx_, ht_update_cnn_model_sync(ctx_, dimage.raster.data(), fcam, dimage.dim().x, dimage.dim().y, segment_scale, 1, apply ? 1 : 0, out.data(), &accepted, cnn_output.data()));
		if (dimage.dim().x != 64 || dimage.dim().y != 64) dimage = segment(dimage, 0xF, { 0.1f, drangey }, segment_scale);
		fill_visualisation(dimage);
		std::vector<Pose> pose;
		if (accepted) { pose.resize(nb_); for (int b = 0; b < nb_; b++) { const float *p = &out[(size_t)b * HT_POSE]; pose[b].position = { p[0], p[1], p[2] }; pose[b].orientation = { p[3], p[4], p[5], p[6] }; } }
		return pose;
	}
	// cnn_input and the drawn parts of cnn_output_analysis, filled on the host from what the device returned (handtrack.h:700, 225, 236-238)
	void fill_visualisation(const Image<unsigned short> &tile)
	{
		const DCamera &c = tile.cam;
		const float dr = drangey - 0.1f;
		cnn_input = Image<float>(c);
		for (size_t i = 0; i < tile.raster.size(); i++) { float v = 1.0f - (tile.raster[i] * c.depth_scale - 0.1f) / dr; cnn_input.raster[i] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
		const DCamera hcam = camsub(c, 4);
		cnn_output_analysis.hmaps.clear();
		for (int m = 0; m < 8; m++)
		{
			Image<unsigned char> h(hcam);
			for (int i = 0; i < 256; i++) { float y = cnn_output[(size_t)256 * m + i] * 255.0f; h.raster[i] = (unsigned char)(y < 0.0f ? 0.0f : (y > 255.0f ? 255.0f : y)); }      // ToGrayScale misc_image.h:169
			cnn_output_analysis.hmaps.push_back(h);
		}
		cnn_output_analysis.vmap = Image<float>(DCamera({ 16, 16 }, { 16.f, 16.f }, { 8.f, 8.f }, c.depth_scale), std::vector<float>(cnn_output.begin() + 2048, cnn_output.end()));
	}
	std::vector<float> flat(const std::vector<Pose> &pose) const
	{
		std::vector<float> f((size_t)nb_ * HT_POSE, 0.f);
		for (int b = 0; b < nb_ && b < (int)pose.size(); b++) { float *p = &f[(size_t)b * HT_POSE]; p[0] = pose[b].position.x; p[1] = pose[b].position.y; p[2] = pose[b].position.z; p[3] = pose[b].orientation.x; p[4] = pose[b].orientation.y; p[5] = pose[b].orientation.z; p[6] = pose[b].orientation.w; }
		return f;
	}
	void pull_params(ht_params &p)
	{
		check(ctx_, ht_get_params(ctx_, &p));
		p.full_reset_on_error = full_reset_on_error; p.angles_only = angles_only; p.always_take_cnn = always_take_cnn; p.drangey = drangey; p.boundary_planes = boundary_planes;
		p.microforce = microforce; p.cloudforce_max_point = cloudforce_max_point; p.cloudforce_max_sum = cloudforce_max_sum; p.mainthreadpasses = mainthreadpasses;
		p.subsample_fraction = subsample_fraction; p.min_point_num = (int)min_point_num; p.accum_error_threshold = accum_error_threshold; p.min_cray_prob = min_cray_prob;
		p.subsample_voxel = subsample_voxel; p.subsample_size = subsample_size;
		p.steps = steps; p.steps_keypoints = steps_keypoints; p.steps_keyangles = steps_keyangles; p.steps_palmangle = steps_palmangle; p.steps_cloudstart = steps_cloudstart; p.steps_unibody = steps_unibody;
	}
	void push_params()
	{
		ht_params p; pull_params(p);
		check(ctx_, ht_set_params(ctx_, &p));
	}
};
// Free function with the reference's signature (handtrack.h:280); it runs on the device of the first live HandTracker.
inline Image<unsigned short> HandSegmentVR(const Image<unsigned short> &depth, int entry_options = 0xF, float2 wrange = { 0.1f, 0.65f }, float diam = 0.17f)
{
	if (!detail::live_ctx()) throw std::runtime_error("HandSegmentVR: construct a HandTracker first (the segmentation runs on its device)");
	return HandTracker::segment_on(detail::live_ctx(), depth, entry_options, wrange, diam);
}

// CNN PoseInitializerCNN(std::string filename) (handtrack.h:103-130): the pose net as an object of its own (own CNN-only device context); like the
// reference, a missing weight file is not an error here -- Eval then fails loudly instead of running on random weights.
inline CNN PoseInitializerCNN(std::string filename, int device = 0)
{
	ht_ctx *ctx = nullptr;
	const int rc = ht_create(nullptr, 1, device, &ctx);
	if (rc != HT_OK) { const std::string msg = ctx ? ht_last_error(ctx) : "ht_create failed"; if (ctx) ht_destroy(ctx); throw std::runtime_error("PoseInitializerCNN: " + msg); }
	CNN cnn; cnn.ctx_ = ctx; cnn.own_ = std::shared_ptr<ht_ctx>(ctx, [](ht_ctx *c) { ht_destroy(c); });
	std::ifstream is(filename, std::ios_base::in | std::ios_base::binary);
	if (is.is_open()) cnn.loadb(is);
	return cnn;
}

// GatherHandExpectedCNN(pose, hcam) (handtrack.h:160-173): the labels the net is trained to produce for a hand pose seen by the 16x16 heat-map camera
struct ExpectedCNN { std::vector<float> cnn_expected; std::vector<float2> image_points; std::vector<Image<unsigned char>> hmaps; Image<unsigned char> vmap; std::vector<float> vals; };
inline ExpectedCNN GatherHandExpectedCNN(const std::vector<Pose> &pose, const DCamera &hcam)
{
	if (pose.size() < 17) throw std::runtime_error("GatherHandExpectedCNN: the landmark table names bones up to 16 (handtrack.h:77-81)");
	std::vector<float> p7(pose.size() * HT_POSE);
	for (size_t b = 0; b < pose.size(); b++) { float *p = &p7[b * HT_POSE]; p[0] = pose[b].position.x; p[1] = pose[b].position.y; p[2] = pose[b].position.z; p[3] = pose[b].orientation.x; p[4] = pose[b].orientation.y; p[5] = pose[b].orientation.z; p[6] = pose[b].orientation.w; }
	// the C-ABI takes the 64x64 tile camera and forms camsub(cam, 4) itself; scaling by 4 and back is exact in binary floating point
	const float cam[HT_CAM] = { hcam.focal().x * 4.0f, hcam.focal().y * 4.0f, hcam.principal().x * 4.0f, hcam.principal().y * 4.0f, hcam.depth_scale,
	                            hcam.pose.position.x, hcam.pose.position.y, hcam.pose.position.z, hcam.pose.orientation.x, hcam.pose.orientation.y, hcam.pose.orientation.z, hcam.pose.orientation.w };
	ExpectedCNN e; e.cnn_expected.resize(HT_CNN_OUT); e.vals.resize(16);
	float ip[16];
	if (ht_expected_cnn_full(p7.data(), cam, e.cnn_expected.data(), ip, e.vals.data()) != HT_OK) throw std::runtime_error("GatherHandExpectedCNN failed");
	for (int k = 0; k < 8; k++) e.image_points.push_back({ ip[2 * k], ip[2 * k + 1] });
	auto gray = [](float v) { v *= 255.0f; return (unsigned char)(v < 0.0f ? 0.0f : v > 255.0f ? 255.0f : v + 0.5f); };      // the labels are bytes / 255: this recovers the byte
	for (int m = 0; m < 8; m++) { Image<unsigned char> h(hcam); for (int i = 0; i < 256; i++) h.raster[i] = gray(e.cnn_expected[(size_t)256 * m + i]); e.hmaps.push_back(h); }
	e.vmap = Image<unsigned char>(DCamera({ 16, 16 }, { 16.f, 16.f }, { 8.f, 8.f }, hcam.depth_scale));
	for (int i = 0; i < 256; i++) e.vmap.raster[i] = gray(e.cnn_expected[2048 + i]);
	return e;
}

// A hand model that is posed, drawn and ray-cast but not tracked (host only): `PhysModel fakehand = LoadHandModel();` (synthetic-tracker.cpp:94)
class PhysModel                                                                               // include/physmodel.h:236-477, the members the applications use
{
	std::shared_ptr<ht_model> m_;
public:
	struct Body { float3 position{ 0, 0, 0 }; float4 orientation{ 0, 0, 0, 1 }; float3 com{ 0, 0, 0 };
		Pose pose() const { Pose p; p.position = position; p.orientation = orientation; return p; }
		float3 PositionUser() const { return pose() * float3{ -com.x, -com.y, -com.z }; } };                          // physics.h:142
	struct ModelHitInfo { bool hit = false; float3 impact{ 0, 0, 0 }, normal{ 0, 0, 0 }; int rb = -1; operator bool() const { return hit; } };      // physmodel.h:280-286
	std::vector<Body> rigidbodies;
	std::vector<Mesh> sdmeshes, meshes;      // physmodel.h:251-252: the subdivision surfaces (rig space) and the collision hulls (centre-of-mass frames)
	explicit PhysModel(const char *jsonfile, bool hand_tweaks = false)
	{
		ht_model *m = nullptr;
		const int rc = ht_model_open(jsonfile, hand_tweaks ? 1 : 0, &m);
		if (rc != HT_OK) { const std::string msg = m ? ht_model_error(m) : "ht_model_open failed"; if (m) ht_model_close(m); throw std::runtime_error("PhysModel: " + msg); }
		m_ = std::shared_ptr<ht_model>(m, [](ht_model *x) { ht_model_close(x); });
		int nb = 0; ht_model_counts(m, &nb, nullptr);
		for (int b = 0; b < nb; b++)
		{
			int nv = 0, nt = 0; float com[3], rest[7];
			ht_model_body(m, b, &nv, &nt, nullptr, com, rest);
			Body rb; rb.position = { rest[0], rest[1], rest[2] }; rb.orientation = { rest[3], rest[4], rest[5], rest[6] }; rb.com = { com[0], com[1], com[2] };
			rigidbodies.push_back(rb);
			Mesh mesh; std::vector<float> v((size_t)nv * 3); std::vector<int> t((size_t)nt * 3);
			ht_model_body_mesh(m, b, v.data(), t.data());
			for (int i = 0; i < nv; i++) mesh.verts.push_back({ v[3 * i], v[3 * i + 1], v[3 * i + 2] });
			for (int i = 0; i < nt; i++) mesh.tris.push_back({ t[3 * i], t[3 * i + 1], t[3 * i + 2] });
			meshes.push_back(mesh);
			sdmeshes.push_back(detail::subdivision_mesh(m, b));
		}
	}
	std::vector<Pose> GetPose() const { std::vector<Pose> p; for (auto &rb : rigidbodies) p.push_back(rb.pose()); return p; }                                            // physmodel.h:433
	std::vector<Pose> GetPoseUser() const { std::vector<Pose> p; for (auto &rb : rigidbodies) { Pose q = rb.pose(); q.position = rb.PositionUser(); p.push_back(q); } return p; }      // :434
	PhysModel &SetPose(const std::vector<Pose> &poses) { for (size_t i = 0; i < poses.size() && i < rigidbodies.size(); i++) { rigidbodies[i].position = poses[i].position; rigidbodies[i].orientation = poses[i].orientation; } return *this; }      // :435
	std::vector<Mesh> &GetMeshes(int show_subdiv = 0)                                                                                                                    // :295-303
	{
		for (size_t i = 0; i < rigidbodies.size(); i++) { Pose u = rigidbodies[i].pose(); u.position = rigidbodies[i].PositionUser(); sdmeshes[i].pose = u; }      // the subdivision meshes are in rig space
		for (size_t i = 0; i < rigidbodies.size(); i++) meshes[i].pose = rigidbodies[i].pose();                                                                   // the hulls in the centre-of-mass frames
		return show_subdiv ? sdmeshes : meshes;
	}
	ModelHitInfo HitCheck(const float3 &v0, const float3 &v1) const                                                                                                      // :287-294
	{
		std::vector<float> p7(rigidbodies.size() * HT_POSE);
		for (size_t b = 0; b < rigidbodies.size(); b++) { const Body &rb = rigidbodies[b]; float *p = &p7[b * HT_POSE]; p[0] = rb.position.x; p[1] = rb.position.y; p[2] = rb.position.z; p[3] = rb.orientation.x; p[4] = rb.orientation.y; p[5] = rb.orientation.z; p[6] = rb.orientation.w; }
		ModelHitInfo h;
		ht_model_hitcheck(m_.get(), p7.data(), &v0.x, &v1.x, &h.impact.x, &h.normal.x, &h.rb);
		h.hit = h.rb >= 0;
		return h;
	}
};
inline PhysModel LoadHandModel(const char *jsonfile = "../assets/model_hand.json") { return PhysModel(jsonfile, true); }                                                    // handtrack.h:347-366
}  // namespace ht_mi355x

#ifdef HT_MI355X_GLOBAL_NAMES      // the reference's headers declare these names at global scope
using ht_mi355x::float2; using ht_mi355x::float3; using ht_mi355x::float4; using ht_mi355x::int2; using ht_mi355x::int3;
using ht_mi355x::Pose; using ht_mi355x::DCamera; using ht_mi355x::Image; using ht_mi355x::Mesh; using ht_mi355x::CNN; using ht_mi355x::HandTracker; using ht_mi355x::PhysModel;
using ht_mi355x::RigidBody; using ht_mi355x::LimitLinear; using ht_mi355x::LimitAngular; using ht_mi355x::PhysicsUpdate; using ht_mi355x::Addresses;
using ht_mi355x::HandSegmentVR; using ht_mi355x::PointCloud; using ht_mi355x::camsub; using ht_mi355x::GatherHandExpectedCNN; using ht_mi355x::PoseInitializerCNN; using ht_mi355x::LoadHandModel;
#endif
