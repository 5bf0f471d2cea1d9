// ht_handtrack.hpp -- C++ wrappers over the C-ABI (ht_mi355x.h) that keep the reference's own names and signatures, so that
// code written against include/handtrack.h / third_party/cnn.h of IntelRealSense/hand_tracking_samples keeps compiling:
//
//     HandTracker htk;                                   // handtrack.h:830 (paths are constructor arguments here)
//     htk.microforce = 3.0f; htk.mainthreadpasses = 3;   // synthetic-tracker.cpp:91-93
//     std::vector<Pose> pose = htk.update(std::move(dimage));      // handtrack.h:748
//     std::vector<float> y = htk.cnn.Eval(x);                        // cnn.h:550
//
// Only the members the per-frame path and synthetic-tracker.cpp touch are provided.  Documented deviation: update()
// runs the CNN job synchronously every frame (the reference polls a background std::async job for 1 ms, which makes its output
// timing dependent, handtrack.h:755-768); this is HandTracker::update_cnn_model followed by the main-thread passes.
// Errors are reported the way the reference's apps expect them: by throwing std::runtime_error (synthetic-tracker.cpp:255-264).
#pragma once
#include <cstdint>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>
#include "ht_mi355x.h"

namespace ht_mi355x
{
struct float2 { float x, y; };
struct int2 { int x, y; };
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
	Image(DCamera cam) : cam(cam), raster((size_t)cam.dim().x * cam.dim().y, T(0)) {}
	Image(DCamera cam, std::vector<T> data) : cam(cam), raster(std::move(data)) {}
	const int2 dim() const { return cam.dim(); }
	T &pixel(int2 p) { return raster[(size_t)p.y * dim().x + p.x]; }
};

namespace detail { inline ht_ctx *&live_ctx() { static ht_ctx *c = nullptr; return c; } }      // a context the free functions below can run on
inline void check(ht_ctx *ctx, int rc) { if (rc != HT_OK) throw std::runtime_error(std::string("ht_mi355x: ") + (ctx ? ht_last_error(ctx) : "no context")); }

class CNN                                                                                    // third_party/cnn.h:100-605 (Eval / Train / loadb / saveb)
{
	ht_ctx *ctx_ = nullptr;
	friend struct HandTracker;
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
	size_t min_point_num = 400; float accum_error_threshold = 0.0f; float min_cray_prob = 0.0f;
	int steps = 5, steps_keypoints = 3, steps_keyangles = 2, steps_palmangle = 2, steps_cloudstart = 1, steps_unibody = 3;
	CNN cnn;
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
		cnn.ctx_ = ctx_;
		if (!detail::live_ctx()) detail::live_ctx() = ctx_;
		ht_model_info(ctx_, &nb_, nullptr, nullptr);
		if (!cnnb_path.empty()) { std::ifstream is(cnnb_path, std::ios_base::in | std::ios_base::binary); if (is.is_open()) cnn.loadb(is); }
		cnn_output.assign(HT_CNN_OUT, 0.01f);
	}
	~HandTracker() { if (detail::live_ctx() == ctx_) detail::live_ctx() = nullptr; if (ctx_) ht_destroy(ctx_); }
	HandTracker(const HandTracker &) = delete; HandTracker &operator=(const HandTracker &) = delete;

	void load_config(const std::string &jsonfile)                                            // handtrack.h:822-828
	{
		ht_params p; pull_params(p);
		float pfe = 0.0f; float seg = segment_scale; int ini = 0;
		check(ctx_, ht_get_tracker_flags(ctx_, 0, 1, &pfe, &ini));
		const float pfe0 = pfe;
		const int rc = ht_config_read(jsonfile.c_str(), &p, &seg, &pfe);
		if (rc == HT_ERR_ARG) throw std::runtime_error("unsupported option (subsample_voxel) in " + jsonfile);
		if (rc != HT_OK) throw std::runtime_error("json parse error - " + jsonfile);
		segment_scale = seg;
		full_reset_on_error = p.full_reset_on_error; angles_only = p.angles_only != 0; always_take_cnn = p.always_take_cnn != 0; drangey = p.drangey; boundary_planes = p.boundary_planes;
		microforce = p.microforce; cloudforce_max_point = p.cloudforce_max_point; cloudforce_max_sum = p.cloudforce_max_sum; mainthreadpasses = p.mainthreadpasses;
		subsample_fraction = p.subsample_fraction; min_point_num = p.min_point_num; accum_error_threshold = p.accum_error_threshold; min_cray_prob = p.min_cray_prob;
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
		{
			const DCamera &fc = dimage.cam;
			float fcam[HT_CAM] = { fc.focal().x, fc.focal().y, fc.principal().x, fc.principal().y, fc.depth_scale, fc.pose.position.x, fc.pose.position.y, fc.pose.position.z,
			                       fc.pose.orientation.x, fc.pose.orientation.y, fc.pose.orientation.z, fc.pose.orientation.w };
			check(ctx_, ht_update_frames_sync(ctx_, dimage.raster.data(), fcam, dimage.dim().x, dimage.dim().y, segment_scale, 1, out.data(), cnn_output.data()));
		}
		if (full) dimage = segment(dimage, 0xF, { 0.1f, drangey }, segment_scale);
		const DCamera &c = dimage.cam;
		// visualisation members, filled on the host from what the device returned (handtrack.h:700, 225, 236-238)
		const float dr = drangey - 0.1f;
		cnn_input = Image<float>(c);
		for (size_t i = 0; i < dimage.raster.size(); i++) { float v = 1.0f - (dimage.raster[i] * c.depth_scale - 0.1f) / dr; cnn_input.raster[i] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
		DCamera hcam({ 16, 16 }, { c.focal().x / 4.0f, c.focal().y / 4.0f }, { c.principal().x / 4.0f, c.principal().y / 4.0f }, c.depth_scale, c.pose);
		cnn_output_analysis.hmaps.clear();
		for (int m = 0; m < 8; m++)
		{
			Image<unsigned char> h(hcam);
			for (int i = 0; i < 256; i++) { float y = cnn_output[(size_t)256 * m + i] * 255.0f; h.raster[i] = (unsigned char)(y < 0.0f ? 0.0f : (y > 255.0f ? 255.0f : y)); }      // ToGrayScale misc_image.h:169
			cnn_output_analysis.hmaps.push_back(h);
		}
		cnn_output_analysis.vmap = Image<float>(DCamera({ 16, 16 }, { 16.f, 16.f }, { 8.f, 8.f }, c.depth_scale), std::vector<float>(cnn_output.begin() + 2048, cnn_output.end()));
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
	// kickstart (handtrack.h:743-746): the CNN job in the calling thread, its pose taken over when it is accepted; no main-thread passes
	void kickstart(Image<unsigned short> dimage) { const int keep = mainthreadpasses; mainthreadpasses = 0; try { update(std::move(dimage)); } catch (...) { mainthreadpasses = keep; throw; } mainthreadpasses = keep; }
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
}  // namespace ht_mi355x
