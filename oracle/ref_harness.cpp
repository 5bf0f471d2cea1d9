//
// ref_harness.cpp -- drives the *reference's own code* (compiled from /root/reference where it lies)
// to produce golden fixtures and reference CPU timings.
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/ref_harness (git-ignored).
// Nothing in the product path links, imports or executes this file or its binary.
// No reference source is copied here: the reference headers are #included from /root/reference
// and only their public functions are called.  The two app-level helpers the reference keeps in
// synthetic-hand-tracker/synthetic-tracker.cpp (an animation-bank reader and the software depth
// rasteriser, :39-55 and :69-76) are not reachable through a header, so this file has its own
// small equivalents built on the reference's PhysModel::HitCheck / DCamera API.
//
// The reference hard-codes "../assets/model_hand.json" etc. (handtrack.h:349,831-832).  The two
// small JSON assets are embedded into this binary at build time (.incbin from /root/reference) and
// written to a temporary directory at start-up, so the binary also runs on the GPU box where
// /root/reference does not exist.
//
// Modes:
//   model  <out.htfx>                              baked 17-bone model (what PhysModel/LoadHandModel build)
//   modelfile <model.json> <out.htfx>              any PhysModel JSON (absolute path) as PhysModel(const char*) builds it
//   scan   <animbank.pose> <stride>                workload statistics per animation row
//   frames <animbank.pose> <first> <stride> <n> <out.htfx>    64x64 frames + cameras + start poses
//   golden <animbank.pose> <rows,comma> <seed> <fc2gain> <out.htfx>   per-stage goldens
//   scale  <animbank.pose> <rows,comma> <seed> <fc2gain> <s> <out.htfx>   HandTracker::scale(s): scaled model (<out>.model) + unit of work
//   train  <animbank.pose> <rows,comma> <seed> <fc2gain> <epochs> <out.htfx>   CNN::Train on the frames' inputs and GatherHandExpectedCNN labels
//   slowfit <animbank.pose> <rows,comma> <out.htfx>   HandTracker::slowfit with several argument sets
//   segment <animbank.pose> <rows,comma> <out.htfx>   320x240 frames and what HandSegmentVR makes of them
//   bench  <frames.htfx> <seed> <fc2gain> <reps>   reference CPU time for the unit of work
//   poses  <frames.htfx> <seed> <fc2gain> <out.htfx>   the unit of work on every frame of the file: user poses, othermodel poses, tracker flags
//   viz <animbank.pose> <row> <out.htfx>         DepthMesh and VisualizeHMaps as the application calls them (synthetic-tracker.cpp:191,204-209)
//   posesfull <frames.htfx> <seed> <fc2gain> <out.htfx>   the same for full-size frames of any size (the tracker segments them) and the staged hand model
//   dataset_write <dir/> <name>                    DepthDataStreamOut (dataset.h:62-106) writes a three-frame set <dir>/<name>.{json,rs,ir,pose,rgb,feye}
//   dataset_read <prefix> <bones> <out.htfx>       load_dataset (dataset.h:109-163) reads <prefix>.* ; everything it returns
//   dataset_header <x.json> <x.pose> <bones> <out.htfx>   DatasetInfo as from_json decodes it and the poses the reference's stream operator reads
//   cnn128 <frames128.htfx> <idx,comma> <seed> <fc2gain> <out.htfx>   the layer list of PoseInitializerCNN on a 128x128 input (SURVEY 8d config 5 ii),
//                                                  assembled from the reference's own layer classes, evaluated on frames of a `fullframes` file
//   e2e128 <frames128.htfx> <idx,comma> <seed> <fc2gain> <out.htfx>   BASELINE configs[4] end to end (SURVEY 8d config 5 i-iii): that net, its decode with
//                                                  camsub(cam, 8) and the tracker's stages called directly on the 128x128 frame, no segmentation
//   bench128 <frames128.htfx> <seed> <fc2gain> <reps> [maxframes]    reference CPU time of that unit of work
//
#include "/root/reference/include/handtrack.h"
#include "/root/reference/include/dataset.h"
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include "htfx.h"

// ---- embedded assets (build artefact only; see oracle/Makefile) -------------------------------
__asm__(
	".section .rodata\n"
	".global ht_asset_model\n ht_asset_model:\n .incbin \"/root/reference/assets/model_hand.json\"\n"
	".global ht_asset_model_end\n ht_asset_model_end:\n .byte 0\n"
	".global ht_asset_vanity\n ht_asset_vanity:\n .incbin \"/root/reference/assets/vanity_bones.json\"\n"
	".global ht_asset_vanity_end\n ht_asset_vanity_end:\n .byte 0\n"
	".text\n");
extern "C" const char ht_asset_model[], ht_asset_model_end[], ht_asset_vanity[], ht_asset_vanity_end[];

static void stage_assets()
{
	char tmpl[] = "/tmp/htref_XXXXXX";
	char *d = mkdtemp(tmpl);
	if (!d) { perror("mkdtemp"); exit(2); }
	std::string root(d);
	mkdir((root + "/assets").c_str(), 0700);
	mkdir((root + "/run").c_str(), 0700);
	if (const char *alt = getenv("HT_REF_MODEL_JSON"))      // another hand model in the reference's schema (tests/golden/make_model_hand26.py); LoadHandModel reads ../assets/model_hand.json
	{
		std::ifstream in(alt, std::ios::binary);
		if (!in.is_open()) { fprintf(stderr, "cannot open %s\n", alt); exit(2); }
		std::ofstream o(root + "/assets/model_hand.json", std::ios::binary); o << in.rdbuf();
	}
	else { std::ofstream o(root + "/assets/model_hand.json", std::ios::binary); o.write(ht_asset_model, ht_asset_model_end - ht_asset_model); }
	{ std::ofstream o(root + "/assets/vanity_bones.json", std::ios::binary); o.write(ht_asset_vanity, ht_asset_vanity_end - ht_asset_vanity); }
	if (chdir((root + "/run").c_str())) { perror("chdir"); exit(2); }
}

// ---- seeded synthetic weights (our own generator; mirrored in hand_tracking_samples_amd/weights.py) ----
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t i)
{
	uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
static std::vector<float> make_cnnb(uint64_t seed, double fc2gain, size_t feat = 2304)
{
	// .cnnb layout (cnn.h:288,454,590): conv1 W[400] B[16]; conv2 W[16384] B[64]; fc1 W[feat*2048] B[2048]; fc2 W[2048*2304] B[2304]; feat = 2304 for the 64x64 net
	struct L { size_t nw, nb; double fan; double gain; } layers[4] = {
		{ 400, 16, 25.0 * 1 + 25.0 * 16, 1.0 }, { 16384, 64, 16.0 * 16 + 16.0 * 64, 1.0 },
		{ feat * 2048, 2048, (double)feat + 2048.0, 1.0 }, { (size_t)2048 * 2304, 2304, 2048.0 + 2304.0, fc2gain } };
	std::vector<float> out;
	uint64_t ctr = 0;
	for (auto &l : layers)
	{
		double range = std::sqrt(6.0 / l.fan) * l.gain;
		for (size_t i = 0; i < l.nw; i++) { double u = (double)(splitmix64_at(seed, ctr++) >> 11) * (1.0 / 9007199254740992.0); out.push_back((float)((2.0 * u - 1.0) * range)); }
		for (size_t i = 0; i < l.nb; i++) { double u = (double)(splitmix64_at(seed, ctr++) >> 11) * (1.0 / 9007199254740992.0); out.push_back((float)((2.0 * u - 1.0) * 0.05 * l.gain)); }
	}
	return out;
}
static void load_weights(HandTracker &htk, uint64_t seed, double fc2gain)
{
	auto w = make_cnnb(seed, fc2gain);
	std::string s((const char*)w.data(), w.size() * sizeof(float));
	std::istringstream is(s, std::ios::binary);
	htk.cnn.loadb(is);
}

// ---- helpers ------------------------------------------------------------------------------------
static std::vector<std::vector<Pose>> read_animbank(const char *fn, size_t nb)
{
	std::vector<std::vector<Pose>> bank;
	FILE *f = fopen(fn, "r");
	if (!f) { fprintf(stderr, "cannot open %s\n", fn); exit(2); }
	for (;;)
	{
		std::vector<Pose> p(nb);
		bool ok = true;
		for (auto &q : p)
			if (fscanf(f, "%f %f %f %f %f %f %f", &q.position.x, &q.position.y, &q.position.z, &q.orientation.x, &q.orientation.y, &q.orientation.z, &q.orientation.w) != 7) { ok = false; break; }
		if (!ok) break;
		bank.push_back(p);
	}
	fclose(f);
	return bank;
}
static Image<unsigned short> raycast_depth(PhysModel &model, const DCamera &cam)   // software rasteriser via PhysModel::HitCheck
{
	Image<unsigned short> depth(cam);
	depth.cam.depth_scale = cam.depth_scale;
	for (int y = 0; y < cam.dim().y; y++) for (int x = 0; x < cam.dim().x; x++)
	{
		auto h = model.HitCheck(float3(0, 0, 0), depth.cam.deprojectz(float2((float)x, (float)y), 4.0f));
		depth.pixel(int2(x, y)) = (unsigned short)(h.impact.z / depth.cam.depth_scale);
	}
	return depth;
}
struct TileFrame { Image<unsigned short> seg; std::vector<Pose> start, gt; };
static TileFrame make_frame(PhysModel &fake, const std::vector<std::vector<Pose>> &bank, size_t k)
{
	DCamera dcam({ 320,240 }, { 305,305 }, { 160,120 }, 0.001f);
	fake.SetPose(bank[k % bank.size()]);
	auto depth = raycast_depth(fake, dcam);
	TileFrame fr;
	fr.seg = HandSegmentVR(depth, 0xF, { 0.1f,0.70f });
	fr.gt = bank[k % bank.size()];
	fr.start = bank[(k + 1) % bank.size()];
	Pose inv = fr.seg.cam.pose.inverse();
	for (auto &p : fr.gt) p = inv * p;
	for (auto &p : fr.start) p = inv * p;
	fr.seg.cam.pose = Pose();
	return fr;
}
static std::vector<float> flat(const std::vector<Pose> &p) { std::vector<float> o; for (auto &q : p) { for (int i = 0; i < 3; i++) o.push_back(q.position[i]); for (int i = 0; i < 4; i++) o.push_back(q.orientation[i]); } return o; }
static std::vector<float> camvec(const DCamera &c) { return { c.focal().x, c.focal().y, c.principal().x, c.principal().y, c.depth_scale, c.pose.position.x, c.pose.position.y, c.pose.position.z, c.pose.orientation.x, c.pose.orientation.y, c.pose.orientation.z, c.pose.orientation.w }; }
static int rbidx(PhysModel &m, const RigidBody *rb) { return rb ? (int)(rb - m.rigidbodies.data()) : -1; }

struct Out
{
	htfx_writer w;
	void f32(const std::string &n, const std::vector<float> &v, std::vector<uint32_t> dims = {}) { if (dims.empty()) dims = { (uint32_t)v.size() }; htfx_put(&w, n.c_str(), HTFX_F32, (uint32_t)dims.size(), dims.data(), v.data()); }
	void i32(const std::string &n, const std::vector<int> &v, std::vector<uint32_t> dims = {}) { if (dims.empty()) dims = { (uint32_t)v.size() }; htfx_put(&w, n.c_str(), HTFX_I32, (uint32_t)dims.size(), dims.data(), v.data()); }
	void u16(const std::string &n, const std::vector<unsigned short> &v, std::vector<uint32_t> dims = {}) { if (dims.empty()) dims = { (uint32_t)v.size() }; htfx_put(&w, n.c_str(), HTFX_U16, (uint32_t)dims.size(), dims.data(), v.data()); }
	void v3(const std::string &n, const std::vector<float3> &v) { std::vector<float> o; for (auto &a : v) { o.push_back(a.x); o.push_back(a.y); o.push_back(a.z); } f32(n, o, { (uint32_t)v.size(), 3 }); }
	void state(const std::string &n, PhysModel &m)   // [nb,13] pos3 quat4 linmom3 angmom3
	{
		std::vector<float> o;
		for (auto &rb : m.rigidbodies) { for (int i = 0; i < 3; i++) o.push_back(rb.position[i]); for (int i = 0; i < 4; i++) o.push_back(rb.orientation[i]); for (int i = 0; i < 3; i++) o.push_back(rb.linear_momentum[i]); for (int i = 0; i < 3; i++) o.push_back(rb.angular_momentum[i]); }
		f32(n, o, { (uint32_t)m.rigidbodies.size(), 13 });
	}
	void linears(const std::string &n, PhysModel &m, const std::vector<LimitLinear> &L)   // [n,16]
	{
		std::vector<float> o;
		for (auto &c : L)
		{
			o.push_back((float)rbidx(m, c.rb0)); o.push_back((float)rbidx(m, c.rb1));
			for (int i = 0; i < 3; i++) o.push_back(c.position0[i]); for (int i = 0; i < 3; i++) o.push_back(c.position1[i]); for (int i = 0; i < 3; i++) o.push_back(c.normal[i]);
			o.push_back(c.targetdist); o.push_back(c.targetspeednobias); o.push_back(c.forcelimit.x); o.push_back(c.forcelimit.y); o.push_back((float)c.friction_master);
		}
		f32(n, o, { (uint32_t)L.size(), 16 });
	}
	void angulars(const std::string &n, PhysModel &m, const std::vector<LimitAngular> &A)   // [n,8]
	{
		std::vector<float> o;
		for (auto &a : A) { o.push_back((float)rbidx(m, a.rb0)); o.push_back((float)rbidx(m, a.rb1)); for (int i = 0; i < 3; i++) o.push_back(a.axis[i]); o.push_back(a.targetspin); o.push_back(a.mintorque); o.push_back(a.maxtorque); }
		f32(n, o, { (uint32_t)A.size(), 8 });
	}
};
static void zero_momenta(PhysModel &m) { for (auto &rb : m.rigidbodies) rb.linear_momentum = rb.angular_momentum = float3(0, 0, 0); }
static void reset_tracker(HandTracker &htk, const std::vector<Pose> &start)
{
	htk.handmodel.SetPose(start); htk.othermodel.SetPose(start);
	zero_momenta(htk.handmodel); zero_momenta(htk.othermodel);
	htk.prev_frame_error = 0.0f; htk.initializing = 0;
}

// the deterministic unit of work: update_cnn_model + mainthreadpasses passes as HandTracker::update runs them (handtrack.h:748-785)
static std::vector<Pose> unit_of_work(HandTracker &htk, const Image<unsigned short> &seg, Out *out = NULL, const std::string &pre = "")
{
	auto points = takesubsample(PointCloud(seg, { 0.1f,htk.drangey }), htk.subsample_fraction, htk.subsample_voxel, htk.subsample_size);
	htk.othermodel.SetPose(htk.handmodel.GetPose());
	auto pose = htk.update_cnn_model(seg);
	if (out)
	{
		out->state(pre + "uw_other_after_cnn", htk.othermodel);
		out->f32(pre + "uw_accept", { (float)pose.size(), htk.prev_frame_error, (float)htk.initializing });
	}
	htk.handmodel.SetPose(pose);
	for (int i = 0; !htk.angles_only && i < htk.mainthreadpasses; i++)
	{
		std::vector<LimitLinear> linears; std::vector<LimitAngular> angulars;
		HandModelEnhancements(htk.handmodel, angulars, false, float3(0, 0, 0), float3(0, 0, 0), 0);
		if (points.size() > htk.min_point_num && htk.boundary_planes)
		{
			std::vector<float3> outdirs = { float3(-1, -0.25f, 0), float3(-1, -1, 0), float3(0, -1, 0), float3(1, -1, 0), float3(1, -0.25f, 0) };
			Append(linears, cloud_chamber(htk.handmodel, points, outdirs, { 0,0,0 }, { 0,0,1 }, 10.0f));
		}
		htk.handmodel.FitPointCloud(points, linears, angulars, htk.microforce);
		if (out) out->state(pre + "uw_hand_pass" + std::to_string(i), htk.handmodel);
	}
	if (points.size() < htk.min_point_num) htk.initializing = 50;
	return htk.handmodel.GetPoseUser();
}

// poses of the 26-bone hand of tests/golden/make_model_hand26.py from 17-bone poses: clone bone = source bone shifted by R(palm)*(0, 0.03, 0)
static void extend_bank(std::vector<std::vector<Pose>> &bank, size_t nb)
{
	static const int src[9] = { 5, 6, 7, 8, 9, 10, 11, 12, 13 };
	if (nb != 26) return;
	for (auto &row : bank)
	{
		const float3 off = qrot(row[1].orientation, float3(0, 0.03f, 0));
		for (int i = 0; i < 9; i++) row.push_back(Pose(row[src[i]].position + off, row[src[i]].orientation));
	}
}
// BASELINE configs[4]: 128x128 frames of whatever hand model is staged (the 26-bone one), through the application's own sequence
// (synthetic-tracker.cpp:204-215: HandSegmentVR, poses re-based into the segment camera, then the tracker on the 64x64 tile)
static int mode_config5(const char *bankfn, const char *rowscsv, uint64_t seed, double gain, const char *outfn)
{
	HandTracker htk;
	htk.microforce = 3.0f; htk.mainthreadpasses = 3; htk.always_take_cnn = 0;
	load_weights(htk, seed, gain);
	PhysModel fake = LoadHandModel();
	const size_t nb = fake.rigidbodies.size();
	auto bank = read_animbank(bankfn, 17);
	extend_bank(bank, nb);
	std::vector<int> rows; { std::stringstream ss(rowscsv); std::string t; while (std::getline(ss, t, ',')) rows.push_back(atoi(t.c_str())); }
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	o.i32("rows", rows); o.f32("weights_seed_gain", { (float)seed, (float)gain });
	DCamera dcam({ 128,128 }, { 163,163 }, { 64,64 }, 0.001f);
	for (size_t fi = 0; fi < rows.size(); fi++)
	{
		std::string pre = "f" + std::to_string(fi) + "/";
		const size_t k = (size_t)rows[fi];
		fake.SetPose(bank[k % bank.size()]);
		auto depth = raycast_depth(fake, dcam);
		auto seg = HandSegmentVR(depth, 0xF, { 0.1f, htk.drangey }, htk.segment_scale);      // arguments of handtrack.h:697-698
		std::vector<Pose> gt = bank[k % bank.size()], start = bank[(k + 1) % bank.size()];
		o.u16(pre + "depth128", depth.raster, { 128,128 }); o.f32(pre + "cam128", camvec(depth.cam));
		o.u16(pre + "tile", seg.raster, { 64,64 }); o.f32(pre + "segcam", camvec(seg.cam));
		o.f32(pre + "startpose_cam", flat(start), { (uint32_t)nb,7 });
		Pose inv = seg.cam.pose.inverse();
		for (auto &p : gt) p = inv * p;
		for (auto &p : start) p = inv * p;
		seg.cam.pose = Pose();
		o.f32(pre + "cam", camvec(seg.cam)); o.f32(pre + "startpose", flat(start), { (uint32_t)nb,7 }); o.f32(pre + "gtpose", flat(gt), { (uint32_t)nb,7 });
		reset_tracker(htk, start);
		auto points = takesubsample(PointCloud(seg, { 0.1f,htk.drangey }), htk.subsample_fraction, htk.subsample_voxel, htk.subsample_size);
		auto pose = unit_of_work(htk, seg, &o, pre);
		o.f32(pre + "cnn_output", htk.cnn_output);
		o.f32(pre + "uw_pose_user", flat(pose), { (uint32_t)nb,7 });
		o.state(pre + "uw_other_final", htk.othermodel);
		// contact load of the final pose (what the extra fingers add)
		{
			std::vector<PhysContact> C; FindShapeShapeContacts(C, Addresses(htk.handmodel.rigidbodies));     // physics.h:451-462
			o.i32(pre + "ncontacts_final", { (int)C.size(), (int)points.size() });
		}
		printf("config5 frame %d row %d P=%d\n", (int)fi, rows[fi], (int)points.size()); fflush(stdout);
	}
	htfx_close(&o.w);
	return 0;
}

// HandTracker::update on frames that are not 64x64 (handtrack.h:693-785): the tracker segments the frame for the CNN itself and fits the
// full-resolution cloud.  Camera `camspec` = "w,h,focal"; the staged hand model may be the 17- or the 26-bone one.
static int mode_fullframe(const char *bankfn, const char *rowscsv, const char *camspec, uint64_t seed, double gain, const char *outfn)
{
	HandTracker htk;
	htk.microforce = 3.0f; htk.mainthreadpasses = 3; htk.always_take_cnn = 0;
	load_weights(htk, seed, gain);
	PhysModel fake = LoadHandModel();
	const size_t nb = fake.rigidbodies.size();
	auto bank = read_animbank(bankfn, 17);
	extend_bank(bank, nb);
	std::vector<int> rows; { std::stringstream ss(rowscsv); std::string t; while (std::getline(ss, t, ',')) rows.push_back(atoi(t.c_str())); }
	int w = 128, h = 128; float focal = 163; sscanf(camspec, "%d,%d,%f", &w, &h, &focal);
	DCamera dcam({ w,h }, { focal,focal }, { w * 0.5f, h * 0.5f }, 0.001f);
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	o.i32("rows", rows); o.f32("weights_seed_gain", { (float)seed, (float)gain }); o.i32("dims", { w, h });
	for (size_t fi = 0; fi < rows.size(); fi++)
	{
		std::string pre = "f" + std::to_string(fi) + "/";
		const size_t k = (size_t)rows[fi];
		fake.SetPose(bank[k % bank.size()]);
		auto depth = raycast_depth(fake, dcam);
		const std::vector<Pose> &start = bank[(k + 1) % bank.size()];
		o.u16(pre + "depth", depth.raster, { (uint32_t)h,(uint32_t)w }); o.f32(pre + "cam", camvec(depth.cam));
		o.f32(pre + "startpose", flat(start), { (uint32_t)nb,7 }); o.f32(pre + "gtpose", flat(bank[k % bank.size()]), { (uint32_t)nb,7 });
		reset_tracker(htk, start);
		auto points = takesubsample(PointCloud(depth, { 0.1f,htk.drangey }), htk.subsample_fraction, htk.subsample_voxel, htk.subsample_size);
		auto pose = unit_of_work(htk, depth, &o, pre);
		o.f32(pre + "cnn_input", htk.cnn_input.raster); o.f32(pre + "cnn_output", htk.cnn_output);
		o.f32(pre + "uw_pose_user", flat(pose), { (uint32_t)nb,7 });
		o.f32(pre + "uw_final", { htk.prev_frame_error, (float)htk.initializing, (float)points.size() });
		// the same frame again with the carried state (streaming use)
		auto pose2 = unit_of_work(htk, depth);
		o.f32(pre + "uw2_pose_user", flat(pose2), { (uint32_t)nb,7 }); o.state(pre + "uw2_hand", htk.handmodel);
		printf("fullframe %dx%d frame %d row %d P=%d accepted %d\n", w, h, (int)fi, rows[fi], (int)points.size(), 0); fflush(stdout);
	}
	htfx_close(&o.w);
	return 0;
}

// bench input for full-size frames: n frames of the staged model seen by camera `camspec` ("w,h,focal"), start pose = the next bank row
static int mode_fullframes(const char *bankfn, int first, int stride, int n, const char *camspec, const char *outfn)
{
	PhysModel fake = LoadHandModel();
	const size_t nb = fake.rigidbodies.size();
	auto bank = read_animbank(bankfn, 17);
	extend_bank(bank, nb);
	int w = 128, h = 128; float focal = 163; sscanf(camspec, "%d,%d,%f", &w, &h, &focal);
	DCamera dcam({ w,h }, { focal,focal }, { w * 0.5f, h * 0.5f }, 0.001f);
	std::vector<unsigned short> depth; std::vector<float> cams, start; std::vector<int> rows;
	for (int i = 0; i < n; i++)
	{
		size_t k = (size_t)(first + (long)i * stride) % bank.size();
		fake.SetPose(bank[k]);
		auto d = raycast_depth(fake, dcam);
		depth.insert(depth.end(), d.raster.begin(), d.raster.end());
		auto c = camvec(d.cam); cams.insert(cams.end(), c.begin(), c.end());
		auto s = flat(bank[(k + 1) % bank.size()]); start.insert(start.end(), s.begin(), s.end());
		rows.push_back((int)k);
		if (i % 16 == 0) { printf("frame %d/%d row %d\n", i, n, (int)k); fflush(stdout); }
	}
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	o.u16("depth", depth, { (uint32_t)n, (uint32_t)h, (uint32_t)w }); o.f32("cam", cams, { (uint32_t)n, 12 });
	o.f32("startpose", start, { (uint32_t)n, (uint32_t)nb, 7 }); o.i32("rows", rows);
	htfx_close(&o.w);
	return 0;
}

// HandTracker::update with the voxel sub-sampling of the main-thread cloud switched on (handtrack.h:535-536,751; physmodel.h:66-118): the hash-table
// averaging of the in-range points into voxels of `size` metres, voxels with fewer than subsample_fraction points dropped.  The CNN job keeps its
// every-4th-point cloud (handtrack.h:703).  Dumps the voxel cloud itself and the unit of work on it.
static int mode_voxel(const char *bankfn, const char *rowscsv, uint64_t seed, double gain, double size, int min_point_num, const char *outfn)
{
	HandTracker htk;
	htk.microforce = 3.0f; htk.mainthreadpasses = 3; htk.always_take_cnn = 0;
	htk.subsample_voxel = 1; htk.subsample_size = (float)size; htk.min_point_num = min_point_num;
	load_weights(htk, seed, gain);
	PhysModel fake = LoadHandModel();
	auto bank = read_animbank(bankfn, fake.rigidbodies.size());
	std::vector<int> rows; { std::stringstream ss(rowscsv); std::string t; while (std::getline(ss, t, ',')) rows.push_back(atoi(t.c_str())); }
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	o.i32("rows", rows); o.f32("weights_seed_gain", { (float)seed, (float)gain }); o.f32("voxel", { (float)size, (float)htk.subsample_fraction, (float)min_point_num });
	for (size_t fi = 0; fi < rows.size(); fi++)
	{
		std::string pre = "f" + std::to_string(fi) + "/";
		TileFrame fr = make_frame(fake, bank, rows[fi]);
		o.u16(pre + "depth", fr.seg.raster, { 64,64 }); o.f32(pre + "cam", camvec(fr.seg.cam)); o.f32(pre + "startpose", flat(fr.start), { 17,7 });
		auto vox = takesubsample(PointCloud(fr.seg, { 0.1f,htk.drangey }), htk.subsample_fraction, htk.subsample_voxel, htk.subsample_size);
		std::vector<float> vp; for (auto &p : vox) { vp.push_back(p.x); vp.push_back(p.y); vp.push_back(p.z); }
		o.f32(pre + "voxel_points", vp, { (uint32_t)vox.size(), 3 });
		reset_tracker(htk, fr.start);
		auto pose = unit_of_work(htk, fr.seg, &o, pre);
		o.f32(pre + "cnn_output", htk.cnn_output);
		o.f32(pre + "uw_pose_user", flat(pose), { 17,7 });
		o.f32(pre + "uw_final", { htk.prev_frame_error, (float)htk.initializing, (float)vox.size() });
		printf("voxel frame %d row %d: %d voxels\n", (int)fi, rows[fi], (int)vox.size()); fflush(stdout);
	}
	htfx_close(&o.w);
	return 0;
}

// ---- modes ---------------------------------------------------------------------------------------
static int dump_model(PhysModel &m, const char *outfn)
{
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	int nb = (int)m.rigidbodies.size(), nj = (int)m.joints.size();
	o.i32("nb", { nb }); o.i32("nj", { nj });
	std::vector<float> bf; std::vector<int> bi, ign(nb * nb, 0), nverts, nplanes;
	for (int b = 0; b < nb; b++)
	{
		auto &rb = m.rigidbodies[b];
		o.v3("b" + std::to_string(b) + "/verts", rb.shapes[0].verts);
		{ std::vector<float3> sv; for (auto &v : m.sdmeshes[b].verts) sv.push_back(v.position); o.v3("b" + std::to_string(b) + "/sdverts", sv); }      // GetMeshes(true): the subdivision surface, rig space (physmodel.h:258,295-303)
		std::vector<float> pl; for (auto &p : rb.shapes[0].planes) for (int i = 0; i < 4; i++) pl.push_back(p[i]);
		o.f32("b" + std::to_string(b) + "/planes", pl, { (uint32_t)rb.shapes[0].planes.size(), 4 });
		std::vector<int> tr; for (auto &t : rb.shapes[0].tris) for (int i = 0; i < 3; i++) tr.push_back(t[i]);
		o.i32("b" + std::to_string(b) + "/tris", tr, { (uint32_t)rb.shapes[0].tris.size(), 3 });
		nverts.push_back((int)rb.shapes[0].verts.size()); nplanes.push_back((int)rb.shapes[0].planes.size());
		// body_f row: mass massinv radius radius_inner damping friction gravscale com3 pos_start3 quat_start4 tensorinv9(column major)
		bf.push_back(rb.mass); bf.push_back(rb.massinv); bf.push_back(rb.radius); bf.push_back(rb.radius_inner); bf.push_back(rb.damping); bf.push_back(rb.friction); bf.push_back(rb.gravscale);
		for (int i = 0; i < 3; i++) bf.push_back(rb.com[i]); for (int i = 0; i < 3; i++) bf.push_back(rb.position_start[i]); for (int i = 0; i < 4; i++) bf.push_back(rb.orientation_start[i]);
		for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) bf.push_back(rb.tensorinv_massless[c][r]);
		bi.push_back(rb.collide);
		for (auto *x : rb.ignore) ign[b * nb + rbidx(m, x)] = 1;
	}
	o.f32("body_f", bf, { (uint32_t)nb, 26 }); o.i32("body_collide", bi); o.i32("ignore", ign, { (uint32_t)nb, (uint32_t)nb });
	o.i32("nverts", nverts); o.i32("nplanes", nplanes);
	std::vector<int> ji; std::vector<float> jf;
	for (auto &j : m.joints)
	{
		ji.push_back(j.rbi0); ji.push_back(j.rbi1);
		for (int i = 0; i < 3; i++) jf.push_back(j.p0[i]); for (int i = 0; i < 3; i++) jf.push_back(j.p1[i]);
		for (int i = 0; i < 3; i++) jf.push_back(j.rangemin[i]); for (int i = 0; i < 3; i++) jf.push_back(j.rangemax[i]); for (int i = 0; i < 4; i++) jf.push_back(j.jointframe[i]);
	}
	o.i32("joint_i", ji, { (uint32_t)nj, 2 }); o.f32("joint_f", jf, { (uint32_t)nj, 16 });
	o.f32("physics", { physics_deltaT, physics_restitution, physics_gravity.x, physics_gravity.y, physics_gravity.z, physics_coloumb, physics_biasfactorjoint, physics_biasfactorpositive, physics_biasfactornegative,
		physics_falltime_to_ballistic, physics_driftmax, physics_damping, (float)physics_iterations, (float)physics_iterations_post, (float)physics_use_collision, physics_weak_force, bone_sum_error_scale, unibody_force });
	o.state("rest_state", m);
	{   // the 0.1 m cube proxy of UnibodyFit (handtrack.h:454-455)
		WingMesh box = WingMeshCube(0.1f);
		RigidBody ub({ Shape(box.verts, box.GenerateTris()) }, float3(0, 0, 0));
		o.v3("unibody/verts", ub.shapes[0].verts);
		std::vector<float> u = { ub.mass, ub.massinv, ub.radius, ub.damping, ub.friction, ub.gravscale, ub.com.x, ub.com.y, ub.com.z };
		for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) u.push_back(ub.tensorinv_massless[c][r]);
		o.f32("unibody/f", u);
	}
	htfx_close(&o.w);
	printf("model: %d bodies %d joints -> %s\n", nb, nj, outfn);
	return 0;
}
static int mode_model(const char *outfn)
{
	HandTracker htk;
	return dump_model(htk.handmodel, outfn);
}
static int mode_modelfile(const char *jsonfn, const char *outfn)     // any PhysModel JSON, without LoadHandModel's post-processing
{
	HandTracker htk;     // sets the physics globals the dump records (handtrack.h:837-838)
	PhysModel m(jsonfn);
	return dump_model(m, outfn);
}

static int mode_scan(const char *bankfn, int stride)
{
	HandTracker htk;
	PhysModel fake = LoadHandModel();
	auto bank = read_animbank(bankfn, fake.rigidbodies.size());
	printf("# animbank rows=%d\n# row inrange P pairs contacts fiterr focal\n", (int)bank.size());
	for (size_t k = 0; k < bank.size(); k += stride)
	{
		TileFrame fr = make_frame(fake, bank, k);
		auto pc = PointCloud(fr.seg, { 0.1f,0.7f });
		auto vp = takesubsample(pc, 4);
		htk.handmodel.SetPose(fr.start); zero_momenta(htk.handmodel);
		float err = FitError(htk.handmodel, vp, fr.seg);
		auto rbs = Addresses(htk.handmodel.rigidbodies);
		int pairs = 0;
		for (auto rb0 : rbs) for (auto rb1 : rbs) if (rb0 < rb1)
		{
			if (length(rb1->position - rb0->position) > rb0->radius + rb1->radius) continue;
			if (std::find(rb0->ignore.begin(), rb0->ignore.end(), rb1) != rb0->ignore.end()) continue;
			pairs++;
		}
		std::vector<PhysContact> contacts; FindShapeShapeContacts(contacts, rbs);
		printf("%d %d %d %d %d %.4f %.2f\n", (int)k, (int)pc.size(), (int)vp.size(), pairs, (int)contacts.size(), err, fr.seg.cam.focal().x);
	}
	return 0;
}

static int mode_frames(const char *bankfn, int first, int stride, int n, const char *outfn)
{
	PhysModel fake = LoadHandModel();
	auto bank = read_animbank(bankfn, fake.rigidbodies.size());
	std::vector<unsigned short> depth; std::vector<float> cams, start, gt; std::vector<int> rows;
	for (int i = 0; i < n; i++)
	{
		size_t k = (size_t)(first + (long)i * stride) % bank.size();
		TileFrame fr = make_frame(fake, bank, k);
		depth.insert(depth.end(), fr.seg.raster.begin(), fr.seg.raster.end());
		auto c = camvec(fr.seg.cam); cams.insert(cams.end(), c.begin(), c.end());
		auto s = flat(fr.start); start.insert(start.end(), s.begin(), s.end());
		auto g = flat(fr.gt); gt.insert(gt.end(), g.begin(), g.end());
		rows.push_back((int)k);
		if (i % 32 == 0) { printf("frame %d/%d row %d\n", i, n, (int)k); fflush(stdout); }
	}
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	o.u16("depth", depth, { (uint32_t)n, 64, 64 }); o.f32("cam", cams, { (uint32_t)n, 12 });
	o.f32("startpose", start, { (uint32_t)n, 17, 7 }); o.f32("gtpose", gt, { (uint32_t)n, 17, 7 }); o.i32("rows", rows);
	htfx_close(&o.w);
	return 0;
}

static void dump_contacts(Out &o, const std::string &n, PhysModel &m, const std::vector<PhysContact> &C)
{
	std::vector<float> v;
	for (auto &c : C)
	{
		v.push_back((float)rbidx(m, c.rb0)); v.push_back((float)rbidx(m, c.rb1));
		for (int i = 0; i < 3; i++) v.push_back(c.normal[i]); for (int i = 0; i < 3; i++) v.push_back(c.p0w[i]); for (int i = 0; i < 3; i++) v.push_back(c.p1w[i]);
		v.push_back(c.separation); for (int i = 0; i < 3; i++) v.push_back(c.p0[i]); for (int i = 0; i < 3; i++) v.push_back(c.p1[i]);
	}
	o.f32(n, v, { (uint32_t)C.size(), 18 });
}

// HandSegmentVR fixtures: the full-size rendered depth frame, its camera, and what the reference's segmentation returns
// (handtrack.h:280-344), for several entry options.  The two intermediate images come from the same reference functions.
static int mode_segment(const char *bankfn, const char *rowscsv, const char *outfn)
{
	PhysModel fake = LoadHandModel();
	auto bank = read_animbank(bankfn, fake.rigidbodies.size());
	std::vector<int> rows; { std::stringstream ss(rowscsv); std::string t; while (std::getline(ss, t, ',')) rows.push_back(atoi(t.c_str())); }
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	o.i32("rows", rows);
	const int opts[] = { 0xF, 1, 2, 4, 8, 5 };
	std::vector<int> optv(opts, opts + 6); o.i32("entry_options", optv);
	for (size_t fi = 0; fi < rows.size(); fi++)
	{
		std::string pre = "s" + std::to_string(fi) + "/";
		DCamera dcam({ 320,240 }, { 305,305 }, { 160,120 }, 0.001f);
		fake.SetPose(bank[rows[fi] % bank.size()]);
		auto depth = raycast_depth(fake, dcam);
		o.u16(pre + "depth", depth.raster, { 240, 320 });
		o.f32(pre + "cam", camvec(depth.cam));
		auto small = DownSampleMin(DownSampleMin(depth));
		o.u16(pre + "small", small.raster, { (uint32_t)small.dim().y, (uint32_t)small.dim().x });
		ushort2 wranged = ushort2(float2(0.1f, 0.70f) / depth.cam.depth_scale);
		auto dt = DistanceTransform(Threshold(small, [wranged](unsigned short d) { return d < wranged.y; }));
		std::vector<int> dti(dt.raster.begin(), dt.raster.end());
		o.i32(pre + "dt", dti, { (uint32_t)dt.dim().y, (uint32_t)dt.dim().x });
		for (int k = 0; k < 6; k++)
		{
			if (k > 0 && fi > 1) break;      // the alternative entry options only on the first two frames
			auto seg = HandSegmentVR(depth, opts[k], { 0.1f,0.70f });
			std::string q = pre + "o" + std::to_string(opts[k]) + "/";
			o.u16(q + "tile", seg.raster, { 64, 64 });
			o.f32(q + "cam", camvec(seg.cam));
		}
	}
	htfx_close(&o.w);
	printf("segment: %d frames -> %s\n", (int)rows.size(), outfn);
	return 0;
}

// HandTracker::scale (handtrack.h:591): the scaled model as the reference holds it, and the unit of work on two frames with that model
static int mode_scale(const char *bankfn, const char *rowscsv, uint64_t seed, double gain, double sc, const char *outfn)
{
	HandTracker htk;
	htk.microforce = 3.0f; htk.mainthreadpasses = 3; htk.always_take_cnn = 0;
	load_weights(htk, seed, gain);
	PhysModel fake = LoadHandModel();
	auto bank = read_animbank(bankfn, fake.rigidbodies.size());
	std::vector<int> rows; { std::stringstream ss(rowscsv); std::string t; while (std::getline(ss, t, ',')) rows.push_back(atoi(t.c_str())); }
	const float seg = htk.scale((float)sc);
	{ std::string mf = std::string(outfn) + ".model"; if (dump_model(htk.handmodel, mf.c_str())) return 2; }
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	o.i32("rows", rows); o.f32("scale_segment", { (float)sc, seg });
	for (size_t fi = 0; fi < rows.size(); fi++)
	{
		std::string pre = "f" + std::to_string(fi) + "/";
		TileFrame fr = make_frame(fake, bank, rows[fi]);
		o.u16(pre + "depth", fr.seg.raster, { 64, 64 }); o.f32(pre + "cam", camvec(fr.seg.cam)); o.f32(pre + "startpose", flat(fr.start), { 17, 7 });
		htk.handmodel.SetPose(fr.start); htk.othermodel.SetPose(fr.start); zero_momenta(htk.handmodel); zero_momenta(htk.othermodel);
		htk.prev_frame_error = 0; htk.initializing = 0;
		auto pose = htk.update_cnn_model(fr.seg);
		if (pose.size()) htk.handmodel.SetPose(pose);
		auto points = takesubsample(PointCloud(fr.seg, { 0.1f,htk.drangey }), htk.subsample_fraction);
		for (int i = 0; i < htk.mainthreadpasses; i++)
		{
			std::vector<LimitAngular> angulars; std::vector<LimitLinear> linears;
			HandModelEnhancements(htk.handmodel, angulars, false, float3(0, 0, 0), float3(0, 0, 0), 0);
			if (htk.boundary_planes && points.size() > htk.min_point_num) linears = cloud_chamber(htk.handmodel, points, { { -1,-0.25f,0 },{ -1,-1,0 },{ 0,-1,0 },{ 1,-1,0 },{ 1,-0.25f,0 } }, float3(0, 0, 0), float3(0, 0, 1), 10.0f);
			htk.handmodel.FitPointCloud(points, linears, angulars, htk.microforce);
		}
		o.state(pre + "hand", htk.handmodel);
		o.f32(pre + "pose_user", flat(htk.handmodel.GetPoseUser()), { 17, 7 });
	}
	htfx_close(&o.w);
	printf("scale %.3f: %d frames -> %s (+ .model)\n", sc, (int)rows.size(), outfn);
	return 0;
}

// HandTracker::slowfit (handtrack.h:786-821), the annotation fit loop: a few argument combinations per frame
static int mode_slowfit(const char *bankfn, const char *rowscsv, const char *outfn)
{
	HandTracker htk;
	htk.microforce = 3.0f;
	PhysModel fake = LoadHandModel();
	auto bank = read_animbank(bankfn, fake.rigidbodies.size());
	std::vector<int> rows; { std::stringstream ss(rowscsv); std::string t; while (std::getline(ss, t, ',')) rows.push_back(atoi(t.c_str())); }
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	o.i32("rows", rows);
	for (size_t fi = 0; fi < rows.size(); fi++)
	{
		std::string pre = "f" + std::to_string(fi) + "/";
		TileFrame fr = make_frame(fake, bank, rows[fi]);
		o.u16(pre + "depth", fr.seg.raster, { 64, 64 }); o.f32(pre + "cam", camvec(fr.seg.cam)); o.f32(pre + "startpose", flat(fr.start), { 17, 7 }); o.f32(pre + "refpose", flat(fr.gt), { 17, 7 });
		auto points = takesubsample(PointCloud(fr.seg, { 0.1f,htk.drangey }), htk.subsample_fraction);
		std::vector<float4> crays;      // unit rays from the camera to the ground-truth feature points, weight 1
		for (auto p : Skin(fr.gt, handmodelfeaturepoints)) crays.push_back(float4(normalize(p), 1.0f));
		std::vector<float> cf; for (auto &c : crays) for (int i = 0; i < 4; i++) cf.push_back(c[i]);
		o.f32(pre + "crays", cf, { 8, 4 });
		const float3 spoint = fr.gt[7] * float3(0, 0, 0.01f), rbpoint = float3(0, 0, 0.01f);
		o.f32(pre + "select", { 7.0f, spoint.x, spoint.y, spoint.z, rbpoint.x, rbpoint.y, rbpoint.z });
		struct { const char *name; int hold, steps; bool sel, rays; } cases[] = { { "plain", 0, 6, false, false }, { "hold1", 1, 6, false, false }, { "hold2", 2, 4, false, false }, { "rays", 0, 6, false, true }, { "nail", 1, 6, true, true } };
		for (auto &c : cases)
		{
			htk.handmodel.SetPose(fr.start); zero_momenta(htk.handmodel);
			htk.slowfit(points, c.hold, fr.gt, c.steps, c.sel ? &htk.handmodel.rigidbodies[7] : NULL, spoint, rbpoint, c.rays ? crays : std::vector<float4>(0));
			o.state(pre + c.name, htk.handmodel);
		}
	}
	htfx_close(&o.w);
	printf("slowfit: %d frames -> %s\n", (int)rows.size(), outfn);
	return 0;
}

// CNN::Train (cnn.h:558-580) with the labels of GatherHandExpectedCNN (handtrack.h:160-173), as train-cnn.cpp:156-162 drives it
static int mode_train(const char *bankfn, const char *rowscsv, uint64_t seed, double gain, int epochs, const char *outfn)
{
	HandTracker htk;
	load_weights(htk, seed, gain);
	PhysModel fake = LoadHandModel();
	auto bank = read_animbank(bankfn, fake.rigidbodies.size());
	std::vector<int> rows; { std::stringstream ss(rowscsv); std::string t; while (std::getline(ss, t, ',')) rows.push_back(atoi(t.c_str())); }
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	o.i32("rows", rows); o.f32("seed_gain_epochs_alpha", { (float)seed, (float)gain, (float)epochs, 0.001f });
	std::vector<std::vector<float>> inputs, labels;
	for (size_t fi = 0; fi < rows.size(); fi++)
	{
		std::string pre = "f" + std::to_string(fi) + "/";
		TileFrame fr = make_frame(fake, bank, rows[fi]);
		float2 drange = { 0.1f, htk.drangey };
		auto cnn_input = Transform(fr.seg, [drange, &fr](unsigned short d) { return (float)clamp(1.0f - (d*fr.seg.cam.depth_scale - drange.x) / (drange.y - drange.x), 0.0f, 1.0f); });
		auto lab = GatherHandExpectedCNN(fr.gt, camsub(fr.seg.cam, 4));
		o.u16(pre + "depth", fr.seg.raster, { 64, 64 }); o.f32(pre + "cam", camvec(fr.seg.cam)); o.f32(pre + "pose", flat(fr.gt), { 17, 7 });
		o.f32(pre + "labels", lab.cnn_expected); o.f32(pre + "vals", lab.vals);
		inputs.push_back(cnn_input.raster); labels.push_back(lab.cnn_expected);
	}
	std::vector<float> mse;
	auto t0 = std::chrono::steady_clock::now();
	for (int e = 0; e < epochs; e++) for (size_t fi = 0; fi < rows.size(); fi++) mse.push_back(htk.cnn.Train(inputs[fi], labels[fi], 0.001f));
	const double train_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	printf("train: %.2f ms per CNN::Train call (1 thread)\n", train_ms / (double)mse.size());
	o.f32("mse", mse);
	o.f32("eval0_after", htk.cnn.Eval(inputs[0]));
	auto *c1 = (CNN::LConv *)htk.cnn.layers[0]; auto *c2 = (CNN::LConv *)htk.cnn.layers[4]; auto *f1 = (CNN::LFull *)htk.cnn.layers[7]; auto *f2 = (CNN::LFull *)htk.cnn.layers[9];
	o.f32("W1", c1->W); o.f32("B1", c1->B); o.f32("B2", c2->B); o.f32("B3", f1->B); o.f32("B4", f2->B);
	o.f32("W2_head", std::vector<float>(c2->W.begin(), c2->W.begin() + 1024));
	std::vector<float> s3, s4; for (size_t i = 0; i < f1->W.size(); i += 9973) s3.push_back(f1->W[i]); for (size_t i = 0; i < f2->W.size(); i += 9973) s4.push_back(f2->W[i]);
	o.f32("W3_every9973", s3); o.f32("W4_every9973", s4);
	htfx_close(&o.w);
	printf("train: %d frames x %d epochs -> %s (mse first %.6g last %.6g)\n", (int)rows.size(), epochs, outfn, mse.front(), mse.back());
	return 0;
}

static int mode_golden(const char *bankfn, const char *rowscsv, uint64_t seed, double gain, const char *outfn)
{
	HandTracker htk;
	htk.microforce = 3.0f; htk.mainthreadpasses = 3; htk.always_take_cnn = 0;     // synthetic-tracker.cpp:91-93
	load_weights(htk, seed, gain);
	PhysModel fake = LoadHandModel();
	auto bank = read_animbank(bankfn, fake.rigidbodies.size());
	std::vector<int> rows; { std::stringstream ss(rowscsv); std::string t; while (std::getline(ss, t, ',')) rows.push_back(atoi(t.c_str())); }
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	o.i32("rows", rows); o.f32("weights_seed_gain", { (float)seed, (float)gain });
	for (size_t fi = 0; fi < rows.size(); fi++)
	{
		std::string pre = "f" + std::to_string(fi) + "/";
		TileFrame fr = make_frame(fake, bank, rows[fi]);
		auto &seg = fr.seg;
		o.u16(pre + "depth", seg.raster, { 64,64 }); o.f32(pre + "cam", camvec(seg.cam)); o.f32(pre + "startpose", flat(fr.start), { 17,7 }); o.f32(pre + "gtpose", flat(fr.gt), { 17,7 });
		float2 drange = { 0.1f, htk.drangey };
		// S1: CNN input / layers / output (handtrack.h:700-701)
		auto cnn_input = Transform(seg, [drange, &seg](unsigned short d) {return (float)clamp(1.0f - (d*seg.cam.depth_scale - drange.x) / (drange.y - drange.x), 0.0f, 1.0f); });
		const bool full = fi < 2;   // bulky per-stage arrays only for the first two frames (fixture size)
		if (full) o.f32(pre + "cnn_input", cnn_input.raster);
		{
			std::vector<float> x = cnn_input.raster;
			for (size_t li = 0; li < htk.cnn.layers.size(); li++)
			{
				x = htk.cnn.layers[li]->forward(x);
				if (fi == 0 && (li == 3 || li >= 6)) o.f32(pre + "cnn_layer" + std::to_string(li), x);   // pooled / FC stages (the 60x60 planes are too bulky)
			}
			o.f32(pre + "cnn_output", x);
			auto y = htk.cnn.Eval(cnn_input.raster);
			if (y != x) { fprintf(stderr, "Eval != layerwise?\n"); return 3; }
		}
		auto cnn_output = htk.cnn.Eval(cnn_input.raster);
		// S2: decode (handtrack.h:218-241)
		DCamera hcam = camsub(seg.cam, 4);
		CNNOutputAnalysis an(cnn_output, hcam);
		{
			std::vector<float> cr, ip; for (auto &c : an.crays) for (int i = 0; i < 4; i++) cr.push_back(c[i]); for (auto &p : an.image_points) { ip.push_back(p.x); ip.push_back(p.y); }
			o.f32(pre + "an_crays", cr, { 8,4 }); o.f32(pre + "an_image_points", ip, { 8,2 }); o.f32(pre + "an_confidence", an.confidence); o.f32(pre + "an_vals", an.vals);
			o.f32(pre + "an_angles", { an.wristroll, an.pitch, an.tilt, an.palmq.x, an.palmq.y, an.palmq.z, an.palmq.w }); o.f32(pre + "an_clenched", an.finger_clenched);
			std::vector<float> top2;   // margin information for tolerance decisions: best and second best value per map
			for (int mi = 0; mi < 8; mi++) { std::vector<float> mm(cnn_output.begin() + 256 * mi, cnn_output.begin() + 256 * (mi + 1)); std::sort(mm.begin(), mm.end()); top2.push_back(mm[255]); top2.push_back(mm[254]); }
			o.f32(pre + "an_top2", top2, { 8,2 });
		}
		// S3: point clouds (misc_image.h:409-417, physmodel.h:58-64)
		auto pc = PointCloud(seg, drange);
		auto vpts = takesubsample(pc, htk.subsample_fraction);
		o.i32(pre + "pc_count", { (int)pc.size(), (int)vpts.size() }); o.v3(pre + "vpts", vpts);
		// S4: FitError at the start pose (handtrack.h:371-399)
		reset_tracker(htk, fr.start);
		o.f32(pre + "fiterror_start", { FitError(htk.handmodel, vpts, seg) });
		// S5: closest + cloud constraints at the start pose (physmodel.h:137-181)
		{
			auto rbs = Addresses(htk.handmodel.rigidbodies);
			std::vector<float> cl;
			for (auto &v : vpts) { auto r = closest(rbs, v); cl.push_back((float)rbidx(htk.handmodel, r.first)); for (int i = 0; i < 4; i++) cl.push_back(r.second[i]); }
			o.f32(pre + "closest_vpts", cl, { (uint32_t)vpts.size(), 5 });
			if (full) o.linears(pre + "cloud_rows_vpts", htk.handmodel, CloudConstraints(rbs, vpts, seg.cam.pose.position));
			o.linears(pre + "cloud_rows_sub", htk.handmodel, CloudConstraints(rbs, takesubsample(vpts), seg.cam.pose.position));
		}
		// S6: joint rows after HandModelEnhancements (handtrack.h:406-441, physmodel.h:321-334)
		{
			std::vector<LimitAngular> extra;
			HandModelEnhancements(htk.handmodel, extra, false, float3(0, 0, 0), float3(0, 0, 0), 0);
			std::vector<float> jr; for (auto &j : htk.handmodel.joints) { for (int i = 0; i < 3; i++) jr.push_back(j.rangemin[i]); for (int i = 0; i < 3; i++) jr.push_back(j.rangemax[i]); }
			o.f32(pre + "joint_ranges", jr, { 16,6 });
			o.linears(pre + "joint_linears", htk.handmodel, htk.handmodel.GetLinearConstraints());
			o.angulars(pre + "joint_angulars", htk.handmodel, htk.handmodel.GetAngularConstraints());
			std::vector<LimitAngular> extra2;
			HandModelEnhancements(htk.handmodel, extra2, false, qrot(seg.cam.pose.orientation, float3(-1, 0, 0)), qrot(seg.cam.pose.orientation, float3(0, -1, 0)));
			o.angulars(pre + "enh_angulars", htk.handmodel, extra2);
			o.angulars(pre + "apply_angles", htk.handmodel, an.ApplyAngles(htk.handmodel, seg.cam.pose, 10000.0f));
			std::vector<float3> outdirs = { float3(-1, -0.25f, 0), float3(-1, -1, 0), float3(0, -1, 0), float3(1, -1, 0), float3(1, -0.25f, 0) };
			o.linears(pre + "chamber_rows", htk.handmodel, cloud_chamber(htk.handmodel, vpts, outdirs, { 0,0,0 }, { 0,0,1 }, 10.0f));
		}
		// S7: body-body contacts at the start pose (physics.h:451-462, gjk.h:607-643)
		{
			std::vector<PhysContact> contacts; FindShapeShapeContacts(contacts, Addresses(htk.handmodel.rigidbodies));
			dump_contacts(o, pre + "contacts_start", htk.handmodel, contacts);
		}
		// S8: two consecutive FitPointCloud passes from the start pose with zero momenta (physmodel.h:345-356)
		reset_tracker(htk, fr.start);
		for (int p = 0; p < 2; p++)
		{
			std::vector<LimitAngular> angulars;
			HandModelEnhancements(htk.handmodel, angulars, false, float3(0, 0, 0), float3(0, 0, 0), 0);
			htk.handmodel.FitPointCloud(vpts, {}, angulars, htk.microforce);
			o.state(pre + "fit_pass" + std::to_string(p), htk.handmodel);
		}
		// S9: MultiStepSim after s=1..5 steps from the start pose (handtrack.h:642-690)
		for (int s = 1; s <= 5; s++)
		{
			reset_tracker(htk, fr.start);
			htk.steps = s;
			htk.MultiStepSim(htk.othermodel, an, vpts, seg.cam.pose);
			o.state(pre + "multistep" + std::to_string(s), htk.othermodel);
		}
		htk.steps = 5;
		// S10: the whole unit of work
		reset_tracker(htk, fr.start);
		auto user = unit_of_work(htk, seg, &o, pre);
		o.f32(pre + "uw_pose_user", flat(user), { 17,7 });
		o.f32(pre + "uw_final", { htk.prev_frame_error, (float)htk.initializing });
		// S10b: a second frame of streaming use (state carried: momenta, prev_frame_error, initializing); same image again
		auto user2 = unit_of_work(htk, seg, NULL, "");
		o.f32(pre + "uw2_pose_user", flat(user2), { 17,7 }); o.state(pre + "uw2_hand", htk.handmodel);
		o.f32(pre + "uw2_final", { htk.prev_frame_error, (float)htk.initializing });
		// S12: forced reset path (handtrack.h:480-506, 451-470)
		if (fi < 3)
		{
			reset_tracker(htk, fr.start);
			PoseFromScratch(htk.othermodel, vpts, an, seg.cam.pose);
			o.state(pre + "scratch", htk.othermodel);
			for (int i = 0; i < 3; i++) { UnibodyFit(htk.othermodel, vpts, seg.cam.pose.position); o.state(pre + "unibody" + std::to_string(i), htk.othermodel); }
			o.f32(pre + "fiterror_scratch", { FitError(htk.othermodel, vpts, seg) });
		}
		printf("golden frame %d row %d P=%d\n", (int)fi, rows[fi], (int)vpts.size()); fflush(stdout);
	}
	// S13: GJK / EPA unit cases on pairs of bodies in hand-made poses (gjk.h:367-437, hull.h:233-310)
	{
		PhysModel &m = htk.handmodel;
		m.Reset();
		struct Case { int a, b; float3 pa; float4 qa; float3 pb; float4 qb; };
		std::vector<Case> cases;
		uint64_t c = 0;
		auto rnd = [&]() { return (float)((double)(splitmix64_at(0xC0FFEEull, c++) >> 11) * (1.0 / 9007199254740992.0)); };
		for (int i = 0; i < 48; i++)
		{
			int a = (int)(rnd() * 17) % 17, b = (int)(rnd() * 17) % 17; if (a == b) b = (b + 1) % 17;
			float4 qa = normalize(float4(rnd() - 0.5f, rnd() - 0.5f, rnd() - 0.5f, rnd() - 0.5f)), qb = normalize(float4(rnd() - 0.5f, rnd() - 0.5f, rnd() - 0.5f, rnd() - 0.5f));
			float sep = (i < 16) ? 0.08f : (i < 32) ? 0.03f : 0.008f;     // far, near/touching, penetrating
			float3 dir = normalize(float3(rnd() - 0.5f, rnd() - 0.5f, rnd() - 0.5f));
			cases.push_back({ a, b, float3(0,0,0.4f), qa, float3(0,0,0.4f) + dir * sep, qb });
		}
		std::vector<float> in, outv;
		for (auto &cs : cases)
		{
			auto &A = m.rigidbodies[cs.a]; auto &B = m.rigidbodies[cs.b];
			A.position = cs.pa; A.orientation = cs.qa; B.position = cs.pb; B.orientation = cs.qb;
			auto h = Separated(SupportFunc(&A, A.shapes[0]), SupportFunc(&B, B.shapes[0]), 1);
			in.push_back((float)cs.a); in.push_back((float)cs.b);
			for (int i = 0; i < 3; i++) in.push_back(cs.pa[i]); for (int i = 0; i < 4; i++) in.push_back(cs.qa[i]); for (int i = 0; i < 3; i++) in.push_back(cs.pb[i]); for (int i = 0; i < 4; i++) in.push_back(cs.qb[i]);
			for (int i = 0; i < 3; i++) outv.push_back(h.normal[i]); for (int i = 0; i < 3; i++) outv.push_back(h.p0w[i]); for (int i = 0; i < 3; i++) outv.push_back(h.p1w[i]); outv.push_back(h.separation);
			auto patch = ContactPatch(SupportFunc(&A, A.shapes[0]), SupportFunc(&B, B.shapes[0]), physics_driftmax);
			outv.push_back((float)patch.count);
			for (int k = 0; k < 5; k++) { for (int i = 0; i < 3; i++) outv.push_back(k < patch.count ? patch[k].p0w[i] : 0); for (int i = 0; i < 3; i++) outv.push_back(k < patch.count ? patch[k].p1w[i] : 0); outv.push_back(k < patch.count ? patch[k].separation : 0); }
		}
		o.f32("gjk_cases_in", in, { (uint32_t)cases.size(), 16 }); o.f32("gjk_cases_out", outv, { (uint32_t)cases.size(), 46 });
		m.Reset();
	}
	htfx_close(&o.w);
	return 0;
}

// minimal HTFX reader for bench mode
struct Arr { std::string name; uint32_t dtype, ndim, dims[4]; std::vector<char> data; };
static std::vector<Arr> htfx_read(const char *fn)
{
	std::vector<Arr> v; FILE *f = fopen(fn, "rb"); if (!f) { fprintf(stderr, "cannot open %s\n", fn); exit(2); }
	char magic[8]; uint32_t count; if (fread(magic, 1, 8, f) != 8 || fread(&count, 4, 1, f) != 1) exit(2);
	for (uint32_t i = 0; i < count; i++)
	{
		Arr a; char nm[48]; uint64_t n;
		if (fread(nm, 1, 48, f) != 48) exit(2); a.name = nm;
		if (fread(&a.dtype, 4, 1, f) != 1 || fread(&a.ndim, 4, 1, f) != 1 || fread(a.dims, 4, 4, f) != 4 || fread(&n, 8, 1, f) != 1) exit(2);
		a.data.resize(n); if (n && fread(a.data.data(), 1, n, f) != n) exit(2);
		fseek(f, (long)((8 - (n & 7)) & 7), SEEK_CUR);
		v.push_back(std::move(a));
	}
	fclose(f); return v;
}
// The 128x128-input variant of the pose net (BASELINE configs[4] / SURVEY 8d "config 5 (ii)").  The reference ships one topology (handtrack.h:108-118, 64x64);
// this is the same list of the reference's OWN layer classes with the dimensions a 128x128 input gives: conv5 -> 124, pool -> 62 -> 31, conv4 -> 28,
// pool -> 14, FC 12544 -> 2048 -> 2304, chunked softmax.  Weights: our seeded generator with the larger first FC layer, loaded through CNN::loadb.
static CNN make_cnn128(uint64_t seed, double gain)
{
	CNN cnn({});
	cnn.layers.push_back(new CNN::LConv({ 128,128,1 }, { 5,5,1,16 }, { 124,124,16 }));
	cnn.layers.push_back(new CNN::LActivation<TanH>(124 * 124 * 16));
	cnn.layers.push_back(new CNN::LMaxPool({ 124,124,16 }));
	cnn.layers.push_back(new CNN::LMaxPool({ 62,62,16 }));
	cnn.layers.push_back(new CNN::LConv({ 31,31,16 }, { 4,4,16,64 }, { 28,28,64 }));
	cnn.layers.push_back(new CNN::LActivation<TanH>(28 * 28 * 64));
	cnn.layers.push_back(new CNN::LMaxPool({ 28,28,64 }));
	cnn.layers.push_back(new CNN::LFull(14 * 14 * 64, 16 * 16 * 8));
	cnn.layers.push_back(new CNN::LActivation<TanH>(16 * 16 * 8));
	cnn.layers.push_back(new CNN::LFull(16 * 16 * 8, 16 * 16 * 8 + 16 * 16));
	cnn.layers.push_back(new CNN::LSoftMaxChunked(concat(std::vector<int>(8, 16 * 16), std::vector<int>(16, 16))));
	auto w = make_cnnb(seed, gain, 14 * 14 * 64);
	std::string s((const char*)w.data(), w.size() * sizeof(float));
	std::istringstream is(s, std::ios::binary);
	cnn.loadb(is);
	return cnn;      // the layer objects are shared by the copies and never freed (CNN holds plain pointers, cnn.h:100-104)
}
static int mode_cnn128(const char *framesfn, const char *idxcsv, uint64_t seed, double gain, const char *outfn)
{
	CNN cnn = make_cnn128(seed, gain);
	auto arrs = htfx_read(framesfn);
	const Arr *ad = NULL, *ac = NULL;
	for (auto &a : arrs) { if (a.name == "depth") ad = &a; if (a.name == "cam") ac = &a; }
	if (!ad || !ac || ad->dims[1] != 128 || ad->dims[2] != 128) { fprintf(stderr, "frames file lacks 128x128 depth / cam\n"); return 2; }
	std::vector<int> idx; { std::stringstream ss(idxcsv); std::string t; while (std::getline(ss, t, ',')) idx.push_back(atoi(t.c_str())); }
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	o.i32("frames", idx); o.f32("weights_seed_gain", { (float)seed, (float)gain });
	double ms = 0;
	for (size_t k = 0; k < idx.size(); k++)
	{
		if (idx[k] < 0 || idx[k] >= (int)ad->dims[0]) { fprintf(stderr, "frame index out of range\n"); return 2; }
		const float *c = (const float*)ac->data.data() + 12 * idx[k];
		DCamera cam({ 128,128 }, { c[0],c[1] }, { c[2],c[3] }, c[4]);
		const unsigned short *d = (const unsigned short*)ad->data.data() + (size_t)128 * 128 * idx[k];
		Image<unsigned short> frame(cam, std::vector<unsigned short>(d, d + 128 * 128));
		const float2 drange = { 0.1f, 0.7f };
		auto cnn_input = Transform(frame, [drange, &frame](unsigned short dd) { return (float)clamp(1.0f - (dd*frame.cam.depth_scale - drange.x) / (drange.y - drange.x), 0.0f, 1.0f); });      // as handtrack.h:700
		std::string pre = "f" + std::to_string(k) + "/";
		auto t0 = std::chrono::steady_clock::now();
		auto y = cnn.Eval(cnn_input.raster);
		ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
		o.f32(pre + "cnn_output", y);
		if (k == 0)
		{
			o.f32(pre + "cnn_input", cnn_input.raster, { 128, 128 });
			std::vector<float> x = cnn_input.raster;
			for (size_t l = 0; l < cnn.layers.size(); l++)
			{
				x = cnn.layers[l]->forward(x);
				if (l == 3 || l == 6 || l == 8 || l == 9) o.f32(pre + "layer" + std::to_string(l), x);
			}
		}
	}
	htfx_close(&o.w);
	printf("cnn128: %d frames -> %s (%.2f ms per CNN::Eval, 1 thread)\n", (int)idx.size(), outfn, ms / (double)idx.size());
	return 0;
}

// BASELINE configs[4] END TO END as SURVEY 8(d) "config 5 (i)-(iii)" defines it: a 128x128 frame of the 26-bone hand, the 128x128-input net, its decode and
// the tracker in ONE unit of work.  HandTracker::update_cnn_model_threadsafe itself (handtrack.h:693-729) cannot run it -- it hands every frame that
// is not 64x64 to HandSegmentVR and decodes with camsub(cam, 4) -- so its statements are restated here one by one with the reference's own
// functions: the frame is its own segment (what HandSegmentVR returns for a frame of the net's size, :283-284), the 16x16 heat-map camera is
// camsub(cam, 8), everything else as written there; then the main-thread passes of HandTracker::update (:769-782) exactly as unit_of_work above.
static std::vector<Pose> unit_of_work_direct(HandTracker &htk, CNN &net, const Image<unsigned short> &dimage, Out *out = NULL, const std::string &pre = "")
{
	const float2 drange = { 0.1f, htk.drangey };
	auto points = takesubsample(PointCloud(dimage, { 0.1f,htk.drangey }), htk.subsample_fraction, htk.subsample_voxel, htk.subsample_size);      // :753
	htk.othermodel.SetPose(htk.handmodel.GetPose());      // :757
	std::vector<Pose> pose;
	{
		const Image<unsigned short> &segment = dimage;
		DCamera hcam = camsub(segment.cam, segment.cam.dim().x / 16);
		auto cnn_input = Transform(segment, [drange, &segment](unsigned short d) {return (float)clamp(1.0f - (d*segment.cam.depth_scale - drange.x) / (drange.y - drange.x), 0.0f, 1.0f); });      // :700
		auto cnn_output = net.Eval(cnn_input.raster);
		auto cnn_output_analysis = CNNOutputAnalysis(cnn_output, hcam);
		auto vpts = takesubsample(PointCloud(dimage, drange), htk.subsample_fraction);
		float olderror = FitError(htk.handmodel, vpts, dimage);
		if (htk.angles_only || olderror > htk.full_reset_on_error)
		{
			PoseFromScratch(htk.othermodel, vpts, cnn_output_analysis, segment.cam.pose);
			for (int i = 0; i < htk.steps_unibody; i++) UnibodyFit(htk.othermodel, vpts, segment.cam.pose.position);
		}
		htk.MultiStepSim(htk.othermodel, cnn_output_analysis, vpts, segment.cam.pose);
		float newerror = FitError(htk.othermodel, vpts, dimage);
		if (newerror > olderror) htk.prev_frame_error = 0.0f; else htk.prev_frame_error += olderror - newerror;
		if ((vpts.size() > htk.min_point_num && htk.initializing) || htk.always_take_cnn || htk.angles_only || htk.prev_frame_error > htk.accum_error_threshold) pose = htk.othermodel.GetPose();
		if (htk.prev_frame_error > htk.accum_error_threshold) htk.prev_frame_error = 0.0f;
		htk.initializing = std::max(htk.initializing - 1, 0);
		htk.cnn_input = cnn_input; htk.cnn_output = cnn_output; htk.cnn_output_analysis = cnn_output_analysis;      // update_cnn_model :737-739
		if (out)
		{
			out->f32(pre + "cnn_output", cnn_output);
			std::vector<float> cr; for (auto &c : cnn_output_analysis.crays) { cr.push_back(c.x); cr.push_back(c.y); cr.push_back(c.z); cr.push_back(c.w); }
			out->f32(pre + "crays", cr, { (uint32_t)cnn_output_analysis.crays.size(), 4 });
			out->f32(pre + "vals", cnn_output_analysis.vals);
			out->f32(pre + "errors", { olderror, newerror, (float)vpts.size(), (float)points.size() });
			out->state(pre + "uw_other_after_cnn", htk.othermodel);
			out->f32(pre + "uw_accept", { (float)pose.size(), htk.prev_frame_error, (float)htk.initializing });
		}
	}
	htk.handmodel.SetPose(pose);      // :764 (an empty pose changes nothing, physmodel.h:277-282)
	for (int i = 0; !htk.angles_only && i < htk.mainthreadpasses; i++)
	{
		std::vector<LimitLinear> linears; std::vector<LimitAngular> angulars;
		HandModelEnhancements(htk.handmodel, angulars, false, float3(0, 0, 0), float3(0, 0, 0), 0);
		if (points.size() > htk.min_point_num && htk.boundary_planes)
		{
			std::vector<float3> outdirs = { float3(-1, -0.25f, 0), float3(-1, -1, 0), float3(0, -1, 0), float3(1, -1, 0), float3(1, -0.25f, 0) };
			Append(linears, cloud_chamber(htk.handmodel, points, outdirs, { 0,0,0 }, { 0,0,1 }, 10.0f));
		}
		htk.handmodel.FitPointCloud(points, linears, angulars, htk.microforce);
		if (out) out->state(pre + "uw_hand_pass" + std::to_string(i), htk.handmodel);
	}
	if (points.size() < htk.min_point_num) htk.initializing = 50;
	return htk.handmodel.GetPoseUser();
}
struct Frames128 { std::vector<Image<unsigned short>> frames; std::vector<std::vector<Pose>> starts; int nb = 0; };
static bool read_frames128(const char *framesfn, Frames128 &F, int maxframes = 0)
{
	auto arrs = htfx_read(framesfn);
	const Arr *ad = NULL, *ac = NULL, *as = NULL;
	for (auto &a : arrs) { if (a.name == "depth") ad = &a; if (a.name == "cam") ac = &a; if (a.name == "startpose") as = &a; }
	if (!ad || !ac || !as || ad->dims[1] != 128 || ad->dims[2] != 128) { fprintf(stderr, "frames file lacks 128x128 depth / cam / startpose\n"); return false; }
	int n = (int)ad->dims[0]; if (maxframes > 0 && n > maxframes) n = maxframes;
	F.nb = (int)as->dims[1];
	for (int i = 0; i < n; i++)
	{
		const float *c = (const float*)ac->data.data() + 12 * i;
		DCamera cam({ 128,128 }, { c[0],c[1] }, { c[2],c[3] }, c[4], Pose({ c[5],c[6],c[7] }, { c[8],c[9],c[10],c[11] }));
		const unsigned short *d = (const unsigned short*)ad->data.data() + (size_t)128 * 128 * i;
		F.frames.push_back(Image<unsigned short>(cam, std::vector<unsigned short>(d, d + 128 * 128)));
		std::vector<Pose> sp(F.nb); const float *s = (const float*)as->data.data() + (size_t)7 * F.nb * i;
		for (int b = 0; b < F.nb; b++) sp[b] = Pose({ s[7 * b],s[7 * b + 1],s[7 * b + 2] }, { s[7 * b + 3],s[7 * b + 4],s[7 * b + 5],s[7 * b + 6] });
		F.starts.push_back(sp);
	}
	return true;
}
// e2e128 <frames128.htfx> <idx,comma> <seed> <fc2gain> <out.htfx>: per-stage dumps of the listed frames (two consecutive updates on the first of them), and
// under "all/" the results of the unit of work on EVERY frame of the file (user poses, othermodel, flags) for the bench's own verification
static int mode_e2e128(const char *framesfn, const char *idxcsv, uint64_t seed, double gain, const char *outfn)
{
	HandTracker htk;
	htk.microforce = 3.0f; htk.mainthreadpasses = 3; htk.always_take_cnn = 0;
	CNN net = make_cnn128(seed, gain);
	Frames128 F; if (!read_frames128(framesfn, F)) return 2;
	if ((size_t)F.nb != htk.handmodel.rigidbodies.size()) { fprintf(stderr, "start poses have %d bones, the model %d\n", F.nb, (int)htk.handmodel.rigidbodies.size()); return 2; }
	std::vector<int> idx; { std::stringstream ss(idxcsv); std::string t; while (std::getline(ss, t, ',')) idx.push_back(atoi(t.c_str())); }
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	o.i32("frames", idx); o.f32("weights_seed_gain", { (float)seed, (float)gain });
	for (size_t k = 0; k < idx.size(); k++)
	{
		if (idx[k] < 0 || idx[k] >= (int)F.frames.size()) { fprintf(stderr, "frame index out of range\n"); return 2; }
		const std::string pre = "f" + std::to_string(k) + "/";
		reset_tracker(htk, F.starts[idx[k]]);
		auto pose = unit_of_work_direct(htk, net, F.frames[idx[k]], &o, pre);
		o.f32(pre + "uw_pose_user", flat(pose), { (uint32_t)F.nb,7 });
		o.state(pre + "uw_other_final", htk.othermodel);
		if (k == 0)      // a second update on the carried state (momenta, prev_frame_error, initializing carried over)
		{
			auto pose2 = unit_of_work_direct(htk, net, F.frames[idx[k]], &o, pre + "second/");
			o.f32(pre + "second/uw_pose_user", flat(pose2), { (uint32_t)F.nb,7 });
		}
		if (k + 1 == idx.size())      // the same frame from a pose far off (another frame's, translated) with always_take_cnn: the full-reset branch (:706-711) and the accept (:721)
		{
			std::vector<Pose> far = F.starts[(idx[k] + F.frames.size() / 2) % F.frames.size()];
			for (auto &q : far) q.position += float3(0.05f, -0.04f, 0.03f);
			reset_tracker(htk, far); htk.always_take_cnn = 1;
			o.f32(pre + "far/startpose", flat(far), { (uint32_t)F.nb,7 });
			auto pose3 = unit_of_work_direct(htk, net, F.frames[idx[k]], &o, pre + "far/");
			o.f32(pre + "far/uw_pose_user", flat(pose3), { (uint32_t)F.nb,7 });
			htk.always_take_cnn = 0;
		}
		printf("e2e128 frame %d done\n", idx[k]); fflush(stdout);
	}
	std::vector<float> user, other, flags;
	for (size_t i = 0; i < F.frames.size(); i++)
	{
		reset_tracker(htk, F.starts[i]);
		auto p = unit_of_work_direct(htk, net, F.frames[i]);
		for (float f : flat(p)) user.push_back(f);
		for (auto &rb : htk.othermodel.rigidbodies) { for (int c = 0; c < 3; c++) other.push_back(rb.position[c]); for (int c = 0; c < 4; c++) other.push_back(rb.orientation[c]); }
		flags.push_back(htk.prev_frame_error); flags.push_back((float)htk.initializing);
	}
	o.f32("all/uw_pose_user", user, { (uint32_t)F.frames.size(), (uint32_t)F.nb, 7 });
	o.f32("all/other_pose", other, { (uint32_t)F.frames.size(), (uint32_t)F.nb, 7 });
	o.f32("all/flags", flags, { (uint32_t)F.frames.size(), 2 });
	htfx_close(&o.w);
	return 0;
}
// bench128 <frames128.htfx> <seed> <fc2gain> <reps> [maxframes]: reference CPU time of that unit of work (and of the 128x128 net alone)
static int mode_bench128(const char *framesfn, uint64_t seed, double gain, int reps, int maxframes)
{
	HandTracker htk;
	htk.microforce = 3.0f; htk.mainthreadpasses = 3; htk.always_take_cnn = 0;
	CNN net = make_cnn128(seed, gain);
	Frames128 F; if (!read_frames128(framesfn, F, maxframes)) return 2;
	if ((size_t)F.nb != htk.handmodel.rigidbodies.size()) { fprintf(stderr, "start poses have %d bones, the model %d\n", F.nb, (int)htk.handmodel.rigidbodies.size()); return 2; }
	const int n = (int)F.frames.size();
	double best_cnn = 1e30, best_uw = 1e30, checksum = 0;
	for (int r = 0; r < reps; r++)
	{
		auto t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < n; i++)
		{
			float2 drange = { 0.1f, htk.drangey }; auto &seg = F.frames[i];
			auto in = Transform(seg, [drange, &seg](unsigned short d) {return (float)clamp(1.0f - (d*seg.cam.depth_scale - drange.x) / (drange.y - drange.x), 0.0f, 1.0f); });
			auto y = net.Eval(in.raster); checksum += y[0];
		}
		auto t1 = std::chrono::steady_clock::now();
		for (int i = 0; i < n; i++) { reset_tracker(htk, F.starts[i]); auto p = unit_of_work_direct(htk, net, F.frames[i]); checksum += p[1].position.x; }
		auto t2 = std::chrono::steady_clock::now();
		best_cnn = std::min(best_cnn, std::chrono::duration<double>(t1 - t0).count() / n); best_uw = std::min(best_uw, std::chrono::duration<double>(t2 - t1).count() / n);
	}
	printf("{\"frames\": %d, \"reps\": %d, \"cnn_ms\": %.4f, \"cnn_fps\": %.2f, \"frame_ms\": %.4f, \"frame_fps\": %.2f, \"checksum\": %.6f}\n", n, reps, best_cnn * 1e3, 1.0 / best_cnn, best_uw * 1e3, 1.0 / best_uw, checksum);
	return 0;
}

static int mode_bench(const char *framesfn, uint64_t seed, double gain, int reps, int maxframes)
{
	HandTracker htk;
	htk.microforce = 3.0f; htk.mainthreadpasses = 3; htk.always_take_cnn = 0;
	load_weights(htk, seed, gain);
	auto arrs = htfx_read(framesfn);
	const Arr *ad = NULL, *ac = NULL, *as = NULL;
	for (auto &a : arrs) { if (a.name == "depth") ad = &a; if (a.name == "cam") ac = &a; if (a.name == "startpose") as = &a; }
	if (!ad || !ac || !as) { fprintf(stderr, "frames file lacks depth/cam/startpose\n"); return 2; }
	int n = (int)ad->dims[0]; if (maxframes > 0 && n > maxframes) n = maxframes;
	// frames of any size (64x64 tiles; 128x128 frames of BASELINE configs[4], which HandTracker segments itself) and any hand model (HT_REF_MODEL_JSON)
	const int h = (int)ad->dims[1], w = (int)ad->dims[2], nb = (int)as->dims[1];
	if ((size_t)nb != htk.handmodel.rigidbodies.size()) { fprintf(stderr, "start poses have %d bones, the model %d\n", nb, (int)htk.handmodel.rigidbodies.size()); return 2; }
	std::vector<Image<unsigned short>> segs; std::vector<std::vector<Pose>> starts;
	for (int i = 0; i < n; i++)
	{
		const float *c = (const float*)ac->data.data() + 12 * i;
		DCamera cam({ w,h }, { c[0],c[1] }, { c[2],c[3] }, c[4], Pose({ c[5],c[6],c[7] }, { c[8],c[9],c[10],c[11] }));
		const unsigned short *d = (const unsigned short*)ad->data.data() + (size_t)w * h * i;
		segs.push_back(Image<unsigned short>(cam, std::vector<unsigned short>(d, d + (size_t)w * h)));
		std::vector<Pose> sp(nb); const float *s = (const float*)as->data.data() + (size_t)7 * nb * i;
		for (int b = 0; b < nb; b++) sp[b] = Pose({ s[7 * b],s[7 * b + 1],s[7 * b + 2] }, { s[7 * b + 3],s[7 * b + 4],s[7 * b + 5],s[7 * b + 6] });
		starts.push_back(sp);
	}
	double best_cnn = 1e30, best_uw = 1e30; double checksum = 0;
	for (int r = 0; r < reps; r++)
	{
		auto t0 = std::chrono::steady_clock::now();
		if (w == 64 && h == 64) for (int i = 0; i < n; i++)
		{
			float2 drange = { 0.1f, htk.drangey }; auto &seg = segs[i];
			auto in = Transform(seg, [drange, &seg](unsigned short d) {return (float)clamp(1.0f - (d*seg.cam.depth_scale - drange.x) / (drange.y - drange.x), 0.0f, 1.0f); });
			auto y = htk.cnn.Eval(in.raster); checksum += y[0];
		}
		auto t1 = std::chrono::steady_clock::now();
		for (int i = 0; i < n; i++)
		{
			reset_tracker(htk, starts[i]);
			auto p = unit_of_work(htk, segs[i]); checksum += p[1].position.x;
		}
		auto t2 = std::chrono::steady_clock::now();
		double c = std::chrono::duration<double>(t1 - t0).count() / n, u = std::chrono::duration<double>(t2 - t1).count() / n;
		best_cnn = std::min(best_cnn, c); best_uw = std::min(best_uw, u);
	}
	printf("{\"frames\": %d, \"reps\": %d, \"cnn_ms\": %.4f, \"cnn_fps\": %.2f, \"frame_ms\": %.4f, \"frame_fps\": %.2f, \"checksum\": %.6f}\n", n, reps, (w == 64 && h == 64) ? best_cnn * 1e3 : 0.0, (w == 64 && h == 64) ? 1.0 / best_cnn : 0.0, best_uw * 1e3, 1.0 / best_uw, checksum);
	return 0;
}

// The unit of work on every frame of a frames file, results only: user-space poses of handmodel, othermodel's state after the CNN job and the
// accept decision.  Built with other compiler flags (HT_REF_FLAGS of tools/ref_flag_spread.sh) it measures how far the REFERENCE moves between its
// own builds (IEEE / FMA-contracted / the Makefile's -Ofast) on the very frames the bench uses.
// `take_cnn`: the application's switch of the same name (synthetic-tracker.cpp:91,127,240): every frame's CNN-driven pose is accepted (handtrack.h:720-722), so the
// user pose depends on the net, its decode and MultiStepSim on every frame (without it the accept branch fires on ~3 % of these frames)
static int mode_poses(const char *framesfn, uint64_t seed, double gain, const char *outfn, int take_cnn = 0)
{
	HandTracker htk;
	htk.microforce = 3.0f; htk.mainthreadpasses = 3; htk.always_take_cnn = take_cnn;
	load_weights(htk, seed, gain);
	auto arrs = htfx_read(framesfn);
	const Arr *ad = NULL, *ac = NULL, *as = NULL;
	for (auto &a : arrs) { if (a.name == "depth") ad = &a; if (a.name == "cam") ac = &a; if (a.name == "startpose") as = &a; }
	if (!ad || !ac || !as) { fprintf(stderr, "frames file lacks depth/cam/startpose\n"); return 2; }
	const int n = (int)ad->dims[0];
	std::vector<float> user, other, accept;
	for (int i = 0; i < n; i++)
	{
		const float *c = (const float*)ac->data.data() + 12 * i;
		DCamera cam({ 64,64 }, { c[0],c[1] }, { c[2],c[3] }, c[4], Pose({ c[5],c[6],c[7] }, { c[8],c[9],c[10],c[11] }));
		const unsigned short *d = (const unsigned short*)ad->data.data() + 4096 * i;
		Image<unsigned short> seg(cam, std::vector<unsigned short>(d, d + 4096));
		std::vector<Pose> sp(17); const float *s = (const float*)as->data.data() + 119 * i;
		for (int b = 0; b < 17; b++) sp[b] = Pose({ s[7 * b],s[7 * b + 1],s[7 * b + 2] }, { s[7 * b + 3],s[7 * b + 4],s[7 * b + 5],s[7 * b + 6] });
		reset_tracker(htk, sp);
		auto points = takesubsample(PointCloud(seg, { 0.1f,htk.drangey }), htk.subsample_fraction, htk.subsample_voxel, htk.subsample_size);
		auto p = unit_of_work(htk, seg);
		for (float f : flat(p)) user.push_back(f);
		for (auto &rb : htk.othermodel.rigidbodies) { for (int k = 0; k < 3; k++) other.push_back(rb.position[k]); for (int k = 0; k < 4; k++) other.push_back(rb.orientation[k]); }
		accept.push_back(htk.prev_frame_error); accept.push_back((float)htk.initializing); accept.push_back((float)points.size());
	}
	Out o; if (htfx_open(&o.w, outfn)) return 3;
	o.f32("uw_pose_user", user, { (uint32_t)n, 17, 7 });
	o.f32("other_pose", other, { (uint32_t)n, 17, 7 });
	o.f32("flags", accept, { (uint32_t)n, 3 });
	htfx_close(&o.w);
	return 0;
}

// posesfull <frames.htfx> <seed> <fc2gain> <out.htfx>: HandTracker's unit of work on every frame of a file of FULL-SIZE frames (any w x h: the tracker segments them
// itself, handtrack.h:697-698) with whatever hand model is staged -- results only, for the bench's verification of BASELINE configs[4] as the reference runs it
static int mode_posesfull(const char *framesfn, uint64_t seed, double gain, const char *outfn)
{
	HandTracker htk;
	htk.microforce = 3.0f; htk.mainthreadpasses = 3; htk.always_take_cnn = 0;
	load_weights(htk, seed, gain);
	auto arrs = htfx_read(framesfn);
	const Arr *ad = NULL, *ac = NULL, *as = NULL;
	for (auto &a : arrs) { if (a.name == "depth") ad = &a; if (a.name == "cam") ac = &a; if (a.name == "startpose") as = &a; }
	if (!ad || !ac || !as) { fprintf(stderr, "frames file lacks depth/cam/startpose\n"); return 2; }
	const int n = (int)ad->dims[0], h = (int)ad->dims[1], w = (int)ad->dims[2], nb = (int)as->dims[1];
	if ((size_t)nb != htk.handmodel.rigidbodies.size()) { fprintf(stderr, "start poses have %d bones, the model %d\n", nb, (int)htk.handmodel.rigidbodies.size()); return 2; }
	std::vector<float> user, other, flags;
	for (int i = 0; i < n; i++)
	{
		const float *c = (const float*)ac->data.data() + 12 * i;
		DCamera cam({ w,h }, { c[0],c[1] }, { c[2],c[3] }, c[4], Pose({ c[5],c[6],c[7] }, { c[8],c[9],c[10],c[11] }));
		const unsigned short *d = (const unsigned short*)ad->data.data() + (size_t)w * h * i;
		Image<unsigned short> frame(cam, std::vector<unsigned short>(d, d + (size_t)w * h));
		std::vector<Pose> sp(nb); const float *s = (const float*)as->data.data() + (size_t)7 * nb * i;
		for (int b = 0; b < nb; b++) sp[b] = Pose({ s[7 * b],s[7 * b + 1],s[7 * b + 2] }, { s[7 * b + 3],s[7 * b + 4],s[7 * b + 5],s[7 * b + 6] });
		reset_tracker(htk, sp);
		auto p = unit_of_work(htk, frame);
		for (float f : flat(p)) user.push_back(f);
		for (auto &rb : htk.othermodel.rigidbodies) { for (int k = 0; k < 3; k++) other.push_back(rb.position[k]); for (int k = 0; k < 4; k++) other.push_back(rb.orientation[k]); }
		flags.push_back(htk.prev_frame_error); flags.push_back((float)htk.initializing);
	}
	Out o; if (htfx_open(&o.w, outfn)) return 3;
	o.f32("uw_pose_user", user, { (uint32_t)n, (uint32_t)nb, 7 });
	o.f32("other_pose", other, { (uint32_t)n, (uint32_t)nb, 7 });
	o.f32("flags", flags, { (uint32_t)n, 2 });
	htfx_close(&o.w);
	return 0;
}

// viz <animbank.pose> <row> <out.htfx>: what the application draws beside the tracker (synthetic-tracker.cpp:191,204-209): DepthMesh of the 320x240 frame, and the
// expected landmark heat-maps laid over the segmented tile by VisualizeHMaps; with the inputs (frame, tile, cameras, pose) the host-side helpers of
// include/ht_handtrack.hpp are checked against
static int mode_viz(const char *bankfn, int row, const char *outfn)
{
	PhysModel fake = LoadHandModel();
	auto bank = read_animbank(bankfn, fake.rigidbodies.size());
	DCamera dcam({ 320,240 }, { 305,305 }, { 160,120 }, 0.001f);
	fake.SetPose(bank[row % bank.size()]);
	auto dimage = raycast_depth(fake, dcam);
	const float2 drange = { 0.1f, 0.7f };
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	o.u16("depth", dimage.raster, { 240, 320 }); o.f32("cam", camvec(dimage.cam)); o.f32("pose", flat(fake.GetPose()), { (uint32_t)fake.rigidbodies.size(), 7 });
	auto dmesh = DepthMesh(dimage, { drange.x, drange.y }, 0.03f, 3);      // synthetic-tracker.cpp:191
	o.v3("dm_verts", dmesh.first);
	std::vector<int> t; for (auto &q : dmesh.second) { t.push_back(q.x); t.push_back(q.y); t.push_back(q.z); }
	o.i32("dm_tris", t, { (uint32_t)dmesh.second.size(), 3 });
	auto segment = HandSegmentVR(dimage);      // :204
	auto segment_f = Transform(segment, [drange, &segment](unsigned short d) {return (float)clamp(1.0f - (d*segment.cam.depth_scale - drange.x) / (drange.y - drange.x), 0.0f, 1.0f); });
	DCamera hcam = camsub(segment_f.cam, 4);
	auto fake_labels = GatherHandExpectedCNN(fake.GetPose(), hcam);
	auto landmark_labels = VisualizeHMaps(fake_labels.hmaps, segment_f);      // :208
	auto angle_labels = ToRGB(UpSample(UpSample(UpSample(fake_labels.vmap))));      // :209
	o.u16("tile", segment.raster, { 64, 64 }); o.f32("segcam", camvec(segment.cam));
	auto bytes = [](const Image<byte3> &im) { std::vector<unsigned short> v; for (auto &c : im.raster) { v.push_back(c.x); v.push_back(c.y); v.push_back(c.z); } return v; };
	o.u16("landmark_labels", bytes(landmark_labels), { (uint32_t)landmark_labels.dim().y, (uint32_t)landmark_labels.dim().x, 3 });
	o.u16("angle_labels", bytes(angle_labels), { (uint32_t)angle_labels.dim().y, (uint32_t)angle_labels.dim().x, 3 });
	htfx_close(&o.w);
	printf("viz: %d mesh vertices, %d triangles, labels %d x %d\n", (int)dmesh.first.size(), (int)dmesh.second.size(), landmark_labels.dim().x, landmark_labels.dim().y);
	return 0;
}

// ---- on-disk dataset formats (include/dataset.h): the reference's own writer and reader -----------------------------------------------------------
static void dump_dataset_info(Out &o, const DatasetInfo &d)
{
	o.f32("info_camera", { (float)d.dcamera.dim().x, (float)d.dcamera.dim().y, d.dcamera.focal().x, d.dcamera.focal().y, d.dcamera.principal().x, d.dcamera.principal().y, d.dcamera.depth_scale });
	o.f32("info_mplane", { d.mplane.x, d.mplane.y, d.mplane.z, d.mplane.w });
	o.f32("info_misc", { d.hasir ? 1.0f : 0.0f, (float)d.rgb_dim.x, (float)d.rgb_dim.y, (float)d.feye_dim.x, (float)d.feye_dim.y, d.segment_scale });
	std::vector<unsigned short> a(d.fname.begin(), d.fname.end()), b(d.camtype.begin(), d.camtype.end());
	o.u16("info_fname", a, { (uint32_t)a.size() }); o.u16("info_camtype", b, { (uint32_t)b.size() });
}
// DepthDataStreamOut (dataset.h:62-106) writes a small three-frame set with every stream (depth, ir, poses, rgb, fish-eye) under <prefix>
static int mode_dataset_write(const char *dir, const char *prefix)
{
	if (chdir(dir)) { fprintf(stderr, "cannot enter %s\n", dir); return 2; }      // the header records the prefix it is given (DatasetInfo::fname): keep it free of this machine's paths
	DatasetInfo dsi{ DCamera({ 16,12 }, { 14.5f,14.25f }, { 8.25f,5.75f }, 0.000125f), float4(0.0f, 0.6f, 0.8f, -0.35f), prefix, "synthetic", false, { 8,6 }, { 4,2 }, 0.165f };
	DepthDataStreamOut out(dsi);
	out.AddRGB().AddFishEye();
	for (int k = 0; k < 3; k++)
	{
		Image<unsigned short> d(dsi.dcamera); Image<unsigned char> ir(dsi.dcamera); Image<byte3> rgb(dsi.rgb_dim); Image<unsigned char> fe(dsi.feye_dim);
		for (size_t i = 0; i < d.raster.size(); i++) { d.raster[i] = (unsigned short)(1000 * k + 7 * i + 1); ir.raster[i] = (unsigned char)(3 * i + k); }
		for (size_t i = 0; i < rgb.raster.size(); i++) rgb.raster[i] = byte3((unsigned char)(i + k), (unsigned char)(2 * i + k), (unsigned char)(255 - i));
		for (size_t i = 0; i < fe.raster.size(); i++) fe.raster[i] = (unsigned char)(17 * i + 5 * k);
		std::vector<Pose> p(17);
		for (int b = 0; b < 17; b++) p[b] = Pose(float3(0.0123456f * b - 0.1f, -0.125f * k + 1e-5f * b, 0.3f + 0.001f * k), normalize(float4(0.1f * b, -0.3f, 0.25f * k, 1.0f)));
		out.SaveFrame(MakeFrame(d, p, ir, rgb, fe));
	}
	return 0;
}
// load_dataset (dataset.h:109-163) reads <prefix>.* and everything it returns is dumped
static int mode_dataset_read(const char *prefix, int nposes, const char *outfn)
{
	auto frames = load_dataset(prefix, (unsigned)nposes);
	DatasetInfo dsi; from_json(dsi, json::parsefile(std::string(prefix) + ".json"));
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	dump_dataset_info(o, dsi);
	o.i32("nframes", { (int)frames.size() });
	for (size_t k = 0; k < frames.size(); k++)
	{
		const std::string pre = "f" + std::to_string(k) + "/"; auto &f = frames[k];
		o.u16(pre + "depth", f.depth.raster, { (uint32_t)f.depth.dim().y, (uint32_t)f.depth.dim().x });
		std::vector<unsigned short> ir(f.ir.raster.begin(), f.ir.raster.end()), fe(f.fisheye.raster.begin(), f.fisheye.raster.end()), rgb;
		for (auto &c : f.rgb.raster) { rgb.push_back(c.x); rgb.push_back(c.y); rgb.push_back(c.z); }
		o.u16(pre + "ir", ir); o.u16(pre + "rgb", rgb); o.u16(pre + "fisheye", fe);
		o.f32(pre + "pose", flat(f.pose), { (uint32_t)f.pose.size(), 7 });
		o.f32(pre + "cam", camvec(f.depth.cam));
		o.i32(pre + "fid", { f.fid });
	}
	htfx_close(&o.w);
	return 0;
}
// the reference-held sample datasets/example/hand_data_example.{json,pose} (its .rs/.ir blobs are stripped): the header as from_json decodes it and every
// pose the reference's stream operator reads from the text
static int mode_dataset_header(const char *jsonfn, const char *posefn, int nposes, const char *outfn)
{
	DatasetInfo dsi; from_json(dsi, json::parsefile(jsonfn));
	Out o; if (htfx_open(&o.w, outfn)) return 2;
	dump_dataset_info(o, dsi);
	std::ifstream in(posefn);
	std::vector<float> all; int n = 0;
	for (;;) { std::vector<Pose> p(nposes); bool ok = true; for (auto &q : p) if (!(in >> q)) { ok = false; break; } if (!ok) break; for (float v : flat(p)) all.push_back(v); n++; }
	o.f32("poses", all, { (uint32_t)n, (uint32_t)nposes, 7 });
	htfx_close(&o.w);
	return 0;
}

int main(int argc, char **argv) try
{
	if (argc < 2) { fprintf(stderr, "usage: see header of ref_harness.cpp\n"); return 1; }
	std::string mode = argv[1];
	// resolve file arguments to absolute paths before we chdir into the staged asset tree
	std::vector<std::string> a;
	for (int i = 2; i < argc; i++) { std::string s = argv[i]; if (s.find('/') != std::string::npos || s.find(".htfx") != std::string::npos || s.find(".pose") != std::string::npos) { char buf[4096]; if (s[0] != '/' && getcwd(buf, sizeof buf)) s = std::string(buf) + "/" + s; } a.push_back(s); }
	stage_assets();
	if (mode == "model" && a.size() == 1) return mode_model(a[0].c_str());
	if (mode == "modelfile" && a.size() == 2) return mode_modelfile(a[0].c_str(), a[1].c_str());
	if (mode == "scan" && a.size() == 2) return mode_scan(a[0].c_str(), atoi(a[1].c_str()));
	if (mode == "frames" && a.size() == 5) return mode_frames(a[0].c_str(), atoi(a[1].c_str()), atoi(a[2].c_str()), atoi(a[3].c_str()), a[4].c_str());
	if (mode == "scale" && a.size() == 6) return mode_scale(a[0].c_str(), a[1].c_str(), strtoull(a[2].c_str(), 0, 0), atof(a[3].c_str()), atof(a[4].c_str()), a[5].c_str());
	if (mode == "train" && a.size() == 6) return mode_train(a[0].c_str(), a[1].c_str(), strtoull(a[2].c_str(), 0, 0), atof(a[3].c_str()), atoi(a[4].c_str()), a[5].c_str());
	if (mode == "slowfit" && a.size() == 3) return mode_slowfit(a[0].c_str(), a[1].c_str(), a[2].c_str());
	if (mode == "segment" && a.size() == 3) return mode_segment(a[0].c_str(), a[1].c_str(), a[2].c_str());
	if (mode == "fullframes" && a.size() == 6) return mode_fullframes(a[0].c_str(), atoi(a[1].c_str()), atoi(a[2].c_str()), atoi(a[3].c_str()), a[4].c_str(), a[5].c_str());
	if (mode == "fullframe" && a.size() == 6) return mode_fullframe(a[0].c_str(), a[1].c_str(), a[2].c_str(), strtoull(a[3].c_str(), 0, 0), atof(a[4].c_str()), a[5].c_str());
	if (mode == "config5" && a.size() == 5) return mode_config5(a[0].c_str(), a[1].c_str(), strtoull(a[2].c_str(), 0, 0), atof(a[3].c_str()), a[4].c_str());
	if (mode == "voxel" && a.size() == 7) return mode_voxel(a[0].c_str(), a[1].c_str(), strtoull(a[2].c_str(), 0, 0), atof(a[3].c_str()), atof(a[4].c_str()), atoi(a[5].c_str()), a[6].c_str());
	if (mode == "golden" && a.size() == 5) return mode_golden(a[0].c_str(), a[1].c_str(), strtoull(a[2].c_str(), 0, 0), atof(a[3].c_str()), a[4].c_str());
	if (mode == "cnn128" && a.size() == 5) return mode_cnn128(a[0].c_str(), a[1].c_str(), strtoull(a[2].c_str(), 0, 0), atof(a[3].c_str()), a[4].c_str());
	if (mode == "e2e128" && a.size() == 5) return mode_e2e128(a[0].c_str(), a[1].c_str(), strtoull(a[2].c_str(), 0, 0), atof(a[3].c_str()), a[4].c_str());
	if (mode == "bench128" && a.size() >= 4) return mode_bench128(a[0].c_str(), strtoull(a[1].c_str(), 0, 0), atof(a[2].c_str()), atoi(a[3].c_str()), a.size() > 4 ? atoi(a[4].c_str()) : 0);
	if (mode == "dataset_write" && a.size() == 2) return mode_dataset_write(a[0].c_str(), a[1].c_str());
	if (mode == "dataset_read" && a.size() == 3) return mode_dataset_read(a[0].c_str(), atoi(a[1].c_str()), a[2].c_str());
	if (mode == "dataset_header" && a.size() == 4) return mode_dataset_header(a[0].c_str(), a[1].c_str(), atoi(a[2].c_str()), a[3].c_str());
	if (mode == "poses" && a.size() == 4) return mode_poses(a[0].c_str(), strtoull(a[1].c_str(), 0, 0), atof(a[2].c_str()), a[3].c_str());
	if (mode == "poses" && a.size() == 5 && a[4] == "takecnn") return mode_poses(a[0].c_str(), strtoull(a[1].c_str(), 0, 0), atof(a[2].c_str()), a[3].c_str(), 1);
	if (mode == "viz" && a.size() == 3) return mode_viz(a[0].c_str(), atoi(a[1].c_str()), a[2].c_str());
	if (mode == "posesfull" && a.size() == 4) return mode_posesfull(a[0].c_str(), strtoull(a[1].c_str(), 0, 0), atof(a[2].c_str()), a[3].c_str());
	if (mode == "bench" && a.size() >= 4) return mode_bench(a[0].c_str(), strtoull(a[1].c_str(), 0, 0), atof(a[2].c_str()), atoi(a[3].c_str()), a.size() > 4 ? atoi(a[4].c_str()) : 0);
	fprintf(stderr, "bad arguments\n");
	return 1;
}
catch (const char *c) { fprintf(stderr, "reference threw: %s\n", c); return 4; }
catch (const std::exception &e) { fprintf(stderr, "reference threw: %s\n", e.what()); return 4; }
