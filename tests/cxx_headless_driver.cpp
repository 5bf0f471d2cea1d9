// A headless stand-in for synthetic-hand-tracker/synthetic-tracker.cpp written against include/ht_handtrack.hpp with the reference's global names:
// the same calls in the same order (tracker set-up :90-93, fake hand :94-96,139, software depth raster :69-76, per-frame sequence :204-216),
// without the window.  Our own code: the reference's application is not copied, only its use of the API is followed.
//
//   driver fakedepth <model> <in.bin> <out.bin>      host only: pose a fake hand, ray-cast a depth frame (FakeDepth)
//   driver pointcloud - <in.bin> <out.bin>           host only: PointCloud(dimage, {0.1, 0.7}) of every frame: i32 n, n x float3
//   driver viz - <in.bin> <out.bin>                  host only: what the application draws beside the tracker (:191, :204-209) on frame 0 of in.bin (320x240) and
//                                                    the 64x64 tile given as frame 1's first 4096 pixels + camera: DepthMesh, VisualizeHMaps of the expected
//                                                    landmark maps over the tile, the angle maps as ToRGB(UpSample^3); out: i32 nv, nt; verts; tris; bytes of both images
//   driver track <model> <weights.cnnb> <in.bin> <out.bin>      the tracking loop on given frames (needs the GPU)
//   driver latency <model> <weights.cnnb> <in.bin> <iters>      time per HandTracker::update call as the application makes it (one frame per call, host buffers in, poses
//                                                    out: synthetic-tracker.cpp:215), then per ht_update_sync call on 8 and 64 trackers; prints p50 / p99 in ms (needs the GPU)
//
// in.bin:  int32 n, w, h, nb; then n records { u16 depth[w*h]; f32 cam[12]; f32 start[nb][7]; f32 gt[nb][7] }
// out.bin (track): n records { f32 pose_user[nb][7]; f32 cnn_output[2304]; f32 labels[2304]; f32 cnn_pose_accepted; f32 handpose_via_facade[nb][7] }
// out.bin (fakedepth): n records { u16 depth[w*h] }
#define HT_MI355X_GLOBAL_NAMES
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include "../include/ht_formats.hpp"

static Image<unsigned short> FakeDepth(PhysModel &model, const DCamera &dcam)      // the application's software rasteriser: one HitCheck per pixel
{
	Image<unsigned short> depth(dcam);
	for (int y = 0; y < dcam.dim().y; y++) for (int x = 0; x < dcam.dim().x; x++)
	{
		// deprojectz(p, 4.0): the point 4 m out along the pixel's ray (misc_image.h:48)
		const float3 far{ ((float)x - dcam.principal().x) / dcam.focal().x * 4.0f, ((float)y - dcam.principal().y) / dcam.focal().y * 4.0f, 4.0f };
		depth.pixel({ x, y }) = (unsigned short)(model.HitCheck({ 0, 0, 0 }, far).impact.z / dcam.depth_scale);
	}
	return depth;
}
struct Record { std::vector<unsigned short> depth; float cam[12]; std::vector<Pose> start, gt; };
static std::vector<Pose> poses_of(const float *p, int nb) { std::vector<Pose> v(nb); for (int b = 0; b < nb; b++) { v[b].position = { p[7 * b], p[7 * b + 1], p[7 * b + 2] }; v[b].orientation = { p[7 * b + 3], p[7 * b + 4], p[7 * b + 5], p[7 * b + 6] }; } return v; }
static void put_poses(FILE *f, const std::vector<Pose> &v) { for (auto &p : v) { const float r[7] = { p.position.x, p.position.y, p.position.z, p.orientation.x, p.orientation.y, p.orientation.z, p.orientation.w }; fwrite(r, 4, 7, f); } }
static std::vector<Record> read_input(const char *fn, int &w, int &h, int &nb)
{
	FILE *f = fopen(fn, "rb"); if (!f) throw std::runtime_error("cannot open input");
	int hdr[4]; if (fread(hdr, 4, 4, f) != 4) throw std::runtime_error("short input");
	w = hdr[1]; h = hdr[2]; nb = hdr[3];
	std::vector<Record> recs(hdr[0]);
	for (auto &r : recs)
	{
		r.depth.resize((size_t)w * h); std::vector<float> a((size_t)nb * 7), b((size_t)nb * 7);
		if (fread(r.depth.data(), 2, r.depth.size(), f) != r.depth.size() || fread(r.cam, 4, 12, f) != 12 || fread(a.data(), 4, a.size(), f) != a.size() || fread(b.data(), 4, b.size(), f) != b.size()) throw std::runtime_error("short record");
		r.start = poses_of(a.data(), nb); r.gt = poses_of(b.data(), nb);
	}
	fclose(f);
	return recs;
}
static DCamera camera_of(const float *c, int w, int h) { Pose p; p.position = { c[5], c[6], c[7] }; p.orientation = { c[8], c[9], c[10], c[11] }; return DCamera({ w, h }, { c[0], c[1] }, { c[2], c[3] }, c[4], p); }

int main(int argc, char **argv)
{
	if (argc < 5) { printf("usage: %s fakedepth <model> <in> <out> | track <model> <cnnb> <in> <out>\n", argv[0]); return 2; }
	try
	{
		const std::string mode = argv[1];
		int w = 0, h = 0, nb = 0;
		if (mode == "fakedepth")
		{
			auto recs = read_input(argv[3], w, h, nb);
			PhysModel fakehand = LoadHandModel(argv[2]);                       // synthetic-tracker.cpp:94
			if ((int)fakehand.rigidbodies.size() != nb) throw std::runtime_error("bone count mismatch");
			FILE *o = fopen(argv[4], "wb");
			for (auto &r : recs)
			{
				fakehand.SetPose(r.gt);                                        // :139
				auto dimage = FakeDepth(fakehand, camera_of(r.cam, w, h));     // :182 (software_rasterizer)
				fwrite(dimage.raster.data(), 2, dimage.raster.size(), o);
			}
			fclose(o);
			printf("fakedepth: %zu frames of %dx%d, %zu mesh triangles on bone 1\n", recs.size(), w, h, fakehand.GetMeshes(true)[1].tris.size());
			return 0;
		}
		if (mode == "pointcloud")      // host only: PointCloud(dimage, {0.1, drangey}) as synthetic-tracker.cpp:233 calls it; argv: pointcloud - <in> <out>
		{
			auto recs = read_input(argv[3], w, h, nb);
			FILE *o = fopen(argv[4], "wb");
			for (auto &r : recs)
			{
				Image<unsigned short> dimage(camera_of(r.cam, w, h), r.depth);
				auto pts = PointCloud(dimage, { 0.1f, 0.7f });
				const int n = (int)pts.size();
				fwrite(&n, 4, 1, o); fwrite(pts.data(), sizeof(float3), pts.size(), o);
			}
			fclose(o);
			printf("pointcloud: %zu frames\n", recs.size());
			return 0;
		}
		if (mode == "viz")
		{
			auto recs = read_input(argv[3], w, h, nb);
			if (recs.size() != 2) throw std::runtime_error("viz wants two records: the frame, and the tile in the second record's first 4096 pixels");
			const float2 drange{ 0.1f, 0.7f };
			Image<unsigned short> dimage(camera_of(recs[0].cam, w, h), recs[0].depth);
			auto dmesh = DepthMesh(dimage, { drange.x, drange.y }, 0.03f, 3);                              // :191
			Image<unsigned short> segment(camera_of(recs[1].cam, 64, 64), std::vector<unsigned short>(recs[1].depth.begin(), recs[1].depth.begin() + 4096));      // :204 (HandSegmentVR runs on the device: its result is an input here)
			auto segment_f = Transform(segment, [drange, &segment](unsigned short d) { const float v = 1.0f - (d * segment.cam.depth_scale - drange.x) / (drange.y - drange.x); return (float)(v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v)); });      // :205
			DCamera hcam = camsub(segment_f.cam, 4);                                                       // :206
			auto fake_labels = GatherHandExpectedCNN(recs[0].gt, hcam);                                    // :207
			auto landmark_labels = VisualizeHMaps(fake_labels.hmaps, segment_f);                           // :208
			auto angle_labels = ToRGB(UpSample(UpSample(UpSample(fake_labels.vmap))));                     // :209
			FILE *o = fopen(argv[4], "wb");
			const int nv = (int)dmesh.first.size(), nt = (int)dmesh.second.size(), dims[4] = { landmark_labels.dim().x, landmark_labels.dim().y, angle_labels.dim().x, angle_labels.dim().y };
			fwrite(&nv, 4, 1, o); fwrite(&nt, 4, 1, o); fwrite(dims, 4, 4, o);
			fwrite(dmesh.first.data(), sizeof(float3), nv, o); fwrite(dmesh.second.data(), sizeof(int3), nt, o);
			for (auto &c : landmark_labels.raster) { const unsigned char b[3] = { c.x, c.y, c.z }; fwrite(b, 1, 3, o); }
			for (auto &c : angle_labels.raster) { const unsigned char b[3] = { c.x, c.y, c.z }; fwrite(b, 1, 3, o); }
			fclose(o);
			printf("viz: %d vertices, %d triangles, labels %dx%d\n", nv, nt, dims[0], dims[1]);
			return 0;
		}
		if (mode == "latency" && argc >= 6)
		{
			auto recs = read_input(argv[4], w, h, nb);
			const int iters = atoi(argv[5]);
			auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
			{
				HandTracker htk(argv[2], argv[3]);
				htk.always_take_cnn = 0; htk.microforce = 3.0f; htk.mainthreadpasses = 3;
				std::vector<double> ms;
				for (int i = 0; i < iters + 20; i++)
				{
					const Record &r = recs[i % recs.size()];
					Image<unsigned short> dimage(camera_of(r.cam, w, h), r.depth);
					if (i % recs.size() == 0) htk.SetPose(r.start);      // a tracker follows its sequence; re-seeded when the frames start over
					const auto t0 = std::chrono::steady_clock::now();
					auto pose = htk.update(std::move(dimage));
					const double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
					if (i >= 20) ms.push_back(dt);
					if (pose.size() != (size_t)nb) throw std::runtime_error("update returned no pose");
				}
				printf("{\"call\": \"HandTracker::update\", \"frames_per_call\": 1, \"iters\": %d, \"p50_ms\": %.4f, \"p99_ms\": %.4f, \"mean_ms\": %.4f}\n", iters, pct(ms, 0.5), pct(ms, 0.99), [&] { double s = 0; for (double v : ms) s += v; return s / ms.size(); }());
			}
			{
				// the reference's structure (handtrack.h:755-768): the CNN job of a frame beside the caller's passes on a second context, collected by a later call
				HandTracker htk(argv[2], argv[3]);
				htk.always_take_cnn = 0; htk.microforce = 3.0f; htk.mainthreadpasses = 3; htk.overlapped_update = true;
				std::vector<double> ms; int collected_late = 0;
				for (int i = 0; i < iters + 20; i++)
				{
					const Record &r = recs[i % recs.size()];
					Image<unsigned short> dimage(camera_of(r.cam, w, h), r.depth);
					if (i % recs.size() == 0) htk.SetPose(r.start);
					const auto t0 = std::chrono::steady_clock::now();
					auto pose = htk.update(std::move(dimage));
					const double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
					if (i >= 20) ms.push_back(dt);
					if (pose.size() != (size_t)nb) throw std::runtime_error("update returned no pose");
				}
				(void)collected_late;
				printf("{\"call\": \"HandTracker::update, overlapped (the CNN job on a second context beside the caller's passes, collected by a later call)\", \"frames_per_call\": 1, \"iters\": %d, \"p50_ms\": %.4f, \"p99_ms\": %.4f, \"mean_ms\": %.4f}\n", iters, pct(ms, 0.5), pct(ms, 0.99), [&] { double s = 0; for (double v : ms) s += v; return s / ms.size(); }());
			}
			for (int B : { 8, 64 })
			{
				ht_ctx *ctx = nullptr;
				if (ht_create(argv[2], B, 0, &ctx) != HT_OK) throw std::runtime_error(ctx ? ht_last_error(ctx) : "ht_create");
				std::vector<float> wts; { std::ifstream is(argv[3], std::ios::binary); is.seekg(0, std::ios::end); wts.resize((size_t)is.tellg() / 4); is.seekg(0); is.read((char *)wts.data(), (std::streamsize)wts.size() * 4); }
				using ht_mi355x::check;
				check(ctx, ht_cnn_load_weights(ctx, wts.data(), wts.size()));
				ht_params par; ht_get_params(ctx, &par); par.microforce = 3.0f; par.mainthreadpasses = 3; check(ctx, ht_set_params(ctx, &par));
				std::vector<unsigned short> depth((size_t)B * w * h); std::vector<float> cams((size_t)B * 12), start((size_t)B * nb * 7), poses((size_t)B * nb * 7);
				for (int k = 0; k < B; k++)
				{
					const Record &r = recs[k % recs.size()];
					memcpy(&depth[(size_t)k * w * h], r.depth.data(), (size_t)w * h * 2); memcpy(&cams[(size_t)k * 12], r.cam, 48);
					for (int b = 0; b < nb; b++) { const float q[7] = { r.start[b].position.x, r.start[b].position.y, r.start[b].position.z, r.start[b].orientation.x, r.start[b].orientation.y, r.start[b].orientation.z, r.start[b].orientation.w }; memcpy(&start[((size_t)k * nb + b) * 7], q, 28); }
				}
				std::vector<double> ms;
				for (int i = 0; i < iters + 20; i++)
				{
					if (i % 16 == 0) check(ctx, ht_tracker_reset(ctx, 0, B, start.data()));
					const auto t0 = std::chrono::steady_clock::now();
					check(ctx, ht_update_sync(ctx, depth.data(), cams.data(), B, poses.data(), nullptr));
					const double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
					if (i >= 20) ms.push_back(dt);
				}
				printf("{\"call\": \"ht_update_sync\", \"frames_per_call\": %d, \"iters\": %d, \"p50_ms\": %.4f, \"p99_ms\": %.4f, \"mean_ms\": %.4f}\n", B, iters, pct(ms, 0.5), pct(ms, 0.99), [&] { double s = 0; for (double v : ms) s += v; return s / ms.size(); }());
				ht_destroy(ctx);
			}
			return 0;
		}
		if (mode != "track" || argc < 6) return 2;
		auto recs = read_input(argv[4], w, h, nb);
		HandTracker htk(argv[2], argv[3]);                                     // :90 (asset paths are arguments here)
		htk.always_take_cnn = 0;                                               // :91
		htk.microforce = 3.0f;                                                 // :92
		htk.mainthreadpasses = 3;                                              // :93
		htk.load_config("../config.json");                                     // :111 (absent: a no-op, as in the reference)
		if (argc >= 7 && std::string(argv[6]) == "overlapped_wait") { htk.overlapped_update = true; htk.overlapped_wait = true; }      // the reference's job / passes structure on two contexts, the job collected before the passes: the synchronous sequence
		FILE *o = fopen(argv[5], "wb");
		for (auto &r : recs)
		{
			Image<unsigned short> dimage(camera_of(r.cam, w, h), r.depth);
			htk.SetPose(r.start);                                              // every record is an independent tracker start (both models, flags cleared)
			auto segment = HandSegmentVR(dimage);                              // :204
			DCamera hcam = camsub(segment.cam, 4);                             // :206
			auto fake_labels = GatherHandExpectedCNN(r.gt, hcam);              // :207
			auto pose = htk.update(std::move(dimage));                         // :215
			put_poses(o, pose);
			fwrite(htk.cnn_output.data(), 4, htk.cnn_output.size(), o);
			fwrite(fake_labels.cnn_expected.data(), 4, fake_labels.cnn_expected.size(), o);
			auto landmark_outputs = VisualizeHMaps(htk.cnn_output_analysis.hmaps, htk.cnn_input);     // :221
			auto angle_outputs = ToRGB(UpSample(UpSample(UpSample(ToGrayScale(htk.cnn_output_analysis.vmap)))));      // :222
			const float shown = (htk.cnn_input.raster.size() == 4096 && htk.cnn_output_analysis.hmaps.size() == 8 && landmark_outputs.raster.size() == (size_t)128 * 1024 && angle_outputs.raster.size() == (size_t)128 * 128) ? 1.0f : 0.0f;      // :218-222 draw these
			fwrite(&shown, 4, 1, o);
			put_poses(o, htk.handmodel.GetPoseUser());                         // :233 reads the tracked model back through the facade
		}
		fclose(o);
		printf("track: %zu frames, %d bones\n", recs.size(), nb);
		return 0;
	}
	catch (const std::exception &e) { printf("error: %s\n", e.what()); return 1; }
}
