// Round trip of the reference's dataset formats through include/ht_formats.hpp (host only; no device call is made).
// Further modes pin the formats on the reference (tests/golden/dataset3/: files its own DepthDataStreamOut wrote and what its own load_dataset / from_json read):
//   read <prefix> <bones> <out.htfx>            load_dataset of ht_formats.hpp, everything it returns, in the array layout of `ref_harness dataset_read`
//   write <dir/> <name>                         DepthDataStreamOut of ht_formats.hpp writes the three-frame set `ref_harness dataset_write` writes
//   header <x.json> <x.pose> <bones> <out.htfx> ReadDatasetInfo + the pose stream, in the layout of `ref_harness dataset_header`
#include <cmath>
#include <cstdio>
#include <unistd.h>
#include "../include/ht_formats.hpp"
#include "../oracle/htfx.h"      // test infrastructure: the fixture container
using namespace ht_mi355x;
static void put_f32(htfx_writer *w, const std::string &n, const std::vector<float> &v, std::vector<uint32_t> dims = {}) { if (dims.empty()) dims = { (uint32_t)v.size() }; htfx_put(w, n.c_str(), HTFX_F32, (uint32_t)dims.size(), dims.data(), v.data()); }
static void put_u16(htfx_writer *w, const std::string &n, const std::vector<unsigned short> &v, std::vector<uint32_t> dims = {}) { if (dims.empty()) dims = { (uint32_t)v.size() }; htfx_put(w, n.c_str(), HTFX_U16, (uint32_t)dims.size(), dims.data(), v.data()); }
static void put_i32(htfx_writer *w, const std::string &n, const std::vector<int> &v) { uint32_t d = (uint32_t)v.size(); htfx_put(w, n.c_str(), HTFX_I32, 1, &d, v.data()); }
static void put_info(htfx_writer *w, const DatasetInfo &d)
{
	put_f32(w, "info_camera", { (float)d.dcamera.dim().x, (float)d.dcamera.dim().y, d.dcamera.focal().x, d.dcamera.focal().y, d.dcamera.principal().x, d.dcamera.principal().y, d.dcamera.depth_scale });
	put_f32(w, "info_mplane", { d.mplane.x, d.mplane.y, d.mplane.z, d.mplane.w });
	put_f32(w, "info_misc", { d.hasir ? 1.0f : 0.0f, (float)d.rgb_dim.x, (float)d.rgb_dim.y, (float)d.feye_dim.x, (float)d.feye_dim.y, d.segment_scale });
	put_u16(w, "info_fname", std::vector<unsigned short>(d.fname.begin(), d.fname.end())); put_u16(w, "info_camtype", std::vector<unsigned short>(d.camtype.begin(), d.camtype.end()));
}
static std::vector<float> flat(const std::vector<Pose> &p) { std::vector<float> o; for (auto &q : p) { o.push_back(q.position.x); o.push_back(q.position.y); o.push_back(q.position.z); o.push_back(q.orientation.x); o.push_back(q.orientation.y); o.push_back(q.orientation.z); o.push_back(q.orientation.w); } return o; }
static float4 unit(float4 q) { const float l = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w); return { q.x / l, q.y / l, q.z / l, q.w / l }; }
static int mode_read(const char *prefix, int bones, const char *outfn)
{
	auto frames = load_dataset(prefix, (unsigned)bones);
	const DatasetInfo dsi = ReadDatasetInfo(std::string(prefix) + ".json");
	htfx_writer w; if (htfx_open(&w, outfn)) return 2;
	put_info(&w, dsi);
	put_i32(&w, "nframes", { (int)frames.size() });
	for (size_t k = 0; k < frames.size(); k++)
	{
		const std::string pre = "f" + std::to_string(k) + "/"; auto &f = frames[k];
		put_u16(&w, pre + "depth", f.depth.raster, { (uint32_t)f.depth.dim().y, (uint32_t)f.depth.dim().x });
		std::vector<unsigned short> ir(f.ir.raster.begin(), f.ir.raster.end()), fe(f.fisheye.raster.begin(), f.fisheye.raster.end()), rgb;
		for (auto &c : f.rgb.raster) { rgb.push_back(c.x); rgb.push_back(c.y); rgb.push_back(c.z); }
		put_u16(&w, pre + "ir", ir); put_u16(&w, pre + "rgb", rgb); put_u16(&w, pre + "fisheye", fe);
		put_f32(&w, pre + "pose", flat(f.pose), { (uint32_t)f.pose.size(), 7 });
		const DCamera &c = f.depth.cam;
		put_f32(&w, pre + "cam", { c.focal().x, c.focal().y, c.principal().x, c.principal().y, c.depth_scale, c.pose.position.x, c.pose.position.y, c.pose.position.z, c.pose.orientation.x, c.pose.orientation.y, c.pose.orientation.z, c.pose.orientation.w });
		put_i32(&w, pre + "fid", { f.fid });
	}
	htfx_close(&w);
	return 0;
}
static int mode_write(const char *dir, const char *name)      // the set of ref_harness.cpp mode_dataset_write, value for value
{
	if (chdir(dir)) return 2;
	DatasetInfo dsi; dsi.dcamera = DCamera({ 16, 12 }, { 14.5f, 14.25f }, { 8.25f, 5.75f }, 0.000125f); dsi.mplane = { 0.0f, 0.6f, 0.8f, -0.35f }; dsi.fname = name; dsi.camtype = "synthetic";
	dsi.hasir = false; dsi.rgb_dim = { 8, 6 }; dsi.feye_dim = { 4, 2 }; dsi.segment_scale = 0.165f;
	DepthDataStreamOut out(dsi);
	out.AddRGB().AddFishEye();
	for (int k = 0; k < 3; k++)
	{
		Image<unsigned short> d(dsi.dcamera); Image<unsigned char> ir(dsi.dcamera); Image<byte3> rgb(DCamera(dsi.rgb_dim, { 0, 0 }, { 0, 0 }, 0.0f)); Image<unsigned char> fe(DCamera(dsi.feye_dim, { 0, 0 }, { 0, 0 }, 0.0f));
		for (size_t i = 0; i < d.raster.size(); i++) { d.raster[i] = (unsigned short)(1000 * k + 7 * i + 1); ir.raster[i] = (unsigned char)(3 * i + k); }
		for (size_t i = 0; i < rgb.raster.size(); i++) { rgb.raster[i].x = (unsigned char)(i + k); rgb.raster[i].y = (unsigned char)(2 * i + k); rgb.raster[i].z = (unsigned char)(255 - i); }
		for (size_t i = 0; i < fe.raster.size(); i++) fe.raster[i] = (unsigned char)(17 * i + 5 * k);
		std::vector<Pose> p(17);
		for (int b = 0; b < 17; b++) { p[b].position = { 0.0123456f * b - 0.1f, -0.125f * k + 1e-5f * b, 0.3f + 0.001f * k }; p[b].orientation = unit({ 0.1f * b, -0.3f, 0.25f * k, 1.0f }); }
		out.SaveFrame(MakeFrame(d, p, ir, rgb, fe));
	}
	return 0;
}
static int mode_header(const char *jsonfn, const char *posefn, int bones, const char *outfn)
{
	const DatasetInfo dsi = ReadDatasetInfo(jsonfn);
	htfx_writer w; if (htfx_open(&w, outfn)) return 2;
	put_info(&w, dsi);
	std::ifstream in(posefn);
	std::vector<float> all; int n = 0;
	for (;;) { std::vector<Pose> p((size_t)bones); bool ok = true; for (auto &q : p) if (!(in >> q)) { ok = false; break; } if (!ok) break; for (float v : flat(p)) all.push_back(v); n++; }
	put_f32(&w, "poses", all, { (uint32_t)n, (uint32_t)bones, 7 });
	htfx_close(&w);
	return 0;
}
int main(int argc, char **argv)
{
	if (argc < 2) { printf("usage: %s prefix [animbank.pose] | read <prefix> <bones> <out.htfx> | write <dir/> <name> | header <json> <pose> <bones> <out.htfx>\n", argv[0]); return 2; }
	try
	{
		if (argc == 5 && std::string(argv[1]) == "read") return mode_read(argv[2], atoi(argv[3]), argv[4]);
		if (argc == 4 && std::string(argv[1]) == "write") return mode_write(argv[2], argv[3]);
		if (argc == 6 && std::string(argv[1]) == "header") return mode_header(argv[2], argv[3], atoi(argv[4]), argv[5]);
		DatasetInfo dsi; dsi.dcamera = DCamera({ 8, 6 }, { 12.5f, 12.5f }, { 4.f, 3.f }, 0.001f); dsi.mplane = { 0, 0, 1, -0.5f }; dsi.fname = argv[1]; dsi.camtype = "synthetic"; dsi.segment_scale = 0.17f;
		std::vector<std::vector<Pose>> poses;
		{
			DepthDataStreamOut out(dsi);
			for (int k = 0; k < 3; k++)
			{
				Image<unsigned short> d(dsi.dcamera); Image<unsigned char> ir(dsi.dcamera);
				for (size_t i = 0; i < d.raster.size(); i++) { d.raster[i] = (unsigned short)(1000 * k + i); ir.raster[i] = (unsigned char)(i + k); }
				std::vector<Pose> p(17);
				for (int b = 0; b < 17; b++) { p[b].position = { 0.01f * b, -0.125f * k, 0.3f }; p[b].orientation = { 0.5f, -0.5f, 0.5f, 0.5f }; }
				poses.push_back(p);
				out.SaveFrame(d, ir, p);
			}
		}
		auto frames = load_dataset(argv[1], 17);
		if (frames.size() != 3) { printf("FAIL frames %zu\n", frames.size()); return 1; }
		for (int k = 0; k < 3; k++)
		{
			if (frames[k].depth.dim().x != 8 || frames[k].depth.raster[5] != 1000 * k + 5 || frames[k].ir.raster[7] != 7 + k) { printf("FAIL raster %d\n", k); return 1; }
			for (int b = 0; b < 17; b++) if (std::fabs(frames[k].pose[b].position.x - poses[k][b].position.x) > 1e-6f || frames[k].pose[b].orientation.y != -0.5f) { printf("FAIL pose %d %d\n", k, b); return 1; }
			if (frames[k].depth.cam.focal().x != 12.5f || frames[k].depth.cam.depth_scale != 0.001f) { printf("FAIL cam\n"); return 1; }
		}
		auto bank = LoadAnimBank(std::string(argv[1]) + ".pose", 17);
		if (bank.size() != 3 || bank[2][16].position.x != poses[2][16].position.x) { printf("FAIL bank %zu\n", bank.size()); return 1; }
		if (argc > 2) { auto ref = LoadAnimBank(argv[2], 17); printf("reference bank rows=%zu q0.w=%g\n", ref.size(), ref.empty() ? 0.f : ref[0][0].orientation.w); if (ref.size() != 2336) return 1; }
		printf("OK\n");
	}
	catch (const std::exception &e) { printf("error: %s\n", e.what()); return 1; }
	return 0;
}
