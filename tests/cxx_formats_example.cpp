// Round trip of the reference's dataset formats through include/ht_formats.hpp (host only; no device call is made).
#include <cmath>
#include <cstdio>
#include "../include/ht_formats.hpp"
using namespace ht_mi355x;
int main(int argc, char **argv)
{
	if (argc < 2) { printf("usage: %s prefix [animbank.pose]\n", argv[0]); return 2; }
	try
	{
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
