// Compile-and-link check of the C++ compatibility header (and, on a GPU box, a one-frame run).
#include <cstdio>
#include <sstream>
#include "../include/ht_handtrack.hpp"
using namespace ht_mi355x;
int main(int argc, char **argv)
{
	if (argc < 2) { printf("usage: %s model.htfx [weights.cnnb]\n", argv[0]); return 2; }
	try
	{
		HandTracker htk(argv[1], argc > 2 ? argv[2] : "");
		htk.microforce = 3.0f; htk.mainthreadpasses = 3;
		Image<unsigned short> dimage(DCamera({ 64, 64 }, { 164.f, 164.f }, { 32.f, 32.f }, 0.001f));
		for (auto &d : dimage.raster) d = 4000;
		std::vector<float> x(HT_CNN_IN, 0.5f);
		if (argc > 2) { auto y = htk.cnn.Eval(x); printf("cnn out[0]=%g\n", y[0]); auto pose = htk.update(std::move(dimage)); printf("bones=%zu\n", pose.size());
			// a frame that is not 64x64 takes the reference's full-frame route (segmentation for the CNN, cloud from the whole frame)
			Image<unsigned short> big(DCamera({ 128, 128 }, { 163.f, 163.f }, { 64.f, 64.f }, 0.001f));
			for (int y = 0; y < 128; y++) for (int x = 0; x < 128; x++) big.raster[y * 128 + x] = (unsigned short)(((x - 64) * (x - 64) + (y - 70) * (y - 70) < 900) ? 450 + (x + y) / 8 : 4000);
			auto pose2 = htk.update(std::move(big)); printf("full frame bones=%zu cnn_input %dx%d\n", pose2.size(), htk.cnn_input.dim().x, htk.cnn_input.dim().y);
			std::vector<float> t(HT_CNN_OUT, 0.0f); for (int m = 0; m < 24; m++) t[(m < 8 ? 256 * m : 2048 + 16 * (m - 8)) + 3] = 1.0f;
			// caller-built rows across the boundary: the reference's own signatures (physmodel.h:345, physics.h:543)
			std::vector<float3> cloud; for (int i = 0; i < 200; i++) cloud.push_back({ 0.02f * (float)((i % 10) - 5), 0.02f * (float)((i / 10) - 10), 0.45f });
			std::vector<LimitLinear> linears = { LimitLinear(nullptr, &htk.handmodel.rigidbodies[1], { 0, 0, 0.45f }, { 0, 0, 0 }, { 0, 0, 1 }, 0.0f, 0.0f, { -5.0f, 5.0f }) };
			std::vector<LimitAngular> angulars = { LimitAngular(nullptr, &htk.handmodel.rigidbodies[0], { 0, 1, 0 }, 0.0f, 0.0f, FLT_MAX) };
			htk.handmodel.FitPointCloud(cloud, linears, angulars, 3.0f);
			auto rbs = Addresses(htk.handmodel.rigidbodies);
			std::vector<LimitLinear> nailed; for (int ax = 0; ax < 3; ax++) nailed.push_back(LimitLinear(rbs[0], rbs[1], { 0, 0, 0.05f }, { 0, 0, -0.05f }, { ax == 0 ? 1.f : 0.f, ax == 1 ? 1.f : 0.f, ax == 2 ? 1.f : 0.f }));
			PhysicsUpdate(rbs, nailed, angulars);
			printf("rows: fit + update ok, palm z %g\n", htk.handmodel.GetPose()[1].position.z);
			float mse = htk.cnn.Train(x, t, 0.001f); std::ostringstream os; htk.cnn.saveb(os); printf("train mse=%g saved=%zu\n", mse, os.str().size()); }
	}
	catch (const std::exception &e) { printf("error: %s\n", e.what()); return 1; }
	return 0;
}
