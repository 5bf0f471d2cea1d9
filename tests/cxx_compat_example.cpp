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
			float mse = htk.cnn.Train(x, t, 0.001f); std::ostringstream os; htk.cnn.saveb(os); printf("train mse=%g saved=%zu\n", mse, os.str().size()); }
	}
	catch (const std::exception &e) { printf("error: %s\n", e.what()); return 1; }
	return 0;
}
