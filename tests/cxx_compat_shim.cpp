// The tracker side of the reference application's translation unit, compiled against include/compat/ (tests/test_cxx_driver.py stages that directory's include/ and
// third_party/ where the reference's stand and this file under synthetic-hand-tracker/): the include block is synthetic-tracker.cpp:15-24 without its two window headers
// (glwin.h, misc_gl.h: the reference's own GL code, out of scope), LoadAnimBank is the application's own function of that name (:39-55, restated), and main() makes the
// tracker-side calls of :90-96, 111, 139, 204-215, 233 in order.  Host only: it is compiled and linked, and run without a device up to the first device call.
#include <exception>
#include <iostream>
#include <fstream>
#include <cctype>    // std::tolower
#include <future>
#include <sstream>

#include "../third_party/geometric.h"
#include "../third_party/mesh.h"
#include "../third_party/misc.h"
#include "../third_party/cnn.h"
#include "../include/misc_image.h"
#include "../include/physmodel.h"
#include "../include/handtrack.h"

std::vector<std::vector<Pose>> LoadAnimBank(std::string filename, size_t pose_array_size)      // the application's own (synthetic-tracker.cpp:39-55): must not collide with the binding's
{
	std::vector<std::vector<Pose>> animbank;
	std::ifstream pfile(filename);
	if (!pfile.is_open()) throw "unable to open animation bank file";
	std::string line;
	while (std::getline(pfile, line) && line != "")
	{
		std::vector<Pose> pose(pose_array_size);
		std::stringstream linestream(line);
		for (auto &p : pose) linestream >> p;
		animbank.push_back(pose);
	}
	return animbank;
}

int main(int argc, char **argv)
{
	if (argc < 2) { printf("compat shim: compiled and linked\n"); return 0; }
	try
	{
		auto animbank = LoadAnimBank(argv[1], 17);
		printf("animbank rows=%zu\n", animbank.size());
		if (argc < 4) return 0;
		HandTracker htk(argv[2], argv[3]);
		htk.always_take_cnn = 0; htk.microforce = 3.0f; htk.mainthreadpasses = 3;
		PhysModel fakehand = LoadHandModel(argv[2]);
		htk.load_config("../config.json");
		fakehand.SetPose(animbank[0]);
		DCamera dcam({ 320, 240 }, { 305, 305 }, { 160, 120 }, 0.001f);
		Image<unsigned short> dimage(dcam);
		auto segment = HandSegmentVR(dimage);
		DCamera hcam = camsub(segment.cam, 4);
		auto fake_labels = GatherHandExpectedCNN(fakehand.GetPose(), hcam);
		auto pose = htk.update(std::move(dimage));
		printf("update: %zu poses, %zu meshes\n", pose.size(), htk.handmodel.GetMeshes(true).size());
		return 0;
	}
	catch (const char *e) { printf("error: %s\n", e); return 1; }
	catch (const std::exception &e) { printf("error: %s\n", e.what()); return 1; }
}
