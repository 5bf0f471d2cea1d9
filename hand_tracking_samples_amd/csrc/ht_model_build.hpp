// ht_model_build.hpp -- init-time model build: model JSON -> the named arrays the context loader consumes.
#pragma once
#include <stdint.h>
#include <map>
#include <string>
#include <vector>

// one named array of the HTFX container (dtype 0 = f32, 1 = i32, 2 = u16, 3 = u8)
struct fx_arr
{
	uint32_t dtype = 0, ndim = 0, dims[4] = { 0, 0, 0, 0 };
	std::vector<unsigned char> data;
	const float *f() const { return (const float *)data.data(); }
	const int *i() const { return (const int *)data.data(); }
};
typedef std::map<std::string, fx_arr> fx_map;

bool fx_load(const char *path, fx_map &out);
bool fx_save(const char *path, const fx_map &in);

// flags for ht_build_model
enum { HT_BUILD_HAND_TWEAKS = 1 };      // LoadHandModel()'s post-processing (handtrack.h:347-366)

// Parses a PhysModel JSON ("controlcages", "joints") and builds bodies, joints, physics constants and the UnibodyFit proxy.
bool ht_build_model(const char *json_path, int flags, fx_map &out, std::string &err);

// numeric top-level members of a JSON object file, as text
bool ht_json_top_level(const char *path, std::map<std::string, std::string> &numbers, std::string &err);
