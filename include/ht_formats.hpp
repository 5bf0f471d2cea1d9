// ht_formats.hpp -- the reference's on-disk formats on either side of the tracking path, header-only host C++.
//
//   .pose   ascii, one frame per line, 7 floats (position xyz, orientation xyzw) per bone
//           writer DepthDataStreamOut::SaveFrame (include/dataset.h:91-100), readers LoadAnimBank (synthetic-hand-tracker/
//           synthetic-tracker.cpp:39-55) and load_dataset (dataset.h:144-146)
//   .rs     raw little-endian u16 depth frames back to back, width x height from the .json header (dataset.h:62-93,118-163)
//   .ir     raw u8 frames, same size (optional)
//   .json   DatasetInfo header: dcamera {dims, focal, principal, depth_scale}, mplane, fname, camtype, hasir, rgb_dim, feyedim,
//           segment_scale (dataset.h:21-37, misc_image.h:57)
//   .cnnb   raw fp32 weights in layer order (cnn.h:97-98,288,454,590-592): see CNN::loadb in ht_handtrack.hpp
// Types come from ht_handtrack.hpp (Pose, DCamera, Image<T>).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include "ht_handtrack.hpp"

namespace ht_mi355x {

// ---- .pose ---------------------------------------------------------------------------------------------------------------
inline std::istream &operator>>(std::istream &in, Pose &p) { return in >> p.position.x >> p.position.y >> p.position.z >> p.orientation.x >> p.orientation.y >> p.orientation.z >> p.orientation.w; }      // geometric.h:139
inline std::ostream &operator<<(std::ostream &out, const Pose &p)      // geometric.h:138
{
	return out << p.position.x << " " << p.position.y << " " << p.position.z << " " << p.orientation.x << " " << p.orientation.y << " " << p.orientation.z << " " << p.orientation.w;
}
// synthetic-tracker.cpp:39-55: one line = one frame, reading stops at the first empty line
inline std::vector<std::vector<Pose>> LoadAnimBank(const std::string &filename, size_t pose_array_size)
{
	std::vector<std::vector<Pose>> animbank;
	std::ifstream pfile(filename);
	if (!pfile.is_open()) throw std::runtime_error("unable to open animation bank file");
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
// the line DepthDataStreamOut::SaveFrame writes (dataset.h:97-99): "px py pz  qx qy qz qw   " per bone
inline void WritePoseLine(std::ostream &out, const std::vector<Pose> &pose)
{
	for (const auto &p : pose) out << p.position.x << " " << p.position.y << " " << p.position.z << "  " << p.orientation.x << " " << p.orientation.y << " " << p.orientation.z << " " << p.orientation.w << "   ";
	out << "\n";
}

// ---- .json header ----------------------------------------------------------------------------------------------------------
namespace detail {
struct jv { char kind = 'n'; std::string text; std::vector<jv> arr; std::vector<std::pair<std::string, jv>> obj;      // kinds: n(ull) b(ool) #(number) s(tring) a(rray) o(bject)
	const jv *get(const std::string &k) const { for (auto &kv : obj) if (kv.first == k) return &kv.second; return nullptr; }
	double num(double def = 0) const { return kind == '#' ? strtod(text.c_str(), nullptr) : def; } };
struct jparse
{
	const char *p, *e;
	void ws() { while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
	[[noreturn]] void fail(const char *m) { throw std::runtime_error(std::string("json parse error - ") + m); }
	std::string str() { std::string s; p++; while (p < e && *p != '"') { if (*p == '\\' && p + 1 < e) p++; s.push_back(*p++); } if (p >= e) fail("unterminated string"); p++; return s; }
	jv value(int depth = 0)
	{
		if (depth > 64) fail("nesting too deep");
		jv v; ws(); if (p >= e) fail("unexpected end");
		if (*p == '{') { v.kind = 'o'; p++; ws(); if (p < e && *p == '}') { p++; return v; }
			for (;;) { ws(); if (p >= e || *p != '"') fail("key expected"); std::string k = str(); ws(); if (p >= e || *p != ':') fail("':' expected"); p++; v.obj.emplace_back(k, value(depth + 1)); ws();
				if (p < e && *p == ',') { p++; continue; } if (p < e && *p == '}') { p++; return v; } fail("',' or '}' expected"); } }
		if (*p == '[') { v.kind = 'a'; p++; ws(); if (p < e && *p == ']') { p++; return v; }
			for (;;) { v.arr.push_back(value(depth + 1)); ws(); if (p < e && *p == ',') { p++; continue; } if (p < e && *p == ']') { p++; return v; } fail("',' or ']' expected"); } }
		if (*p == '"') { v.kind = 's'; v.text = str(); return v; }
		if (e - p >= 4 && !std::string(p, 4).compare("true")) { v.kind = 'b'; v.text = "1"; p += 4; return v; }
		if (e - p >= 5 && !std::string(p, 5).compare("false")) { v.kind = 'b'; v.text = "0"; p += 5; return v; }
		if (e - p >= 4 && !std::string(p, 4).compare("null")) { p += 4; return v; }
		const char *s = p; while (p < e && (std::string("+-.eE").find(*p) != std::string::npos || (*p >= '0' && *p <= '9'))) p++;
		if (p == s) fail("unexpected character"); v.kind = '#'; v.text.assign(s, p); return v;
	}
};
inline jv parse_json_file(const std::string &path)
{
	std::ifstream in(path, std::ios::binary); if (!in.is_open()) throw std::runtime_error("file not found: " + path);
	std::stringstream ss; ss << in.rdbuf(); std::string text = ss.str();
	jparse jp{ text.data(), text.data() + text.size() }; return jp.value();
}
inline float jf(const jv *v, size_t i, float def = 0) { return v && v->kind == 'a' && i < v->arr.size() ? (float)strtof(v->arr[i].text.c_str(), nullptr) : def; }
}  // namespace detail

struct DatasetInfo      // dataset.h:21-37
{
	DCamera dcamera; float4 mplane{ 0, 0, 0, 0 }; std::string fname, camtype; bool hasir = false; int2 rgb_dim{ 0, 0 }, feye_dim{ 0, 0 }; float segment_scale = 0;
};
inline DatasetInfo ReadDatasetInfo(const std::string &jsonfile)
{
	const detail::jv root = detail::parse_json_file(jsonfile);
	DatasetInfo d;
	if (const detail::jv *c = root.get("dcamera"))      // visit_fields(DCamera) misc_image.h:57
	{
		const detail::jv *dims = c->get("dims"), *focal = c->get("focal"), *pr = c->get("principal"), *ds = c->get("depth_scale");
		d.dcamera = DCamera({ (int)detail::jf(dims, 0), (int)detail::jf(dims, 1) }, { detail::jf(focal, 0), detail::jf(focal, 1) }, { detail::jf(pr, 0), detail::jf(pr, 1) }, ds ? (float)strtof(ds->text.c_str(), nullptr) : 0.0f);
	}
	const detail::jv *m = root.get("mplane"); d.mplane = { detail::jf(m, 0), detail::jf(m, 1), detail::jf(m, 2), detail::jf(m, 3) };
	if (const detail::jv *v = root.get("fname")) d.fname = v->text;
	if (const detail::jv *v = root.get("camtype")) d.camtype = v->text;
	if (const detail::jv *v = root.get("hasir")) d.hasir = v->kind == 'b' ? v->text == "1" : v->num() != 0;
	const detail::jv *r = root.get("rgb_dim"), *f = root.get("feyedim"); d.rgb_dim = { (int)detail::jf(r, 0), (int)detail::jf(r, 1) }; d.feye_dim = { (int)detail::jf(f, 0), (int)detail::jf(f, 1) };
	if (const detail::jv *v = root.get("segment_scale")) d.segment_scale = (float)strtof(v->text.c_str(), nullptr);
	return d;
}
inline void WriteDatasetInfo(const std::string &jsonfile, const DatasetInfo &d)
{
	std::ofstream o(jsonfile);
	o << "{\n  \"dcamera\": {\n    \"dims\": [" << d.dcamera.dim().x << "," << d.dcamera.dim().y << "],\n    \"focal\": [" << d.dcamera.focal().x << "," << d.dcamera.focal().y << "],\n    \"principal\": ["
	  << d.dcamera.principal().x << "," << d.dcamera.principal().y << "],\n    \"depth_scale\": " << d.dcamera.depth_scale << "\n  },\n  \"mplane\": [" << d.mplane.x << "," << d.mplane.y << "," << d.mplane.z << "," << d.mplane.w
	  << "],\n  \"fname\": \"" << d.fname << "\",\n  \"camtype\": \"" << d.camtype << "\",\n  \"hasir\": " << (d.hasir ? "true" : "false") << ",\n  \"rgb_dim\": [" << d.rgb_dim.x << "," << d.rgb_dim.y << "],\n  \"feyedim\": ["
	  << d.feye_dim.x << "," << d.feye_dim.y << "],\n  \"segment_scale\": " << d.segment_scale << "\n}\n";
}

// ---- .rs / .ir / .pose datasets ------------------------------------------------------------------------------------------
struct Frame { Image<unsigned short> depth; std::vector<Pose> pose; Image<unsigned char> ir; std::string fname; int fid = 0; };      // dataset.h:40-51 (depth, pose, ir)
// dataset.h:118-163
inline std::vector<Frame> load_dataset(const std::string &bname, unsigned int pose_array_size)
{
	std::ifstream file_in_depth(bname + ".rs", std::ios_base::binary | std::ios_base::in);
	if (!file_in_depth.is_open()) throw std::runtime_error("unable to open .rs file");
	const DatasetInfo dsi = ReadDatasetInfo(bname + ".json");
	std::ifstream file_in_pose(bname + ".pose", std::ios_base::in), file_in_ir(bname + ".ir", std::ios_base::in | std::ios_base::binary);
	const size_t npix = (size_t)dsi.dcamera.dim().x * dsi.dcamera.dim().y;
	std::vector<Frame> frames;
	for (int k = 0;; k++)
	{
		std::vector<unsigned short> dbuf(npix); std::vector<unsigned char> ibuf(npix, (unsigned char)0);
		if (!file_in_depth.read((char *)dbuf.data(), (std::streamsize)(npix * sizeof(unsigned short)))) break;
		if (dsi.hasir && !file_in_depth.read((char *)ibuf.data(), (std::streamsize)npix)) break;      // interleaved depth and ir (deprecated layout)
		if (file_in_ir.is_open()) file_in_ir.read((char *)ibuf.data(), (std::streamsize)npix);
		std::vector<Pose> pose(pose_array_size);
		if (file_in_pose.is_open()) for (auto &p : pose) file_in_pose >> p;
		Frame f; f.depth = Image<unsigned short>(dsi.dcamera, std::move(dbuf)); f.ir = Image<unsigned char>(dsi.dcamera, std::move(ibuf)); f.pose = std::move(pose); f.fname = bname; f.fid = k;
		frames.push_back(std::move(f));
	}
	return frames;
}
// dataset.h:62-104 (depth, ir, pose streams + the .json header)
struct DepthDataStreamOut
{
	std::string prefix; std::ofstream file_out_depth, file_out_poses, file_out_ir;
	explicit DepthDataStreamOut(const std::string &prefix_) : prefix(prefix_)
	{
		file_out_depth.open(prefix + ".rs", std::ios::binary | std::ios::trunc); file_out_ir.open(prefix + ".ir", std::ios::binary | std::ios::trunc); file_out_poses.open(prefix + ".pose", std::ios::out | std::ios::trunc);
	}
	explicit DepthDataStreamOut(const DatasetInfo &dsi) : DepthDataStreamOut(dsi.fname) { WriteDatasetInfo(dsi.fname + ".json", dsi); }
	void SaveFrame(const Image<unsigned short> &dimage, const Image<unsigned char> &irimage, const std::vector<Pose> &pose)
	{
		if (!file_out_depth.is_open()) throw std::runtime_error("hey file wasn't opened");
		file_out_depth.write((const char *)dimage.raster.data(), (std::streamsize)(dimage.raster.size() * sizeof(unsigned short)));
		file_out_ir.write((const char *)irimage.raster.data(), (std::streamsize)irimage.raster.size());
		WritePoseLine(file_out_poses, pose);
	}
};
}  // namespace ht_mi355x
