// ht_formats.hpp -- the reference's on-disk formats on either side of the tracking path, header-only host C++.
//
//   .pose   ascii, one frame per line, 7 floats (position xyz, orientation xyzw) per bone
//           writer DepthDataStreamOut::SaveFrame (include/dataset.h:91-100), readers LoadAnimBank (synthetic-hand-tracker/
//           synthetic-tracker.cpp:39-55) and load_dataset (dataset.h:144-146)
//   .rs     raw little-endian u16 depth frames back to back, width x height from the .json header (dataset.h:62-93,118-163)
//   .ir     raw u8 frames, same size (optional)
//   .rgb    raw 3 x u8 colour frames of the header's rgb_dim, .feye raw u8 fish-eye frames of its feyedim (both optional, dataset.h:77-78,102-105,134-139)
//   .json   DatasetInfo header: dcamera {dims, focal, principal, depth_scale}, mplane, fname, camtype, hasir, rgb_dim, feyedim,
//           segment_scale (dataset.h:21-37, misc_image.h:57)
//   .cnnb   raw fp32 weights in layer order (cnn.h:97-98,288,454,590-592): see CNN::loadb in ht_handtrack.hpp
// Types come from ht_handtrack.hpp (Pose, DCamera, Image<T>).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include "ht_handtrack.hpp"
#include "ht_json.hpp"

namespace ht_mi355x {

// ---- .pose ---------------------------------------------------------------------------------------------------------------
// The format (what DepthDataStreamOut::SaveFrame writes, include/dataset.h:97-99, and what synthetic-tracker.cpp:39-55 / dataset.h:144-146 read):
// plain text, numbers separated by blanks, 7 per bone in the order position x y z, orientation x y z w; an animation bank holds one frame
// per line and ends at the first empty line, a dataset's pose file is read as one stream of numbers, frame after frame.
namespace detail {
// the whole file as bytes; *found tells whether it could be opened
inline std::string file_bytes(const std::string &path, bool *found = nullptr)
{
	std::string bytes;
	FILE *f = std::fopen(path.c_str(), "rb");
	if (found) *found = f != nullptr;
	if (!f) return bytes;
	char chunk[1 << 16];
	for (size_t got; (got = std::fread(chunk, 1, sizeof chunk, f)) > 0;) bytes.append(chunk, got);
	std::fclose(f);
	return bytes;
}
// Reads up to `want` numbers from the text in [at, end) into out[]; stops early where no number follows.  Returns how many were read.
inline size_t take_numbers(const char *&at, const char *end, float *out, size_t want)
{
	size_t n = 0;
	while (n < want)
	{
		while (at < end && (*at == ' ' || *at == '\t' || *at == '\r' || *at == '\n')) at++;
		if (at >= end) break;
		char *stop = nullptr;
		const float v = std::strtof(at, &stop);      // the text ends in a NUL (std::string), so strtof cannot run past `end`
		if (stop == at) break;
		out[n++] = v; at = stop;
	}
	return n;
}
// fills as many of the bones as there are numbers for; a bone that is cut short keeps the numbers it got (the rest stay at the identity pose)
inline size_t take_poses(const char *&at, const char *end, std::vector<Pose> &bones)
{
	size_t whole = 0;
	for (auto &b : bones)
	{
		float v[7] = { b.position.x, b.position.y, b.position.z, b.orientation.x, b.orientation.y, b.orientation.z, b.orientation.w };
		const size_t got = take_numbers(at, end, v, 7);
		b.position = { v[0], v[1], v[2] }; b.orientation = { v[3], v[4], v[5], v[6] };
		if (got < 7) break;
		whole++;
	}
	return whole;
}
}  // namespace detail

inline std::istream &operator>>(std::istream &in, Pose &p) { return in >> p.position.x >> p.position.y >> p.position.z >> p.orientation.x >> p.orientation.y >> p.orientation.z >> p.orientation.w; }      // geometric.h:139
inline std::ostream &operator<<(std::ostream &out, const Pose &p)      // geometric.h:138
{
	return out << p.position.x << " " << p.position.y << " " << p.position.z << " " << p.orientation.x << " " << p.orientation.y << " " << p.orientation.z << " " << p.orientation.w;
}
// Same call as synthetic-tracker.cpp:39 (one line = one frame of pose_array_size bones; the bank ends at the first empty line or at the end of the file)
inline std::vector<std::vector<Pose>> LoadAnimBank(const std::string &filename, size_t pose_array_size)
{
	bool found = false;
	const std::string text = detail::file_bytes(filename, &found);
	if (!found) throw std::runtime_error("animation bank not found: " + filename);
	std::vector<std::vector<Pose>> bank;
	for (size_t from = 0; from < text.size();)
	{
		size_t to = text.find('\n', from);
		if (to == std::string::npos) to = text.size();
		size_t len = to - from;
		if (len && text[from + len - 1] == '\r') len--;      // a bank edited on Windows
		if (len == 0) break;
		const std::string line = text.substr(from, len);      // own NUL-terminated copy: numbers never run into the next line
		const char *at = line.c_str();
		bank.emplace_back(pose_array_size);
		detail::take_poses(at, at + line.size(), bank.back());
		from = to + 1;
	}
	return bank;
}
// one line of a .pose file: "px py pz  qx qy qz qw   " per bone (the spacing DepthDataStreamOut::SaveFrame uses, dataset.h:97-99)
inline void WritePoseLine(std::ostream &out, const std::vector<Pose> &pose)
{
	for (const auto &p : pose) out << p.position.x << " " << p.position.y << " " << p.position.z << "  " << p.orientation.x << " " << p.orientation.y << " " << p.orientation.z << " " << p.orientation.w << "   ";
	out << "\n";
}

// ---- .json header (parsed by include/ht_json.hpp, the product's one JSON reader; a syntax error is thrown here as the reference's reader throws) ----
namespace detail {
inline ht_json::jnode parse_json_file(const std::string &path)
{
	std::ifstream in(path, std::ios::binary); if (!in.is_open()) throw std::runtime_error("file not found: " + path);
	std::stringstream ss; ss << in.rdbuf(); const std::string text = ss.str();
	ht_json::jnode root; std::string err;
	if (!ht_json::parse(text.data(), text.size(), root, &err)) throw std::runtime_error("json parse error - " + err);
	return root;
}
inline float jf(const ht_json::jnode *v, size_t i, float def = 0) { return v && v->kind == ht_json::jnode::ARR && i < v->items.size() ? v->items[i].as_float() : def; }
}  // namespace detail

struct DatasetInfo      // dataset.h:21-37
{
	DCamera dcamera; float4 mplane{ 0, 0, 0, 0 }; std::string fname, camtype; bool hasir = false; int2 rgb_dim{ 0, 0 }, feye_dim{ 0, 0 }; float segment_scale = 0;
};
inline DatasetInfo ReadDatasetInfo(const std::string &jsonfile)
{
	const ht_json::jnode root = detail::parse_json_file(jsonfile);
	DatasetInfo d;
	if (const ht_json::jnode *c = root.get("dcamera"))      // visit_fields(DCamera) misc_image.h:57
	{
		const ht_json::jnode *dims = c->get("dims"), *focal = c->get("focal"), *pr = c->get("principal"), *ds = c->get("depth_scale");
		d.dcamera = DCamera({ (int)detail::jf(dims, 0), (int)detail::jf(dims, 1) }, { detail::jf(focal, 0), detail::jf(focal, 1) }, { detail::jf(pr, 0), detail::jf(pr, 1) }, ds ? (float)strtof(ds->text.c_str(), nullptr) : 0.0f);
	}
	const ht_json::jnode *m = root.get("mplane"); d.mplane = { detail::jf(m, 0), detail::jf(m, 1), detail::jf(m, 2), detail::jf(m, 3) };
	if (const ht_json::jnode *v = root.get("fname")) d.fname = v->text;
	if (const ht_json::jnode *v = root.get("camtype")) d.camtype = v->text;
	if (const ht_json::jnode *v = root.get("hasir")) d.hasir = v->kind == ht_json::jnode::BOOL ? v->text == "1" : (v->kind == ht_json::jnode::NUM && strtod(v->text.c_str(), nullptr) != 0);
	const ht_json::jnode *r = root.get("rgb_dim"), *f = root.get("feyedim"); d.rgb_dim = { (int)detail::jf(r, 0), (int)detail::jf(r, 1) }; d.feye_dim = { (int)detail::jf(f, 0), (int)detail::jf(f, 1) };
	if (const ht_json::jnode *v = root.get("segment_scale")) d.segment_scale = (float)strtof(v->text.c_str(), nullptr);
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
// A dataset is four files with one base name (dataset.h:62-163): base.json = the DatasetInfo header above; base.rs = the depth frames, raw
// little-endian u16, width x height of the header's camera, back to back (with the deprecated hasir flag each depth frame is followed by its
// u8 infra-red frame in the same file); base.ir = the infra-red frames, raw u8, same size, optional; base.pose = the poses as text.
// byte3 (linalg.h:355) comes with the image helpers of ht_handtrack.hpp
struct Frame      // dataset.h:40-51
{
	Image<unsigned short> depth; std::vector<Pose> pose, startpose; float4 mplane{ 0, 0, 0, 3.402823466e+38f }; Image<unsigned char> ir; std::string fname; int fid = 0;
	Image<byte3> rgb; Image<unsigned char> fisheye;
};
inline Frame MakeFrame(Image<unsigned short> depth, std::vector<Pose> pose) { Frame f; f.depth = std::move(depth); f.pose = f.startpose = std::move(pose); return f; }      // dataset.h:53-60
inline Frame MakeFrame(Image<unsigned short> depth, std::vector<Pose> pose, float4 mplane) { Frame f = MakeFrame(std::move(depth), std::move(pose)); f.mplane = mplane; return f; }
inline Frame MakeFrame(Image<unsigned short> depth, std::vector<Pose> pose, Image<unsigned char> ir, Image<byte3> rgb, Image<unsigned char> fisheye)
{
	Frame f = MakeFrame(std::move(depth), std::move(pose)); f.ir = std::move(ir); f.rgb = std::move(rgb); f.fisheye = std::move(fisheye); return f;
}

// Same call as dataset.h:118: every complete frame of the dataset, in file order.
inline std::vector<Frame> load_dataset(const std::string &bname, unsigned int pose_array_size)
{
	bool have_rs = false, have_ir = false, have_pose = false, have_rgb = false, have_feye = false;
	const std::string rs = detail::file_bytes(bname + ".rs", &have_rs);
	if (!have_rs) throw std::runtime_error("dataset has no depth file: " + bname + ".rs");
	const DatasetInfo info = ReadDatasetInfo(bname + ".json");
	const std::string irb = detail::file_bytes(bname + ".ir", &have_ir), posetext = detail::file_bytes(bname + ".pose", &have_pose);
	const std::string rgbb = detail::file_bytes(bname + ".rgb", &have_rgb), feyeb = detail::file_bytes(bname + ".feye", &have_feye);
	const size_t rgb_px = (size_t)(info.rgb_dim.x > 0 ? info.rgb_dim.x : 0) * (size_t)(info.rgb_dim.y > 0 ? info.rgb_dim.y : 0), feye_px = (size_t)(info.feye_dim.x > 0 ? info.feye_dim.x : 0) * (size_t)(info.feye_dim.y > 0 ? info.feye_dim.y : 0);
	const size_t px = (size_t)info.dcamera.dim().x * (size_t)info.dcamera.dim().y;
	const size_t depth_bytes = px * sizeof(unsigned short), record = depth_bytes + (info.hasir ? px : 0);
	const size_t count = record ? rs.size() / record : 0;
	const char *pose_at = posetext.c_str(), *pose_end = pose_at + posetext.size();
	std::vector<Frame> frames(count);
	for (size_t k = 0; k < count; k++)
	{
		Frame &f = frames[k];
		const char *rec = rs.data() + k * record;
		std::vector<unsigned short> d(px);
		std::memcpy(d.data(), rec, depth_bytes);
		std::vector<unsigned char> ir(px, (unsigned char)0);
		if (info.hasir) std::memcpy(ir.data(), rec + depth_bytes, px);
		if (have_ir && (k + 1) * px <= irb.size()) std::memcpy(ir.data(), irb.data() + k * px, px);      // the separate file wins, as it is read last in the reference
		f.depth = Image<unsigned short>(info.dcamera, std::move(d));
		f.ir = Image<unsigned char>(info.dcamera, std::move(ir));
		f.pose.assign(pose_array_size, Pose());
		if (have_pose) detail::take_poses(pose_at, pose_end, f.pose);
		f.startpose = f.pose;
		// colour and fish-eye frames have the header's rgb_dim / feyedim (dataset.h:134-139); a frame the stream no longer holds stays black, as a short read leaves it in the reference
		f.rgb = Image<byte3>(DCamera(info.rgb_dim, { 0, 0 }, { 0, 0 }, 0.0f)); f.fisheye = Image<unsigned char>(DCamera(info.feye_dim, { 0, 0 }, { 0, 0 }, 0.0f));
		if (have_rgb && rgb_px && (k + 1) * rgb_px * 3 <= rgbb.size()) std::memcpy(f.rgb.raster.data(), rgbb.data() + k * rgb_px * 3, rgb_px * 3);
		if (have_feye && feye_px && (k + 1) * feye_px <= feyeb.size()) std::memcpy(f.fisheye.raster.data(), feyeb.data() + k * feye_px, feye_px);
		f.fname = bname; f.fid = (int)k;
	}
	return frames;
}
// Same surface as dataset.h:62-104: opened on a base name (or on a header, which is then written), SaveFrame appends one frame to all three streams.
class DepthDataStreamOut
{
	std::FILE *depth_out = nullptr, *ir_out = nullptr, *pose_out = nullptr, *rgb_out = nullptr, *feye_out = nullptr;
	static void put(std::FILE *f, const void *p, size_t n, const char *what) { if (n && std::fwrite(p, 1, n, f) != n) throw std::runtime_error(std::string("dataset: short write to the ") + what + " stream"); }
public:
	const std::string prefix;
	explicit DepthDataStreamOut(const std::string &base) : prefix(base)
	{
		depth_out = std::fopen((base + ".rs").c_str(), "wb"); ir_out = std::fopen((base + ".ir").c_str(), "wb"); pose_out = std::fopen((base + ".pose").c_str(), "w");
	}
	explicit DepthDataStreamOut(const DatasetInfo &header) : DepthDataStreamOut(header.fname) { WriteDatasetInfo(header.fname + ".json", header); }
	DepthDataStreamOut(const DepthDataStreamOut &) = delete;
	DepthDataStreamOut &operator=(const DepthDataStreamOut &) = delete;
	~DepthDataStreamOut() { for (std::FILE *f : { depth_out, ir_out, pose_out, rgb_out, feye_out }) if (f) std::fclose(f); }
	DepthDataStreamOut &AddRGB() { if (!rgb_out) rgb_out = std::fopen((prefix + ".rgb").c_str(), "wb"); return *this; }              // dataset.h:77
	DepthDataStreamOut &AddFishEye() { if (!feye_out) feye_out = std::fopen((prefix + ".feye").c_str(), "wb"); return *this; }       // dataset.h:78
	void SaveFrame(const Image<unsigned short> &dimage, const Image<unsigned char> &irimage, const std::vector<Pose> &pose)
	{
		if (!depth_out || !ir_out || !pose_out) throw std::runtime_error("dataset: cannot write " + prefix + ".rs/.ir/.pose");
		put(depth_out, dimage.raster.data(), dimage.raster.size() * sizeof(unsigned short), "depth");
		put(ir_out, irimage.raster.data(), irimage.raster.size(), "infra-red");
		std::ostringstream line;
		WritePoseLine(line, pose);
		put(pose_out, line.str().data(), line.str().size(), "pose");
	}
	void SaveFrame(const Frame &frame)                                                                                                 // dataset.h:101-107
	{
		SaveFrame(frame.depth, frame.ir, frame.pose);
		if (rgb_out && !frame.rgb.raster.empty()) put(rgb_out, frame.rgb.raster.data(), frame.rgb.raster.size() * 3, "colour");
		if (feye_out && !frame.fisheye.raster.empty()) put(feye_out, frame.fisheye.raster.data(), frame.fisheye.raster.size(), "fish-eye");
	}
	void SaveFrames(const std::vector<Frame> &frames) { for (auto &f : frames) SaveFrame(f); }
};
static_assert(sizeof(byte3) == 3, "colour frames are packed 3 bytes per pixel");
}  // namespace ht_mi355x
