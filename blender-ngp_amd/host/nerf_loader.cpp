#include "nerf_loader.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <filesystem>
#include <future>

#include "mini_json.h"
#include "png_reader.h"

namespace fs = std::filesystem;

namespace ngp {

static const float PI_F = 3.14159265358979323846f;
static float fov_to_focal_length(int resolution, float degrees) { return 0.5f * (float)resolution / tanf(0.5f * degrees * PI_F / 180); }  // common_device.cuh:473-475

static std::string lower(std::string s) { std::transform(s.begin(), s.end(), s.begin(), ::tolower); return s; }

std::vector<std::string> resolve_nerf_json_paths(const std::string& data_path) {  // testbed_nerf.cu:2736-2753
	std::vector<std::string> out;
	const fs::path p(data_path);
	if (fs::is_directory(p)) {
		for (const auto& e : fs::directory_iterator(p)) if (e.is_regular_file() && lower(e.path().extension().string()) == ".json") out.push_back(e.path().string());
		std::sort(out.begin(), out.end());   // directory order is unspecified; sorted keeps runs reproducible
	} else if (lower(p.extension().string()) == ".json") {
		out.push_back(data_path);
	} else {
		throw std::runtime_error{"NeRF data path must either be a json file or a directory containing json files."};
	}
	return out;
}

void convert_rgba32_host(size_t n, uint8_t* px, bool white_transparent, bool black_transparent, uint32_t mask_color) {
	for (size_t i = 0; i < n; ++i) {
		uint8_t* rgba = px + i * 4;
		if (white_transparent && rgba[0] == 255 && rgba[1] == 255 && rgba[2] == 255) rgba[3] = 0;   // NSVF: white = transparent
		if (black_transparent && rgba[0] == 0 && rgba[1] == 0 && rgba[2] == 0) rgba[3] = 0;
		uint32_t v; memcpy(&v, rgba, 4);
		if (mask_color != 0 && mask_color == v) { rgba[0] = 0xFF; rgba[1] = 0x00; rgba[2] = 0xFF; rgba[3] = 0x00; }
	}
}

struct LensState { int mode = 0; float params[7] = {0, 0, 0, 0, 0, 0, 0}; };

static void read_lens(const Json& j, LensState& lens, float principal_point[2], float rolling_shutter[4]) {  // nerf_loader.cu:197-270
	int mode = 0;
	const char* keys[4] = {"k1", "k2", "p1", "p2"};
	for (int k = 0; k < 4; ++k) {
		if (j.contains(keys[k])) { lens.params[k] = (float)j[keys[k]].number(); if (lens.params[k] != 0.f) mode = 1; }
	}
	if (j.contains("cx")) principal_point[0] = (float)j["cx"].number() / (float)j["w"].number();
	if (j.contains("cy")) principal_point[1] = (float)j["cy"].number() / (float)j["h"].number();
	if (j.contains("rolling_shutter")) {
		const Json& rs = j["rolling_shutter"];
		const float motionblur = rs.size() >= 4 ? (float)rs[(size_t)3].number() : 0.f;
		rolling_shutter[0] = (float)rs[(size_t)0].number(); rolling_shutter[1] = (float)rs[(size_t)1].number(); rolling_shutter[2] = (float)rs[(size_t)2].number(); rolling_shutter[3] = motionblur;
	}
	if (j.contains("ftheta_p0")) {
		for (int k = 0; k < 5; ++k) lens.params[k] = (float)j["ftheta_p" + std::to_string(k)].number();
		lens.params[5] = (float)j["w"].number(); lens.params[6] = (float)j["h"].number();
		mode = 2;
	}
	if (j.contains("latlong")) mode = 3;
	if (mode != 0) lens.mode = mode;   // an outer distortion mode is not overridden by nothing
}

static bool read_focal_length(const Json& j, float focal_length[2], const int res[2]) {  // nerf_loader.cu:272-301
	auto read = [&](int resolution, const std::string& axis) -> float {
		if (j.contains(axis + "_fov")) return fov_to_focal_length(resolution, (float)j[axis + "_fov"].number());
		if (j.contains("fl_" + axis)) return (float)j["fl_" + axis].number();
		if (j.contains("camera_angle_" + axis)) return fov_to_focal_length(resolution, (float)j["camera_angle_" + axis].number() * 180 / PI_F);
		return 0.0f;
	};
	const float x_fl = read(res[0], "x"), y_fl = read(res[1], "y");   // x_fov is in degrees, camera_angle_x in radians
	if (x_fl != 0) { focal_length[0] = focal_length[1] = x_fl; if (y_fl != 0) focal_length[1] = y_fl; }
	else if (y_fl != 0) { focal_length[0] = focal_length[1] = y_fl; }
	else return false;
	return true;
}

static Mat34 nerf_matrix_to_ngp(const Mat34& nerf_matrix, float scale, const Vec3& offset, bool from_mitsuba) {  // nerf_loader.h:113-132
	NerfDataset d;
	d.scale = scale; d.offset = offset; d.from_mitsuba = from_mitsuba;
	return d.nerf_matrix_to_ngp(nerf_matrix);
}

LoadedNerfData load_nerf_host(const std::vector<std::string>& jsonpaths, float sharpen_amount) {  // nerf_loader.cu:303-747
	if (jsonpaths.empty()) throw std::runtime_error{"Cannot load NeRF data from an empty set of paths."};
	LoadedNerfData result;
	std::vector<Json> jsons;
	for (const std::string& p : jsonpaths) jsons.push_back(Json::parse_file(p));
	if (jsons.front().contains("camera") && jsons.front()["camera"].is_array()) throw std::runtime_error{"hdf5 is no longer supported. please use the hdf52nerf.py conversion script"};

	// pass 1 (347-405): frame ordering, n_frames culling, sharpness filter; frames_of[i] is what the second pass iterates
	std::vector<std::vector<Json>> frames_of(jsons.size());
	for (size_t i = 0; i < jsons.size(); ++i) {
		const Json& json = jsons[i];
		const fs::path basepath = fs::path(jsonpaths[i]).parent_path();
		if (!json.contains("frames") || !json["frames"].is_array()) { fprintf(stderr, "  %s does not contain any frames. Skipping.\n", jsonpaths[i].c_str()); continue; }
		std::vector<Json> frames(json["frames"].elements());
		const float sharpness_discard_threshold = (float)json.value("sharpness_discard_threshold", 0.0);
		std::sort(frames.begin(), frames.end(), [](const Json& a, const Json& b) { return a["file_path"].str() < b["file_path"].str(); });
		if (json.contains("n_frames")) frames.resize(std::min(frames.size(), (size_t)json["n_frames"].number()));
		if (!frames.empty() && frames[0].contains("sharpness")) {
			const std::vector<Json> copy = frames;
			frames.clear();
			const int neighborhood_size = 3;
			for (int k = 0; k < (int)copy.size(); ++k) {
				float mean_sharpness = 0.0f;
				const int mean_start = std::max(0, k - neighborhood_size), mean_end = std::min(k + neighborhood_size, (int)copy.size() - 1);
				for (int j = mean_start; j < mean_end; ++j) mean_sharpness += (float)copy[j]["sharpness"].number();
				mean_sharpness /= (float)(mean_end - mean_start);
				Json f = copy[k];
				std::string fp = f["file_path"].str();
				std::replace(fp.begin(), fp.end(), '\\', '/');   // Windows paths on Linux
				f["file_path"] = Json(fp);
				if (fs::exists(basepath / fp) && (float)f["sharpness"].number() > sharpness_discard_threshold * mean_sharpness) frames.push_back(f);
			}
		}
		for (const Json& f : frames) result.paths.push_back(f["file_path"].str());
		result.n_images += frames.size();
		frames_of[i] = std::move(frames);
	}
	if (result.n_images == 0) throw std::invalid_argument{"No training images were found for NeRF training!"};
	result.xforms.resize(result.n_images);
	result.metadata.assign(result.n_images, NgpImageMeta{});
	result.pixels.resize(result.n_images);

	// pass 2 (430-700): dataset-level keys, then the frames of each json
	bool white_transparent = false, black_transparent = false;
	size_t image_idx = 0;
	std::vector<std::future<void>> futures;
	for (size_t i = 0; i < jsons.size(); ++i) {
		const Json& json = jsons[i];
		const fs::path basepath = fs::path(jsonpaths[i]).parent_path();
		const std::string jp = jsonpaths[i];
		size_t lastdot = jp.find_last_of('.'); if (lastdot == std::string::npos) lastdot = jp.length();
		size_t lastunderscore = jp.find_last_of('_'); if (lastunderscore == std::string::npos) lastunderscore = lastdot; else lastunderscore++;
		const std::string part_after_underscore = lastunderscore <= lastdot ? jp.substr(lastunderscore, lastdot - lastunderscore) : std::string();

		if (json.contains("normal_mts_args")) result.from_mitsuba = true;
		if (result.from_mitsuba) { result.scale = 0.66f; result.offset = Vec3{0.25f * result.scale, 0.25f * result.scale, 0.25f * result.scale}; }
		if (json.contains("render_aabb")) {
			for (int k = 0; k < 3; ++k) { result.render_aabb.min[k] = (float)json["render_aabb"][(size_t)0][(size_t)k].number(); result.render_aabb.max[k] = (float)json["render_aabb"][(size_t)1][(size_t)k].number(); }
		}
		if (json.contains("sharpen")) sharpen_amount = (float)json["sharpen"].number();
		if (json.contains("white_transparent")) white_transparent = json["white_transparent"].boolean();
		if (json.contains("black_transparent")) black_transparent = json["black_transparent"].boolean();
		if (json.contains("scale")) result.scale = (float)json["scale"].number();
		if (json.contains("importance_sampling")) result.wants_importance_sampling = json["importance_sampling"].boolean();
		if (json.contains("n_extra_learnable_dims") && json["n_extra_learnable_dims"].number() != 0) throw std::runtime_error{"n_extra_learnable_dims > 0 is outside the NeRF hot path of this build (SURVEY.md §8 f4)"};
		result.sharpen_amount = sharpen_amount;

		LensState lens;
		float principal_point[2] = {0.5f, 0.5f}, rolling_shutter[4] = {0, 0, 0, 0};
		read_lens(json, lens, principal_point, rolling_shutter);
		if (json.contains("aabb_scale")) result.aabb_scale = (int)json["aabb_scale"].number();
		if (json.contains("offset")) {
			const Json& o = json["offset"];
			result.offset = o.is_array() ? Vec3{(float)o[(size_t)0].number(), (float)o[(size_t)1].number(), (float)o[(size_t)2].number()} : Vec3{(float)o.number(), (float)o.number(), (float)o.number()};
		}
		if (json.contains("aabb")) {  // isotropic fit of the given box into the unit cube (500-506)
			const Json& a = json["aabb"];
			auto A = [&](int r, int c) { return (float)a[(size_t)r][(size_t)c].number(); };
			const float length = std::max(0.000001f, std::max(std::max(std::abs(A(1, 0) - A(0, 0)), std::abs(A(1, 1) - A(0, 1))), std::abs(A(1, 2) - A(0, 2))));
			result.scale = 1.f / length;
			result.offset = Vec3{((A(1, 0) + A(0, 0)) * 0.5f) * -result.scale + 0.5f, ((A(1, 1) + A(0, 1)) * 0.5f) * -result.scale + 0.5f, ((A(1, 2) + A(0, 2)) * 0.5f) * -result.scale + 0.5f};
		}
		if (json.contains("up")) { result.up = Vec3{(float)json["up"][(size_t)1].number(), (float)json["up"][(size_t)2].number(), (float)json["up"][(size_t)0].number()}; }  // axes permuted like the xforms
		if (json.contains("envmap")) throw std::runtime_error{"environment maps are outside the NeRF hot path of this build (SURVEY.md §8 f4)"};

		const float scale = result.scale;
		const Vec3 offset = result.offset;
		for (size_t k = 0; k < frames_of[i].size(); ++k) {
			const size_t i_img = k + image_idx;
			const Json frame = frames_of[i][k];
			futures.push_back(std::async(std::launch::async, [&result, &json, frame, basepath, i_img, k, part_after_underscore, white_transparent, black_transparent, lens, principal_point, rolling_shutter, scale, offset, from_mitsuba = result.from_mitsuba]() {
				std::string json_provided_path = frame["file_path"].str();
				if (json_provided_path.empty()) { char buf[256]; snprintf(buf, 256, "%s_%03d/rgba.png", part_after_underscore.c_str(), (int)k); json_provided_path = buf; }
				fs::path path = basepath / json_provided_path;
				if (path.extension().empty()) {
					path.replace_extension("png");
					if (!fs::exists(path)) path.replace_extension("exr");
					if (!fs::exists(path)) throw std::runtime_error{"Could not find image file: " + path.string()};
				}
				if (lower(path.extension().string()) == ".exr") throw std::runtime_error{"EXR training images are not part of this build (PNG only): " + path.string()};
				int w = 0, h = 0;
				std::vector<uint8_t>& img = result.pixels[i_img];
				read_png_rgba8(path.string(), w, h, img);
				uint32_t mask_color = 0;
				const fs::path maskpath = path.parent_path() / ("dynamic_mask_" + path.stem().string() + ".png");
				if (fs::exists(maskpath)) {  // 604-622
					int wa = 0, ha = 0; std::vector<uint8_t> mask;
					read_png_rgba8(maskpath.string(), wa, ha, mask);
					if (wa != w || ha != h) throw std::runtime_error{"Dynamic mask " + maskpath.string() + " has wrong resolution."};
					mask_color = 0x00FF00FF;   // hot pink
					for (size_t p = 0; p < (size_t)w * h; ++p) if (mask[p * 4] != 0) memcpy(&img[p * 4], &mask_color, 4);
				}
				convert_rgba32_host((size_t)w * h, img.data(), white_transparent, black_transparent, mask_color);

				NgpImageMeta& m = result.metadata[i_img];
				m.image_data_type = 1;   // EImageDataType::Byte
				m.res[0] = w; m.res[1] = h;
				bool got_fl = read_focal_length(json, m.focal_length, m.res);
				got_fl |= read_focal_length(frame, m.focal_length, m.res);
				if (!got_fl) throw std::runtime_error{"Couldn't read fov."};
				const Json& ms = frame.contains("transform_matrix_start") ? frame["transform_matrix_start"] : frame["transform_matrix"];
				const Json& me = frame.contains("transform_matrix_end") ? frame["transform_matrix_end"] : ms;
				Mat34 start, end;
				for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) { start.m[c * 3 + r] = (float)ms[(size_t)r][(size_t)c].number(); end.m[c * 3 + r] = (float)me[(size_t)r][(size_t)c].number(); }
				memcpy(m.rolling_shutter, rolling_shutter, 16);
				memcpy(m.principal_point, principal_point, 8);
				LensState fl = lens;
				read_lens(frame, fl, m.principal_point, m.rolling_shutter);   // per-frame override
				m.lens_mode = fl.mode; memcpy(m.lens_params, fl.params, sizeof(fl.params));
				const Mat34 s = nerf_matrix_to_ngp(start, scale, offset, from_mitsuba), e = nerf_matrix_to_ngp(end, scale, offset, from_mitsuba);
				memcpy(result.xforms[i_img].start, s.m, sizeof(s.m));
				memcpy(result.xforms[i_img].end, e.m, sizeof(e.m));
			}));
			if (futures.size() >= 16) { for (auto& f : futures) f.get(); futures.clear(); }   // bounded fan-out
		}
		image_idx += frames_of[i].size();
	}
	for (auto& f : futures) f.get();
	return result;
}

} // namespace ngp
