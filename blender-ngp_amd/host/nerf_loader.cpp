#include "nerf_loader.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <fstream>
#include <future>

#include <dirent.h>
#include <sys/stat.h>

#include "image_io.h"
#include "mini_json.h"
#include "png_reader.h"
#include "snapshot.h"

namespace ngp {

// ---- the few path operations the loader needs (the reference uses wjakob's filesystem::path; plain POSIX here, C++14)
static bool path_exists(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }
static bool path_is_directory(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
static bool path_is_file(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }
static std::string path_parent(const std::string& p) { const size_t k = p.find_last_of('/'); return k == std::string::npos ? std::string(".") : (k == 0 ? std::string("/") : p.substr(0, k)); }
static std::string path_filename(const std::string& p) { const size_t k = p.find_last_of('/'); return k == std::string::npos ? p : p.substr(k + 1); }
static std::string path_extension(const std::string& p) {   // without the dot, like filesystem::path::extension()
	const std::string f = path_filename(p); const size_t k = f.find_last_of('.');
	return (k == std::string::npos || k == 0) ? std::string() : f.substr(k + 1);
}
static std::string path_basename(const std::string& p) { const std::string f = path_filename(p); const size_t k = f.find_last_of('.'); return (k == std::string::npos || k == 0) ? f : f.substr(0, k); }
static std::string path_with_extension(const std::string& p, const std::string& ext) {
	const std::string dir = path_parent(p), base = path_basename(p);
	return (p.find('/') == std::string::npos ? std::string() : dir + "/") + base + "." + ext;
}
static std::string lower_extension(const std::string& p) { std::string e = path_extension(p); for (char& c : e) c = (char)std::tolower((unsigned char)c); return e; }
static float srgb_to_linear_host(float srgb) { return srgb <= 0.04045f ? srgb / 12.92f : std::pow((srgb + 0.055f) / 1.055f, 2.4f); }   // common_device.cuh:31-37
static std::string path_join(const std::string& a, const std::string& b) { if (!b.empty() && b[0] == '/') return b; return a + "/" + b; }

static const float PI_F = 3.14159265358979323846f;
static float fov_to_focal_length(int resolution, float degrees) { return 0.5f * (float)resolution / tanf(0.5f * degrees * PI_F / 180); }  // common_device.cuh:473-475

static std::string lower(std::string s) { std::transform(s.begin(), s.end(), s.begin(), ::tolower); return s; }

std::vector<std::string> resolve_nerf_json_paths(const std::string& data_path) {  // testbed_nerf.cu:2736-2753
	std::vector<std::string> out;
	if (path_is_directory(data_path)) {
		if (DIR* dir = opendir(data_path.c_str())) {
			while (const dirent* e = readdir(dir)) {
				const std::string f = path_join(data_path, e->d_name);
				if (lower(path_extension(f)) == "json" && path_is_file(f)) out.push_back(f);
			}
			closedir(dir);
		}
		std::sort(out.begin(), out.end());   // directory order is unspecified; sorted keeps runs reproducible
	} else if (lower(path_extension(data_path)) == "json") {
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
		const std::string basepath = path_parent(jsonpaths[i]);
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
				if (path_exists(path_join(basepath, fp)) && (float)f["sharpness"].number() > sharpness_discard_threshold * mean_sharpness) frames.push_back(f);
			}
		}
		for (const Json& f : frames) result.paths.push_back(f["file_path"].str());
		result.n_images += frames.size();
		frames_of[i] = std::move(frames);
	}
	if (result.n_images == 0) throw std::invalid_argument{"No training images were found for NeRF training!"};
	result.xforms.resize(result.n_images);
	result.metadata.assign(result.n_images, NgpImageMeta{}); std::vector<Vec3> light_dirs(result.n_images, Vec3{0.f, 0.f, 0.f});
	result.pixels.resize(result.n_images);

	result.image_type.assign(result.n_images, 1);
	result.depth16.resize(result.n_images);
	result.depth_scale.assign(result.n_images, -1.f);
	result.rays.resize(result.n_images);

	// pass 2 (430-700): dataset-level keys, then the frames of each json
	bool white_transparent = false, black_transparent = false, fix_premult = false, enable_ray_loading = true, enable_depth_loading = true;
	float info_depth_scale = -1.f;   // LoadedImageInfo::depth_scale (325), set by `integer_depth_scale`
	size_t image_idx = 0;
	std::vector<std::future<void>> futures;
	std::atomic<bool> any_rays{false}, any_exr{false}, any_light_dirs{false};
	for (size_t i = 0; i < jsons.size(); ++i) {
		const Json& json = jsons[i];
		const std::string basepath = path_parent(jsonpaths[i]);
		const std::string jp = jsonpaths[i];
		size_t lastdot = jp.find_last_of('.'); if (lastdot == std::string::npos) lastdot = jp.length();
		size_t lastunderscore = jp.find_last_of('_'); if (lastunderscore == std::string::npos) lastunderscore = lastdot; else lastunderscore++;
		const std::string part_after_underscore = lastunderscore <= lastdot ? jp.substr(lastunderscore, lastdot - lastunderscore) : std::string();

		if (json.contains("enable_ray_loading")) enable_ray_loading = json["enable_ray_loading"].boolean();
		if (json.contains("enable_depth_loading")) enable_depth_loading = json["enable_depth_loading"].boolean();
		if (json.contains("normal_mts_args")) result.from_mitsuba = true;
		if (json.contains("fix_premult")) fix_premult = json["fix_premult"].boolean();
		if (result.from_mitsuba) { result.scale = 0.66f; result.offset = Vec3{0.25f * result.scale, 0.25f * result.scale, 0.25f * result.scale}; }
		if (json.contains("render_aabb")) {
			for (int k = 0; k < 3; ++k) { result.render_aabb.min[k] = (float)json["render_aabb"][(size_t)0][(size_t)k].number(); result.render_aabb.max[k] = (float)json["render_aabb"][(size_t)1][(size_t)k].number(); }
		}
		if (json.contains("sharpen")) sharpen_amount = (float)json["sharpen"].number();
		if (json.contains("white_transparent")) white_transparent = json["white_transparent"].boolean();
		if (json.contains("black_transparent")) black_transparent = json["black_transparent"].boolean();
		if (json.contains("scale")) result.scale = (float)json["scale"].number();
		if (json.contains("importance_sampling")) result.wants_importance_sampling = json["importance_sampling"].boolean();
		if (json.contains("n_extra_learnable_dims")) result.n_extra_learnable_dims = (uint32_t)json["n_extra_learnable_dims"].number();   // nerf_loader.cu:480-481
		result.sharpen_amount = sharpen_amount;

		LensState lens;
		float principal_point[2] = {0.5f, 0.5f}, rolling_shutter[4] = {0, 0, 0, 0};
		if (json.contains("integer_depth_scale")) info_depth_scale = (float)json["integer_depth_scale"].number();
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
		if (json.contains("envmap") && result.envmap_resolution[0] == 0 && result.envmap_resolution[1] == 0) {   // nerf_loader.cu:533-546: the first json that names one wins
			const std::string envmap_path = path_join(basepath, json["envmap"].str());
			if (!path_exists(envmap_path)) throw std::runtime_error{"Environment map " + envmap_path + " does not exist."};
			int w = 0, h = 0;
			if (lower_extension(envmap_path) == "exr") {            // load_exr (common_device.cu:39-47): the file's floats as they are
				read_exr_rgba_f32(envmap_path, w, h, result.envmap_data);
				result.is_hdr = true;
			} else {                                                // load_stbi (common_device.cu:49-80): 8-bit -> from_rgba32<float> (sRGB decode, premultiplied)
				if (lower_extension(envmap_path) == "hdr") throw std::runtime_error{"Radiance .hdr environment maps (stbi_loadf) are not decoded by this build; use .exr"};
				std::vector<uint8_t> px;
				read_image_rgba8(envmap_path, w, h, px);
				if (w == 0 || h == 0) throw std::runtime_error{"Image has zero pixels."};
				result.envmap_data.resize((size_t)w * h * 4);
				for (size_t i = 0; i < (size_t)w * h; ++i) {
					const float alpha = px[i * 4 + 3] * (1.0f / 255.0f);
					for (int c = 0; c < 3; ++c) result.envmap_data[i * 4 + c] = srgb_to_linear_host(px[i * 4 + c] * (1.0f / 255.0f)) * alpha;
					result.envmap_data[i * 4 + 3] = alpha;
				}
			}
			result.envmap_resolution[0] = w; result.envmap_resolution[1] = h;
		}

		const float scale = result.scale;
		const Vec3 offset = result.offset;
		const bool from_mitsuba = result.from_mitsuba;
		for (size_t k = 0; k < frames_of[i].size(); ++k) {
			const size_t i_img = k + image_idx;
			const Json frame = frames_of[i][k];
			futures.push_back(std::async(std::launch::async, [&result, &json, &any_rays, &any_exr, &any_light_dirs, &light_dirs, frame, basepath, i_img, k, part_after_underscore, white_transparent, black_transparent, fix_premult,
			                                                  enable_ray_loading, enable_depth_loading, info_depth_scale, lens, principal_point, rolling_shutter, scale, offset, from_mitsuba]() {
				std::string json_provided_path = frame["file_path"].str();
				if (json_provided_path.empty()) { char buf[256]; snprintf(buf, 256, "%s_%03d/rgba.png", part_after_underscore.c_str(), (int)k); json_provided_path = buf; }
				std::string path = path_join(basepath, json_provided_path);
				if (path_extension(path).empty()) {
					path = path_with_extension(path, "png");
					if (!path_exists(path)) path = path_with_extension(path, "exr");
					if (!path_exists(path)) throw std::runtime_error{"Could not find image file: " + path};
				}
				int w = 0, h = 0;
				NgpImageMeta& m = result.metadata[i_img];
				if (lower(path_extension(path)) == "exr") {
					// load_exr_to_gpu (tinyexr_wrapper.cu:136-232): RGBA fp16, colour multiplied by alpha when `fix_premult`, missing alpha = 1
					std::vector<float> rgba;
					read_exr_rgba_f32(path, w, h, rgba);
					std::vector<uint8_t>& img = result.pixels[i_img];
					img.resize((size_t)w * h * 8);
					uint16_t* dst = (uint16_t*)img.data();
					for (size_t p = 0; p < (size_t)w * h; ++p) {
						const float alpha = rgba[p * 4 + 3], fix = fix_premult ? alpha : 1.0f;
						for (int c = 0; c < 3; ++c) dst[p * 4 + c] = float_to_half_bits(rgba[p * 4 + c] * fix);
						dst[p * 4 + 3] = float_to_half_bits(alpha);
					}
					result.image_type[i_img] = 2;   // EImageDataType::Half
					any_exr = true;
				} else {
					std::vector<uint8_t>& img = result.pixels[i_img];
					read_image_rgba8(path, w, h, img);   // stbi_load(path, .., 4): PNG or JPEG by content
					const std::string alphapath = path_join(basepath, json_provided_path + ".alpha." + path_extension(path));   // fmt "{}.alpha.{}" (585)
					if (path_exists(alphapath)) {
						int wa = 0, ha = 0; std::vector<uint8_t> alpha;
						read_image_rgba8(alphapath, wa, ha, alpha);
						if (wa != w || ha != h) throw std::runtime_error{"Alpha image " + alphapath + " has wrong resolution."};
						for (size_t p = 0; p < (size_t)w * h; ++p) {   // red channel of the alpha image, sRGB -> linear (595-597)
							const float sv = alpha[p * 4] * (1.f / 255.f);
							const float lin = sv <= 0.04045f ? sv / 12.92f : std::pow((sv + 0.055f) / 1.055f, 2.4f);
							img[p * 4 + 3] = (uint8_t)(255.0f * lin);
						}
					}
					uint32_t mask_color = 0;
					const std::string maskpath = path_join(path_parent(path), "dynamic_mask_" + path_basename(path) + ".png");
					if (path_exists(maskpath)) {  // 600-620
						int wa = 0, ha = 0; std::vector<uint8_t> mask;
						read_image_rgba8(maskpath, wa, ha, mask);
						if (wa != w || ha != h) throw std::runtime_error{"Dynamic mask " + maskpath + " has wrong resolution."};
						mask_color = 0x00FF00FF;   // hot pink
						for (size_t p = 0; p < (size_t)w * h; ++p) if (mask[p * 4] != 0) memcpy(&img[p * 4], &mask_color, 4);
					}
					convert_rgba32_host((size_t)w * h, img.data(), white_transparent, black_transparent, mask_color);
					result.image_type[i_img] = 1;   // EImageDataType::Byte
				}
				m.image_data_type = result.image_type[i_img];
				m.res[0] = w; m.res[1] = h;

				// depth (630-645): a 16-bit image whose values are scaled by integer_depth_scale * scale at upload (727)
				if (enable_depth_loading && info_depth_scale > 0.f && frame.contains("depth_path")) {
					const std::string depthpath = path_join(basepath, frame["depth_path"].str());
					if (path_exists(depthpath)) {
						int wa = 0, ha = 0;
						read_png_gray16(depthpath, wa, ha, result.depth16[i_img]);
						if (wa != w || ha != h) throw std::runtime_error{"Depth image " + depthpath + " has wrong resolution."};
					}
				}
				result.depth_scale[i_img] = info_depth_scale;

				// per-pixel rays (647-668): rays_<image name>.dat next to the image, w * h records of 6 floats, converted to the NGP frame
				const std::string rayspath = path_join(path_parent(path), "rays_" + path_basename(path) + ".dat");
				if (enable_ray_loading && path_exists(rayspath)) {
					std::vector<NgpRay>& rays = result.rays[i_img];
					rays.resize((size_t)w * h);
					std::ifstream f(rayspath, std::ios::binary);
					f.read((char*)rays.data(), (std::streamsize)(rays.size() * sizeof(NgpRay)));
					if ((size_t)f.gcount() != rays.size() * sizeof(NgpRay)) throw std::runtime_error{"Rays file " + rayspath + " is too short."};
					for (NgpRay& r : rays) {   // nerf_ray_to_ngp (nerf_loader.h:165-180): origin scaled + offset, both cycled (x, y, z) <- (y, z, x)
						const float o[3] = {r.o[0] * scale + offset.x, r.o[1] * scale + offset.y, r.o[2] * scale + offset.z};
						const float d[3] = {r.d[0], r.d[1], r.d[2]};
						r.o[0] = o[1]; r.o[1] = o[2]; r.o[2] = o[0];
						r.d[0] = d[1]; r.d[1] = d[2]; r.d[2] = d[0];
					}
					any_rays = true;
				}
				if (frame.contains("driver_parameters")) {   // nerf_loader.cu:671-680: the frame's light direction, three extra network inputs; replaces the latent codes
					const Json& dp = frame["driver_parameters"];
					float l[3] = {(float)dp.value("LightX", 0.0), (float)dp.value("LightY", 0.0), (float)dp.value("LightZ", 0.0)};
					const float n = std::sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
					if (n > 0.f) { l[0] /= n; l[1] /= n; l[2] /= n; }
					// nerf_direction_to_ngp (nerf_loader.h:103-111)
					light_dirs[i_img] = from_mitsuba ? Vec3{-l[0], -l[1], -l[2]} : Vec3{l[1], l[2], l[0]};
					any_light_dirs = true;
				}

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
	result.has_rays = any_rays;
	if (any_light_dirs) { result.has_light_dirs = true; result.n_extra_learnable_dims = 0; result.light_dirs = light_dirs; }   // nerf_loader.cu:678-679
	result.is_hdr = result.is_hdr || any_exr;   // an .exr envmap already set it (nerf_loader.cu:542 keeps it)
	return result;
}

} // namespace ngp
