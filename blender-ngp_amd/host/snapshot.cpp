#include "snapshot.h"

#include <algorithm>
#include <cstring>

#include "testbed.h"

namespace ngp {

uint16_t float_to_half_bits(float f) {
	uint32_t x; memcpy(&x, &f, 4);
	const uint32_t sign = (x >> 16) & 0x8000u;
	x &= 0x7fffffffu;
	if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0u));   // inf / nan
	if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                        // rounds to inf
	if (x < 0x33000001u) return (uint16_t)sign;                                                     // rounds to zero
	if (x < 0x38800000u) {                                                                          // subnormal half
		const int shift = 126 - (int)(x >> 23);              // 14..24
		const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
		uint32_t r = mant >> shift;
		const uint32_t rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
		if (rem > half || (rem == half && (r & 1u))) ++r;
		return (uint16_t)(sign | r);
	}
	uint32_t r = ((x - 0x38000000u) >> 13);
	const uint32_t rem = x & 0x1fffu;
	if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
	return (uint16_t)(sign | r);
}

float half_bits_to_float(uint16_t h) {
	const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
	uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
	if (e == 0) {
		if (m == 0) x = sign;
		else { int s = 0; while (!(m & 0x400u)) { m <<= 1; ++s; } x = sign | ((uint32_t)(113 - s) << 23) | ((m & 0x3ffu) << 13); }
	} else if (e == 31) x = sign | 0x7f800000u | (m << 13);
	else x = sign | ((e + 112u) << 23) | (m << 13);
	float f; memcpy(&f, &x, 4);
	return f;
}

Json vec_to_json(const float* v, int n) { Json j = Json::array(); for (int i = 0; i < n; ++i) j.push_back(Json((double)v[i])); return j; }
void vec_from_json(const Json& j, float* v, int n) { for (int i = 0; i < n && (size_t)i < j.size(); ++i) v[i] = (float)j[(size_t)i].number(); }

Json mat_to_json(const float* m, int rows, int cols) {
	Json j = Json::array();
	for (int r = 0; r < rows; ++r) {
		Json row = Json::array();
		for (int c = 0; c < cols; ++c) row.push_back(Json((double)m[c * rows + r]));
		j.push_back(row);
	}
	return j;
}
void mat_from_json(const Json& j, float* m, int rows, int cols) {
	for (int r = 0; r < rows && (size_t)r < j.size(); ++r) {
		const Json& row = j[(size_t)r];
		for (int c = 0; c < cols && (size_t)c < row.size(); ++c) m[c * rows + r] = (float)row[(size_t)c].number();
	}
}

Json aabb_to_json(const NgpAabb& b) { Json j = Json::object(); j["min"] = vec_to_json(b.min, 3); j["max"] = vec_to_json(b.max, 3); return j; }
NgpAabb aabb_from_json(const Json& j) { NgpAabb b{}; vec_from_json(j.at("min"), b.min, 3); vec_from_json(j.at("max"), b.max, 3); return b; }

static Json lens_to_json(const NgpImageMeta& m) {  // json_binding.h:83-98
	Json j = Json::object();
	if (m.lens_mode == 1) { j["k1"] = Json((double)m.lens_params[0]); j["k2"] = Json((double)m.lens_params[1]); j["p1"] = Json((double)m.lens_params[2]); j["p2"] = Json((double)m.lens_params[3]); }
	else if (m.lens_mode == 2) {
		for (int k = 0; k < 5; ++k) j["ftheta_p" + std::to_string(k)] = Json((double)m.lens_params[k]);
		j["w"] = Json((double)m.lens_params[5]); j["h"] = Json((double)m.lens_params[6]);
	}
	return j;
}
static void lens_from_json(const Json& j, NgpImageMeta& m) {  // json_binding.h:100-119
	if (j.contains("k1")) {
		m.lens_mode = 1;
		m.lens_params[0] = (float)j.at("k1").number(); m.lens_params[1] = (float)j.at("k2").number();
		m.lens_params[2] = (float)j.at("p1").number(); m.lens_params[3] = (float)j.at("p2").number();
	} else if (j.contains("ftheta_p0")) {
		m.lens_mode = 2;
		for (int k = 0; k < 5; ++k) m.lens_params[k] = (float)j.at("ftheta_p" + std::to_string(k)).number();
		m.lens_params[5] = (float)j.at("w").number(); m.lens_params[6] = (float)j.at("h").number();
	} else m.lens_mode = 0;
}

Json dataset_to_json(const NerfDataset& d) {
	Json j = Json::object();
	j["n_images"] = Json((unsigned long long)d.n_images);
	Json paths = Json::array(), metadata = Json::array(), xforms = Json::array();
	for (size_t i = 0; i < d.n_images; ++i) {
		paths.push_back(Json(i < d.paths.size() ? d.paths[i] : std::string()));
		const NgpImageMeta& m = d.metadata[i];
		Json mj = Json::object();
		mj["focal_length"] = vec_to_json(m.focal_length, 2);
		mj["lens"] = lens_to_json(m);
		mj["principal_point"] = vec_to_json(m.principal_point, 2);
		mj["rolling_shutter"] = vec_to_json(m.rolling_shutter, 4);
		Json res = Json::array(); res.push_back(Json((int)m.res[0])); res.push_back(Json((int)m.res[1]));
		mj["resolution"] = res;
		metadata.push_back(mj);
		Json xj = Json::object();
		xj["start"] = mat_to_json(d.xforms[i].start, 3, 4);
		xj["end"] = mat_to_json(d.xforms[i].end, 3, 4);
		xforms.push_back(xj);
	}
	j["paths"] = paths; j["metadata"] = metadata; j["xforms"] = xforms;
	j["render_aabb"] = aabb_to_json(d.render_aabb);
	const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	j["render_aabb_to_local"] = mat_to_json(ident, 3, 3);
	const float up[3] = {d.up.x, d.up.y, d.up.z}, off[3] = {d.offset.x, d.offset.y, d.offset.z};
	j["up"] = vec_to_json(up, 3);
	j["offset"] = vec_to_json(off, 3);
	Json env = Json::array(); env.push_back(Json(0)); env.push_back(Json(0));
	j["envmap_resolution"] = env;
	j["scale"] = Json((double)d.scale);
	j["aabb_scale"] = Json(d.aabb_scale);
	j["from_mitsuba"] = Json(d.from_mitsuba);
	j["is_hdr"] = Json(d.is_hdr);
	j["wants_importance_sampling"] = Json(true);
	return j;
}

void dataset_from_json(const Json& j, NerfDataset& d) {
	d = NerfDataset{};
	d.n_images = (size_t)j.at("n_images").number();
	d.metadata.assign(d.n_images, NgpImageMeta{});
	d.xforms.resize(d.n_images);
	d.pixelmemory.clear(); d.pixelmemory.resize(d.n_images);
	d.paths.assign(d.n_images, std::string());
	if (j.contains("paths")) for (size_t i = 0; i < d.n_images && i < j["paths"].size(); ++i) d.paths[i] = j["paths"][i].str();
	for (size_t i = 0; i < d.n_images; ++i) {
		NgpImageMeta& m = d.metadata[i];
		m.principal_point[0] = m.principal_point[1] = 0.5f;
		if (j.contains("lens")) lens_from_json(j["lens"], m);
		if (j.contains("camera_distortion")) lens_from_json(j["camera_distortion"], m);
		if (j.contains("principal_point")) vec_from_json(j["principal_point"], m.principal_point, 2);
		if (j.contains("rolling_shutter")) vec_from_json(j["rolling_shutter"], m.rolling_shutter, 4);
		if (j.contains("focal_length")) vec_from_json(j["focal_length"], m.focal_length, 2);
		mat_from_json(j.at("xforms")[i].at("start"), d.xforms[i].start, 3, 4);
		mat_from_json(j.at("xforms")[i].at("end"), d.xforms[i].end, 3, 4);
		if (j.contains("focal_lengths")) vec_from_json(j["focal_lengths"][i], m.focal_length, 2);
		if (j.contains("metadata")) {
			const Json& ji = j["metadata"][i];
			m.res[0] = (int32_t)ji.at("resolution")[(size_t)0].number(); m.res[1] = (int32_t)ji.at("resolution")[(size_t)1].number();
			vec_from_json(ji.at("focal_length"), m.focal_length, 2);
			vec_from_json(ji.at("principal_point"), m.principal_point, 2);
			if (ji.contains("lens")) lens_from_json(ji["lens"], m);
			if (ji.contains("camera_distortion")) lens_from_json(ji["camera_distortion"], m);
		}
	}
	d.render_aabb = aabb_from_json(j.at("render_aabb"));
	float up[3] = {0, 1, 0}, off[3] = {0, 0, 0};
	vec_from_json(j.at("up"), up, 3); vec_from_json(j.at("offset"), off, 3);
	d.up = Vec3{up[0], up[1], up[2]}; d.offset = Vec3{off[0], off[1], off[2]};
	d.scale = (float)j.at("scale").number();
	d.aabb_scale = (int)j.at("aabb_scale").number();
	d.from_mitsuba = j.at("from_mitsuba").boolean();
	d.is_hdr = j.value("is_hdr", false);
}

void snapshot_read_params(const Json& snapshot, std::vector<uint16_t>& params_fp16, std::vector<float>& params_fp32) {
	const std::string type = snapshot.value("params_type", "__half");
	const std::vector<uint8_t>& b = snapshot.at("params_binary").bin();
	if (type == "float") {
		params_fp32.resize(b.size() / 4);
		memcpy(params_fp32.data(), b.data(), params_fp32.size() * 4);
		params_fp16.resize(params_fp32.size());
		for (size_t i = 0; i < params_fp32.size(); ++i) params_fp16[i] = float_to_half_bits(params_fp32[i]);
	} else if (type == "__half") {
		params_fp16.resize(b.size() / 2);
		memcpy(params_fp16.data(), b.data(), params_fp16.size() * 2);
		params_fp32.resize(params_fp16.size());
		for (size_t i = 0; i < params_fp16.size(); ++i) params_fp32[i] = half_bits_to_float(params_fp16[i]);
	} else {
		throw std::runtime_error{"Snapshot uses an unknown parameter type '" + type + "'."};
	}
	if (snapshot.contains("n_params") && (size_t)snapshot["n_params"].number() != params_fp16.size()) throw std::runtime_error{"Snapshot parameter count does not match its binary blob."};
}

void snapshot_read_density_grid(const Json& snapshot, std::vector<float>& grid) {
	const std::vector<uint8_t>& b = snapshot.at("density_grid_binary").bin();
	grid.resize(b.size() / 2);
	const uint16_t* h = (const uint16_t*)b.data();
	for (size_t i = 0; i < grid.size(); ++i) grid[i] = half_bits_to_float(h[i]);
}

static bool ends_with_ci(const std::string& s, const std::string& suffix) {
	if (s.size() < suffix.size()) return false;
	std::string t = s.substr(s.size() - suffix.size());
	std::transform(t.begin(), t.end(), t.begin(), ::tolower);
	return t == suffix;
}

Json load_config_or_snapshot(const std::string& path) {
	if (ends_with_ci(path, ".msgpack")) return Json::from_msgpack_file(path);  // parent pointers are already resolved in snapshots
	Json result = Json::parse_file(path);
	while (result.contains("parent")) {  // merge_parent_network_config (testbed.cu:77-88)
		const std::string parent = result["parent"].str();
		const std::string base = path.substr(0, path.find_last_of("/\\") + 1);
		Json p = Json::parse_file(base + parent);
		result.erase("parent");
		p.merge_patch(result);
		result = p;
	}
	return result;
}

} // namespace ngp
