#include "nerf_renderer.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "snapshot.h"

namespace ngp {

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
static void check(int rc, const char* what) { if (rc != 0) throw std::runtime_error(std::string(what) + " failed: " + ngp_hip_last_error()); }

// ---- small linear algebra -----------------------------------------------------------------------------------------------------
Mat4 Mat4::operator*(const Mat4& o) const {
	Mat4 r;
	for (int c = 0; c < 4; ++c) for (int row = 0; row < 4; ++row) {
		float s = 0.f;
		for (int k = 0; k < 4; ++k) s += m[k * 4 + row] * o.m[c * 4 + k];
		r.m[c * 4 + row] = s;
	}
	return r;
}

Mat4 Mat4::inverse() const {  // general 4x4 inverse by cofactors, in double (Eigen's inverse() is float: results agree to ~1e-7 relative)
	double a[16], inv[16];
	for (int i = 0; i < 16; ++i) a[i] = m[i];
	inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
	inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
	inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
	inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
	inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
	inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
	inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
	inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
	inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
	inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
	inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
	inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
	inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
	inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
	inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
	inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
	const double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
	Mat4 r;
	const double idet = det != 0.0 ? 1.0 / det : 0.0;
	for (int i = 0; i < 16; ++i) r.m[i] = (float)(inv[i] * idet);
	return r;
}

void BoundingBox::enlarge(const Vec3& p) {
	min = Vec3{std::min(min.x, p.x), std::min(min.y, p.y), std::min(min.z, p.z)};
	max = Vec3{std::max(max.x, p.x), std::max(max.y, p.y), std::max(max.z, p.z)};
}
void BoundingBox::inflate(float amount) { min = Vec3{min.x - amount, min.y - amount, min.z - amount}; max = Vec3{max.x + amount, max.y + amount, max.z + amount}; }
Vec3 BoundingBox::relative_pos(const Vec3& p) const { const Vec3 d = diag(); return Vec3{(p.x - min.x) / d.x, (p.y - min.y) / d.y, (p.z - min.z) / d.z}; }

// ---- masks ------------------------------------------------------------------------------------------------------------------------
Mask3D::Mask3D(EMaskShape shape, const Mat4& transform, EMaskMode mode, const float config[6], float feather, float opacity) {
	pod.mode = (int)mode; pod.shape = (int)shape;
	memcpy(pod.transform, transform.m, 64);
	const Mat4 inv = transform.inverse();
	memcpy(pod.itransform, inv.m, 64);
	memcpy(pod.config, config, 24);
	pod.feather = feather; pod.opacity = opacity;
}
Mask3D Mask3D::All(EMaskMode mode) { const float c[6] = {0, 0, 0, 0, 0, 0}; return Mask3D(EMaskShape::All, Mat4{}, mode, c, 0.0f, 1.0f); }
Mask3D Mask3D::Box(const Vec3& d, const Mat4& t, EMaskMode mode, float feather, float opacity) { const float c[6] = {d.x, d.y, d.z, 0, 0, 0}; return Mask3D(EMaskShape::Box, t, mode, c, feather, opacity); }
Mask3D Mask3D::Cylinder(float r, float h, const Mat4& t, EMaskMode mode, float feather, float opacity) { const float c[6] = {r, h, 0, 0, 0, 0}; return Mask3D(EMaskShape::Cylinder, t, mode, c, feather, opacity); }
Mask3D Mask3D::Sphere(float r, const Mat4& t, EMaskMode mode, float feather, float opacity) { const float c[6] = {r, 0, 0, 0, 0, 0}; return Mask3D(EMaskShape::Sphere, t, mode, c, feather, opacity); }
Mask3D Mask3D::transformed_by(const Mat4& matrix) const {  // mask_3D.cuh:248-253
	Mask3D copy = *this;
	Mat4 t; memcpy(t.m, pod.transform, 64);
	const Mat4 nt = matrix * t, ni = nt.inverse();
	memcpy(copy.pod.transform, nt.m, 64); memcpy(copy.pod.itransform, ni.m, 64);
	return copy;
}

DownsampleInfo DownsampleInfo::MakeFromMip(int rx, int ry, uint32_t mip) {  // common.h:337-355
	DownsampleInfo d;
	NgpDownsampleInfo& ds = d.pod;
	ds.max_pixels = (uint32_t)(rx * ry);
	ds.max_res[0] = rx; ds.max_res[1] = ry;
	if (mip == 0) {
		ds.scaled_pixels = ds.max_pixels; ds.scaled_res[0] = rx; ds.scaled_res[1] = ry; ds.skip[0] = ds.skip[1] = 1;
	} else {
		ds.skip[0] = ds.skip[1] = 1 << mip;
		ds.scaled_res[0] = (rx + ds.skip[0] - 1) / ds.skip[0]; ds.scaled_res[1] = (ry + ds.skip[1] - 1) / ds.skip[1];
		ds.scaled_pixels = (uint32_t)(ds.scaled_res[0] * ds.scaled_res[1]);
	}
	return d;
}

static Vec3 avg4(const Vec3& a, const Vec3& b, const Vec3& c, const Vec3& d) { return Vec3{(a.x + b.x + c.x + d.x) / 4.0f, (a.y + b.y + c.y + d.y) / 4.0f, (a.z + b.z + c.z + d.z) / 4.0f}; }
Vec3 Quadrilateral3D::center() const { return avg4(tl, tr, bl, br); }
Vec3 QuadrilateralHexahedron::center() const { const Vec3 f = front.center(), b = back.center(); return Vec3{(f.x + b.x) / 2.0f, (f.y + b.y) / 2.0f, (f.z + b.z) / 2.0f}; }

static bool same(const Vec3& a, const Vec3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static bool same(const Quadrilateral3D& a, const Quadrilateral3D& b) { return same(a.tl, b.tl) && same(a.tr, b.tr) && same(a.bl, b.bl) && same(a.br, b.br); }
bool RenderCameraProperties::operator==(const RenderCameraProperties& o) const {
	return memcmp(transform.m, o.transform.m, sizeof(transform.m)) == 0 && model == o.model && focal_length == o.focal_length && near_distance == o.near_distance &&
	       aperture_size == o.aperture_size && focus_z == o.focus_z && spherical_quadrilateral.width == o.spherical_quadrilateral.width &&
	       spherical_quadrilateral.height == o.spherical_quadrilateral.height && spherical_quadrilateral.curvature == o.spherical_quadrilateral.curvature &&
	       same(quadrilateral_hexahedron.front, o.quadrilateral_hexahedron.front) && same(quadrilateral_hexahedron.back, o.quadrilateral_hexahedron.back);
}
NgpRenderCamera RenderCameraProperties::pod() const {
	NgpRenderCamera c{};
	memcpy(c.transform, transform.m, sizeof(c.transform));
	c.model = (int)model; c.focal_length = focal_length;
	c.sq_width = spherical_quadrilateral.width; c.sq_height = spherical_quadrilateral.height; c.sq_curvature = spherical_quadrilateral.curvature;
	const Quadrilateral3D* q[2] = {&quadrilateral_hexahedron.front, &quadrilateral_hexahedron.back};
	float* dst[2] = {c.qh_front, c.qh_back};
	for (int k = 0; k < 2; ++k) {
		const Vec3 v[4] = {q[k]->tl, q[k]->tr, q[k]->bl, q[k]->br};
		for (int i = 0; i < 4; ++i) { dst[k][3 * i] = v[i].x; dst[k][3 * i + 1] = v[i].y; dst[k][3 * i + 2] = v[i].z; }
	}
	c.near_distance = near_distance; c.aperture_size = aperture_size; c.focus_z = focus_z;
	return c;
}

// ---- fields -----------------------------------------------------------------------------------------------------------------------
void NeuralRadianceField::load_snapshot(void* stream) {
	if (is_loaded) return;
	if (snapshot_path.empty()) throw std::runtime_error{"No snapshot path specified."};
	Json config = Json::from_msgpack_file(snapshot_path);
	if (!config.contains("snapshot")) throw std::runtime_error{"File " + snapshot_path + " does not contain a snapshot."};
	const Json& snapshot = config["snapshot"];
	if (snapshot.value("version", 0) < 1) throw std::runtime_error{"Snapshot uses an old format."};
	grid_size = (uint32_t)snapshot.at("density_grid_size").number();
	if (grid_size != NGP_NERF_GRIDSIZE) throw std::runtime_error{"Incompatible grid size."};
	if (snapshot.contains("aabb")) train_aabb = aabb_from_json(snapshot["aabb"]);
	aabb_scale = (uint32_t)snapshot.at("nerf").at("aabb_scale").number();
	max_cascade = 0;
	while ((1u << max_cascade) < aabb_scale) ++max_cascade;
	cone_angle_constant = aabb_scale <= 1 ? 0.0f : (1.0f / 256.0f);   // fixed-size stepping in unit-cube scenes (:183)

	std::vector<float> grid;
	snapshot_read_density_grid(snapshot, grid);
	const size_t volume = (size_t)grid_size * grid_size * grid_size;
	density_grid_bitfield.resize(volume * num_cascades / 8);
	density_grid_mean.resize(4);
	if (grid.size() == volume * (max_cascade + 1)) {
		density_grid.resize(grid.size() * 4);
		density_grid.copy_from_host(grid.data(), grid.size() * 4);
		// NeuralRadianceField::update_density_grid_mean_and_bitfield (nerf_data.cu:61-88)
		check(ngp_hip_density_grid_mean(stream, density_grid.as<float>(), (uint32_t)volume, density_grid_mean.as<float>()), "density_grid_mean");
		check(ngp_hip_grid_to_bitfield_and_pool(stream, density_grid.as<float>(), max_cascade + 1, density_grid_mean.as<float>(), density_grid_bitfield.as<uint8_t>()), "grid_to_bitfield_and_pool");
	} else if (!grid.empty()) {
		throw std::runtime_error{"Incompatible number of grid cascades."};
	} else {
		density_grid_bitfield.memset(0, stream);
	}

	// network geometry from the stored config (the fused gfx950 kernels cover the configs/nerf/base.json family)
	const Json empty = Json::object();
	const Json& enc = config.contains("encoding") ? config["encoding"] : empty;
	const uint32_t n_levels = (uint32_t)enc.value("n_levels", 16);
	if (n_levels != 16 || (uint32_t)enc.value("n_features_per_level", 2) != 2) throw std::runtime_error{"snapshot network not supported by the gfx950 fused kernels: HashGrid must have 16 levels x 2 features"};
	const uint32_t log2_hashmap_size = (uint32_t)enc.value("log2_hashmap_size", 15);
	uint32_t base_resolution = (uint32_t)enc.value("base_resolution", 0);
	if (!base_resolution) base_resolution = 1u << (log2_hashmap_size / 3);
	float per_level_scale = (float)enc.value("per_level_scale", 0.0);
	if (per_level_scale <= 0.0f) per_level_scale = std::exp(std::log(2048.0f * (float)aabb_scale / (float)base_resolution) / (float)(n_levels - 1));
	check(ngp_hip_net_make_desc_host(n_levels, log2_hashmap_size, base_resolution, per_level_scale, &desc), "ngp_hip_net_make_desc_host");
	desc_gpu.resize(sizeof(NgpNetDesc));
	desc_gpu.copy_from_host(&desc, sizeof(NgpNetDesc));

	// the colour network's depth (configs/nerf/base_{0,1,2,3}layer.json): anything but two hidden layers runs the generic kernels (NgpNetVariant)
	n_rgb_hidden_layers = 2;
	if (config.contains("rgb_network")) {
		const int h = config["rgb_network"].value("n_hidden_layers", 2);
		if (h < 0 || h > 3 || (h > 0 && config["rgb_network"].value("n_neurons", 64) != 64)) throw std::runtime_error{"snapshot network not supported: rgb_network must be 64 neurons wide with 0..3 hidden layers"};
		n_rgb_hidden_layers = (uint32_t)h;
	}
	NgpNetVariant nv{0u, n_rgb_hidden_layers, nullptr, nullptr, nullptr, 0u};
	std::vector<uint16_t> p16; std::vector<float> p32;
	snapshot_read_params(snapshot, p16, p32);
	const size_t n_params = (size_t)ngp_hip_net_mlp_params_host(&nv) + 2u * (size_t)desc.n_grid_entries;
	if (p16.size() != n_params) throw std::runtime_error{"Snapshot has " + std::to_string(p16.size()) + " parameters, its network config needs " + std::to_string(n_params) + "."};
	params.resize(n_params * 2);
	params.copy_from_host(p16.data(), n_params * 2);
	HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
	is_loaded = true;
}

// ---- render loop (nerf_renderer.cu:565-791) -----------------------------------------------------------------------------------------
uint64_t NerfRenderer::render(RenderBuffer& rb, const RenderRequest& request, void* stream) {
	if (request.nerfs.empty()) return 0;
	hipStream_t st = (hipStream_t)stream;
	const bool trace = this->trace;   // stage markers on stderr (pyngp: render_trace)
	if (trace) fprintf(stderr, "multi render: %zu nerfs\n", request.nerfs.size());

	// RenderData::update_nerfs (render_data.cuh:45-82): drop fields no descriptor refers to, add new ones, one proxy per descriptor
	m_fields.erase(std::remove_if(m_fields.begin(), m_fields.end(), [&](const std::unique_ptr<NeuralRadianceField>& f) {
		return std::none_of(request.nerfs.begin(), request.nerfs.end(), [&](const NerfDescriptor& d) { return d.snapshot_path == f->snapshot_path; });
	}), m_fields.end());
	std::vector<NeuralRadianceField*> field_of;
	for (const NerfDescriptor& d : request.nerfs) {
		auto it = std::find_if(m_fields.begin(), m_fields.end(), [&](const std::unique_ptr<NeuralRadianceField>& f) { return f->snapshot_path == d.snapshot_path; });
		if (it == m_fields.end()) { m_fields.emplace_back(new NeuralRadianceField(d.snapshot_path)); it = m_fields.end() - 1; }
		field_of.push_back(it->get());
	}
	// RenderData::copy_from_host (:84-93): lazy snapshot load, masks (local, then the request's global masks brought into the NeRF's frame,
	// with the implicit `All` mask in front: render_modifiers.cuh:30-62) and one NerfProps per proxy
	const uint32_t n_nerfs = (uint32_t)request.nerfs.size();
	std::vector<NgpMask3D> all_masks;
	std::vector<uint32_t> mask_first(n_nerfs), mask_count(n_nerfs);
	for (uint32_t n = 0; n < n_nerfs; ++n) {
		const NerfDescriptor& d = request.nerfs[n];
		field_of[n]->load_snapshot(stream);
		std::vector<Mask3D> masks(d.modifiers.masks);
		const Mat4 itransform = d.transform.inverse();
		for (const Mask3D& gm : request.modifiers.masks) masks.push_back(gm.transformed_by(itransform));
		if (!masks.empty() && masks[0].pod.shape != (int)EMaskShape::All) {
			const EMaskMode mode = masks[0].pod.mode == (int)EMaskMode::Add ? EMaskMode::Subtract : EMaskMode::Add;
			masks.insert(masks.begin(), Mask3D::All(mode));
		}
		mask_first[n] = (uint32_t)all_masks.size(); mask_count[n] = (uint32_t)masks.size();
		for (const Mask3D& m : masks) all_masks.push_back(m.pod);
	}
	m_masks_gpu.enlarge(std::max<size_t>(all_masks.size(), 1) * sizeof(NgpMask3D));
	if (!all_masks.empty()) m_masks_gpu.copy_from_host(all_masks.data(), all_masks.size() * sizeof(NgpMask3D));
	std::vector<NgpNerfProps> props(n_nerfs);
	for (uint32_t n = 0; n < n_nerfs; ++n) {
		const NerfDescriptor& d = request.nerfs[n];
		const NeuralRadianceField& f = *field_of[n];
		NgpNerfProps& p = props[n];
		memset(&p, 0, sizeof(p));
		memcpy(p.transform, d.transform.m, 64);
		const Mat4 inv = d.transform.inverse();
		memcpy(p.itransform, inv.m, 64);
		p.density_grid_bitfield = f.density_grid_bitfield.as<uint8_t>();
		p.grid_size = f.grid_size; p.grid_volume = f.grid_size * f.grid_size * f.grid_size;
		p.render_aabb = d.aabb.pod(); p.train_aabb = f.train_aabb;
		p.masks = m_masks_gpu.as<NgpMask3D>() + mask_first[n]; p.n_masks = mask_count[n];
		p.cone_angle = f.cone_angle_constant; p.min_cone_stepsize = f.min_cone_step_size(); p.max_cone_stepsize = f.max_cone_step_size();
		p.nerf_cascades = f.num_cascades; p.opacity = d.opacity;
	}
	m_props_gpu.enlarge(n_nerfs * sizeof(NgpNerfProps));
	m_props_gpu.copy_from_host(props.data(), n_nerfs * sizeof(NgpNerfProps));
	HIP_TRY(hipStreamSynchronize(st));
	if (trace) fprintf(stderr, "multi render: fields loaded, props uploaded\n");

	// RenderDataWorkspace::enlarge (render_data_workspace.cuh:71-113)
	const NgpDownsampleInfo& ds = request.output.ds.pod;
	const uint32_t n_pixels = ds.scaled_pixels;
	if (n_pixels == 0) return 0;
	const uint32_t stride = (n_pixels + 127u) / 128u * 128u;
	// nerf_renderer.cu marches clamp(n_rays / n_alive, 1, 8) steps per pass.  The per-ray sample sequence does not depend on how it is cut
	// into passes and n_alive * n_steps <= n_pixels keeps every buffer as sized for one step per pixel, so the cap is raised: fewer passes
	// (each costs a host round trip and 3-5 launches per NeRF) once few rays are left — same image (tests/test_bl_render_gpu.py).
	const uint32_t max_steps = 64, OUT_STRIDE = 4;
	for (int b = 0; b < 2; ++b) { m_global[b].enlarge((size_t)stride * sizeof(NgpGlobalRay)); m_proxy[b].enlarge((size_t)stride * n_nerfs * sizeof(NgpProxyRay)); }
	m_hit.enlarge((size_t)stride * sizeof(NgpGlobalRay));
	m_net_in.enlarge((size_t)stride * 8 * sizeof(NgpCoord));
	m_net_out.enlarge((size_t)stride * 8 * OUT_STRIDE * 2);
	m_counters.enlarge((size_t)(2 + 2 * (1 + n_nerfs)) * 4 + 8);   // (both loops' words: sized once, the pointers below stay valid)
	m_enc_ws.enlarge(ngp_hip_nerf_encode_workspace_bytes(stride * 8));
	uint32_t* alive_counter = m_counters.as<uint32_t>();
	uint32_t* hit_counter = alive_counter + 1;

	// init_rays_from_camera (:613-660)
	const NgpRenderCamera cam = request.camera.pod();
	check(ngp_hip_multi_init_global_rays(stream, 0 /* todo in the reference: sample index */, m_global[0].as<NgpGlobalRay>(), rb.depth_buffer.as<float>(), &ds, &cam), "multi_init_global_rays");
	const NgpNerfProps* props_dev = m_props_gpu.as<NgpNerfProps>();
	for (uint32_t n = 0; n < n_nerfs; ++n) {
		check(ngp_hip_multi_init_proxy_rays(stream, n_pixels, m_global[0].as<NgpGlobalRay>(), m_proxy[0].as<NgpProxyRay>() + (size_t)n * stride, props_dev + n), "multi_init_proxy_rays");
	}

	// march_rays_and_accumulate_colors (:664-791)
	uint32_t n_alive = n_pixels, i = 1, dbi = 0;
	uint64_t n_samples = 0;
	HIP_TRY(hipMemsetAsync(hit_counter, 0, 4, st));
	const float cam_pos[3] = {cam.transform[9], cam.transform[10], cam.transform[11]};
	last_n_passes = 0;
	if (fused_passes && n_nerfs <= 30) {
		// ---- fused pass loop.  Per pass: advance (march the proxies that moved, cull, compact, per-NeRF lists, counts to the host mailbox) -> poll -> per NeRF that is sampled
		// { next inputs, network, composite }.  n_steps follows the stock tracer's budget (pass_samples_factor x pixels per pass).  The fork's sampler is NOT independent
		// of how a ray is cut into passes: inside a pass it emits the sample at t + dt BEFORE testing that position (nerf_renderer.cu:350-366), at a pass boundary
		// march_active_rays moves the ray to the next occupied voxel first — so a boundary that falls next to an empty voxel trades one sample in the gap for one at the
		// landing point, and where NeRFs overlap the cull between two passes decides who samples next.  The reference's own schedule (clamp(n_rays / n_alive, 1, 8))
		// depends on resolution and content in the same way; `reference_schedule` reproduces the unfused loop's frame bit for bit, the default differs from it in < 1 %
		// of the pixels by < 0.03 (tests/test_bl_render_gpu.py).
		const uint32_t n_words = 1 + n_nerfs;
		if (m_host_words_n < n_words) {
			if (m_host_words) (void)hipHostFree(m_host_words);
			m_host_words = nullptr; m_host_words_n = 0;
			HIP_TRY(hipHostMalloc((void**)&m_host_words, (size_t)n_words * 8, hipHostMallocMapped | hipHostMallocCoherent));
			memset(m_host_words, 0, (size_t)n_words * 8);
			m_host_words_n = n_words;
		}
		uint64_t* mailbox_dev = nullptr;
		HIP_TRY(hipHostGetDevicePointer((void**)&mailbox_dev, m_host_words, 0));
		// device words: [0] finished rays, [1] workgroups done, then two sets of {alive, active per NeRF}
		uint32_t* words = m_counters.as<uint32_t>();
		HIP_TRY(hipMemsetAsync(words, 0, (size_t)(2 + 2 * n_words) * 4, st));
		uint32_t* final_counter = words, *blocks_done = words + 1;
		uint32_t* sets[2] = {words + 2, words + 2 + n_words};
		m_active_lists.enlarge((size_t)stride * n_nerfs * 4);
		const float factor = reference_schedule ? 1.0f : std::min(std::max(pass_samples_factor, 1.0f), 6.0f);
		const uint32_t pass_samples = (uint32_t)((float)n_pixels * factor);
		const size_t max_elems = (size_t)pass_samples + 128u * n_nerfs + 128u;
		m_net_in.enlarge(max_elems * sizeof(NgpCoord));
		m_net_out.enlarge(max_elems * OUT_STRIDE * 2);
		m_enc_ws.enlarge(ngp_hip_nerf_encode_workspace_bytes((uint32_t)max_elems));
		std::vector<uint32_t> n_active(n_nerfs);
		auto read_mailbox = [&]() {
			const uint32_t want = m_sequence;
			const auto t0 = std::chrono::steady_clock::now();
			for (uint32_t k = 0; k < n_words; ++k) {
				uint64_t word;
				for (uint32_t spins = 1; (uint32_t)((word = __atomic_load_n(m_host_words + k, __ATOMIC_ACQUIRE)) >> 32) != want; ++spins) {
					if ((spins & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {   // a launch that died posts nothing: ask the stream
						HIP_TRY(hipStreamSynchronize(st));
						if ((uint32_t)((word = __atomic_load_n(m_host_words + k, __ATOMIC_ACQUIRE)) >> 32) != want) throw std::runtime_error{"multi render: the advance launch finished without posting its counts"};
						break;
					}
					__builtin_ia32_pause();
				}
				if (k == 0) n_alive = (uint32_t)word; else n_active[k - 1] = (uint32_t)word;
			}
		};
		uint32_t n_prev = n_pixels;
		int src = 0;
		bool first = true;
		while (i < 10000 && n_prev > 0) {
			const int dst = src ^ 1;
			const uint32_t n_entered = n_prev;
			++m_sequence;
			check(ngp_hip_multi_advance(stream, n_prev, n_nerfs, m_global[src].as<NgpGlobalRay>(), m_proxy[src].as<NgpProxyRay>(), m_global[dst].as<NgpGlobalRay>(), m_proxy[dst].as<NgpProxyRay>(), stride,
			                            m_hit.as<NgpGlobalRay>(), cam_pos, props_dev, reference_schedule ? 0u : max_skips_per_pass, sets[last_n_passes & 1], sets[(last_n_passes + 1) & 1], final_counter, m_active_lists.as<uint32_t>(),
			                            blocks_done, mailbox_dev, m_sequence, first ? (uint32_t)ds.scaled_res[0] : 0u, first ? (uint32_t)ds.scaled_res[1] : 0u), "multi_advance");
			first = false;
			++last_n_passes;
			read_mailbox();
			if (trace) fprintf(stderr, "multi render (fused): pass %u i=%u n_alive=%u n_active[0]=%u\n", last_n_passes, i, n_alive, n_active[0]);
			src = dst;
			n_prev = n_alive;
			if (n_alive == 0) break;
			NgpGlobalRay* g = m_global[src].as<NgpGlobalRay>();
			NgpProxyRay* px = m_proxy[src].as<NgpProxyRay>();
			const uint32_t n_steps = reference_schedule ? std::min(std::max(n_pixels / n_entered, 1u), max_steps) : std::min(std::max(pass_samples / n_alive, 1u), std::max(max_steps_per_pass, 1u));
			bool sampled = false;
			for (uint32_t n = 0; n < n_nerfs; ++n) {
				if (n_active[n] == 0) continue;
				sampled = true;
				const NeuralRadianceField& f = *field_of[n];
				NgpProxyRay* pn = px + (size_t)n * stride;
				const uint32_t* list = m_active_lists.as<uint32_t>() + (size_t)n * stride;
				const uint32_t n_network_elements = (n_active[n] * n_steps + 127u) / 128u * 128u;
				check(ngp_hip_multi_generate_next_inputs_list(stream, n_active[n], list, g, pn, m_net_in.as<NgpCoord>(), n_steps, props_dev + n), "multi_generate_next_inputs");
				NgpNetVariant nv{0u, f.n_rgb_hidden_layers, nullptr, nullptr, nullptr, 0u};
				check(ngp_hip_nerf_inference_ws(stream, f.desc_gpu.as<NgpNetDesc>(), f.params.as<uint16_t>(), m_net_in.as<float>(), 7, n_network_elements, m_net_out.as<uint16_t>(), OUT_STRIDE,
				                                m_enc_ws.data(), m_enc_ws.bytes(), f.n_rgb_hidden_layers == 2 ? nullptr : &nv), "nerf_inference (multi)");
				n_samples += n_network_elements;
				check(ngp_hip_multi_composite_list(stream, n_active[n], list, i, g, pn, m_net_in.as<NgpCoord>(), m_net_out.as<uint16_t>(), OUT_STRIDE, n_steps, (int)f.rgb_activation, (int)f.density_activation,
				                                   f.min_transmittance, props_dev + n), "multi_composite");
			}
			if (sampled) i += n_steps;   // (a pass in which every ray rested samples nothing)
		}
		uint32_t n_hit = 0;
		HIP_TRY(hipMemcpyAsync(&n_hit, final_counter, 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		check(ngp_hip_multi_shade(stream, n_hit, m_hit.as<NgpGlobalRay>(), 0 /* :602 passes train_in_linear_colors = false */, rb.frame_buffer.as<float>(), rb.depth_buffer.as<float>(), &ds,
		                          request.output.flip_y ? 1 : 0), "multi_shade");
		return n_samples;
	}
	while (i < 10000) {
		const int tmp = dbi % 2, cur = (dbi + 1) % 2;
		++dbi;
		++last_n_passes;
		HIP_TRY(hipMemsetAsync(alive_counter, 0, 4, st));
		check(ngp_hip_multi_compact_rays(stream, n_alive, m_global[tmp].as<NgpGlobalRay>(), m_global[cur].as<NgpGlobalRay>(), m_proxy[tmp].as<NgpProxyRay>(), m_proxy[cur].as<NgpProxyRay>(),
		                                 n_nerfs, stride, m_hit.as<NgpGlobalRay>(), alive_counter, hit_counter), "multi_compact_rays");
		HIP_TRY(hipMemcpyAsync(&n_alive, alive_counter, 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		if (trace) fprintf(stderr, "multi render: pass i=%u n_alive=%u\n", i, n_alive);
		if (n_alive == 0) break;
		NgpGlobalRay* g = m_global[cur].as<NgpGlobalRay>();
		NgpProxyRay* px = m_proxy[cur].as<NgpProxyRay>();
		check(ngp_hip_multi_march_active_rays(stream, n_alive, n_nerfs, g, px, stride, props_dev), "multi_march_active_rays");
		// cull + per-NeRF lists of the rays that sample each NeRF in this pass (a ray samples at most one): the network then runs on
		// exactly those rays instead of on every alive ray for every NeRF as nerf_renderer.cu:735-768 does
		m_active_lists.enlarge((size_t)stride * n_nerfs * 4); m_active_counts.enlarge((size_t)n_nerfs * 4);
		check(ngp_hip_multi_cull_rays_collect(stream, n_alive, n_nerfs, g, px, stride, cam_pos, props_dev, m_active_lists.as<uint32_t>(), m_active_counts.as<uint32_t>()), "multi_cull_rays");
		std::vector<uint32_t> n_active(n_nerfs);
		HIP_TRY(hipMemcpyAsync(n_active.data(), m_active_counts.data(), (size_t)n_nerfs * 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		const uint32_t n_steps = std::min(std::max(n_pixels / n_alive, 1u), max_steps);
		for (uint32_t n = 0; n < n_nerfs; ++n) {
			if (n_active[n] == 0) continue;
			const NeuralRadianceField& f = *field_of[n];
			NgpProxyRay* pn = px + (size_t)n * stride;
			const uint32_t* list = m_active_lists.as<uint32_t>() + (size_t)n * stride;
			const uint32_t n_network_elements = (n_active[n] * n_steps + 127u) / 128u * 128u;
			check(ngp_hip_multi_generate_next_inputs_list(stream, n_active[n], list, g, pn, m_net_in.as<NgpCoord>(), n_steps, props_dev + n), "multi_generate_next_inputs");
			NgpNetVariant nv{0u, f.n_rgb_hidden_layers, nullptr, nullptr, nullptr, 0u};
			check(ngp_hip_nerf_inference_ws(stream, f.desc_gpu.as<NgpNetDesc>(), f.params.as<uint16_t>(), m_net_in.as<float>(), 7, n_network_elements, m_net_out.as<uint16_t>(), OUT_STRIDE,
			                                m_enc_ws.data(), m_enc_ws.bytes(), f.n_rgb_hidden_layers == 2 ? nullptr : &nv), "nerf_inference (multi)");
			n_samples += n_network_elements;
			check(ngp_hip_multi_composite_list(stream, n_active[n], list, i, g, pn, m_net_in.as<NgpCoord>(), m_net_out.as<uint16_t>(), OUT_STRIDE, n_steps, (int)f.rgb_activation, (int)f.density_activation,
			                                   f.min_transmittance, props_dev + n), "multi_composite");
		}
		i += n_steps;
	}
	uint32_t n_hit = 0;
	HIP_TRY(hipMemcpyAsync(&n_hit, hit_counter, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	check(ngp_hip_multi_shade(stream, n_hit, m_hit.as<NgpGlobalRay>(), 0 /* :602 passes train_in_linear_colors = false */, rb.frame_buffer.as<float>(), rb.depth_buffer.as<float>(), &ds,
	                          request.output.flip_y ? 1 : 0), "multi_shade");
	return n_samples;
}

NerfRenderer::~NerfRenderer() { if (m_host_words) (void)hipHostFree(m_host_words); }

} // namespace ngp
