// nerf_renderer.h — host side of the Blender add-on's multi-NeRF renderer: the request schema that crosses pybind11
// (include/neural-graphics-primitives/nerf/render_request.cuh, nerf_descriptor.cuh, render_modifiers*.cuh, mask_3D.cuh,
// camera_models.cuh, common.h:300-355), one NeuralRadianceField per snapshot path (neural_radiance_field.cuh) and the render loop
// (src/nerf_renderer.cu:565-791) over the ngp_hip_multi_* kernels.
#pragma once
#include <algorithm>
#include <cmath>

#include <memory>
#include <string>
#include <vector>

#include "ngp_hip.h"
#include "testbed.h"

namespace ngp {

struct Mat4 {  // column-major 4x4 (Eigen::Matrix4f)
	float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
	Mat4 inverse() const;
	Mat4 operator*(const Mat4& o) const;
};

struct BoundingBox {  // bounding_box.cuh:43-268 (the members the request schema and the add-on use)
	Vec3 min{1e30f, 1e30f, 1e30f}, max{-1e30f, -1e30f, -1e30f};
	BoundingBox() = default;
	BoundingBox(const Vec3& a, const Vec3& b) : min(a), max(b) {}
	Vec3 center() const { return Vec3{0.5f * (min.x + max.x), 0.5f * (min.y + max.y), 0.5f * (min.z + max.z)}; }
	Vec3 diag() const { return Vec3{max.x - min.x, max.y - min.y, max.z - min.z}; }
	bool contains(const Vec3& p) const { return p.x >= min.x && p.x <= max.x && p.y >= min.y && p.y <= max.y && p.z >= min.z && p.z <= max.z; }
	void enlarge(const Vec3& p);
	void enlarge(const BoundingBox& o) { enlarge(o.min); enlarge(o.max); }
	void inflate(float amount);
	Vec3 relative_pos(const Vec3& p) const;
	NgpAabb pod() const { return NgpAabb{{min.x, min.y, min.z}, {max.x, max.y, max.z}}; }
	bool is_empty() const { return max.x < min.x || max.y < min.y || max.z < min.z; }
	BoundingBox intersection(const BoundingBox& o) const {   // bounding_box.cuh:94-103
		return BoundingBox(Vec3{std::max(min.x, o.min.x), std::max(min.y, o.min.y), std::max(min.z, o.min.z)}, Vec3{std::min(max.x, o.max.x), std::min(max.y, o.max.y), std::min(max.z, o.max.z)});
	}
	bool intersects(const BoundingBox& o) const { return !intersection(o).is_empty(); }
	float distance_sq(const Vec3& p) const {                  // bounding_box.cuh:246-248
		const float d[3] = {std::max(std::max(min.x - p.x, p.x - max.x), 0.f), std::max(std::max(min.y - p.y, p.y - max.y), 0.f), std::max(std::max(min.z - p.z, p.z - max.z), 0.f)};
		return d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
	}
	float signed_distance(const Vec3& p) const {              // bounding_box.cuh:250-253 (as written there: |p - min| - diag)
		const float q[3] = {std::fabs(p.x - min.x) - (max.x - min.x), std::fabs(p.y - min.y) - (max.y - min.y), std::fabs(p.z - min.z) - (max.z - min.z)};
		const float m0 = std::max(q[0], 0.f), m1 = std::max(q[1], 0.f), m2 = std::max(q[2], 0.f);
		return std::sqrt(m0 * m0 + m1 * m1 + m2 * m2) + std::min(std::max(q[0], std::max(q[1], q[2])), 0.0f);
	}
	void ray_intersect(const Vec3& pos, const Vec3& dir, float t[2]) const {   // bounding_box.cuh:163-210 (the slab test; miss = both FLT_MAX)
		const float FMAX = 3.402823466e+38f;
		float tmin = (min.x - pos.x) / dir.x, tmax = (max.x - pos.x) / dir.x;
		if (tmin > tmax) std::swap(tmin, tmax);
		float tymin = (min.y - pos.y) / dir.y, tymax = (max.y - pos.y) / dir.y;
		if (tymin > tymax) std::swap(tymin, tymax);
		if (tmin > tymax || tymin > tmax) { t[0] = t[1] = FMAX; return; }
		if (tymin > tmin) tmin = tymin;
		if (tymax < tmax) tmax = tymax;
		float tzmin = (min.z - pos.z) / dir.z, tzmax = (max.z - pos.z) / dir.z;
		if (tzmin > tzmax) std::swap(tzmin, tzmax);
		if (tmin > tzmax || tzmin > tmax) { t[0] = t[1] = FMAX; return; }
		if (tzmin > tmin) tmin = tzmin;
		if (tzmax < tmax) tmax = tzmax;
		t[0] = tmin; t[1] = tmax;
	}
};

enum class EMaskMode : int { Add, Subtract };
enum class EMaskShape : int { Box, Cylinder, Sphere, All };
enum class ECameraModel : int { Perspective, QuadrilateralHexahedron, SphericalQuadrilateral };  // camera_models.cuh:27-31

struct Mask3D {  // mask_3D.cuh:129-255
	NgpMask3D pod{};
	Mask3D() { pod.mode = 0; pod.shape = 0; Mat4 id; memcpy(pod.transform, id.m, 64); memcpy(pod.itransform, id.m, 64); }
	Mask3D(EMaskShape shape, const Mat4& transform, EMaskMode mode, const float config[6], float feather, float opacity);
	static Mask3D All(EMaskMode mode);
	static Mask3D Box(const Vec3& dims, const Mat4& transform, EMaskMode mode, float feather, float opacity);
	static Mask3D Cylinder(float radius, float height, const Mat4& transform, EMaskMode mode, float feather, float opacity);
	static Mask3D Sphere(float radius, const Mat4& transform, EMaskMode mode, float feather, float opacity);
	Mask3D transformed_by(const Mat4& matrix) const;
};

struct RenderModifiersDescriptor { std::vector<Mask3D> masks; };  // render_modifiers_descriptor.cuh

struct DownsampleInfo {  // common.h:300-355
	NgpDownsampleInfo pod{};
	static DownsampleInfo MakeFromMip(int res_x, int res_y, uint32_t mip);
};

struct Quadrilateral3D { Vec3 tl, tr, bl, br; Vec3 center() const; };
struct QuadrilateralHexahedron { Quadrilateral3D front, back; Vec3 center() const; };
struct SphericalQuadrilateral { float width = 0, height = 0, curvature = 0; };

struct RenderOutputProperties {  // render_request.cuh:17-52
	int32_t resolution[2] = {0, 0};
	DownsampleInfo ds;
	uint32_t spp = 1;
	EColorSpace color_space = EColorSpace::Linear;
	ETonemapCurve tonemap_curve = ETonemapCurve::Identity;
	float exposure = 0.f;
	float background_color[4] = {0, 0, 0, 0};
	bool flip_y = false;
};

struct RenderCameraProperties {  // render_request.cuh:55-103
	Mat34 transform;
	ECameraModel model = ECameraModel::Perspective;
	float focal_length = 1.f;
	SphericalQuadrilateral spherical_quadrilateral;
	QuadrilateralHexahedron quadrilateral_hexahedron;
	float near_distance = 0.f, aperture_size = 0.f, focus_z = 1.f;
	bool operator==(const RenderCameraProperties& o) const;
	bool operator!=(const RenderCameraProperties& o) const { return !(*this == o); }
	NgpRenderCamera pod() const;
};

struct NerfDescriptor {  // nerf_descriptor.cuh:15-35
	std::string snapshot_path;
	BoundingBox aabb;
	Mat4 transform;
	RenderModifiersDescriptor modifiers;
	float opacity = 1.f;
};

struct RenderRequest {  // render_request.cuh:105-125
	RenderOutputProperties output;
	RenderCameraProperties camera;
	RenderModifiersDescriptor modifiers;
	std::vector<NerfDescriptor> nerfs;
	BoundingBox aabb;
};

// one trained field per snapshot file, loaded lazily and kept across requests (render_data.cuh:38-93)
class NeuralRadianceField {
public:
	explicit NeuralRadianceField(const std::string& path) : snapshot_path(path) {}
	void load_snapshot(void* stream);  // neural_radiance_field.cuh:153-298 + nerf_data.cu:61-88
	std::string snapshot_path;
	bool is_loaded = false;
	NgpNetDesc desc{};
	uint32_t n_rgb_hidden_layers = 2;   // of the snapshot's rgb_network
	DeviceBuffer desc_gpu, params, density_grid, density_grid_bitfield, density_grid_mean;
	NgpAabb train_aabb{{0, 0, 0}, {1, 1, 1}};
	uint32_t grid_size = 128, max_cascade = 0, num_cascades = 8, aabb_scale = 1;
	float cone_angle_constant = 1.0f / 256.0f;
	float min_transmittance = 0.01f;
	ENerfActivation rgb_activation = ENerfActivation::Logistic, density_activation = ENerfActivation::Exponential;
	float min_cone_step_size() const { return 1.73205080757f / 1024.0f; }                                             // :66-72
	float max_cone_step_size() const { return min_cone_step_size() * (float)(1u << (num_cascades - 1)) * 1024.0f / (float)grid_size; }
};

class NerfRenderer {
public:
	// returns the number of network samples evaluated; rb.frame_buffer / depth_buffer must be cleared by the caller (bl_render_frame)
	uint64_t render(RenderBuffer& rb, const RenderRequest& request, void* stream);
	size_t n_loaded_fields() const { return m_fields.size(); }
	~NerfRenderer();
	bool trace = false;   // stage markers / pass structure on stderr
	// the pass loop: fused (default) = one ngp_hip_multi_advance launch (march + cull + compact + lists) and one host-mailbox poll per pass, the stock tracer's
	// sample budget per pass; unfused = the launch sequence of nerf_renderer.cu:664-791 with its two blocking read-backs (kept as the checker: same pixels)
	bool fused_passes = true;
	uint32_t max_skips_per_pass = 96;     // empty voxels a marched proxy may cross in the advance launch before the ray sits the pass out (0: never rests)
	float pass_samples_factor = 4.0f;     // network samples per pass = factor x pixels (the reference: 1 x, at most 8 steps)
	uint32_t max_steps_per_pass = 64;
	bool reference_schedule = false;      // fused loop on the unfused loop's schedule: n_steps from the rays that ENTERED the pass (clamp(pixels / n, 1, max_steps)), no rest — the same frame, bit for bit
	uint32_t last_n_passes = 0;
private:
	uint64_t* m_host_words = nullptr;     // mapped host memory: the advance launch's mailbox
	uint32_t m_host_words_n = 0, m_sequence = 0;
	std::vector<std::unique_ptr<NeuralRadianceField>> m_fields;
	DeviceBuffer m_global[2], m_proxy[2], m_hit, m_net_in, m_net_out, m_counters, m_props_gpu, m_masks_gpu, m_enc_ws, m_active_lists, m_active_counts;
};

} // namespace ngp
