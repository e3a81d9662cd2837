// nerf_loader.h — transforms.json + image ingest (src/nerf_loader.cu:197-747, Testbed::load_nerf at src/testbed_nerf.cu:2735-2759).
// Host-only first stage (JSON keys, path resolution, frame ordering / culling, PNG / JPEG / EXR decode, alpha / mask / depth / ray files, RGBA8 fix-ups of convert_rgba32), so that it can be
// tested without a GPU; Testbed::load_training_data uploads the result.
#pragma once
#include <string>
#include <vector>

#include "ngp_hip.h"
#include "testbed.h"

namespace ngp {

struct LoadedNerfData {
	size_t n_images = 0;
	std::vector<std::string> paths;                 // as written in the json (NerfDataset::paths)
	std::vector<NgpXForm> xforms;                   // already in the NGP convention (nerf_matrix_to_ngp)
	std::vector<NgpImageMeta> metadata;             // .pixels unset
	std::vector<std::vector<uint8_t>> pixels;       // per image: RGBA8 (EImageDataType::Byte, PNG / JPEG) or RGBA fp16 bits (Half, EXR)
	std::vector<int> image_type;                    // 1 = Byte, 2 = Half
	std::vector<std::vector<uint16_t>> depth16;     // `depth_path` images (16-bit, one channel); empty = none
	std::vector<float> depth_scale;                 // `integer_depth_scale` (LoadedImageInfo::depth_scale, -1 = unset); multiplied by `scale` at upload
	std::vector<std::vector<NgpRay>> rays;          // rays_<image>.dat, already in the NGP frame; empty = none
	bool has_rays = false;
	float sharpen_amount = 0.f;                     // `sharpen` key or the caller's default: applied on the device after upload (nerf_loader.cu:803-825)
	float scale = 1.0f;                             // NERF_SCALE (nerf_loader.h:28)
	Vec3 offset{0.f, 0.f, 0.f};
	int aabb_scale = 1;
	bool from_mitsuba = false, is_hdr = false, wants_importance_sampling = true;
	NgpAabb render_aabb{{1e30f, 1e30f, 1e30f}, {-1e30f, -1e30f, -1e30f}};
	Vec3 up{0.0f, 1.0f, 0.0f};
	uint32_t n_extra_learnable_dims = 0;            // `n_extra_learnable_dims` (nerf_loader.cu:480-481)
	bool has_light_dirs = false;                    // some frame carries `driver_parameters` (671-680): three light-direction dims, no latent codes
	std::vector<Vec3> light_dirs;                   // per image, NGP frame, normalised
	std::vector<float> envmap_data;                 // `envmap` key (nerf_loader.cu:533-546): linear premultiplied RGBA fp32, [h][w][4]
	int envmap_resolution[2] = {0, 0};
};

// a .json file, or a directory (every *.json in it), like Testbed::load_nerf
std::vector<std::string> resolve_nerf_json_paths(const std::string& data_path);
LoadedNerfData load_nerf_host(const std::vector<std::string>& jsonpaths, float sharpen_amount = 0.f);
// convert_rgba32 (nerf_loader.cu:59-81) on the host
void convert_rgba32_host(size_t n_pixels, uint8_t* rgba, bool white_transparent, bool black_transparent, uint32_t mask_color);

} // namespace ngp
