// plumbing.cpp — the two "plumbing" configurations of SURVEY.md §8a next to the NeRF path: P1 image fitting (src/testbed_image.cu) and P2 SDF fitting
// (the training step of src/testbed_sdf.cu:1229-1252 on user-provided samples).  Both are a tcnn NetworkWithInputEncoding — HashGrid over 2 / 3
// dimensions -> FullyFusedMLP 64 x 2 hidden -> 16 — trained by Trainer::training_step + optimizer_step(128); here: ngp_hip_gridmlp_* + ngp_hip_loss_and_gradient
// + the same Adam / ExponentialDecay / Ema kernel as the NeRF path, on the Testbed's own parameter / optimizer buffers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "image_io.h"
#include "snapshot.h"
#include "testbed.h"

namespace ngp {

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
static constexpr uint32_t GM_OUT_STRIDE = 4;
static constexpr float LOSS_SCALE = 128.0f;                 // testbed.h:272, optimizer_step(stream, 128)
static constexpr uint32_t BATCH_SIZE_GRANULARITY = 128;     // tcnn::batch_size_granularity
static inline uint32_t next_multiple(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

void Testbed::reset_network_gridmlp() {  // testbed.cu:2249-2470, Image / Sdf branch (2397-2445)
	join_side_ema();
	tuner_reset();
	m_rng = Pcg32(m_seed);
	m_windowless_render_surface.reset_accumulation();
	const Json& config = m_network_config;
	const Json empty = Json::object();
	const Json& enc = config.contains("encoding") ? config["encoding"] : empty;
	auto require = [](bool ok, const std::string& what) { if (!ok) throw std::runtime_error{"network config not supported by the gfx950 fused kernels: " + what}; };
	std::string enc_type = enc.value("otype", "HashGrid"); std::transform(enc_type.begin(), enc_type.end(), enc_type.begin(), ::tolower);
	require(enc_type == "hashgrid", "encoding.otype must be HashGrid");
	require((uint32_t)enc.value("n_features_per_level", 2) == 2 && (uint32_t)enc.value("n_levels", 16) == 16, "HashGrid must have 16 levels x 2 features");
	if (config.contains("network")) require(config["network"].value("n_neurons", 64) == 64 && config["network"].value("n_hidden_layers", 2) == 2, "network must be 64 neurons x 2 hidden layers");
	m_num_levels = 16;
	const uint32_t n_dims = gridmlp_n_dims();
	const uint32_t log2_hashmap_size = (uint32_t)enc.value("log2_hashmap_size", 15);
	m_base_grid_resolution = (uint32_t)enc.value("base_resolution", 0);
	if (!m_base_grid_resolution) m_base_grid_resolution = 1u << (log2_hashmap_size / n_dims);
	float desired_resolution = 2048.0f;   // finest level over the unit cube (2315); Image: half the image resolution (2317)
	if (m_testbed_mode == ETestbedMode::Image) desired_resolution = (float)std::max(m_image.resolution[0], m_image.resolution[1]) / 2.0f;
	m_per_level_scale = (float)enc.value("per_level_scale", 0.0);
	if (m_per_level_scale <= 0.0f) m_per_level_scale = std::exp(std::log(desired_resolution * 1.0f / (float)m_base_grid_resolution) / (float)(m_num_levels - 1));
	check(ngp_hip_gridmlp_make_desc_host(n_dims, m_num_levels, log2_hashmap_size, m_base_grid_resolution, m_per_level_scale, &m_desc), "ngp_hip_gridmlp_make_desc_host");
	m_n_params = ngp_hip_gridmlp_n_params_host(&m_desc);
	m_n_matrix_params = NGP_GRIDMLP_N_PARAMS;
	m_desc_gpu.resize(sizeof(NgpNetDesc));
	m_desc_gpu.copy_from_host(&m_desc, sizeof(NgpNetDesc));
	std::string loss = config.contains("loss") ? config["loss"].value("otype", "L2") : std::string("L2");
	std::transform(loss.begin(), loss.end(), loss.begin(), ::tolower);
	if (loss == "l2") m_nerf.training.loss_type = ELossType::L2;
	else if (loss == "relativel2") m_nerf.training.loss_type = ELossType::RelativeL2;
	else if (loss == "l1") m_nerf.training.loss_type = ELossType::L1;
	else if (loss == "mape") m_nerf.training.loss_type = ELossType::Mape;
	else throw std::runtime_error{"loss '" + loss + "' is not implemented for the grid -> MLP configs (L2, RelativeL2, L1, MAPE are)"};
	if (config.contains("optimizer")) parse_optimizer_config(config["optimizer"]);
	m_optimizer_step = 0;
	m_params.resize(m_n_params * 2); m_inference_params.resize(m_n_params * 2); m_grads.resize(m_n_params * 2);
	m_master.resize(m_n_params * 4); m_first_moments.resize(m_n_params * 4); m_second_moments.resize(m_n_params * 4); m_ema.resize(m_n_params * 4);
	m_first_moments.memset(0, m_stream); m_second_moments.memset(0, m_stream); m_ema.memset(0, m_stream); m_grads.memset(0, m_stream);
	check(ngp_hip_gridmlp_init_params(m_stream, &m_desc, m_seed, m_master.as<float>(), m_params.as<uint16_t>(), m_inference_params.as<uint16_t>()), "ngp_hip_gridmlp_init_params");
	m_loss_scalar_gpu.resize(4);
	m_training_step = 0;
	m_loss_scalar = 0.f;
	sync();
}

// Trainer::training_step (forward, loss, backward) + optimizer_step(128) + Trainer::loss (sum of the per-element values)
void Testbed::gridmlp_training_step(const float* pos, uint32_t n_dims, const float* targets, uint32_t dims, uint32_t n, bool get_loss_scalar) {
	if (n % 256) throw std::runtime_error{"training batch size must be a multiple of 256"};
	m_gm_out.enlarge((size_t)n * GM_OUT_STRIDE * 2); m_x_saved.enlarge((size_t)n * 32 * 2); m_dloss.enlarge((size_t)n * GM_OUT_STRIDE * 2);
	m_gm_values.enlarge((size_t)n * dims * 4);
	m_bwd_scratch.enlarge(ngp_hip_gridmlp_backward_scratch_bytes(n));
	const NgpNetDesc* desc = m_desc_gpu.as<NgpNetDesc>();
	profile_begin(PK_FORWARD);
	// fused or encode + MLP kernel: measured on this workload (testbed.h ENetworkPass) — random SDF points want the two-kernel pass, stratified image batches the fused one
	if (tuner_pick(m_stream)) {
		m_enc_ws.enlarge(ngp_hip_nerf_encode_workspace_bytes(n));
		check(ngp_hip_gridmlp_forward_ws(m_stream, n_dims, desc, m_params.as<uint16_t>(), pos, n_dims, n, m_gm_out.as<uint16_t>(), GM_OUT_STRIDE, m_x_saved.as<uint16_t>(), m_enc_ws.data(), m_enc_ws.bytes()), "gridmlp_forward (two kernels)");
	} else {
		check(ngp_hip_gridmlp_forward(m_stream, n_dims, desc, m_params.as<uint16_t>(), pos, n_dims, n, m_gm_out.as<uint16_t>(), GM_OUT_STRIDE, m_x_saved.as<uint16_t>()), "gridmlp_forward");
	}
	tuner_done(m_stream);
	profile_end(PK_FORWARD, n);
	check(ngp_hip_loss_and_gradient(m_stream, (int)m_nerf.training.loss_type, n, dims, LOSS_SCALE, m_gm_out.as<uint16_t>(), GM_OUT_STRIDE, targets, m_gm_values.as<float>(), m_dloss.as<uint16_t>(),
	                                GM_OUT_STRIDE), "loss_and_gradient");
	profile_begin(PK_BACKWARD);
	check(ngp_hip_gridmlp_backward(m_stream, n_dims, desc, m_params.as<uint16_t>(), pos, n_dims, n, m_x_saved.as<uint16_t>(), m_dloss.as<uint16_t>(), GM_OUT_STRIDE, m_grads.as<uint16_t>(),
	                               m_bwd_scratch.data(), m_bwd_scratch.bytes()), "gridmlp_backward");
	profile_end(PK_BACKWARD, n);
	if (get_loss_scalar) {
		check(ngp_hip_reduce_sum_f32(m_stream, m_gm_values.as<float>(), n * dims, m_loss_scalar_gpu.as<float>()), "reduce_sum");
		sync();
		m_loss_scalar_gpu.copy_to_host(&m_loss_scalar, 4);
	}
	optimizer_step();
	++m_training_step;
}

// ---- P1: image ------------------------------------------------------------------------------------------------------------------------
void Testbed::load_image(const std::string& path) {  // Testbed::load_image (testbed_image.cu:362-434): .exr -> float RGBA, .bin -> half RGBA, else the 8-bit decoders
	auto ends_with = [&](const char* ext) { const size_t n = strlen(ext); if (path.size() < n) return false; std::string tail = path.substr(path.size() - n); std::transform(tail.begin(), tail.end(), tail.begin(), ::tolower); return tail == ext; };
	if (ends_with(".exr")) {   // load_exr_image (385-397)
		int w = 0, h = 0; std::vector<float> px;
		read_exr_rgba_f32(path, w, h, px);
		set_image_data(w, h, px.data());
		m_data_path = path;
		return;
	}
	if (!ends_with(".bin")) {  // load_stbi_image (399-412): load_stbi hands back linear premultiplied floats (tinyexr_wrapper.cu load_stbi: srgb_to_linear on rgb, alpha / 255)
		int w = 0, h = 0; std::vector<uint8_t> px8;
		read_image_rgba8(path, w, h, px8);
		std::vector<float> px((size_t)w * h * 4);
		for (size_t i = 0; i < (size_t)w * h; ++i) {
			const float a = (float)px8[i * 4 + 3] * (1.0f / 255.0f);
			for (int c = 0; c < 3; ++c) {
				const float v = (float)px8[i * 4 + c] * (1.0f / 255.0f);
				px[i * 4 + c] = (v <= 0.04045f ? v / 12.92f : std::pow((v + 0.055f) / 1.055f, 2.4f)) * a;
			}
			px[i * 4 + 3] = a;
		}
		set_image_data(w, h, px.data());
		m_data_path = path;
		return;
	}
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) throw std::runtime_error{path + " does not exist."};
	int32_t hw[2];
	if (fread(hw, 4, 2, f) != 2 || hw[0] <= 0 || hw[1] <= 0) { fclose(f); throw std::runtime_error{"bad .bin image header in " + path}; }
	const size_t n_pixels = (size_t)hw[0] * hw[1];
	std::vector<uint16_t> px(n_pixels * 4);
	const size_t got = fread(px.data(), 2, px.size(), f);
	fclose(f);
	if (got != px.size()) throw std::runtime_error{"truncated .bin image " + path};
	m_image.resolution[1] = hw[0]; m_image.resolution[0] = hw[1];
	m_image.data.resize(px.size() * 2);
	m_image.data.copy_from_host(px.data(), px.size() * 2);
	m_image.type = 2;
	m_data_path = path;
	m_training_data_available = true;
}

void Testbed::set_image_data(int w, int h, const float* rgba_host) {
	if (m_testbed_mode != ETestbedMode::Image) throw std::runtime_error{"set_image_data: TestbedMode.Image only"};
	m_image.resolution[0] = w; m_image.resolution[1] = h;
	m_image.data.resize((size_t)w * h * 16);
	m_image.data.copy_from_host(rgba_host, (size_t)w * h * 16);
	m_image.type = 3;
	m_training_data_available = true;
}

void Testbed::train_image(uint32_t batch_size, bool get_loss_scalar) {  // testbed_image.cu:220-291
	const uint32_t n = batch_size;
	m_image.positions.enlarge((size_t)n * 8); m_image.targets.enlarge((size_t)n * 12);
	// generate_random_uniform<float>(stream, m_rng, n * 2, positions) [tcnn], then stratification when the batch is a square power of two (236-247)
	check(ngp_hip_generate_random_uniform(m_stream, m_rng.state, m_rng.inc, n * 2, m_image.positions.as<float>()), "generate_random_uniform");
	m_rng.advance((int64_t)n * 2);
	if (m_image.stratified) {
		uint32_t log2_batch = 0;
		while ((1u << log2_batch) < n) ++log2_batch;
		if ((1u << log2_batch) == n && log2_batch % 2 == 0) check(ngp_hip_image_stratify2(m_stream, n, log2_batch, m_image.positions.as<float>()), "stratify2");
	}
	check(ngp_hip_image_eval_and_snap(m_stream, n, m_image.data.data(), m_image.type, m_image.positions.as<float>(), m_image.resolution, m_image.targets.as<float>(), 3,
	                                  m_image.snap_to_pixel_centers, m_image.linear_colors), "eval_image_and_snap");
	gridmlp_training_step(m_image.positions.as<float>(), 2, m_image.targets.as<float>(), 3, n, get_loss_scalar);
}

void Testbed::render_image(RenderBuffer& rb) {  // testbed_image.cu:293-360 (no activation visualisation)
	const uint32_t n_pixels = (uint32_t)rb.res[0] * (uint32_t)rb.res[1];
	const uint32_t n_elements = next_multiple(n_pixels, BATCH_SIZE_GRANULARITY);
	m_image.render_coords.enlarge((size_t)n_elements * 8); m_image.render_out.enlarge((size_t)n_elements * GM_OUT_STRIDE * 2);
	const float sc[2] = {m_screen_center[0] - 0.5f, m_screen_center[1] - 0.5f};
	HIP_TRY(hipMemsetAsync(m_image.render_coords.data(), 0, (size_t)n_elements * 8, (hipStream_t)m_stream));
	check(ngp_hip_image_init_coords(m_stream, m_image.render_coords.as<float>(), rb.res, m_image.resolution, m_scale, m_image.pos, sc, m_snap_to_pixel_centers, rb.spp), "image_init_coords");
	join_side_ema();
	check(ngp_hip_gridmlp_forward(m_stream, 2, m_desc_gpu.as<NgpNetDesc>(), m_inference_params.as<uint16_t>(), m_image.render_coords.as<float>(), 2, n_elements, m_image.render_out.as<uint16_t>(),
	                              GM_OUT_STRIDE, nullptr), "gridmlp_forward (render)");
	check(ngp_hip_image_shade(m_stream, rb.res, m_image.render_coords.as<float>(), m_image.render_out.as<uint16_t>(), GM_OUT_STRIDE, rb.frame_buffer.as<float>(), rb.depth_buffer.as<float>(),
	                          m_image.linear_colors), "image_shade");
}

float Testbed::compute_image_mse(bool quantize_to_byte) {  // testbed_image.cu:461-523
	if (m_testbed_mode != ETestbedMode::Image || m_n_params == 0 || !m_image.type) throw std::runtime_error{"compute_image_mse: needs Image mode with an image and a network"};
	const uint32_t n_elements = (uint32_t)m_image.resolution[0] * (uint32_t)m_image.resolution[1];
	const uint32_t max_batch = 1u << 20;
	m_image.se.enlarge((size_t)(n_elements + 256) * 4);
	DeviceBuffer pos, targets, pred;
	pos.resize((size_t)max_batch * 8); targets.resize((size_t)max_batch * 12); pred.resize((size_t)max_batch * GM_OUT_STRIDE * 2);
	for (uint32_t offset = 0; offset < n_elements; offset += max_batch) {
		const uint32_t count = std::min(max_batch, n_elements - offset);
		const uint32_t batch = (count + 255u) & ~255u;
		check(ngp_hip_image_coords_from_idx(m_stream, batch, offset, pos.as<float>(), m_image.resolution), "image_coords_from_idx");
		check(ngp_hip_image_eval_and_snap(m_stream, batch, m_image.data.data(), m_image.type, pos.as<float>(), m_image.resolution, targets.as<float>(), 3, 1, m_image.linear_colors), "eval_image_and_snap");
		join_side_ema();
		check(ngp_hip_gridmlp_forward(m_stream, 2, m_desc_gpu.as<NgpNetDesc>(), m_inference_params.as<uint16_t>(), pos.as<float>(), 2, batch, pred.as<uint16_t>(), GM_OUT_STRIDE, nullptr), "gridmlp_forward (mse)");
		check(ngp_hip_image_mse(m_stream, count, targets.as<float>(), pred.as<uint16_t>(), GM_OUT_STRIDE, m_image.se.as<float>() + offset, quantize_to_byte), "image_mse");
	}
	check(ngp_hip_reduce_sum_f32(m_stream, m_image.se.as<float>(), n_elements, m_loss_scalar_gpu.as<float>()), "reduce_sum");
	sync();
	float sum = 0.f;
	m_loss_scalar_gpu.copy_to_host(&sum, 4);
	return sum / (float)n_elements;
}

// ---- P2: sdf (training step on provided samples) --------------------------------------------------------------------------------------------
void Testbed::override_sdf_training_data(const float* points, const float* distances, size_t n) {  // python_api.cu:74-100
	if (m_testbed_mode != ETestbedMode::Sdf) throw std::runtime_error{"override_sdf_training_data: TestbedMode.Sdf only"};
	if (n == 0) throw std::runtime_error{"Invalid Points<->Distances data"};
	// the reference maps the points into the unit cube of the loaded mesh (raw_aabb, mesh_scale); without a mesh the caller provides unit-cube data
	m_sdf.positions.resize(n * 12); m_sdf.distances.resize(n * 4);
	m_sdf.positions.copy_from_host(points, n * 12);
	m_sdf.distances.copy_from_host(distances, n * 4);
	m_sdf.n_samples = (uint32_t)n; m_sdf.cursor = 0;
	m_training_data_available = true;
}

void Testbed::train_sdf(uint32_t batch_size, bool get_loss_scalar) {  // testbed_sdf.cu:1229-1252
	if (m_sdf.n_samples < batch_size) throw std::runtime_error{"train(): fewer SDF samples provided than the batch size"};
	if (m_sdf.cursor + batch_size > m_sdf.n_samples) m_sdf.cursor = 0;   // walk the provided set batch by batch
	const float* pos = m_sdf.positions.as<float>() + (size_t)m_sdf.cursor * 3;
	const float* dist = m_sdf.distances.as<float>() + m_sdf.cursor;
	m_sdf.cursor += batch_size;
	gridmlp_training_step(pos, 3, dist, 1, batch_size, get_loss_scalar);
}

} // namespace ngp
