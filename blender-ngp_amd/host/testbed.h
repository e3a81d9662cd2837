// testbed.h — host-side orchestrator: the NeRF part of the reference's `Testbed` (include/neural-graphics-primitives/testbed.h,
// src/testbed.cu, src/testbed_nerf.cu host functions), re-implemented over the C ABI of libngp_hip.so (include/ngp_hip.h).
// No Eigen, no tiny-cuda-nn, no CUDA: plain C++17 + the HIP runtime API for memory / streams.
//
// Kept from the reference: names and meaning of the public methods / properties that scripts/run.py and the Blender add-on use
// (src/python_api.cu:540-732), the training schedule (Testbed::train, src/testbed.cu:2527-2587), the step structure
// (train_nerf / train_nerf_step, src/testbed_nerf.cu:2896-3385), the render loop (NerfTracer, 2047-2267) and error behaviour
// (std::runtime_error -> Python RuntimeError).
#pragma once

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

#include "cam_adam.h"
#include "mini_json.h"
#include "ngp_hip.h"

namespace ngp {

// ---- enums (include/neural-graphics-primitives/common.h:65-160)
enum class ETestbedMode : int { Nerf, Sdf, Image, Volume };
enum class ERenderMode : int { AO, Shade, Normals, Positions, Depth, Distortion, Cost, Slice };
enum class ELossType : int { L2, L1, Mape, Smape, Huber, LogL1, RelativeL2 };
enum class ENerfActivation : int { None, ReLU, Logistic, Exponential };
enum class EColorSpace : int { Linear, SRGB, VisPosNeg };
enum class ETonemapCurve : int { Identity, ACES, Hable, Reinhard };
enum class ELensMode : int { Perspective, OpenCV, FTheta, LatLong };

struct Vec3 { float x = 0, y = 0, z = 0; };
struct Mat34 { float m[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0}; };  // column-major 3x4 (Eigen default)

// ---- RAII device memory (replaces tcnn::GPUMemory / GPUMemoryArena)
class DeviceBuffer {
public:
	DeviceBuffer() = default;
	~DeviceBuffer();
	DeviceBuffer(const DeviceBuffer&) = delete;
	DeviceBuffer& operator=(const DeviceBuffer&) = delete;
	DeviceBuffer(DeviceBuffer&& o) noexcept { *this = std::move(o); }
	DeviceBuffer& operator=(DeviceBuffer&& o) noexcept;
	void resize(size_t bytes);            // reallocates if size differs; contents undefined
	void enlarge(size_t bytes);           // grows only
	void free();
	void memset(int value, void* stream = nullptr);
	void copy_from_host(const void* src, size_t bytes, size_t dst_offset = 0);
	void copy_to_host(void* dst, size_t bytes, size_t src_offset = 0) const;
	template <typename T> T* as() const { return (T*)m_ptr; }
	void* data() const { return m_ptr; }
	size_t bytes() const { return m_bytes; }
	static size_t total_allocated();
private:
	void* m_ptr = nullptr;
	size_t m_bytes = 0;
};

// host pcg32 (tcnn pcg32.h), same stream as the device one
struct Pcg32 {
	uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
	Pcg32() = default;
	explicit Pcg32(uint64_t seed) { this->seed(seed, 1); }
	void seed(uint64_t initstate, uint64_t initseq) { state = 0; inc = (initseq << 1u) | 1u; next_uint(); state += initstate; next_uint(); }
	uint32_t next_uint() {
		uint64_t old = state;
		state = old * 0x5851f42d4c957f2dULL + inc;
		uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
		return (xs >> rot) | (xs << ((~rot + 1u) & 31));
	}
	float next_float() { uint32_t u = (next_uint() >> 9) | 0x3f800000u; float f; memcpy(&f, &u, 4); return f - 1.0f; }
	void advance(int64_t delta_ = (1ll << 32)) {
		uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = inc, acc_mult = 1u, acc_plus = 0u, delta = (uint64_t)delta_;
		while (delta > 0) {
			if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
			cur_plus = (cur_mult + 1) * cur_plus; cur_mult *= cur_mult; delta /= 2;
		}
		state = acc_mult * state + acc_plus;
	}
};

// ---- dataset (include/neural-graphics-primitives/nerf_loader.h:47-185)
struct NerfDataset {
	size_t n_images = 0;
	std::vector<NgpXForm> xforms;
	std::vector<std::string> paths;              // image paths, carried through snapshots (json_binding.h:133)
	std::vector<NgpImageMeta> metadata;          // host copy; .pixels are device pointers into pixelmemory
	std::vector<DeviceBuffer> pixelmemory;
	std::vector<DeviceBuffer> depthmemory;       // fp32 per pixel, already multiplied by depth_scale (nerf_loader.cu:785-802); empty = no depth
	std::vector<DeviceBuffer> raymemory;         // per-pixel rays (rays_<image>.dat), NGP frame; empty = pinhole / lens model (nerf_loader.cu:835-841)
	DeviceBuffer metadata_gpu;
	float scale = 1.0f;                          // NERF_SCALE = 1.0 in this fork (nerf_loader.h:28)
	Vec3 offset;                                 // {0,0,0} (nerf_loader.cu:186)
	int aabb_scale = 1;
	bool is_hdr = false;
	bool from_mitsuba = false;
	bool has_rays = false;
	NgpAabb render_aabb{{1e30f, 1e30f, 1e30f}, {-1e30f, -1e30f, -1e30f}};
	Vec3 up{0.0f, 1.0f, 0.0f};
	// per-image extra network inputs (nerf_loader.h:94-99): `n_extra_learnable_dims` latent codes, or 3 light-direction dims when frames carry `driver_parameters`
	// per image a 128 x 72 grid of tile sharpness (variance of the Laplacian; nerf_loader.cu:129-169, 178-179, 829-834), computed on the device when
	// include_sharpness_in_error first asks for it and again after an image changed
	DeviceBuffer sharpness_data;
	int sharpness_resolution[2] = {128, 72};
	bool sharpness_valid = false;
	void update_sharpness();
	uint32_t n_extra_learnable_dims = 0;
	bool has_light_dirs = false;
	std::vector<Vec3> light_dirs;                // per image, NGP frame, normalised (TrainingImageMetadata::light_dir)
	uint32_t n_extra_dims() const { return (has_light_dirs ? 3u : 0u) + n_extra_learnable_dims; }
	std::vector<float> envmap_data;              // `envmap` key of transforms.json (nerf_loader.h envmap_data; host copy, uploaded by reset_network like testbed.cu:2459-2461)
	int envmap_resolution[2] = {0, 0};

	Mat34 nerf_matrix_to_ngp(const Mat34& nerf_matrix, bool scale_columns = false) const;   // nerf_loader.h:113-132
	Mat34 ngp_matrix_to_nerf(const Mat34& ngp_matrix, bool scale_columns = false) const;    // nerf_loader.h:134-152
	void set_training_image(int frame_idx, int w, int h, const void* pixels_host, int image_data_type, const float* depth_host = nullptr, float depth_scale = -1.f); // nerf_loader.cu:749-
	void sharpen_training_image(int frame_idx, float sharpen_amount);   // nerf_loader.cu:803-825: Byte -> half4, then the unsharp filter
	void update_metadata(int first = 0, int last = -1);         // nerf_loader.cu:851-867
};

struct NerfCounters {  // testbed.h:369-381
	DeviceBuffer numsteps_counter, loss;   // (numsteps_counter_compacted: Testbed::m_gen_counters, word 2 of the step's slot)
	uint32_t rays_per_batch = 1 << 12;
	uint32_t n_rays_total = 0;
	uint32_t measured_batch_size = 0;
	uint32_t measured_batch_size_before_compaction = 0;
};

// trainable_buffer.cuh together with the tcnn Trainer / optimizer the reference pairs it with: an all-fp32 [h][w][n_dims] image trained by
// [Ema o] ExponentialDecay o Adam.  The environment map (N = 4; testbed.cu:2447-2462, testbed.h:936-944) and the lens-distortion map (N = 2;
// testbed.cu:2386-2396, testbed.h:946-952) are the two on the NeRF path.
struct TrainableBuffer {
	int resolution[2] = {0, 0}; uint32_t n_dims = 0;
	DeviceBuffer params, ema, gradients, gradient_weights, first_moments, second_moments;
	bool use_ema = false, has_decay = false;
	float ema_decay = 0.99f, decay_base = 1.f, base_learning_rate = 1e-3f, learning_rate = 1e-3f, beta1 = 0.9f, beta2 = 0.999f, epsilon = 1e-8f;
	uint32_t decay_start = 0, decay_interval = 1, decay_end = 0, step = 0;
	ELossType loss_type = ELossType::L2;
	size_t n_params() const { return (size_t)resolution[0] * (size_t)resolution[1] * n_dims; }
	float* params_inference() const { return (use_ema && step > 0) ? ema.as<float>() : params.as<float>(); }   // Ema's custom weights once it has stepped
	void reset(int w, int h, uint32_t dims, const Json& optimizer_config, void* stream);    // zero-initialised (trainable_buffer.cuh:64-69)
	void set_params(const float* host, size_t n);                                            // Trainer::set_params_full_precision
	void clear_gradients(void* stream, bool weights_too);
	void optimizer_step(void* stream, float loss_scale);                                     // Trainer::optimizer_step(stream, loss_scale)
};

class Testbed;

struct NerfTraining {
	Testbed* owner = nullptr;
	NerfDataset dataset;
	int n_images_for_training = 0;
	int n_images_for_training_prev = 0;
	std::vector<NgpXForm> transforms;
	DeviceBuffer transforms_gpu;
	DeviceBuffer cam_exposure_gpu;               // [n_images][3] log2 exposures the loss kernel applies to the targets (zeros until optimised)
	// optimize_exposure (testbed.h:647, 658-666; testbed_nerf.cu:2916-2919, 3056-3135): per-image Adam on the exposure gradient of the loss kernel
	struct ExposureAdam { uint32_t iter = 0; float m[3] = {0, 0, 0}, v[3] = {0, 0, 0}, x[3] = {0, 0, 0}; };
	std::vector<ExposureAdam> cam_exposure;
	DeviceBuffer cam_exposure_gradient_gpu;
	bool optimize_exposure = false;
	float exposure_l2_reg = 0.0f;
	// optimize_extrinsics (testbed.h:633-644, 651; testbed_nerf.cu:1600-1712, 2598-2633, 3056-3093): per-image position / angle-axis rotation offsets on top of
	// dataset.xforms, trained with the ray gradients the backward pass hands to ngp_hip_compute_cam_gradient, applied by update_transforms
	bool optimize_extrinsics = false;
	std::vector<Vec3Adam> cam_pos_offset;
	std::vector<RotationAdam> cam_rot_offset;
	DeviceBuffer cam_pos_gradient_gpu, cam_rot_gradient_gpu;       // [n_images][3] fp32 each, accumulated over n_steps_between_cam_updates steps
	bool cam_gradient_window_open = false;                         // the buffers hold this window's sums (cleared when the switch comes on mid-window)
	std::vector<float> cam_pos_gradient, cam_rot_gradient;         // host copies of the last update (python_api.cu does not expose them; tests do)
	void reset_camera_extrinsics();                                // testbed_nerf.cu:2543-2555 (optimizer states only; the transforms follow at the next update_transforms)
	// optimize_focal_length (testbed.h:652; testbed_nerf.cu:3095-3103): compute_cam_gradient_train_nerf receives cam_focal_length_gradient and never writes it, so
	// the reference's Adam step sees a zero gradient on a zero variable and the offset stays exactly 0 — and nothing reads the offset.  The switch is accepted and
	// trains nothing, like there.
	bool optimize_focal_length = false;
	// per-image latent codes (testbed.h:637-641, 660; testbed_nerf.cu:2297-2318, 2925-2931, 3029-3054): [(n_images + 1)][n_extra_dims] fp32, the last row is the
	// scratch slot of get_inference_extra_dims; every image has its own host Adam (AdamOptimizer<ArrayXf>, lr 1e-4 at construction, reset per step)
	bool optimize_extra_dims = false, optimize_distortion = false;
	DeviceBuffer extra_dims_gpu, extra_dims_gradient_gpu;
	struct ExtraDimsAdam { uint32_t iter = 0; std::vector<float> m, v, x; };
	std::vector<ExtraDimsAdam> extra_dims_opt;
	void reset_extra_dims(Pcg32& rng);
	bool distortion_gradient_window_open = false;                  // the distortion gradients of the current n_steps_between_cam_updates window are being accumulated
	bool train_envmap = false;                                     // testbed.h:656 (the reference sets it from its GUI only; exposed on pyngp here)
	bool include_sharpness_in_error = false;                       // testbed.h:670; 1476-1485, 2901-2912
	DeviceBuffer sharpness_grid;                                   // cascaded 128^3 x NERF_CASCADES grid of the sharpest image tile that has seen each cell
	float extrinsic_l2_reg = 1e-4f, extrinsic_learning_rate = 1e-3f, intrinsic_l2_reg = 1e-4f;   // testbed.h:673-678
	int view = 0;                                                  // current training view of the GUI navigation (testbed.h:636)
	uint32_t n_steps_between_cam_updates = 16, n_steps_since_cam_update = 0;
	NerfCounters counters_rgb;
	Pcg32 density_grid_rng;
	float near_distance = 0.2f;                  // testbed.h:676
	float density_grid_decay = 0.95f;            // testbed.h:675
	bool random_bg_color = true;                 // testbed.h:663
	bool linear_colors = false;                  // testbed.h:664
	bool snap_to_pixel_centers = false;          // testbed.h:665
	ELossType loss_type = ELossType::L2;
	ELossType depth_loss_type = ELossType::L1;   // testbed.h:654
	float depth_supervision_lambda = 0.f;        // testbed.h:680
	// error map (always-on accumulation, testbed_nerf.cu:2933-2939, 2971-3023)
	DeviceBuffer error_map_data;
	int32_t error_map_res[2] = {0, 0};
	uint32_t n_steps_since_error_map_update = 0;
	uint32_t n_steps_between_error_map_updates = 128;
	uint32_t n_rays_since_error_map_update = 0;
	// CDFs built from the error map every n_steps_between_error_map_updates steps (2973-3023); sampling from them is off by default
	DeviceBuffer cdf_x_cond_y, cdf_y, cdf_img;
	int32_t cdf_res[2] = {0, 0};
	std::vector<float> pmf_img_cpu;
	bool is_cdf_valid = false;
	bool sample_focal_plane_proportional_to_error = false;   // testbed.h:668
	bool sample_image_proportional_to_error = false;         // testbed.h:669
	uint32_t cdf_mode() const { return is_cdf_valid ? (sample_focal_plane_proportional_to_error ? 1u : 0u) | (sample_image_proportional_to_error ? 2u : 0u) : 0u; }
	const NgpErrorMapCdf* error_map_cdf(NgpErrorMapCdf& storage) const;   // NULL when both switches are off (3211-3212, 3243-3245)

	void set_image(int frame_idx, int w, int h, const float* rgba_host, const float* depth_host = nullptr, float depth_scale = -1.f);   // python_api.cu:53-72 (float RGBA, optional float depth)
	void set_image_rgba8(int frame_idx, int w, int h, const uint8_t* rgba_host);         // Byte images as the PNG loader stores them (nerf_loader.cu:622-623)
	void set_camera_extrinsics(int frame_idx, const Mat34& camera_to_world, bool convert_to_ngp = true); // testbed_nerf.cu:2539-2541
	Mat34 get_camera_extrinsics(int frame_idx) const;
	void set_camera_intrinsics(int frame_idx, float fx, float fy, float cx, float cy, float k1, float k2, float p1, float p2); // testbed_nerf.cu:2502-2516
	void update_transforms(int first = 0, int last = -1);                                // testbed_nerf.cu:2598-2633 (dataset.xforms + the extrinsic offsets)
};

struct Nerf {
	NerfTraining training;
	DeviceBuffer density_grid;            // fp32 [(max_cascade+1) * 128^3], Morton order
	DeviceBuffer bitfield_brick_summary; bool brick_summary_valid = false;   // ngp_hip_bitfield_brick_summary of the current bitfield (the march's empty-space shortcut)
	DeviceBuffer density_grid_bitfield;   // 8 cascades * 128^3 / 8 bytes
	DeviceBuffer density_grid_mean;       // 1 float
	uint32_t max_cascade = 0;
	uint32_t density_grid_ema_step = 0;
	ENerfActivation rgb_activation = ENerfActivation::Exponential;   // testbed.h:709
	ENerfActivation density_activation = ENerfActivation::Exponential; // testbed.h:710
	float cone_angle_constant = 1.f / 256.f;
	float render_min_transmittance = 0.01f;  // testbed.h:725
	uint32_t render_n_streams = 1;           // >1 traces the frame as independent pixel ranges on separate streams (measured slower on ROCm 7.0: 16 -> 28 ms at 2)
	uint32_t render_max_skips_per_pass = 96; // fused-compaction tracer: empty voxels a ray may step over per pass before it rests until the next one (ngp_hip_generate_next_inputs); 0: no limit, as the reference.  Same image.
	float render_pass_samples_factor = 4.0f; // fused-compaction tracer: network samples per pass = this x the frame's pixels (the reference: 1); [1, 4].  Same image; 800x800 on MI355X: 1 -> 7.3 ms, 2 -> 5.8, 3 -> 5.4, 4 -> 5.35 (fewer, larger passes: tools/render_probe.py)  Round 3 re-sweep at HEAD (tools/render_probe.py NGP_PROBE_SWEEP): 3 -> 5.63 ms, 4 -> 5.48 ms: 4.
	uint32_t render_max_steps_per_pass = 0;  // 0 = by the scene (round 5): 64 with one cascade (lego stand-in: 5.2 ms per 800 x 800 frame; 5.6 at 16, 6.9 at 8), 8 — the reference's
	                                         // m_max_steps_inbetween_compactions (testbed.h NerfTracer) — with several (fox photographs: 18.1 ms per 1080 x 1920 frame; 18.4 at 16, 19.2 at 64); same pixels
	uint32_t render_steps_cap() const { return render_max_steps_per_pass ? render_max_steps_per_pass : (max_cascade > 0 ? 8u : 64u); }
	bool render_with_lens_distortion = false;
	float sharpen = 0.f;
	int show_accel = -1;
	bool render_fused_network = false;       // the tracer's network pass through the fused kernel (gathers inside) instead of XCD-affine encode + MLP kernel: same samples, same bits
	bool render_fused_compaction = true;     // compaction folded into advance_pos / composite (NgpCompactOut) instead of a pass of its own; off: the reference's loop
	bool render_tile_order = true;           // the tracer's rays in 8 x 8 pixel tiles (NgpRenderExtras.tile_order) when the frame's size allows: same pixels, more coherent gathers
	Vec3 light_dir{0.5f, 0.5f, 0.5f};        // testbed.h:712: the light direction presented at inference time when the dataset has light directions
	uint32_t extra_dim_idx_for_inference = 0; // testbed.h:713: which training image's latent code is presented at inference time
	bool visualize_cameras = false;          // GUI-side (stored)
	float glow_y_cutoff = 0.f; int glow_mode = 0;   // testbed.h:730-731; the glow shading of composite_kernel_nerf (testbed_nerf.cu:843-939)
	NgpImageMeta render_lens_proxy{};        // only lens_mode / lens_params are used (render_lens)
};

// frame / accumulate / output buffers (src/render_buffer.cu CudaRenderBuffer, windowless surface)
struct RenderBuffer {
	DeviceBuffer frame_buffer, depth_buffer, accumulate_buffer, surface;
	int32_t res[2] = {0, 0};
	uint32_t spp = 0;
	EColorSpace color_space = EColorSpace::Linear;
	ETonemapCurve tonemap_curve = ETonemapCurve::Identity;
	void resize(int w, int h);
	void reset_accumulation() { spp = 0; }
};

struct TrainStats { float training_prep_ms = 0, training_ms = 0, render_ms = 0; };

class NerfRenderer;
struct RenderRequest;
class ShmCounterExchange;

class Testbed {
public:
	explicit Testbed(ETestbedMode mode = ETestbedMode::Nerf);
	~Testbed();

	// ---- data (python_api.cu:546, 619)
	void load_training_data(const std::string& path);
	void create_empty_nerf_dataset(size_t n_images, int aabb_scale = 1, bool is_hdr = false);
	void load_nerf_post();                                             // testbed_nerf.cu:2643-2733

	// ---- network (testbed.cu:120-194, 2249-2470)
	void reload_network_from_file(const std::string& path = "");
	void reload_network_from_json(const Json& json, const std::string& config_base_path = "");
	Json load_network_config(const std::string& path);
	void reset_network(bool clear_density_grid = true);
	void reset(bool reset_density_grid = true) { reset_network(reset_density_grid); }
	size_t n_params() const { return m_n_params; }
	size_t n_encoding_params() const { return m_n_params - m_n_matrix_params; }

	// ---- training (testbed.cu:2044-2090, 2527-2587; testbed_nerf.cu:2761-3401)
	bool frame();
	void train(uint32_t batch_size);
	void training_prep_nerf(uint32_t batch_size);
	void update_density_grid_nerf(float decay, uint32_t n_uniform, uint32_t n_nonuniform);
	void update_density_grid_mean_and_bitfield(bool bitfield_is_current = false);   // true: the update's fused tail already left mean + bitfield; only the brick summary is rebuilt
	void train_nerf(uint32_t target_batch_size, bool get_loss_scalar);
	// data-parallel split of train_nerf: begin = everything up to and including backward (gradients ready in
	// gradients()), end = optimizer step + counter feedback given the GLOBAL (all-rank summed) counters.
	void set_distributed(uint32_t rank, uint32_t world_size);
	// The whole data-parallel step inside train(): ranks of ONE node, shared-memory counter exchange + an RCCL communicator of this Testbed's own
	// (ngp_rccl_*).  `key` names the rendezvous (same on every rank of the job, e.g. MASTER_PORT); strong_scaling: train(B) back-propagates B / world
	// samples per rank (the reference's convergence per step); otherwise B per rank (world x the global batch).  After this frame() / train() work
	// at world_size > 1 like at 1.
	void init_data_parallel(uint32_t rank, uint32_t world_size, const std::string& key, bool strong_scaling);
	// a frame rendered in row shards (testbed.cpp "a frame rendered by several ranks"): with a data-parallel communicator render() is sharded over its ranks
	// automatically; set_render_shard makes this Testbed trace the rows of (rank, world) only, the rest of the surface stays background
	void set_render_shard(uint32_t rank, uint32_t world);
	void render_shard_rows(int height, int& row_begin, int& row_end) const;
	void fetch_render_surface(RenderBuffer& rb, float* out);
	void download(const void* device_src, size_t bytes, void* host_dst);   // device -> pinned staging (kept) -> host_dst; never a DMA into the caller's pageable memory (testbed.cpp)
	void* m_pinned = nullptr; size_t m_pinned_bytes = 0;
	std::mutex m_download_mutex;             // download() may be entered by the caller's thread and by the async render worker at once
	uint32_t m_render_shard_rank = 0, m_render_shard_world = 1;
	DeviceBuffer m_render_gather;
	// the data-parallel optimizer step (testbed.cpp optimizer_step_sharded): reduce-scatter -> Adam on the rank's shard -> all-gather; false: fp16 all-reduce + replicated step
	bool m_dp_sharded_optimizer = true;
	void set_dp_sharded_optimizer(bool on);   // only without a live communicator
	bool m_dp_state_stale = false;            // sharded steps ran since the last dp_gather_optimizer_state: fp32 state outside this rank's shard is old
	bool m_dp_fp16_wire = true;               // sharded step's gradient exchange: fp16 slices point to point + fp32 sum in rank order (false: widen -> fp32 reduce-scatter -> narrow, rounds 3-4)
	bool m_dp_sharded_ema = true;             // sharded step: the Ema stage on this rank's shard only; inference weights gathered on demand (false: Ema over all parameters on every rank)
	bool m_dp_inference_stale = false;        // sharded-Ema steps ran since the last dp_gather_inference_params: the inference weights outside this rank's shard are old
	void set_dp_sharded_ema(bool on);         // only without a live communicator
	void dp_gather_inference_params();        // COLLECTIVE: in-place all-gather of the fp16 inference weights
	void inference_params_from_training_weights(const char* who);   // stale sharded Ema weights and no communicator left: fall back to the (whole) training weights
	void require_inference_params(const char* who, bool collective);
	bool m_render_sharded = false;            // render() is a collective over the data-parallel ranks (rows per rank + all-gather); off: local, whole frame
	bool render_is_collective() const { return m_dp_comm && m_render_sharded; }   // (a one-rank communicator runs the same path: split, gather buffer, RCCL call)
	static constexpr size_t DP_PARAM_SLACK = 1024;   // elements behind the weights / gradients: world x shard (shard a multiple of 8) may exceed n_params by < 8 x world
	DeviceBuffer m_dp_grads_f32, m_dp_shard_f32;
	void optimizer_step_sharded();
	void dp_gather_optimizer_state();   // collective: the whole fp32 optimizer state on every rank (before a snapshot with optimizer state / leaving data-parallel mode)
	void shutdown_data_parallel();
	bool m_dp_strong_scaling = false;
	bool m_dp_march_behind_exchange = false;   // set by init_data_parallel when world_size > 1 (pyngp: dp_march_behind_exchange, settable for one-rank tests)
	void set_dp_march_behind_exchange(bool on) { drop_prefetch(); m_dp_march_behind_exchange = on; if (on) m_want_grid_grad_event = true; }
	void set_dp_strong_scaling(bool on) { drop_prefetch(); m_dp_strong_scaling = on; }   // every rank, between two steps
	// step = begin (samples, inference, loss/compaction; returns the LOCAL counters) -> [all-reduce counters + loss]
	//      -> backward (counter feedback with the GLOBAL sums, next step's march on stream B, forward + backward; gradients ready)
	//      -> [all-reduce gradients] -> end (optimizer, bookkeeping)
	void train_nerf_dp_begin(uint32_t target_batch_size, uint32_t counters_out[2], bool get_loss_scalar = false, bool wait_for_counters = true);
	// data-parallel hooks: a device buffer of 3 doubles that receives {samples, compacted samples, loss sum} of every step begun (the
	// operand of the host's counter all-reduce), and stream-order waits on "counters posted" / "grid gradients final"
	void set_dp_counter_buffer(void* three_doubles_dev) { m_dp_counters_dev = three_doubles_dev; }
	void stream_wait_counters(void* other_stream);
	void stream_wait_grid_gradients(void* other_stream);
	void train_nerf_dp_backward(uint32_t target_batch_size, uint32_t global_measured_before, uint32_t global_measured, bool get_loss_scalar, float global_loss_sum);
	void train_nerf_dp_end();
	void invalidate_training_inputs();
	// ---- plumbing configs: 2-D image fitting and SDF fitting through one grid -> MLP network (host/plumbing.cpp)
	struct ImageState {   // testbed.h Image (m_image)
		DeviceBuffer data, positions, targets, render_coords, render_out, se;
		int32_t resolution[2] = {0, 0};
		int type = 0;                  // 2 = half4, 3 = float4 (EDataType)
		float pos[2] = {0.f, 0.f};
		bool snap_to_pixel_centers = false, linear_colors = false;   // m_image.training.*
		bool stratified = true;        // ERandomMode::Stratified (testbed.h default)
	} m_image;
	struct SdfState {     // the part of m_sdf that override_sdf_training_data feeds (python_api.cu:74-100)
		DeviceBuffer positions, distances;
		uint32_t n_samples = 0, cursor = 0;
		float mesh_scale = 1.0f;
	} m_sdf;
	void load_image(const std::string& path);                                         // testbed_image.cu:362-434 (.bin: int32 h, int32 w, fp16 RGBA)
	void set_image_data(int w, int h, const float* rgba_host);                        // fp32 RGBA (what load_exr_image / load_stbi_image leave on the device)
	void override_sdf_training_data(const float* points, const float* distances, size_t n);
	void reset_network_gridmlp();
	void train_image(uint32_t batch_size, bool get_loss_scalar);                      // testbed_image.cu:220-291
	void train_sdf(uint32_t batch_size, bool get_loss_scalar);                        // testbed_sdf.cu:1229-1252 on user-provided (pos, distance) pairs
	void render_image(RenderBuffer& rb);                                              // testbed_image.cu:293-360
	float compute_image_mse(bool quantize_to_byte);                                   // testbed_image.cu:461-523
	uint32_t gridmlp_n_dims() const { return m_testbed_mode == ETestbedMode::Image ? 2u : 3u; }
	void gridmlp_training_step(const float* pos, uint32_t n_dims, const float* targets, uint32_t dims, uint32_t n, bool get_loss_scalar);
	uint32_t m_n_matrix_params = NGP_MLP_N_PARAMS;     // parameters that get weight decay (the MLPs): 10240 NeRF (base family), 7168 grid -> MLP
	// the network the config + dataset ask for (src/testbed.cu:2337-2363): extra dims behind the direction encoding, hidden layers of the colour network.  The base
	// family (0, 2) runs the fused kernels; anything else the generic ones (NgpNetVariant, include/ngp_hip.h)
	uint32_t m_n_extra_dims = 0, m_n_rgb_hidden_layers = 2;
	bool m_netx_scalar_kernels = false;                // network variants on the scalar checker kernels (NGP_NETX_SCALAR) instead of the MFMA kernels
	bool net_is_base_family() const { return m_n_extra_dims == 0 && m_n_rgb_hidden_layers == 2; }
	const NgpNetVariant* net_variant(NgpNetVariant& storage, const float* extra_dims = nullptr, const uint32_t* sample_slot = nullptr, float* dL_dextra = nullptr) const {
		if (net_is_base_family()) return nullptr;
		storage.n_extra_dims = m_n_extra_dims; storage.n_rgb_hidden_layers = m_n_rgb_hidden_layers; storage.extra_dims = extra_dims; storage.sample_slot = sample_slot; storage.dL_dextra = dL_dextra;
		storage.flags = m_netx_scalar_kernels ? (uint32_t)NGP_NETX_SCALAR : 0u;
		return &storage;
	}
	const float* get_inference_extra_dims();           // testbed_nerf.cu:2320-2337
	DeviceBuffer m_ray_image, m_sample_slot_all, m_sample_slot, m_dl_dextra, m_extra_numsteps;   // per kept ray: its image; per sample (pre-compaction / compacted): the extra-dims row; dL/d(extra dims) [B][n_extra]
	DeviceBuffer m_gm_out, m_gm_values;

	// ---- Blender multi-NeRF requests (python_api.cu:192-260, testbed.cu:2675-2693)
	void bl_render_frame(RenderBuffer& rb, const RenderRequest& request);
	std::vector<float> bl_request_nerf_render_sync(const RenderRequest& request);      // H*W*4 floats; zeros while another render is running
	bool bl_request_nerf_render_sync(const RenderRequest& request, float* out);        // into the caller's array; false (array untouched) while another render is running
	bool bl_try_begin_render();                                                        // false if a render is already in flight
	void bl_end_render();
	uint64_t m_bl_render_samples = 0;
	uint32_t m_bl_render_passes = 0;
	// the backward pass over the live samples of the batch (round 5): samples whose loss gradient is zero in all four channels (ray tails in fp16: 30-45 % of a batch) are
	// left out of the MFMA kernel and of the hash-grid binning (ngp_hip_compact_live_samples + ngp_hip_nerf_backward_live; same hash-grid gradients bit for bit)
	bool m_compact_backward = false;         // pyngp: compact_backward (off: measured neutral on the whole step, profiles/r05_experiments.md section 8)
	DeviceBuffer m_coords_live, m_live_index, m_live_count;   // the live samples' coordinate rows next to each other, their rows in the batch; two counters (step parity: one is read while the other is cleared)
	uint32_t m_live_parity = 0;
	float backward_live_fraction();          // pyngp (read-only; drains the stream): live samples / batch of the last compacted step
	uint32_t m_live_last_batch = 0, m_live_last_parity = 0;
	bool m_ema_on_side_stream = false;       // pyngp: ema_on_side_stream — the optimizer step's Ema stage on stream B (optimizer_step()); off: measured slower on three of four workloads
	void join_side_ema();                    // stream A waits for the pending Ema stage (readers of m_ema / m_inference_params call it; so does sync())
	bool m_bl_fused_passes = true;            // NerfRenderer::fused_passes and its schedule knobs (nerf_renderer.h)
	uint32_t m_bl_max_skips_per_pass = 96, m_bl_max_steps_per_pass = 64;
	float m_bl_pass_samples_factor = 4.0f;
	bool m_bl_reference_schedule = false;
	RenderBuffer m_bl_render_surface;
	// request_nerf_render_async workers: detached like the reference's (python_api.cu:228-229), but counted, so that the Testbed can wait for them
	// (a worker may itself queue the next request from its callback — nobody ever joins a thread, least of all itself)
	void bl_start_async(std::function<void()> job);
	void bl_wait_for_renders();
	std::mutex m_render_mutex; std::condition_variable m_render_cv; int m_render_workers = 0;
	void* stream() const { return m_stream; }          // hipStream_t all training work is queued on
	                                                   // instead of evaluating every marched sample; same kept samples, measured at par with the flat pass (its tiles pack worse), so off
	bool m_trace_sync = false;                         // debugging aid: drain both streams behind every launch group and name it on stderr (finds a kernel that never returns)
	bool m_render_trace = false;                       // debugging aid: the tracers' pass structure (alive rays, steps per pass) on stderr
	bool m_morton_grid_samples = true;                 // occupancy-grid update: generate the samples in Morton order of their cells (coherent gathers in the density pass and the splat; the same samples, the same grid)
	bool m_x_row_index_mode = true;                    // training step, base network family: the compaction leaves an index into the uncompacted batch's encoding rows instead of copying the rows (the same bits reach the backward pass)
	std::vector<uint16_t> debug_x_saved(size_t batch);
	bool m_enable_prefetch = true;                     // march step n+1 on a second stream while step n back-propagates
	bool m_separate_forward = false;                   // dev / test: run the reference's second network pass over the compacted batch as well
	uint64_t m_prefetch_hits = 0;
	uint32_t m_grid_prefetch_hits = 0;                 // occupancy-grid updates whose sample positions were generated ahead on stream B
	// test hook: the next update's samples as stream B generated them ahead (false when none is pending), or (regenerate) generated now in stream order from the same generator state
	bool debug_grid_update_samples(bool regenerate, std::vector<float>& positions, std::vector<uint32_t>& indices, uint32_t& step);
	uint16_t* gradients() const { return m_grads.as<uint16_t>(); }
	float local_loss_sum();
	// ---- test hook: a stage-by-stage record of ONE training step of the product path (tests/test_baseline_configs_gpu.py replays it through the
	// oracle at the BASELINE sizes).  Armed by debug_capture_next_step(); the step then copies, in stream order, the parameters it starts from,
	// the march's outputs before the loss kernel rewrites them, and the loss kernel's outputs before the roll-over.
	struct StepCapture {
		bool armed = false, valid = false;
		uint32_t step = 0, R = 0, max_inference = 0, n_rays_global = 0, ray_offset = 0, target_batch_size = 0;
		uint64_t rng_state = 0, rng_inc = 0;
		DeviceBuffer params, ray_indices, rays, numsteps, coords, gen_counters, numsteps_compacted, coords_compacted, dloss, density_grid_mean, bitfield;
		bool prefetch_hit = false;
	} m_capture;
	void debug_capture_next_step() { m_capture.armed = true; m_capture.valid = false; }
	const DeviceBuffer& debug_buffer(const std::string& name) const;   // step scratch by name: "mlp_out", "coords_compacted", "dloss", "x_saved", "grads", "coords"

	// ---- rendering (python_api.cu:132-190; testbed.cu:2695-2911; testbed_nerf.cu:2047-2267, 2354-2500)
	std::vector<float> render_to_cpu(int width, int height, int spp, bool linear);
	void render_to_cpu(int width, int height, int spp, bool linear, float* out);      // into the caller's array (pyngp: the numpy array it returns)
	void render_frame(const Mat34& cam0, const Mat34& cam1, const float rolling_shutter[4], RenderBuffer& rb, bool to_srgb);
	void render_nerf(RenderBuffer& rb, const float focal_length[2], const Mat34& cam0, const Mat34& cam1, const float rolling_shutter[4], const float screen_center[2]);
	void set_nerf_camera_matrix(const Mat34& cam) { m_camera = m_nerf.training.dataset.nerf_matrix_to_ngp(cam); } // testbed.cu:219-221
	void reset_camera();
	float fov() const;
	void set_fov(float val);
	void fov_xy(float out[2]) const;                                   // testbed.cu:2161-2167
	void set_fov_xy(const float val[2]);
	// camera helpers of testbed.cu:223-243 (column-major 3x4: columns 0..2 = right / up(down) / view direction, column 3 = position)
	Vec3 view_pos() const { return Vec3{m_camera.m[9], m_camera.m[10], m_camera.m[11]}; }
	Vec3 view_dir() const { return Vec3{m_camera.m[6], m_camera.m[7], m_camera.m[8]}; }
	Vec3 look_at() const;
	void set_look_at(const Vec3& pos);
	void set_view_dir(const Vec3& dir);
	float scale() const { return m_scale; }
	void set_scale(float scale);
	void set_camera_to_training_view(int trainview);                   // testbed.cu:273-281
	void first_training_view(); void last_training_view(); void previous_training_view(); void next_training_view();   // testbed.cu:245-271
	Mat34 crop_box(bool nerf_space) const;                             // testbed.cu:395-445
	void set_crop_box(Mat34 m, bool nerf_space);
	std::vector<Vec3> crop_box_corners(bool nerf_space) const;
	// python_api.cu:262-275: spp frames between two camera poses, per-ray time A + B u + C v + D t from `rolling_shutter`
	std::vector<float> render_with_rolling_shutter_to_cpu(const Mat34& camera_transform_start, const Mat34& camera_transform_end, const float rolling_shutter[4], int width, int height, int spp, bool linear);
	void render_with_rolling_shutter_to_cpu(const Mat34& camera_transform_start, const Mat34& camera_transform_end, const float rolling_shutter[4], int width, int height, int spp, bool linear, float* out);

	// ---- snapshots (testbed.cu:3006-3106) — next-row f1, see DESIGN.md
	void save_snapshot(const std::string& path, bool include_optimizer_state);
	void load_snapshot(const std::string& path);

	float loss() const { return m_loss_scalar; }
	uint32_t training_step() const { return m_training_step; }
	void sync();

	// ---- state (names follow testbed.h)
	ETestbedMode m_testbed_mode;
	Nerf m_nerf;
	bool m_train = false;
	bool m_train_encoding = true, m_train_network = true;   // testbed.h:914-915: Adam's optimize_non_matrix_params / optimize_matrix_params (testbed.cu:2556-2563)
	bool m_training_data_available = false;
	uint32_t m_training_step = 0;
	uint32_t m_training_batch_size = 1 << 18;          // testbed.h:909
	float m_loss_scalar = 0.f;
	uint64_t m_seed = 1337;                             // testbed.h:567
	Pcg32 m_rng;
	NgpAabb m_aabb{}, m_raw_aabb{}, m_render_aabb{};
	float m_bounding_radius = 1.0f;                    // testbed.h:556 (snapshot field)
	float m_render_aabb_to_local[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	float m_background_color[4] = {0.f, 0.f, 0.f, 1.f}; // testbed.h:875
	EColorSpace m_color_space = EColorSpace::Linear;
	float m_exposure = 0.f;
	bool m_snap_to_pixel_centers = false;
	float m_render_near_distance = 0.0f;
	ERenderMode m_render_mode = ERenderMode::Shade;    // every ERenderMode of the stock tracer (ngp_hip_composite_ex / ngp_hip_init_rays_ex / the Slice kernels)
	int m_device = 0;               // the HIP device this Testbed's streams and buffers live on
	TrainableBuffer m_envmap;       // testbed.h:936-944: resolution of the dataset's `envmap` image ((0, 0): none)
	TrainableBuffer m_distortion;   // testbed.h:946-952: 32 x 32 x 2 zeros unless trained; passed unconditionally to the ray generator
	std::vector<NgpMask3D> m_render_masks;             // python_api.cu:694: crop masks of the STOCK renderer (testbed_nerf.cu:2339-2352, 833-840, 1943-1956)
	uint32_t m_n_render_masks = 0;
	void prepare_nerf_masks();                         // testbed_nerf.cu:2339-2352
	int m_quilting_dims[2] = {1, 1};                   // testbed.h:549 (set by the reference's VR / HoloPlay GUI paths only)
	NgpRenderCamera m_render_camera_models{};          // stock renderer: model (0 Perspective) + the SphericalQuadrilateral / QuadrilateralHexahedron shapes (python_api.cu:691-693)
	float m_aperture_size = 0.0f;                      // depth of field: radius of the lens disk (testbed.h m_aperture_size; python `dof` / `aperture_size`)
	float m_slice_plane_z = 0.0f;                      // focus distance is m_slice_plane_z + m_scale (testbed_nerf.cu:2355)
	bool m_autofocus = false;
	float m_autofocus_target[3] = {0.5f, 0.5f, 0.5f};
	void autofocus();                                  // testbed.cu:2933-2941
	Mat34 m_camera;
	bool m_camera_smoothing = false, m_loop_animation = false, m_dynamic_res = false;   // GUI-side state kept for script compatibility
	float m_dynamic_res_target_fps = 20.0f; int m_fixed_res_factor = 8;                 // testbed.h:521-522
	bool m_imgui_enabled = true, m_visualize_unit_cube = false, m_floor_enable = false, m_dlss = false; float m_dlss_sharpening = 0.0f;
	bool m_render_ground_truth = false; int m_ground_truth_render_mode = 0;             // testbed.h:880-881 (GUI overlay of the training images)
	int m_visualized_dimension = -1; uint32_t m_visualized_layer = 0;                   // > -1: EncodingVis / neuron visualisation (testbed_nerf.cu:2360)
	Vec3 m_sun_dir{0.57735027f, 0.57735027f, 0.57735027f}; float m_parallax_shift[3] = {0.f, 0.f, 0.f};
	Vec3 m_up_dir{0.f, 1.f, 0.f};
	float m_relative_focal_length[2] = {1.f, 1.f};
	uint32_t m_fov_axis = 1;
	float m_zoom = 1.f;
	float m_screen_center[2] = {0.5f, 0.5f};
	float m_scale = 1.5f;
	bool m_max_level_rand_training = false;
	Json m_network_config;
	std::string m_network_config_path;
	std::string m_data_path;
	TrainStats m_stats;
	RenderBuffer m_windowless_render_surface;
	uint64_t m_render_samples_evaluated = 0;          // network samples of the last render_to_cpu (for MP/s + roofline accounting)

	// ---- live kernel timing (HIP events on m_stream, the stream the kernels are launched on): bench.py's roofline numbers
	enum ProfKernel { PK_GEN_SAMPLES = 0, PK_INFERENCE, PK_LOSS, PK_FORWARD, PK_BACKWARD, PK_OPTIMIZER, PK_GRID_PREP, PK_GRAD_EXCHANGE, PK_PARAM_GATHER, PK_COUNT };
	struct ProfAccum { double ms = 0; uint64_t launches = 0; uint64_t units = 0; };
	bool m_counters_host_seen = true;                  // the host polled the counters of the step begun last (else the run-ahead march waits for the counters event)
	bool m_async_training_steps = false;              // frame() returns with the step's tail (backward, optimizer) still running on the stream
	std::chrono::steady_clock::time_point m_last_step_return{};
	bool m_profile_enabled = false;
	uint32_t m_profile_mask = ~0u;   // bit k: bracket the launches of ProfKernel k with events (each bracket costs a few us of dispatch gap)
	uint32_t m_profile_every = 1;    // bracket the launches of every n-th training step only (a sampled live measurement: 1/n of the dispatch gaps)
	ProfAccum m_prof[PK_COUNT];
	void reset_profile();
	void profile_begin(int k, void* stream = nullptr);
	void profile_end(int k, uint64_t units, void* stream = nullptr);
	void profile_collect(bool only_finished = false);                            // after a stream sync: fold pending event pairs into m_prof

	// ---- organisation of the network pass over a training batch.  The library has two, bit for bit the same function (tests/test_network_gpu.py,
	// tests/test_gridmlp_gpu.py): Fused = ngp_hip_nerf_forward / ngp_hip_gridmlp_forward (gathers inside the MLP kernel: the L1 serves neighbouring samples of a ray),
	// TwoKernel = ..._ws (XCD-affine encode into level planes + MLP kernel: every XCD's L2 holds the tables it walks).  Which is faster is the WORKLOAD's property —
	// profiles/r05_a_*: lego stand-in (constant step, one cascade) 171 us fused / 198 two-kernel; fox photographs (cone stepping, three cascades) 270 / 201; SDF batches
	// of random points 171 / 95 — so Auto (the default) measures: a calibration brackets the pass with HIP events for 2 x N_SAMPLES consecutive steps, alternating the
	// organisations, and keeps the one with the lower median (the other must win by 3 % to take over).  Calibrations run early (behind the first occupancy updates),
	// again once the step cadence has settled, and then rarely; a reset_network / new data starts over.
	enum class ENetworkPass : int { Auto = 0, Fused = 1, TwoKernel = 2 };
	struct NetworkPassTuner {
		static constexpr int N_SAMPLES = 6;
		ENetworkPass chosen = ENetworkPass::Fused;
		uint32_t next_calibration_step = 0;      // set by tuner_reset
		uint32_t n_calibrations = 0, last_calibration_step = 0;
		float last_us[2] = {0.f, 0.f};           // medians of the last calibration: [0] fused, [1] two-kernel
		int remaining = 0, measuring = -1;       // launches the running calibration still has to issue; organisation bracketed right now (-1: none)
		float us[2][N_SAMPLES]; int count[2] = {0, 0};
		struct Pending { void* e0; void* e1; int org; };
		std::vector<Pending> pending;
	};
	ENetworkPass m_network_pass = ENetworkPass::Auto;
	NetworkPassTuner m_pass_tuner;
	void tuner_reset();
	bool tuner_pick(void* stream);               // true: this launch runs the two-kernel organisation; starts / continues a calibration
	void tuner_done(void* stream);               // behind the launch
	void tuner_collect(bool wait);

	// network + optimizer state
	NgpNetDesc m_desc{};
	DeviceBuffer m_desc_gpu, m_params, m_inference_params, m_master, m_first_moments, m_second_moments, m_ema, m_grads;
	size_t m_n_params = 0;
	uint32_t m_optimizer_step = 0;
	float m_learning_rate = 1e-2f, m_base_learning_rate = 1e-2f, m_beta1 = 0.9f, m_beta2 = 0.99f, m_epsilon = 1e-15f, m_l2_reg = 1e-6f;
	float m_ema_decay = 0.95f;
	bool m_use_ema = true;
	uint32_t m_decay_start = 20000, m_decay_interval = 10000, m_decay_end = 0;
	float m_decay_base = 0.33f;
	bool m_has_decay = true;
	float m_per_level_scale = 0.f;
	uint32_t m_base_grid_resolution = 16, m_num_levels = 16;

	// distributed
	uint32_t m_rank = 0, m_world_size = 1;
	void* m_dp_comm = nullptr;                         // ngp_rccl_init handle (init_data_parallel)
	std::unique_ptr<class ShmCounterExchange> m_dp_shm;
	uint64_t m_dp_exchange_step = 0;

private:
	std::unique_ptr<NerfRenderer> m_renderer;
	std::atomic<bool> m_currently_rendering{false};
	void* m_stream = nullptr;
	void* m_stream_b = nullptr;                        // second stream: sample generation one step ahead
	struct PrefetchedSamples { bool valid = false; uint32_t step = 0, R = 0, max_inference = 0, batch = 0, cdf_mode = 0; uint64_t rng_state = 0, version = 0; int n_images = 0; int slot = 0; };
	PrefetchedSamples m_prefetch;
	void* m_prefetch_event = nullptr;
	// the occupancy-grid update's sample positions, generated ahead on stream B during the step before the update (they depend on the grid and its generator,
	// not on the parameters that step is still training)
	struct PrefetchedGridSamples { bool valid = false; uint32_t step = 0, n_uniform = 0, n_nonuniform = 0, ema_step = 0; uint64_t rng_state = 0, rng_inc = 0, version = 0; int n_images = 0; };
	PrefetchedGridSamples m_grid_prefetch;
	void* m_grid_prefetch_event = nullptr;
	// the Ema stage of the optimizer step on stream B (round 5, opt-in): it reads the new fp16 weights and writes only what renderers and snapshots read, so it leaves the training
	// chain; m_adam_event orders it behind the Adam stage, m_ema_event orders the next Adam stage (which overwrites the weights it reads) and every reader behind it
	void* m_adam_event = nullptr;
	void* m_ema_event = nullptr;
	std::atomic<bool> m_ema_pending{false}; std::mutex m_ema_mutex;              // stream A has not yet been ordered behind the last Ema launch
	void maybe_prefetch_grid_samples(uint32_t next_step);
	void launch_grid_samples(void* stream, uint32_t n_uniform, uint32_t n_nonuniform);   // memset of the splat buffer + the two generators; advances density_grid_rng twice
	void* m_counters_event = nullptr;
	bool m_want_counters_event = false, m_counters_event_recorded = false, m_want_grid_grad_event = false, m_grid_grad_event_recorded = false;
	void* m_host_words = nullptr;                      // 4 pinned, device-mapped words: {numsteps, numsteps_compacted, loss sum, -}
	float m_local_loss_sum = 0.f;
	int m_gen_slot = 0;
	int m_compact_slot_zeroed = -1;                    // ... and whose compacted-samples counter
	int m_next_slot_zeroed = -1;                       // slot whose march counters the last post_words launch cleared
	uint32_t m_post_tag = 0;
	void* m_dp_counters_dev = nullptr;
	void* m_grid_grad_event = nullptr;
	uint64_t m_state_version = 0;
	bool m_train_continues = true;
	DeviceBuffer m_gen_counters;                       // 2 slots x {ray counter, numsteps counter, compacted numsteps counter, pad}
	uint32_t next_max_inference(uint32_t target_batch_size) const;
	void launch_generate(void* stream, int slot, uint32_t R, uint32_t max_inference, const Pcg32& rng, bool next_to_backward);
	void maybe_prefetch_next(uint32_t target_batch_size);
	void drop_prefetch();
	// step scratch (replaces the GPUMemoryArena carve-out of train_nerf_step 3144-3170 and update_density_grid_nerf 2770-2776)
	DeviceBuffer m_ray_indices, m_rays, m_numsteps, m_coords, m_mlp_out, m_dloss, m_coords_compacted, m_x_saved, m_bwd_scratch, m_ray_counter;
	DeviceBuffer m_cam_rays, m_cam_ray_indices, m_coords_gradient;   // optimize_extrinsics: the step's rays kept past the next march, and the network's input gradient [sample][6]
	DeviceBuffer m_x_index;                            // slot of the compacted batch -> row of m_x_all (x_row_index_mode)
	bool m_last_step_used_x_index = false;
	DeviceBuffer m_x_all;                              // encodings of the uncompacted samples (carried through the compaction by the loss kernel)
	DeviceBuffer m_enc_ws;                             // level planes of the XCD-affine encode (ngp_hip_nerf_*_ws)
	DeviceBuffer m_grid_positions, m_grid_indices, m_grid_tmp, m_grid_mlp_out;
	DeviceBuffer m_grid_mean_ws;             // per-workgroup partial sums of the mean (the update's fused tail)
	DeviceBuffer m_grid_sample_counters;     // per-workgroup sample counts of the Morton-ordered generators (one workspace per half)
	DeviceBuffer m_loss_scalar_gpu;
	// tracer scratch (NerfTracer::enlarge 2270-2295)
	std::vector<void*> m_render_streams;
	void* m_render_event = nullptr;
	void* m_render_host_words = nullptr;
	uint32_t m_render_sequence = 0;   // NgpCompactOut::sequence of the last tracer pass issued
	std::vector<DeviceBuffer> m_tr_enc_ws;
	DeviceBuffer m_render_masks_gpu, m_tr_vis_scratch, m_tr_vis_rgba;
	DeviceBuffer m_tr_payload[2], m_tr_rgba[2], m_tr_depth[2], m_tr_hit_payload, m_tr_hit_rgba, m_tr_hit_depth, m_tr_net_in, m_tr_net_out, m_tr_counters;
	struct ProfPending { int k; void* e0; void* e1; uint64_t units; };
	std::vector<ProfPending> m_prof_pending;
	std::vector<void*> m_prof_event_pool;
	void* prof_event();
	void check(int rc, const char* what);
	void optimizer_step();
	void update_after_training(uint32_t target_batch_size, uint32_t counter, uint32_t compacted_counter, bool get_loss_scalar, float loss_sum);
	void parse_optimizer_config(const Json& opt);
};

} // namespace ngp
